// trw_kernels_wire.hpp -- the wire format of the multi-GPU gather (VERDICT r02 missing 2).
//
// The reference is single-device (csrc/cuda/rw_cuda.cu:198); the one exchange this path has is the gather of the
// [B, L+1] int64 row blocks to the root.  Node ids are < n_nodes, so while n_nodes <= 2^32 a row travels as uint32
// -- half the bytes over xGMI, whose per-link rate is what bounds the gather -- and is widened again where it lands.
// Both kernels are plain streams: 32 bytes of int64 <-> 16 bytes of uint32 per thread, fully coalesced, ONE trip per
// thread (a grid of count / 4 threads: on this part a copy whose threads make one trip each sustains 6.2 TB/s, a
// persistent grid-stride loop over the same loads 4.7 -- scripts/machine_probe.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

typedef unsigned long long u64x2_w __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_w __attribute__((ext_vector_type(4)));

// rows[0..count) -> wire[0..count): the low 32 bits; a value that does not fit sets bit 0 of *overflow (optional).
__global__ void __launch_bounds__(256) rows_narrow_kernel(const int64_t *__restrict__ rows, int64_t count,
                                                          uint32_t *__restrict__ wire,
                                                          unsigned long long *__restrict__ overflow)
{
    const int64_t quads = count >> 2;
    unsigned long long high = 0;
    const u64x2_w *src = reinterpret_cast<const u64x2_w *>(rows);
    u32x4_w *dst = reinterpret_cast<u32x4_w *>(wire);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < quads) {
        const u64x2_w a = src[2 * i], b = src[2 * i + 1];
        high |= (a.x | a.y | b.x | b.y) >> 32;
        const u32x4_w v = {(uint32_t)a.x, (uint32_t)a.y, (uint32_t)b.x, (uint32_t)b.y};
        dst[i] = v;
    }
    const int64_t tail = (quads << 2) + i;  // the last count % 4 values (threads 0..2 of the grid)
    if (i < 4 && tail < count) {
        const unsigned long long v = (unsigned long long)rows[tail];
        high |= v >> 32;
        wire[tail] = (uint32_t)v;
    }
    if (overflow && high) atomicOr(overflow, 1ull);
}

// wire[0..count) -> rows[0..count), zero-extended.
__global__ void __launch_bounds__(256) rows_widen_kernel(const uint32_t *__restrict__ wire, int64_t count,
                                                         int64_t *__restrict__ rows)
{
    const int64_t quads = count >> 2;
    const u32x4_w *src = reinterpret_cast<const u32x4_w *>(wire);
    u64x2_w *dst = reinterpret_cast<u64x2_w *>(rows);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < quads) {
        const u32x4_w v = src[i];
        const u64x2_w a = {v.x, v.y}, b = {v.z, v.w};
        dst[2 * i] = a;
        dst[2 * i + 1] = b;
    }
    const int64_t tail = (quads << 2) + i;
    if (i < 4 && tail < count) rows[tail] = (int64_t)wire[tail];
}

// The same two streams one element per lane, for blocks whose address is not 16-byte aligned (a slice that starts at
// an odd element of a larger tensor).
__global__ void __launch_bounds__(256) rows_narrow_scalar_kernel(const int64_t *__restrict__ rows, int64_t count,
                                                                 uint32_t *__restrict__ wire,
                                                                 unsigned long long *__restrict__ overflow)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    unsigned long long high = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) {
        const unsigned long long v = (unsigned long long)rows[i];
        high |= v >> 32;
        wire[i] = (uint32_t)v;
    }
    if (overflow && high) atomicOr(overflow, 1ull);
}

__global__ void __launch_bounds__(256) rows_widen_scalar_kernel(const uint32_t *__restrict__ wire, int64_t count,
                                                                int64_t *__restrict__ rows)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) rows[i] = (int64_t)wire[i];
}

}  // namespace
