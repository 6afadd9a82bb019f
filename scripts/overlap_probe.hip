// A stand-in for RCCL's point-to-point kernels: a copy kernel with a small, fixed grid (a few dozen
// workgroups, as a send/recv group uses), to check on ONE GPU whether such a kernel gets wave slots while
// the persistent biased walk kernel holds the whole chip (scripts/overlap_probe.py).  Measurement tooling only.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o /tmp/liboverlap_probe.so scripts/overlap_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>

// eight 16-byte loads in flight per thread, as a collective's copy loops keep
__global__ void __launch_bounds__(512) probe_copy_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16)
{
    extern __shared__ uint32_t lds[];  // dynamic: the launcher asks for as much LDS per workgroup as it wants to model
    if (n16 == ~(size_t)0) lds[threadIdx.x] = 1u;  // (keeps the allocation alive; never true)
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 7 * stride < n16; i += 8 * stride) {
        uint4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = src[i + k * stride];
#pragma unroll
        for (int k = 0; k < 8; ++k) dst[i + k * stride] = v[k];
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

// lds_bytes: LDS each workgroup claims (a collective's kernel holds some; how much decides whether it fits beside
// a walk kernel that has most of a CU's 160 KB)
extern "C" int probe_copy_lds(const void *src, void *dst, size_t bytes, int blocks, int lds_bytes, void *stream)
{
    probe_copy_kernel<<<blocks, 512, lds_bytes, (hipStream_t)stream>>>((const uint4 *)src, (uint4 *)dst, bytes / 16);
    return (int)hipGetLastError();
}

extern "C" int probe_copy(const void *src, void *dst, size_t bytes, int blocks, void *stream)
{
    return probe_copy_lds(src, dst, bytes, blocks, 0, stream);
}
