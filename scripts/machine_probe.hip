// Machine probe for the walk kernels' roofline: what does one MI355X sustain for
//   (1) a streaming device copy (the "achievable HBM bandwidth" SURVEY §8(d) asks to quote next to the 8 TB/s peak), and
//   (2) divergent dependent gathers of 4/8/16 bytes over footprints from cache-resident to many GB, at the
//       occupancy of the walk kernels (one gather in flight per lane, 8 waves per SIMD) and with more in flight.
// The walk kernels are bound by (2): DESIGN.md §4 quotes their L2-miss rate against the ceiling printed here.
//
//   hipcc --offload-arch=gfx950 -O3 -o machine_probe scripts/machine_probe.hip && ./machine_probe
//
// Measurement tooling only: not part of libtrw_hip.so.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x)                                                                         \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

__global__ void fill_kernel(uint32_t *buf, size_t n_words)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += stride) buf[i] = mix((uint32_t)i);
}

// Every lane runs kIlp independent chains; the next address of a chain depends on the word just loaded.
template <int kBytes, int kIlp>
__global__ void __launch_bounds__(256) gather_kernel(const uint32_t *__restrict__ buf, uint64_t slot_mask, int iters,
                                                      uint32_t *__restrict__ sink)
{
    uint32_t s[kIlp];
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int k = 0; k < kIlp; ++k) s[k] = mix(gid * kIlp + k + 1u);
    for (int it = 0; it < iters; ++it) {
        uint32_t got[kIlp];
#pragma unroll
        for (int k = 0; k < kIlp; ++k) {
            // 40-bit slot index from two mixes so that footprints beyond 2^32 slots would work too
            const uint64_t slot = (((uint64_t)mix(s[k] ^ 0x9e3779b9u) << 32) | mix(s[k])) & slot_mask;
            const uint32_t *p = buf + slot * (kBytes / 4);
            if (kBytes == 4) {
                got[k] = *p;
            } else if (kBytes == 8) {
                const u32x2 v = *reinterpret_cast<const u32x2 *>(p);
                got[k] = v.x ^ v.y;
            } else if (kBytes == 16) {
                const u32x4 v = *reinterpret_cast<const u32x4 *>(p);
                got[k] = v.x ^ v.y ^ v.z ^ v.w;
            } else {  // 32 bytes: two 16-byte loads of one aligned 32-byte record (the biased walk's fat entries)
                const u32x4 v = *reinterpret_cast<const u32x4 *>(p), w = *reinterpret_cast<const u32x4 *>(p + 4);
                got[k] = v.x ^ v.y ^ v.z ^ v.w ^ w.x ^ w.y ^ w.z ^ w.w;
            }
        }
#pragma unroll
        for (int k = 0; k < kIlp; ++k) s[k] = s[k] * 0x2c1b3c6du + got[k] + (uint32_t)it;
    }
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < kIlp; ++k) acc ^= s[k];
    if (acc == 0x12345678u) sink[0] = acc;  // keeps the chains alive
}

template <int kBytes, int kIlp>
static double run_gather(const uint32_t *buf, size_t footprint_bytes, int blocks, int iters, uint32_t *sink)
{
    const uint64_t slots = footprint_bytes / kBytes;  // power of two
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    gather_kernel<kBytes, kIlp><<<blocks, 256>>>(buf, slots - 1, iters / 4 + 1, sink);  // warm-up
    CHECK(hipEventRecord(a));
    gather_kernel<kBytes, kIlp><<<blocks, 256>>>(buf, slots - 1, iters, sink);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    CHECK(hipEventDestroy(a));
    CHECK(hipEventDestroy(b));
    return (double)blocks * 256.0 * iters * kIlp / (ms * 1e-3);
}

// float4 copy kernels (the yardstick MI355X_MICROARCH.md quotes: 6.29 TB/s read+write): every thread moves `per`
// float4 per trip, the trips of a block are contiguous; plain or non-temporal accesses.
typedef float f32x4_p __attribute__((ext_vector_type(4)));
template <int kPer, bool kNt>
__global__ void __launch_bounds__(256) copy_kernel(const float4 *__restrict__ src_, float4 *__restrict__ dst_, size_t n16)
{
    const f32x4_p *__restrict__ src = reinterpret_cast<const f32x4_p *>(src_);
    f32x4_p *__restrict__ dst = reinterpret_cast<f32x4_p *>(dst_);
    const size_t stride = (size_t)gridDim.x * 256 * kPer;
    for (size_t base = (size_t)blockIdx.x * 256 * kPer + threadIdx.x; base < n16; base += stride) {
        f32x4_p v[kPer];
#pragma unroll
        for (int k = 0; k < kPer; ++k)
            if (base + (size_t)k * 256 < n16)
                v[k] = kNt ? __builtin_nontemporal_load(src + base + (size_t)k * 256) : src[base + (size_t)k * 256];
#pragma unroll
        for (int k = 0; k < kPer; ++k)
            if (base + (size_t)k * 256 < n16) {
                if (kNt) __builtin_nontemporal_store(v[k], dst + base + (size_t)k * 256);
                else dst[base + (size_t)k * 256] = v[k];
            }
    }
}

// write-only stream (what the window builders' outputs can hope for): 16 or 8 bytes per thread, one trip per thread
template <int kBytes>
__global__ void __launch_bounds__(256) write_kernel(unsigned long long *__restrict__ dst, size_t n8, unsigned long long v)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (kBytes == 16) {
        typedef unsigned long long u64x2_p __attribute__((ext_vector_type(2)));
        if (2 * i + 1 < n8) reinterpret_cast<u64x2_p *>(dst)[i] = u64x2_p{v + i, v};
    } else {
        if (i < n8) dst[i] = v + i;
    }
}

template <int kBytes>
static double time_write(void *dst, size_t bytes, int reps)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    const size_t n8 = bytes / 8, threads = kBytes == 16 ? n8 / 2 : n8;
    const unsigned blocks = (unsigned)((threads + 255) / 256);
    write_kernel<kBytes><<<blocks, 256>>>((unsigned long long *)dst, n8, 1ull);
    CHECK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) write_kernel<kBytes><<<blocks, 256>>>((unsigned long long *)dst, n8, 2ull + i);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    CHECK(hipEventDestroy(a));
    CHECK(hipEventDestroy(b));
    return ms / reps;
}

template <int kPer, bool kNt>
static double time_copy(const void *src, void *dst, size_t bytes, int blocks, int reps)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    copy_kernel<kPer, kNt><<<blocks, 256>>>((const float4 *)src, (float4 *)dst, bytes / 16);
    CHECK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) copy_kernel<kPer, kNt><<<blocks, 256>>>((const float4 *)src, (float4 *)dst, bytes / 16);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    CHECK(hipEventDestroy(a));
    CHECK(hipEventDestroy(b));
    return ms / reps;
}

int main(int argc, char **argv)
{
    const size_t max_mb = argc > 1 ? strtoull(argv[1], nullptr, 10) : 16384;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("# device: %s, %d CUs\n", prop.name, prop.multiProcessorCount);

    // ---- (0) "one <footprint MB> <bytes>": a single launch with a known number of gathers, to be run under
    //      rocprofv3 --pmc FETCH_SIZE / TCC_MISS_sum (scripts/probe_fetch_granule.sh): FETCH_SIZE*1024/gathers is
    //      what ONE L2 miss of a random gather moves from HBM on this part (64 or 128 bytes?)
    if (argc > 3 && std::string(argv[1]) == "one") {
        const size_t fp = (size_t)strtoull(argv[2], nullptr, 10) << 20;
        const int bytes = atoi(argv[3]);
        uint32_t *buf, *sink;
        CHECK(hipMalloc(&buf, fp));
        CHECK(hipMalloc(&sink, 256));
        fill_kernel<<<4096, 256>>>(buf, fp / 4);
        CHECK(hipDeviceSynchronize());
        const int blocks = prop.multiProcessorCount * 8, iters = 400;
        double rate = 0;
        if (bytes == 4) rate = run_gather<4, 1>(buf, fp, blocks, iters, sink);
        else if (bytes == 16) rate = run_gather<16, 1>(buf, fp, blocks, iters, sink);
        else rate = run_gather<32, 1>(buf, fp, blocks, iters, sink);
        // run_gather launches a warm-up of iters/4+1 and the timed launch of iters: the LARGER dispatch is the timed one
        printf("one: footprint %zu MB, %d-byte gathers, timed launch = %.0f gathers, %.3e gathers/s\n", fp >> 20, bytes,
               (double)blocks * 256.0 * iters, rate);
        return 0;
    }

    // ---- (1) streaming copy ----
    {
        const size_t bytes = (size_t)4 << 30;
        void *src, *dst;
        CHECK(hipMalloc(&src, bytes));
        CHECK(hipMalloc(&dst, bytes));
        CHECK(hipMemset(src, 1, bytes));
        CHECK(hipMemset(dst, 2, bytes));
        hipEvent_t a, b;
        CHECK(hipEventCreate(&a));
        CHECK(hipEventCreate(&b));
        CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0));
        CHECK(hipEventRecord(a));
        const int reps = 10;
        for (int i = 0; i < reps; ++i) CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0));
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        printf("copy_variant hipMemcpyAsync(D2D): %.3f ms per copy, %.0f GB/s read+write\n", ms / reps,
               2.0 * bytes * reps / (ms * 1e-3) / 1e9);
        double best = ms / reps;
        const char *best_name = "hipMemcpyAsync";
        // hand-written float4 copies: blocks per CU x float4 per thread and trip x plain / non-temporal
        const int cus = prop.multiProcessorCount;
        struct { const char *name; double ms; } v[] = {
            {"float4 x1, 8 blocks/CU", time_copy<1, false>(src, dst, bytes, cus * 8, reps)},
            {"float4 x1, 32 blocks/CU", time_copy<1, false>(src, dst, bytes, cus * 32, reps)},
            {"float4 x4, 8 blocks/CU", time_copy<4, false>(src, dst, bytes, cus * 8, reps)},
            {"float4 x4, 16 blocks/CU", time_copy<4, false>(src, dst, bytes, cus * 16, reps)},
            {"float4 x1 nt, 8 blocks/CU", time_copy<1, true>(src, dst, bytes, cus * 8, reps)},
            {"float4 x4 nt, 8 blocks/CU", time_copy<4, true>(src, dst, bytes, cus * 8, reps)},
            {"float4 x4 nt, 16 blocks/CU", time_copy<4, true>(src, dst, bytes, cus * 16, reps)},
            {"float4 x8 nt, 8 blocks/CU", time_copy<8, true>(src, dst, bytes, cus * 8, reps)},
            {"float4 x1, one trip per thread", time_copy<1, false>(src, dst, bytes, (int)(bytes / 16 / 256), reps)},
            {"float4 x4, one trip per thread", time_copy<4, false>(src, dst, bytes, (int)(bytes / 16 / 1024), reps)},
            {"float4 x4 nt, one trip per thread", time_copy<4, true>(src, dst, bytes, (int)(bytes / 16 / 1024), reps)},
        };
        for (auto &x : v) {
            printf("copy_variant %s: %.3f ms per copy, %.0f GB/s read+write\n", x.name, x.ms, 2.0 * bytes / (x.ms * 1e-3) / 1e9);
            if (x.ms < best) {
                best = x.ms;
                best_name = x.name;
            }
        }
        const double w16 = time_write<16>(dst, bytes, reps), w8 = time_write<8>(dst, bytes, reps);
        printf("stream_write: 4 GiB, 16 B per thread %.3f ms = %.0f GB/s; 8 B per thread %.3f ms = %.0f GB/s\n", w16,
               bytes / (w16 * 1e-3) / 1e9, w8, bytes / (w8 * 1e-3) / 1e9);
        printf("stream_copy: 4 GiB device-to-device, %.3f ms per copy, %.0f GB/s read+write (best: %s)\n", best,
               2.0 * bytes / (best * 1e-3) / 1e9, best_name);
        CHECK(hipFree(src));
        CHECK(hipFree(dst));
    }

    // ---- (2) dependent divergent gathers ----
    const size_t max_bytes = max_mb << 20;
    uint32_t *buf, *sink;
    CHECK(hipMalloc(&buf, max_bytes));
    CHECK(hipMalloc(&sink, 256));
    fill_kernel<<<4096, 256>>>(buf, max_bytes / 4);
    CHECK(hipDeviceSynchronize());
    const int blocks = prop.multiProcessorCount * 8;  // 8 waves per SIMD, as the walk kernels run
    printf("# gathers/s by footprint; lanes=%d (8 waves/SIMD); bytes per gather x chains in flight per lane\n",
           blocks * 256);
    printf("%10s %12s %12s %12s %12s %12s %12s %12s\n", "footprint", "4Bx1", "8Bx1", "16Bx1", "16Bx2", "16Bx4", "4Bx4",
           "32Bx1");
    for (size_t mb = 4; mb <= max_mb; mb *= 4) {
        const size_t fp = mb << 20;
        const int iters = 400;
        const double r41 = run_gather<4, 1>(buf, fp, blocks, iters, sink);
        const double r81 = run_gather<8, 1>(buf, fp, blocks, iters, sink);
        const double r161 = run_gather<16, 1>(buf, fp, blocks, iters, sink);
        const double r162 = run_gather<16, 2>(buf, fp, blocks, iters / 2, sink);
        const double r164 = run_gather<16, 4>(buf, fp, blocks, iters / 4, sink);
        const double r44 = run_gather<4, 4>(buf, fp, blocks, iters / 4, sink);
        const double r321 = run_gather<32, 1>(buf, fp, blocks, iters, sink);
        printf("%8zu MB %12.3e %12.3e %12.3e %12.3e %12.3e %12.3e %12.3e\n", mb, r41, r81, r161, r162, r164, r44, r321);
        fflush(stdout);
    }
    CHECK(hipFree(buf));
    CHECK(hipFree(sink));
    return 0;
}
