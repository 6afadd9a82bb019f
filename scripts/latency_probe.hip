// Dependent-load latency of ONE wave on an otherwise idle MI355X: what a chain of gathers that cannot overlap costs per link.
// (The serial replay of the reference's biased stream, biased_glibc_wave_kernel, has one such link per walk step.)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/latency_probe scripts/latency_probe.hip && /tmp/latency_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// every lane follows its own chain through `next` (a random cyclic permutation in units of `stride` 32-byte entries)
__global__ void __launch_bounds__(64) chase(const uint4 *__restrict__ ent, uint32_t start_stride, int links, int lanes, uint32_t *out,
                                            unsigned long long *cycles)
{
    uint32_t at = threadIdx.x * start_stride;
    const bool live = (int)threadIdx.x < lanes;
    const unsigned long long t0 = clock64();
    for (int i = 0; i < links; ++i) {
        if (live) {
            const uint4 a = ent[2 * (size_t)at], b = ent[2 * (size_t)at + 1];  // a 32-byte entry, as the walk's
            at = a.x ^ (b.x & 0u);
        }
    }
    const unsigned long long t1 = clock64();
    out[threadIdx.x] = at;
    if (threadIdx.x == 0) *cycles = t1 - t0;
}

int main()
{
    for (size_t mb : {16, 256, 4096}) {
        const size_t n = mb * 1024 * 1024 / 32;  // 32-byte entries
        std::vector<uint32_t> perm(n);
        for (size_t i = 0; i < n; ++i) perm[i] = (uint32_t)i;
        uint64_t s = 88172645463325252ull;
        for (size_t i = n - 1; i > 0; --i) {  // Sattolo: one cycle through every entry
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            const size_t j = s % i;
            std::swap(perm[i], perm[j]);
        }
        std::vector<uint32_t> host(n * 8, 0u);
        for (size_t i = 0; i < n; ++i) host[8 * i] = perm[i];
        uint4 *ent;
        uint32_t *out;
        unsigned long long *cyc;
        CHECK(hipMalloc(&ent, n * 32));
        CHECK(hipMalloc(&out, 256));
        CHECK(hipMalloc(&cyc, 8));
        CHECK(hipMemcpy(ent, host.data(), n * 32, hipMemcpyHostToDevice));
        for (int lanes : {1, 64}) {
            const int links = 20000;
            hipEvent_t a, b;
            CHECK(hipEventCreate(&a));
            CHECK(hipEventCreate(&b));
            chase<<<1, 64>>>(ent, (uint32_t)(n / 64), 1000, lanes, out, cyc);  // warm-up
            CHECK(hipEventRecord(a));
            chase<<<1, 64>>>(ent, (uint32_t)(n / 64), links, lanes, out, cyc);
            CHECK(hipEventRecord(b));
            CHECK(hipDeviceSynchronize());
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, a, b));
            unsigned long long c = 0;
            CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
            printf("footprint %5zu MB, %2d lane(s) of one wave: %7.1f ns per dependent 32-byte gather (%.0f shader clocks)\n", mb, lanes,
                   ms * 1e6 / links, (double)c / links);
        }
        CHECK(hipFree(ent));
        CHECK(hipFree(out));
        CHECK(hipFree(cyc));
    }
    return 0;
}
