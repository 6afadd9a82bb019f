// What does a walk kernel's output cost in HBM traffic?  (measurement tooling only)
// The biased walk writes its [N, L+1] int64 rows one 8-byte column at a time, each lane into a different row, a row's next
// column a fraction of a millisecond later.  This probe reproduces that pattern without the walk: `walkers` rows of
// `cols` int64, every resident lane owns one row at a time and stores its columns in order, with `gap` dependent
// random 16-byte gathers over a 4 GB buffer between two stores of a lane (the walk's own gathers: they are what pushes a
// row's line out of L2 before its next column arrives), in one of these ways:
//     mode 0   plain 8-byte stores                         (what biased_pool_kernel does)
//     mode 1   8-byte nontemporal stores
//     mode 2   16-byte stores (pairs of columns; rows are 8-byte aligned)
//     mode 3   whole aligned 64-byte lines, assembled by the lane (8 columns at a time; head and tail of a row as 8-byte stores)
//     mode 4   as 3, nontemporal
// Run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (scripts/probe_store_traffic.sh): bytes fetched and written per row byte.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_probe scripts/store_probe.hip && /tmp/store_probe <mode> [walkers] [cols] [gap]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef int64_t i64x2_a8 __attribute__((ext_vector_type(2), aligned(8)));

__device__ const uint4 *g_table;
__device__ uint32_t g_mask;

__device__ __forceinline__ uint32_t spin(uint32_t x, int gap)
{
    for (int i = 0; i < gap; ++i) {
        x = x * 1664525u + 1013904223u;
        const uint4 v = g_table[(x ^ (x >> 13)) & g_mask];
        x ^= v.x + v.w;
    }
    return x;
}

template <int kMode>
__global__ void __launch_bounds__(256) store_kernel(int64_t *__restrict__ out, int64_t walkers, int cols, int gap,
                                                    unsigned long long *__restrict__ next)
{
    const int lane = threadIdx.x & 63;
    uint32_t x = threadIdx.x + blockIdx.x * 977u;
    while (true) {
        long long base = 0;
        if (lane == 0) base = (long long)atomicAdd(next, 64ull);
        base = __shfl(base, 0);
        if (base >= walkers) break;
        const int64_t w = base + lane;
        if (w >= walkers) continue;
        int64_t *row = out + w * cols;
        if (kMode <= 1) {
            for (int c = 0; c < cols; ++c) {
                x = spin(x, gap);
                if (kMode == 0) row[c] = (int64_t)x;
                else __builtin_nontemporal_store((int64_t)x, row + c);
            }
        } else if (kMode == 2) {
            int c = 0;
            for (; c + 1 < cols; c += 2) {
                x = spin(x, 2 * gap);
                i64x2_a8 two = {(int64_t)x, (int64_t)(x + 1u)};
                *reinterpret_cast<i64x2_a8 *>(row + c) = two;
            }
            if (c < cols) row[c] = (int64_t)x;
        } else {
            int c = 0;
            // head: up to the first 64-byte boundary
            while (c < cols && (((uintptr_t)(row + c)) & 63u) != 0) {
                x = spin(x, gap);
                row[c] = (int64_t)x;
                ++c;
            }
            for (; c + 8 <= cols; c += 8) {
                x = spin(x, 8 * gap);
                int64_t *p = row + c;
#pragma unroll
                for (int k = 0; k < 8; k += 2) {
                    i64x2_a8 two = {(int64_t)(x + k), (int64_t)(x + k + 1)};
                    if (kMode == 3) *reinterpret_cast<i64x2_a8 *>(p + k) = two;
                    else __builtin_nontemporal_store(two, reinterpret_cast<i64x2_a8 *>(p + k));
                }
            }
            for (; c < cols; ++c) {
                x = spin(x, gap);
                row[c] = (int64_t)x;
            }
        }
    }
    if (x == 0x12345678u) out[0] = 1;
}

int main(int argc, char **argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const int64_t walkers = argc > 2 ? atoll(argv[2]) : 10000000;
    const int cols = argc > 3 ? atoi(argv[3]) : 81;
    const int gap = argc > 4 ? atoi(argv[4]) : 2;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int64_t *out;
    unsigned long long *next;
    const size_t bytes = (size_t)walkers * cols * 8;
    CHECK(hipMalloc(&out, bytes));
    CHECK(hipMalloc(&next, 8));
    CHECK(hipMemset(out, 0, bytes));
    uint4 *table;
    const size_t table_bytes = (size_t)4 << 30;
    CHECK(hipMalloc(&table, table_bytes));
    CHECK(hipMemset(table, 1, table_bytes));
    const uint32_t mask = (uint32_t)(table_bytes / 16 - 1);
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_table), &table, sizeof(table)));
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_mask), &mask, sizeof(mask)));
    const int blocks = prop.multiProcessorCount * 4;  // 16 waves per CU, as the walker pool has
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipMemset(next, 0, 8));
        CHECK(hipEventRecord(a));
        switch (mode) {
        case 0: store_kernel<0><<<blocks, 256>>>(out, walkers, cols, gap, next); break;
        case 1: store_kernel<1><<<blocks, 256>>>(out, walkers, cols, gap, next); break;
        case 2: store_kernel<2><<<blocks, 256>>>(out, walkers, cols, gap, next); break;
        case 3: store_kernel<3><<<blocks, 256>>>(out, walkers, cols, gap, next); break;
        default: store_kernel<4><<<blocks, 256>>>(out, walkers, cols, gap, next); break;
        }
        CHECK(hipEventRecord(b));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventElapsedTime(&ms, a, b));
    }
    printf("store: mode %d, %lld rows x %d columns = %.3f GB, %d gathers (%.3f GB of 64-byte lines) between stores: %.2f ms per launch "
           "(two launches)\n", mode, (long long)walkers, cols, bytes / 1e9, gap, (double)walkers * cols * gap * 64 / 1e9, ms);
    return 0;
}
