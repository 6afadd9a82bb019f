// Issue cost of the integer instructions the walk kernels are made of, on this GPU (measurement tooling only).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/alu_probe scripts/alu_probe.hip && /tmp/alu_probe
// Each kernel runs a long unrolled run of ONE instruction (four independent chains per lane, so latency is covered)
// with 1, 2 and 4 waves per SIMD resident on every CU; prints shader cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kIters = 4096, kUnroll = 8;  // instructions per lane = kIters * kUnroll * 4

enum Op { OP_XOR = 0, OP_ADD, OP_MAD64, OP_MUL_LO, OP_MUL_HI, OP_MUL24, OP_CNDMASK, OP_BFE, OP_LSHL_ADD, OP_MIN, OP_RCP, OP_CND_SGPR, OP_CMP_VCC, OP_CMP_SGPR, OP_CMP_CND, OP_CND_INDEP, OP_ADDC, OP_MED3, OP_BFI, OP_COUNT };
static const char *kNames[OP_COUNT] = {"v_xor_b32", "v_add_u32", "v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24",
                                       "v_cndmask_b32", "v_bfe_u32", "v_lshl_add_u32", "v_min_u32", "v_rcp_f32", "v_cndmask(sgpr)",
                                       "v_cmp->vcc", "v_cmp->sgpr", "v_cmp+v_cndmask", "v_cndmask(indep)", "v_addc_co_u32", "v_med3_u32",
                                       "v_bfi_b32"};

template <int kOp>
__global__ void __launch_bounds__(256) alu_kernel(uint32_t *out, uint32_t seed, unsigned long long *cycles)
{
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9E3779B9u, c = a + 77u, d = b + 1234567u;
    const uint32_t m = seed | 0xD2511F53u;
    uint64_t wa = a, wb = b, wc = c, wd = d;
    const unsigned long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < kIters; ++i) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            if (kOp == OP_XOR) {
                asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            } else if (kOp == OP_ADD) {
                asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            } else if (kOp == OP_MAD64) {
                asm volatile("v_mad_u64_u32 %0, s[10:11], %4, %5, 0\n v_mad_u64_u32 %1, s[10:11], %4, %6, 0\n"
                             "v_mad_u64_u32 %2, s[10:11], %4, %7, 0\n v_mad_u64_u32 %3, s[10:11], %4, %8, 0"
                             : "+v"(wa), "+v"(wb), "+v"(wc), "+v"(wd) : "v"(m), "v"(a), "v"(b), "v"(c), "v"(d) : "s10", "s11");
                a ^= (uint32_t)(wa >> 32);  // (keeps the products live; one xor per four multiplies)
            } else if (kOp == OP_MUL_LO) {
                asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            } else if (kOp == OP_MUL_HI) {
                asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            } else if (kOp == OP_MUL24) {
                asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            } else if (kOp == OP_CNDMASK) {
                asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n"
                             "v_cndmask_b32 %3, %3, %4, vcc"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "vcc");
            } else if (kOp == OP_BFE) {
                asm volatile("v_bfe_u32 %0, %0, %4, 31\n v_bfe_u32 %1, %1, %4, 31\n v_bfe_u32 %2, %2, %4, 31\n v_bfe_u32 %3, %3, %4, 31"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m & 1u));
            } else if (kOp == OP_LSHL_ADD) {
                asm volatile("v_lshl_add_u32 %0, %0, 1, %4\n v_lshl_add_u32 %1, %1, 1, %4\n v_lshl_add_u32 %2, %2, 1, %4\n"
                             "v_lshl_add_u32 %3, %3, 1, %4"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            } else if (kOp == OP_MIN) {
                asm volatile("v_min_u32 %0, %0, %4\n v_min_u32 %1, %1, %4\n v_min_u32 %2, %2, %4\n v_min_u32 %3, %3, %4"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            } else if (kOp == OP_CND_SGPR) {
                asm volatile("v_cndmask_b32_e64 %0, %0, %4, s[10:11]\n v_cndmask_b32_e64 %1, %1, %4, s[10:11]\n"
                             "v_cndmask_b32_e64 %2, %2, %4, s[10:11]\n v_cndmask_b32_e64 %3, %3, %4, s[10:11]"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "s10", "s11");
            } else if (kOp == OP_CMP_VCC) {
                asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_cmp_lt_u32 vcc, %1, %4\n v_cmp_lt_u32 vcc, %2, %4\n v_cmp_lt_u32 vcc, %3, %4"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "vcc");
            } else if (kOp == OP_CMP_SGPR) {
                asm volatile("v_cmp_lt_u32_e64 s[10:11], %0, %4\n v_cmp_lt_u32_e64 s[12:13], %1, %4\n"
                             "v_cmp_lt_u32_e64 s[14:15], %2, %4\n v_cmp_lt_u32_e64 s[16:17], %3, %4"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17");
            } else if (kOp == OP_CMP_CND) {  // two instructions per count of two: the pair a select compiles to
                asm volatile("v_cmp_lt_u32 vcc, %0, %4\n s_nop 1\n v_cndmask_b32 %0, %0, %4, vcc\n"
                             "v_cmp_lt_u32 vcc, %1, %4\n s_nop 1\n v_cndmask_b32 %1, %1, %4, vcc"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "vcc");
            } else if (kOp == OP_CND_INDEP) {
                asm volatile("v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %4, %5, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n"
                             "v_cndmask_b32 %3, %4, %5, vcc"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(seed) : "vcc");
            } else if (kOp == OP_ADDC) {
                asm volatile("v_addc_co_u32 %0, vcc, %0, %4, vcc\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n"
                             "v_addc_co_u32 %2, vcc, %2, %4, vcc\n v_addc_co_u32 %3, vcc, %3, %4, vcc"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "vcc");
            } else if (kOp == OP_MED3) {
                asm volatile("v_med3_u32 %0, %0, %4, %1\n v_med3_u32 %1, %1, %4, %2\n v_med3_u32 %2, %2, %4, %3\n"
                             "v_med3_u32 %3, %3, %4, %0"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            } else if (kOp == OP_BFI) {
                asm volatile("v_bfi_b32 %0, %4, %0, %1\n v_bfi_b32 %1, %4, %1, %2\n v_bfi_b32 %2, %4, %2, %3\n"
                             "v_bfi_b32 %3, %4, %3, %0"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            } else if (kOp == OP_RCP) {
                asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            }
        }
    }
    const unsigned long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) atomicMax(cycles, t1 - t0);
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ (uint32_t)wa ^ (uint32_t)wb ^ (uint32_t)wc ^ (uint32_t)wd;
}

template <int kOp>
static void run(uint32_t *out, unsigned long long *cycles, int cus)
{
    for (int waves_per_simd = 1; waves_per_simd <= 4; waves_per_simd *= 2) {
        const int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
        CHECK(hipMemset(cycles, 0, sizeof(*cycles)));
        alu_kernel<kOp><<<blocks, 256>>>(out, 1u, cycles);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemset(cycles, 0, sizeof(*cycles)));
        alu_kernel<kOp><<<blocks, 256>>>(out, 2u, cycles);
        CHECK(hipDeviceSynchronize());
        unsigned long long c = 0;
        CHECK(hipMemcpy(&c, cycles, sizeof(c), hipMemcpyDeviceToHost));
        const double per_wave = (double)kIters * kUnroll * 4;  // instructions one wave issued
        printf("%-16s %d wave(s)/SIMD: %6.2f cycles per instruction per wave, %6.2f per instruction per SIMD\n", kNames[kOp],
               waves_per_simd, c / per_wave, c / (per_wave * waves_per_simd));
    }
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t *out;
    unsigned long long *cycles;
    CHECK(hipMalloc(&out, (size_t)cus * 4 * 256 * 4));
    CHECK(hipMalloc(&cycles, sizeof(*cycles)));
    printf("%s, %d CUs; %d instructions per lane and kernel (clock64 = shader clock)\n", prop.name, cus, kIters * kUnroll * 4);
    run<OP_XOR>(out, cycles, cus);
    run<OP_ADD>(out, cycles, cus);
    run<OP_CNDMASK>(out, cycles, cus);
    run<OP_BFE>(out, cycles, cus);
    run<OP_LSHL_ADD>(out, cycles, cus);
    run<OP_MIN>(out, cycles, cus);
    run<OP_MUL24>(out, cycles, cus);
    run<OP_MUL_LO>(out, cycles, cus);
    run<OP_MUL_HI>(out, cycles, cus);
    run<OP_MAD64>(out, cycles, cus);
    run<OP_RCP>(out, cycles, cus);
    run<OP_CND_SGPR>(out, cycles, cus);
    run<OP_CND_INDEP>(out, cycles, cus);
    run<OP_CMP_VCC>(out, cycles, cus);
    run<OP_CMP_SGPR>(out, cycles, cus);
    run<OP_CMP_CND>(out, cycles, cus);
    run<OP_ADDC>(out, cycles, cus);
    run<OP_MED3>(out, cycles, cus);
    run<OP_BFI>(out, cycles, cus);
    return 0;
}
