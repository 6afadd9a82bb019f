// Issue cost of more VALU instructions of the walk kernels on this GPU (measurement tooling; extends alu_probe.hip).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/alu_probe2 scripts/alu_probe2.hip && /tmp/alu_probe2
// One kernel per instruction: a long unrolled
// run of it on four independent chains per lane, 1 / 2 / 4 waves per SIMD on every CU; prints shader-clock cycles
// per wave-instruction per SIMD.  What matters to the kernels: which integer ops are in the ~2.6-cycle class
// (xor/add were) and which in the ~4.4-cycle class (bfe/min/mul/cndmask/cmp were).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kIters = 2048, kUnroll = 8;  // instructions per lane = kIters * kUnroll * 4

template <int kOp>
__global__ void __launch_bounds__(256) alu_kernel(uint32_t *out, uint32_t seed, unsigned long long *cycles)
{
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9E3779B9u, c = a + 77u, d = b + 1234567u;
    const uint32_t m = seed | 0xD2511F53u;
    uint64_t wa = a, wb = b, wc = c, wd = d;
    asm volatile("s_mov_b32 s20, 0x12345\n s_mov_b32 s10, 0x55555555\n s_mov_b32 s11, 0x55555555" ::: "s20", "s10", "s11");
    const unsigned long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < kIters; ++i) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            if (kOp == 0) asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 1) asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 2) asm volatile("v_sub_u32 %0, %0, %4\n v_sub_u32 %1, %1, %4\n v_sub_u32 %2, %2, %4\n v_sub_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 3) asm volatile("v_subrev_u32 %0, %0, %4\n v_subrev_u32 %1, %1, %4\n v_subrev_u32 %2, %2, %4\n v_subrev_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 4) asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 5) asm volatile("v_or_b32 %0, %0, %4\n v_or_b32 %1, %1, %4\n v_or_b32 %2, %2, %4\n v_or_b32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 6) asm volatile("v_xnor_b32 %0, %0, %4\n v_xnor_b32 %1, %1, %4\n v_xnor_b32 %2, %2, %4\n v_xnor_b32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 7) asm volatile("v_lshrrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1\n v_lshrrev_b32 %2, 1, %2\n v_lshrrev_b32 %3, 1, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 8) asm volatile("v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 9) asm volatile("v_lshrrev_b32 %0, %4, %0\n v_lshrrev_b32 %1, %4, %1\n v_lshrrev_b32 %2, %4, %2\n v_lshrrev_b32 %3, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 10) asm volatile("v_ashrrev_i32 %0, 1, %0\n v_ashrrev_i32 %1, 1, %1\n v_ashrrev_i32 %2, 1, %2\n v_ashrrev_i32 %3, 1, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 11) asm volatile("v_max_u32 %0, %0, %4\n v_max_u32 %1, %1, %4\n v_max_u32 %2, %2, %4\n v_max_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 12) asm volatile("v_min_u32 %0, %0, %4\n v_min_u32 %1, %1, %4\n v_min_u32 %2, %2, %4\n v_min_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 13) asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 14) asm volatile("v_not_b32 %0, %0\n v_not_b32 %1, %1\n v_not_b32 %2, %2\n v_not_b32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 15) asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_add_co_u32 %1, vcc, %1, %4\n v_add_co_u32 %2, vcc, %2, %4\n v_add_co_u32 %3, vcc, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "vcc", "s10", "s11");
            if (kOp == 16) asm volatile("v_bfe_u32 %0, %0, %4, 31\n v_bfe_u32 %1, %1, %4, 31\n v_bfe_u32 %2, %2, %4, 31\n v_bfe_u32 %3, %3, %4, 31" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 17) asm volatile("v_bfe_i32 %0, %0, %4, 31\n v_bfe_i32 %1, %1, %4, 31\n v_bfe_i32 %2, %2, %4, 31\n v_bfe_i32 %3, %3, %4, 31" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 18) asm volatile("v_bfi_b32 %0, %4, %0, %1\n v_bfi_b32 %1, %4, %1, %2\n v_bfi_b32 %2, %4, %2, %3\n v_bfi_b32 %3, %4, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 19) asm volatile("v_alignbit_b32 %0, %0, %4, 7\n v_alignbit_b32 %1, %1, %4, 7\n v_alignbit_b32 %2, %2, %4, 7\n v_alignbit_b32 %3, %3, %4, 7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 20) asm volatile("v_perm_b32 %0, %0, %4, %1\n v_perm_b32 %1, %1, %4, %2\n v_perm_b32 %2, %2, %4, %3\n v_perm_b32 %3, %3, %4, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 21) asm volatile("v_mad_u32_u24 %0, %0, %4, %1\n v_mad_u32_u24 %1, %1, %4, %2\n v_mad_u32_u24 %2, %2, %4, %3\n v_mad_u32_u24 %3, %3, %4, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 22) asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 23) asm volatile("v_mul_hi_u32_u24 %0, %0, %4\n v_mul_hi_u32_u24 %1, %1, %4\n v_mul_hi_u32_u24 %2, %2, %4\n v_mul_hi_u32_u24 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 24) asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 25) asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 26) asm volatile("v_add3_u32 %0, %0, %4, %1\n v_add3_u32 %1, %1, %4, %2\n v_add3_u32 %2, %2, %4, %3\n v_add3_u32 %3, %3, %4, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 27) asm volatile("v_lshl_or_b32 %0, %0, 3, %4\n v_lshl_or_b32 %1, %1, 3, %4\n v_lshl_or_b32 %2, %2, 3, %4\n v_lshl_or_b32 %3, %3, 3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 28) asm volatile("v_lshl_add_u32 %0, %0, 1, %4\n v_lshl_add_u32 %1, %1, 1, %4\n v_lshl_add_u32 %2, %2, 1, %4\n v_lshl_add_u32 %3, %3, 1, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 29) asm volatile("v_add_lshl_u32 %0, %0, %4, 1\n v_add_lshl_u32 %1, %1, %4, 1\n v_add_lshl_u32 %2, %2, %4, 1\n v_add_lshl_u32 %3, %3, %4, 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 30) asm volatile("v_and_or_b32 %0, %0, %4, %1\n v_and_or_b32 %1, %1, %4, %2\n v_and_or_b32 %2, %2, %4, %3\n v_and_or_b32 %3, %3, %4, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 31) asm volatile("v_or3_b32 %0, %0, %4, %1\n v_or3_b32 %1, %1, %4, %2\n v_or3_b32 %2, %2, %4, %3\n v_or3_b32 %3, %3, %4, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 32) asm volatile("v_xad_u32 %0, %0, %4, %1\n v_xad_u32 %1, %1, %4, %2\n v_xad_u32 %2, %2, %4, %3\n v_xad_u32 %3, %3, %4, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 33) asm volatile("v_med3_u32 %0, %0, %4, %1\n v_med3_u32 %1, %1, %4, %2\n v_med3_u32 %2, %2, %4, %3\n v_med3_u32 %3, %3, %4, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 34) asm volatile("v_mbcnt_lo_u32_b32 %0, %0, %4\n v_mbcnt_lo_u32_b32 %1, %1, %4\n v_mbcnt_lo_u32_b32 %2, %2, %4\n v_mbcnt_lo_u32_b32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 35) asm volatile("v_bcnt_u32_b32 %0, %0, %4\n v_bcnt_u32_b32 %1, %1, %4\n v_bcnt_u32_b32 %2, %2, %4\n v_bcnt_u32_b32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 36) asm volatile("v_ffbl_b32 %0, %0\n v_ffbl_b32 %1, %1\n v_ffbl_b32 %2, %2\n v_ffbl_b32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 37) asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "vcc", "s10", "s11");
            if (kOp == 38) asm volatile("v_cndmask_b32_e64 %0, %0, %4, s[10:11]\n v_cndmask_b32_e64 %1, %1, %4, s[10:11]\n v_cndmask_b32_e64 %2, %2, %4, s[10:11]\n v_cndmask_b32_e64 %3, %3, %4, s[10:11]" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 39) asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_cmp_lt_u32 vcc, %1, %4\n v_cmp_lt_u32 vcc, %2, %4\n v_cmp_lt_u32 vcc, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "vcc", "s10", "s11");
            if (kOp == 40) asm volatile("v_cmp_lt_u32_e64 s[10:11], %0, %4\n v_cmp_lt_u32_e64 s[10:11], %1, %4\n v_cmp_lt_u32_e64 s[10:11], %2, %4\n v_cmp_lt_u32_e64 s[10:11], %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "vcc", "s10", "s11");
            if (kOp == 41) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 42) asm volatile("v_add_u32_sdwa %0, %0, %4 dst_sel:DWORD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %1, %1, %4 dst_sel:DWORD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %2, %2, %4 dst_sel:DWORD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %3, %3, %4 dst_sel:DWORD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 43) asm volatile("v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %1, %1, %4\n v_pk_add_u16 %2, %2, %4\n v_pk_add_u16 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 44) asm volatile("v_dot4_u32_u8 %0, %0, %4, %1\n v_dot4_u32_u8 %1, %1, %4, %2\n v_dot4_u32_u8 %2, %2, %4, %3\n v_dot4_u32_u8 %3, %3, %4, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 45) asm volatile("v_xor_b32 %0, s20, %0\n v_xor_b32 %1, s20, %1\n v_xor_b32 %2, s20, %2\n v_xor_b32 %3, s20, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 46) asm volatile("v_xor_b32 %0, 0x9E3779B9, %0\n v_xor_b32 %1, 0x9E3779B9, %1\n v_xor_b32 %2, 0x9E3779B9, %2\n v_xor_b32 %3, 0x9E3779B9, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 47) asm volatile("v_add_u32 %0, 0x9E3779B9, %0\n v_add_u32 %1, 0x9E3779B9, %1\n v_add_u32 %2, 0x9E3779B9, %2\n v_add_u32 %3, 0x9E3779B9, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 48) asm volatile("v_and_b32 %0, 0x0FFFFFFF, %0\n v_and_b32 %1, 0x0FFFFFFF, %1\n v_and_b32 %2, 0x0FFFFFFF, %2\n v_and_b32 %3, 0x0FFFFFFF, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 49) asm volatile("v_lshrrev_b64 %0, 1, %0\n v_lshrrev_b64 %1, 1, %1\n v_lshrrev_b64 %2, 1, %2\n v_lshrrev_b64 %3, 1, %3" : "+v"(wa), "+v"(wb), "+v"(wc), "+v"(wd));
            if (kOp == 50) { asm volatile("v_mad_u64_u32 %0, s[12:13], %4, %5, 0\n v_mad_u64_u32 %1, s[12:13], %4, %6, 0\n v_mad_u64_u32 %2, s[12:13], %4, %7, 0\n v_mad_u64_u32 %3, s[12:13], %4, %8, 0" : "+v"(wa), "+v"(wb), "+v"(wc), "+v"(wd) : "v"(m), "v"(a), "v"(b), "v"(c), "v"(d) : "s12", "s13"); a ^= (uint32_t)(wa >> 32); }
            if (kOp == 51) asm volatile("v_readfirstlane_b32 s21, %0\n v_readfirstlane_b32 s22, %1\n v_readfirstlane_b32 s23, %2\n v_readfirstlane_b32 s24, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "s21", "s22", "s23", "s24");
            if (kOp == 52) asm volatile("v_cmp_lt_u32_e64 s[12:13], %0, %4\n v_cmp_lt_u32_e64 s[14:15], %1, %4\n v_cndmask_b32_e64 %0, %0, %4, s[12:13]\n v_cndmask_b32_e64 %1, %1, %4, s[14:15]" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "s12", "s13", "s14", "s15");
            if (kOp == 54) asm volatile("v_bitop3_b32 %0, %0, s20, %1 bitop3:0x96\n v_bitop3_b32 %1, %1, s20, %2 bitop3:0x96\n v_bitop3_b32 %2, %2, s20, %3 bitop3:0x96\n v_bitop3_b32 %3, %3, s20, %0 bitop3:0x96" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (kOp == 55) asm volatile("v_cndmask_b32_e64 %0, %0, %4, vcc\n v_cndmask_b32_e64 %1, %1, %4, vcc\n v_cndmask_b32_e64 %2, %2, %4, vcc\n v_cndmask_b32_e64 %3, %3, %4, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "vcc", "s10", "s11");
            if (kOp == 56) { asm volatile("s_mov_b64 vcc, s[10:11]\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "vcc", "s10", "s11"); }
            if (kOp == 57) { asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cmp_lt_u32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "vcc", "s10", "s11"); }
            if (kOp == 53) asm volatile("v_bitop3_b32 %0, %0, %4, %1 bitop3:0x96\n v_bitop3_b32 %1, %1, %4, %2 bitop3:0x96\n v_bitop3_b32 %2, %2, %4, %3 bitop3:0x96\n v_bitop3_b32 %3, %3, %4, %0 bitop3:0x96" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
        }
    }
    const unsigned long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) atomicMax(cycles, t1 - t0);
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ (uint32_t)wa ^ (uint32_t)wb ^ (uint32_t)wc ^ (uint32_t)wd;
}

static const char *kNames[] = {"v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xnor_b32", "v_lshrrev_b32", "v_lshlrev_b32", "v_lshrrev_b32(v)", "v_ashrrev_i32", "v_max_u32", "v_min_u32", "v_mov_b32", "v_not_b32", "v_add_co_u32", "v_bfe_u32", "v_bfe_i32", "v_bfi_b32", "v_alignbit_b32", "v_perm_b32", "v_mad_u32_u24", "v_mul_u32_u24", "v_mul_hi_u32_u24", "v_mul_lo_u32", "v_mul_hi_u32", "v_add3_u32", "v_lshl_or_b32", "v_lshl_add_u32", "v_add_lshl_u32", "v_and_or_b32", "v_or3_b32", "v_xad_u32", "v_med3_u32", "v_mbcnt_lo", "v_bcnt_u32", "v_ffbl_b32", "v_cndmask(vcc)", "v_cndmask(sgpr)", "v_cmp_lt_u32->vcc", "v_cmp_lt_u32->sgpr", "v_mov_dpp quad_perm", "v_add_u32_sdwa", "v_pk_add_u16", "v_dot4_u32_u8", "v_xor_b32(sgpr)", "v_xor_b32(lit)", "v_add_u32(lit)", "v_and_b32(lit)", "v_lshrrev_b64", "v_mad_u64_u32", "v_readfirstlane", "s_nop-free cmp+cndmask pair", "v_bitop3_b32 (a^b^c)", "v_bitop3_b32 (a^sgpr^c)", "v_cndmask_e64(vcc)", "s_mov vcc + 4 v_cndmask(vcc) [5 instr]", "cmp->vcc, cndmask(vcc) x2 [4 instr]"};

template <int kOp>
static void run(uint32_t *out, unsigned long long *cycles, int cus)
{
    double per_simd[3];
    int k = 0;
    for (int waves_per_simd = 1; waves_per_simd <= 4; waves_per_simd *= 2) {
        const int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
        unsigned long long c = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipMemset(cycles, 0, sizeof(*cycles)));
            alu_kernel<kOp><<<blocks, 256>>>(out, 1u + rep, cycles);
            CHECK(hipDeviceSynchronize());
        }
        CHECK(hipMemcpy(&c, cycles, sizeof(c), hipMemcpyDeviceToHost));
        const double per_wave = (double)kIters * kUnroll * 4;
        per_simd[k++] = c / (per_wave * waves_per_simd);
    }
    printf("%-28s %6.2f %6.2f %6.2f\n", kNames[kOp], per_simd[0], per_simd[1], per_simd[2]);
    fflush(stdout);
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
    printf("%s, %d CUs; %d instructions per lane and kernel; cycles per wave-instruction per SIMD at 1 / 2 / 4 waves per SIMD\n",
           prop.name, cus, kIters * kUnroll * 4);
    run<0>(out, cycles, cus);
    run<1>(out, cycles, cus);
    run<2>(out, cycles, cus);
    run<3>(out, cycles, cus);
    run<4>(out, cycles, cus);
    run<5>(out, cycles, cus);
    run<6>(out, cycles, cus);
    run<7>(out, cycles, cus);
    run<8>(out, cycles, cus);
    run<9>(out, cycles, cus);
    run<10>(out, cycles, cus);
    run<11>(out, cycles, cus);
    run<12>(out, cycles, cus);
    run<13>(out, cycles, cus);
    run<14>(out, cycles, cus);
    run<15>(out, cycles, cus);
    run<16>(out, cycles, cus);
    run<17>(out, cycles, cus);
    run<18>(out, cycles, cus);
    run<19>(out, cycles, cus);
    run<20>(out, cycles, cus);
    run<21>(out, cycles, cus);
    run<22>(out, cycles, cus);
    run<23>(out, cycles, cus);
    run<24>(out, cycles, cus);
    run<25>(out, cycles, cus);
    run<26>(out, cycles, cus);
    run<27>(out, cycles, cus);
    run<28>(out, cycles, cus);
    run<29>(out, cycles, cus);
    run<30>(out, cycles, cus);
    run<31>(out, cycles, cus);
    run<32>(out, cycles, cus);
    run<33>(out, cycles, cus);
    run<34>(out, cycles, cus);
    run<35>(out, cycles, cus);
    run<36>(out, cycles, cus);
    run<37>(out, cycles, cus);
    run<38>(out, cycles, cus);
    run<39>(out, cycles, cus);
    run<40>(out, cycles, cus);
    run<41>(out, cycles, cus);
    run<42>(out, cycles, cus);
    run<43>(out, cycles, cus);
    run<44>(out, cycles, cus);
    run<45>(out, cycles, cus);
    run<46>(out, cycles, cus);
    run<47>(out, cycles, cus);
    run<48>(out, cycles, cus);
    run<49>(out, cycles, cus);
    run<50>(out, cycles, cus);
    run<51>(out, cycles, cus);
    run<52>(out, cycles, cus);
    run<53>(out, cycles, cus);
    run<54>(out, cycles, cus);
    run<55>(out, cycles, cus);
    run<56>(out, cycles, cus);
    run<57>(out, cycles, cus);
    return 0;
}
