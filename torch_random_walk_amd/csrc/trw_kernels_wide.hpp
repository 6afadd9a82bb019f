// trw_kernels_wide.hpp -- int64 CSR walk kernels (any graph size).
//
// These index the caller's int64 row_ptr / col_idx directly and write the
// [N, L+1] int64 result in place.  They serve graphs beyond 2^31 nodes / 2^32
// entries and the uniform walk on the per-walker stream; everything else runs
// the narrow kernels of trw_kernels_narrow.hpp.  Included by trw_hip.hip only.
#pragma once

namespace {

using trw::kGlDeg;
using trw::kGlSep;
using trw::kJumpDigits;
using trw::kJumpRadix;
using trw::kJumpRow;

constexpr int kBlock = 256;          // 4 waves of 64
constexpr int kStatusBytes = 256;    // status block at the head of the workspace
constexpr uint32_t kTrialCap = 1u << 20;
constexpr int kStDescents = 8;    // status slot: adjacent pairs col[e] > col[e+1]   (uint32 shadow pass)
constexpr int kStBoundary = 9;    // status slot: such pairs that straddle a row boundary
// Rows are all sorted <=> the int64 probe raised no flag (status[1]) and every descent the shadow pass counted sits
// on a row boundary; whichever of the two passes did not run left its slots zero.
__device__ __forceinline__ bool rows_sorted(const int64_t *status)
{
    return status[1] == 0 && status[kStDescents] == status[kStBoundary];
}
constexpr int64_t kStTrialCap = TRW_ST_TRIAL_CAP;

// x^(d*256^k) mod (x^31 - x^28 - 1), see glibc_jump.hpp.  Uploaded once per device.
__device__ uint32_t g_jump_tab[kJumpDigits * kJumpRadix * kJumpRow];

struct SeedWindow {
    uint32_t w[kGlDeg];
};

// The reference's one serial rand() stream, stepped by a single lane: the 31-word generator window
// lives in LDS (it is indexed by a running slot), seeded from the host-computed window of srand(seed).
struct GlibcSerial {
    uint32_t *w;
    uint32_t slot;
    __device__ __forceinline__ void init(uint32_t *lds31, const SeedWindow &sw)
    {
        w = lds31;
        slot = 0;
        for (int j = 0; j < kGlDeg; ++j) w[j] = sw.w[j];
    }
    __device__ __forceinline__ uint32_t next()  // one rand(): r[k] = r[k-31] + r[k-3], value >> 1
    {
        uint32_t k = slot + (kGlDeg - kGlSep);
        if (k >= kGlDeg) k -= kGlDeg;
        const uint32_t v = w[slot] + w[k];
        w[slot] = v;
        slot = (slot + 1 == kGlDeg) ? 0 : slot + 1;
        return v >> 1;
    }
};

struct Thresholds {
    uint32_t t0, t1, t2;  // accept iff rand31 < t_k  (k = distance class 0,1,2)
};

// ---------------------------------------------------------------------------
// Shared device helpers
// ---------------------------------------------------------------------------

// r % deg with the reference's int64 semantics (rw_cpu.cpp:20) in 32-bit math:
// r < 2^31, so whenever deg > r the result is r itself.
__device__ __forceinline__ uint32_t draw_mod(uint32_t r, int64_t deg)
{
    return ((uint64_t)deg > (uint64_t)r) ? r : r % (uint32_t)deg;
}

// sample_neighbor (rw_cpu.cpp:7-30) given the row of `cur` already in registers.
// Data faults do not trap: the walker stays on `cur` and a status bit is raised.
__device__ __forceinline__ int64_t pick_neighbor(int64_t cur, int64_t start, int64_t deg, uint32_t r,
                                                 const int64_t *__restrict__ col_idx, int64_t n_nodes,
                                                 int64_t nnz, uint32_t &st)
{
    (void)nnz;
    if (deg <= 0) {  // rw_cpu.cpp:20 divides by zero here; rows outside [0,nnz] were emptied by load_row,
        st |= TRW_ST_DEGREE0;  // which is where the reference's bounds fallback (rw_cpu.cpp:23-28) lands too
        return cur;
    }
    const int64_t idx = start + (int64_t)draw_mod(r, deg);
    const int64_t nx = col_idx[idx];
    if ((uint64_t)nx >= (uint64_t)n_nodes) {
        st |= TRW_ST_NODE_RANGE;
        return cur;
    }
    return nx;
}

// The CSR row of `node`.  A row that is not inside [0, nnz] is reported and
// emptied so that nothing downstream indexes col_idx through it.
__device__ __forceinline__ void load_row(const int64_t *__restrict__ row_ptr, int64_t node, int64_t nnz,
                                         int64_t &start, int64_t &deg, uint32_t &st)
{
    const int64_t a = row_ptr[node];
    const int64_t b = row_ptr[node + 1];
    start = a;
    deg = b - a;
    if (a < 0 || b < a || b > nnz) {
        st |= TRW_ST_BAD_CSR;
        start = 0;
        deg = 0;
    }
}

__device__ __forceinline__ void mulhilo32(uint32_t a, uint32_t b, uint32_t &hi, uint32_t &lo)
{
    unsigned long long prod, carry;
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(prod), "=s"(carry) : "v"(a), "v"(b));
    hi = (uint32_t)(prod >> 32);
    lo = (uint32_t)prod;
}

// hi1 ^ c1 ^ k: gfx950's three-input boolean v_bitop3_b32 (truth table 0x96 = a ^ b ^ c) runs at the rate of ONE v_xor_b32
// between vector registers (2.6 cycles per wave, scripts/alu_probe2.hip -> profiles/r05_alu_probe2.txt; 4.35 with a scalar
// operand, which is still less than the v_xor pair's 6.9)
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
// (a & b) | (~a & c): v_bfi_b32 at the fast rate when all three operands are vector registers
__device__ __forceinline__ uint32_t select_bits(uint32_t mask, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(mask, b, c, 0xCA); }

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&o)[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one v_mad_u64_u32 per product: the compiler's own lowering of a 32 x 32 -> 64 multiply by a constant is a
        // v_mul_hi_u32 and a v_mul_lo_u32 (both quarter rate) for the same thing
        uint32_t hi0, lo0, hi1, lo1;
        mulhilo32(0xD2511F53u, c0, hi0, lo0);
        mulhilo32(0xCD9E8D57u, c2, hi1, lo1);
        const uint32_t n0 = xor3(hi1, c1, k0);
        const uint32_t n2 = xor3(hi0, c3, k1);
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

// The same ten rounds with the key schedule handed in (k0r[r] = seed + r * 0x9E3779B9, k1r[r] = r * 0xBB67AE85, r = 0..9;
// philox_keys_v) -- by the walker pools, which keep it in VECTOR registers: an operation with a scalar operand costs 4.3
// cycles per wave where one between vector registers costs 2.6 (scripts/alu_probe2.hip), and a block has twenty of them.
struct PhiloxKeysV {
    uint32_t k0[10], k1[10];
};
__device__ __forceinline__ void philox_keys_v(uint32_t seed, PhiloxKeysV &k)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        k.k0[r] = seed + (uint32_t)r * 0x9E3779B9u;
        k.k1[r] = (uint32_t)r * 0xBB67AE85u;
        asm volatile("" : "+v"(k.k0[r]), "+v"(k.k1[r]));  // (pinned: the compiler would fold them back into scalars / literals)
    }
}
__device__ __forceinline__ void philox4x32_10_keyed(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                    const PhiloxKeysV &k, uint32_t (&o)[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        mulhilo32(0xD2511F53u, c0, hi0, lo0);
        mulhilo32(0xCD9E8D57u, c2, hi1, lo1);
        const uint32_t n0 = xor3(hi1, c1, k.k0[r]);
        const uint32_t n2 = r == 0 ? (hi0 ^ c3) : xor3(hi0, c3, k.k1[r]);  // (the second key word starts at 0)
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

// Block `blk` of walker `w`'s stream: key (seed, 0), counter (w_lo, w_hi, blk, 0).
__device__ __forceinline__ void walker_block(uint64_t w, uint32_t blk, uint32_t seed, uint32_t (&o)[4])
{
    philox4x32_10((uint32_t)w, (uint32_t)(w >> 32), blk, 0u, seed, 0u, o);
}

// ---------------------------------------------------------------------------
// glibc jump-ahead: one 31-word window per lane of the uniform kernel
// ---------------------------------------------------------------------------

__device__ __forceinline__ void window_apply_dev(const uint32_t *__restrict__ g, uint32_t (&w)[kGlDeg])
{
    uint32_t ext[2 * kGlDeg - 1];
#pragma unroll
    for (int j = 0; j < kGlDeg; ++j) ext[j] = w[j];
#pragma unroll
    for (int t = 0; t < kGlDeg - 1; ++t) ext[kGlDeg + t] = ext[t] + ext[t + kGlDeg - kGlSep];
    uint32_t acc[kGlDeg];
#pragma unroll
    for (int m = 0; m < kGlDeg; ++m) acc[m] = 0;
#pragma unroll
    for (int i = 0; i < kGlDeg; ++i) {
        const uint32_t gi = g[i];
#pragma unroll
        for (int m = 0; m < kGlDeg; ++m) acc[m] += gi * ext[i + m];
    }
#pragma unroll
    for (int m = 0; m < kGlDeg; ++m) w[m] = acc[m];
}

// win[j*nlanes + lane] = generator word j in front of draw (pos0 + lane*stride).
__global__ void __launch_bounds__(kBlock)
glibc_windows_kernel(SeedWindow sw, uint64_t pos0, uint64_t stride, int64_t nlanes, uint32_t *__restrict__ win)
{
    const int64_t lane = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (lane >= nlanes) return;
    const uint64_t pos = pos0 + (uint64_t)lane * stride;
    uint32_t w[kGlDeg];
#pragma unroll
    for (int j = 0; j < kGlDeg; ++j) w[j] = sw.w[j];
#pragma unroll 1
    for (int k = 0; k < kJumpDigits; ++k) {
        const uint32_t d = (uint32_t)(pos >> (8 * k)) & 255u;
        if (d) window_apply_dev(&g_jump_tab[(k * kJumpRadix + d) * kJumpRow], w);
    }
#pragma unroll
    for (int j = 0; j < kGlDeg; ++j) win[(int64_t)j * nlanes + lane] = w[j];
}

// ---------------------------------------------------------------------------
// Uniform walk, reference-exact draws (uniform_walk, rw_cpu.cpp:58-107)
// ---------------------------------------------------------------------------
// Lane l owns walkers [l*G, l*G+G): their draws are G*L consecutive words of the
// glibc stream, produced in registers by the lagged-Fibonacci recurrence.  The
// loop is unrolled by the generator order so every window index is static.
__global__ void __launch_bounds__(kBlock)
uniform_glibc_wide_kernel(const int64_t *__restrict__ row_ptr, const int64_t *__restrict__ col_idx,
                     const int64_t *__restrict__ targets, int64_t n_nodes, int64_t nnz, int64_t n_walkers,
                     int32_t L, int32_t G, int64_t nlanes, const uint32_t *__restrict__ win,
                     int64_t *__restrict__ out, int64_t *__restrict__ status)
{
    const int64_t lane = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (lane == 0) status[2] = TRW_RNG_GLIBC;
    if (lane >= nlanes) return;

    uint32_t w[kGlDeg];
#pragma unroll
    for (int j = 0; j < kGlDeg; ++j) w[j] = win[(int64_t)j * nlanes + lane];

    uint32_t st = 0;
    int64_t walker = lane * G;
    int32_t s = 0;  // next column to write; 0 = start a new walker
    int64_t cur = 0;
    int64_t *row_out = out;
    bool live = false, mine = false;
    const int32_t total = G * L;

    // One draw = one add on a statically indexed window; one step = two dependent gathers.
    auto step = [&](uint32_t r) {
        if (s == 0) {  // wave-uniform: every lane starts its next walker together
            mine = walker < n_walkers;
            live = mine;
            if (mine) {
                cur = targets[walker];
                row_out = out + walker * (int64_t)(L + 1);
                row_out[0] = cur;
                if ((uint64_t)cur >= (uint64_t)n_nodes) {  // reported; the row is filled with the bad id
                    st |= TRW_ST_NODE_RANGE;
                    live = false;
                }
            }
            s = 1;
        }
        if (live) {
            int64_t start, deg;
            load_row(row_ptr, cur, nnz, start, deg, st);
            cur = pick_neighbor(cur, start, deg, r, col_idx, n_nodes, nnz, st);
        }
        if (mine) row_out[s] = cur;
        if (++s > L) {
            s = 0;
            ++walker;
        }
    };
    int32_t d0 = 0;
    for (; d0 + kGlDeg <= total; d0 += kGlDeg) {
#pragma unroll
        for (int j = 0; j < kGlDeg; ++j) {
            const uint32_t v = w[j] + w[(j + kGlDeg - kGlSep) % kGlDeg];
            w[j] = v;
            step(v >> 1);
        }
    }
#pragma unroll
    for (int j = 0; j < kGlDeg - 1; ++j) {
        if (d0 + j < total) {  // wave-uniform tail
            const uint32_t v = w[j] + w[(j + kGlDeg - kGlSep) % kGlDeg];
            w[j] = v;
            step(v >> 1);
        }
    }
    if (st) atomicOr((unsigned long long *)&status[0], (unsigned long long)st);
}

// Uniform walk on the per-walker Philox stream: draw s-1 drives step s.
__global__ void __launch_bounds__(kBlock)
uniform_walker_kernel(const int64_t *__restrict__ row_ptr, const int64_t *__restrict__ col_idx,
                      const int64_t *__restrict__ targets, int64_t n_nodes, int64_t nnz, int64_t n_walkers,
                      int64_t walker_base, int32_t L, uint32_t seed, int64_t *__restrict__ out,
                      int64_t *__restrict__ status)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i == 0) status[2] = TRW_RNG_WALKER;
    if (i >= n_walkers) return;
    uint32_t st = 0;
    int64_t cur = targets[i];
    int64_t *row_out = out + i * (int64_t)(L + 1);
    row_out[0] = cur;
    if ((uint64_t)cur >= (uint64_t)n_nodes) {
        for (int32_t c = 1; c <= L; ++c) row_out[c] = cur;
        atomicOr((unsigned long long *)&status[0], (unsigned long long)TRW_ST_NODE_RANGE);
        return;
    }
    const uint64_t gw = (uint64_t)walker_base + (uint64_t)i;
    uint32_t blk[4];
    for (int32_t s0 = 0; s0 < L; s0 += 4) {
        walker_block(gw, (uint32_t)(s0 >> 2), seed, blk);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (s0 + k >= L) break;
            int64_t start, deg;
            load_row(row_ptr, cur, nnz, start, deg, st);
            cur = pick_neighbor(cur, start, deg, blk[k] >> 1, col_idx, n_nodes, nnz, st);
            row_out[s0 + k + 1] = cur;
        }
    }
    if (st) atomicOr((unsigned long long *)&status[0], (unsigned long long)st);
}

// ---------------------------------------------------------------------------
// Row order probe: may is_neighbor bisect?
// ---------------------------------------------------------------------------
// status[1] is raised when some row of the CSR is not in non-decreasing order.
// Rows longer than a wave are swept by the whole wave, 64 entries per pass.
__global__ void __launch_bounds__(kBlock)
csr_rows_sorted_kernel(const int64_t *__restrict__ row_ptr, const int64_t *__restrict__ col_idx, int64_t n_nodes,
                       int64_t nnz, int64_t *__restrict__ status)
{
    const int lane = threadIdx.x & 63;
    int64_t node = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    bool unsorted = false;
    bool bad = false;
    // every lane of a wave runs the same number of iterations so the ballots below are safe
    const int64_t iters = (n_nodes + stride - 1) / stride;
    for (int64_t it = 0; it < iters; ++it, node += stride) {
        int64_t a = 0, b = 0;
        if (node < n_nodes) {
            a = row_ptr[node];
            b = row_ptr[node + 1];
            if (a < 0 || b < a || b > nnz) {
                bad = true;
                b = a;
            }
        }
        const int64_t deg = b - a;
        const bool longrow = deg > 64;
        if (!longrow) {
            for (int64_t e = a + 1; e < b; ++e)
                if (col_idx[e - 1] > col_idx[e]) unsorted = true;
        }
        unsigned long long todo = __ballot(longrow);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int64_t ra = __shfl(a, src);
            const int64_t rb = __shfl(b, src);
            for (int64_t e = ra + 1 + lane; e < rb; e += 64)
                if (col_idx[e - 1] > col_idx[e]) unsorted = true;
        }
    }
    if (__any(unsorted) && lane == 0) status[1] = 1;
    if (__any(bad) && lane == 0) atomicOr((unsigned long long *)&status[0], (unsigned long long)TRW_ST_BAD_CSR);
}

// is_neighbor(x, t) (rw_cpu.cpp:32-56): does t's row contain x?  The reference
// scans linearly with early exit; the result is a pure membership boolean, so a
// bisection over a sorted row returns the same bit.
// `probes` counts the col_idx entries this call reads; `ref_scan` the entries the
// reference's linear early-exit scan would have read for the same answer (the
// first occurrence of x is at offset lo, so it reads lo+1; a miss reads the row).
__device__ __forceinline__ bool row_contains(const int64_t *__restrict__ col_idx, int64_t start, int64_t deg,
                                             int64_t x, bool sorted, uint32_t &probes, uint64_t &ref_scan)
{
    if (sorted) {
        int64_t lo = 0, n = deg;
        while (n > 0) {
            const int64_t half = n >> 1;
            const int64_t v = col_idx[start + lo + half];
            ++probes;
            if (v < x) {
                lo += half + 1;
                n -= half + 1;
            } else {
                n = half;
            }
        }
        bool found = false;
        if (lo < deg) {
            ++probes;
            found = col_idx[start + lo] == x;
        }
        ref_scan += found ? (uint64_t)lo + 1 : (uint64_t)deg;
        return found;
    }
    for (int64_t e = start; e < start + deg; ++e) {
        ++probes;
        if (col_idx[e] == x) {
            ref_scan += (uint64_t)(e - start) + 1;
            return true;
        }
    }
    ref_scan += (uint64_t)deg;
    return false;
}

__device__ __forceinline__ uint64_t wave_sum(uint64_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}

// ---------------------------------------------------------------------------
// Biased (node2vec) walk on the per-walker stream (biased_walk, rw_cpu.cpp:109-201)
// ---------------------------------------------------------------------------
// One walker per lane.  Draw 0 picks the first neighbour; trial k >= 0 of the
// rejection loop consumes draws 2k+1 (candidate) and 2k+2 (acceptance), both
// always, in that order (rw_cpu.cpp:168-169).  Lanes advance through their own
// steps independently, so a lane that accepts early does not wait for the
// wave's slowest trial chain at every step, only at the end of the walk.
template <bool kCount>
__global__ void __launch_bounds__(kBlock)
biased_walker_wide_kernel(const int64_t *__restrict__ row_ptr, const int64_t *__restrict__ col_idx,
                     const int64_t *__restrict__ targets, int64_t n_nodes, int64_t nnz, int64_t n_walkers,
                     int64_t walker_base, int32_t L, uint32_t seed, Thresholds th, int64_t *__restrict__ out,
                     int64_t *__restrict__ status)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i == 0) status[2] = TRW_RNG_WALKER;
    const bool sorted = rows_sorted(status);
    uint32_t st = 0;
    uint32_t n_trials = 0, n_member = 0, n_probes = 0;
    uint64_t n_refscan = 0;
    bool run = i < n_walkers;

    int64_t *row_out = out + (run ? i : 0) * (int64_t)(L + 1);
    const int64_t target = run ? targets[i] : 0;
    if (run) {
        row_out[0] = target;
        if ((uint64_t)target >= (uint64_t)n_nodes) {
            for (int32_t c = 1; c <= L; ++c) row_out[c] = target;
            st |= TRW_ST_NODE_RANGE;
            run = false;
        }
    }
    const uint64_t gw = (uint64_t)walker_base + (uint64_t)i;
    uint32_t blk[4];
    uint32_t bi = 0;
    walker_block(gw, bi, seed, blk);
  if (run) {

    // step 1: uniform neighbour of the start node (rw_cpu.cpp:159)
    int64_t t = target, t_start, t_deg;
    load_row(row_ptr, t, nnz, t_start, t_deg, st);
    int64_t prev = pick_neighbor(t, t_start, t_deg, blk[0] >> 1, col_idx, n_nodes, nnz, st);
    row_out[1] = prev;
    int64_t p_start, p_deg;
    load_row(row_ptr, prev, nnz, p_start, p_deg, st);

    int32_t s = 2;
    uint32_t rejected = 0;
    bool odd = false;  // which half of the Philox block the next trial draws from
    while (s <= L) {
        uint32_t rc, ru;
        if (!odd) {
            rc = blk[1] >> 1;
            ru = blk[2] >> 1;
        } else {
            rc = blk[3] >> 1;
            walker_block(gw, ++bi, seed, blk);
            ru = blk[0] >> 1;
        }
        odd = !odd;

        uint32_t st_trial = 0;
        const int64_t x = pick_neighbor(prev, p_start, p_deg, rc, col_idx, n_nodes, nnz, st_trial);
        ++n_trials;
        bool accept;
        if (st_trial) {  // dead end or corrupt row: take it, the status bit tells the caller
            st |= st_trial;
            accept = true;
        } else if (x == t) {
            accept = ru < th.t0;
        } else {
            ++n_member;
            accept = ru < (row_contains(col_idx, t_start, t_deg, x, sorted, n_probes, n_refscan) ? th.t1 : th.t2);
        }
        if (!accept && ++rejected >= kTrialCap) {
            st |= (uint32_t)kStTrialCap;
            accept = true;
        }
        if (accept) {
            row_out[s] = x;
            t = prev;
            t_start = p_start;
            t_deg = p_deg;
            prev = x;
            if (s < L) load_row(row_ptr, prev, nnz, p_start, p_deg, st);
            ++s;
            rejected = 0;
        }
    }
  }
    if (st) atomicOr((unsigned long long *)&status[0], (unsigned long long)st);
    if (kCount) {  // whole waves reach this point together
        // same slots as the narrow kernel: trials, candidate loads (= trials here), membership
        // tests performed, col_idx entries those tests read
        (void)n_refscan;
        const uint64_t a = wave_sum(n_trials), b = wave_sum(n_member), c = wave_sum(n_probes);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd((unsigned long long *)&status[4], (unsigned long long)a);
            atomicAdd((unsigned long long *)&status[5], (unsigned long long)a);
            atomicAdd((unsigned long long *)&status[6], (unsigned long long)b);
            atomicAdd((unsigned long long *)&status[7], (unsigned long long)c);
        }
    }
}

// ---------------------------------------------------------------------------
// Biased walk, reference-exact: the serial glibc chain replayed by one lane
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
biased_glibc_seq_kernel(const int64_t *__restrict__ row_ptr, const int64_t *__restrict__ col_idx,
                        const int64_t *__restrict__ targets, int64_t n_nodes, int64_t nnz, int64_t n_walkers,
                        int32_t L, SeedWindow sw, Thresholds th, int32_t only_unsorted, int64_t *__restrict__ out,
                        int64_t *__restrict__ status)
{
    __shared__ uint32_t w[kGlDeg];
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const bool sorted = rows_sorted(status);
    if (only_unsorted && sorted) return;  // the wave-cooperative replay (biased_glibc_wave_kernel) has this graph
    status[2] = TRW_RNG_GLIBC;
    GlibcSerial glibc;
    glibc.init(w, sw);
    auto next = [&]() -> uint32_t { return glibc.next(); };
    uint32_t st = 0;
    uint32_t n_probes = 0;
    uint64_t n_refscan = 0;
    for (int64_t i = 0; i < n_walkers; ++i) {
        int64_t *row_out = out + i * (int64_t)(L + 1);
        const int64_t target = targets[i];
        row_out[0] = target;
        if ((uint64_t)target >= (uint64_t)n_nodes) {
            for (int32_t c = 1; c <= L; ++c) row_out[c] = target;
            st |= TRW_ST_NODE_RANGE;
            break;  // the chain's draw positions are undefined past a fault
        }
        const uint32_t th_lo = th.t1 < th.t2 ? th.t1 : th.t2, th_hi = th.t1 < th.t2 ? th.t2 : th.t1;
        int64_t t = target, t_start, t_deg;
        load_row(row_ptr, t, nnz, t_start, t_deg, st);
        int64_t prev = pick_neighbor(t, t_start, t_deg, next(), col_idx, n_nodes, nnz, st);
        row_out[1] = prev;
        int64_t p_start, p_deg;
        load_row(row_ptr, prev, nnz, p_start, p_deg, st);
        for (int32_t s = 2; s <= L && !st; ++s) {
            uint32_t rejected = 0;
            int64_t x;
            while (true) {
                const uint32_t rc = next();
                const uint32_t ru = next();
                x = pick_neighbor(prev, p_start, p_deg, rc, col_idx, n_nodes, nnz, st);
                if (st) break;
                bool accept;
                if (x == t) accept = ru < th.t0;
                else if (ru < th_lo) accept = true;    // below both thresholds: is_neighbor cannot change the answer
                else if (ru >= th_hi) accept = false;  // above both
                else accept = ru < (row_contains(col_idx, t_start, t_deg, x, sorted, n_probes, n_refscan) ? th.t1 : th.t2);
                if (accept) break;
                if (++rejected >= kTrialCap) {
                    st |= (uint32_t)kStTrialCap;
                    break;
                }
            }
            row_out[s] = x;
            t = prev;
            t_start = p_start;
            t_deg = p_deg;
            prev = x;
            load_row(row_ptr, prev, nnz, p_start, p_deg, st);
        }
        if (st) break;
    }
    if (st) atomicOr((unsigned long long *)&status[0], (unsigned long long)st);
}


}  // namespace
