// trw_hip.hip -- MI355X (gfx950) CSR random-walk kernels and their C ABI.
//
// One hot path of Sachinx0e/torch_random_walk, written from scratch for CDNA4:
// the CSR walker behind torch_rw.rw.walk (reference csrc/cpu/rw_cpu.cpp is the
// behaviour, csrc/cuda/rw_cuda.cu the shape of the GPU entry point).  This is
// integer gather work bounded by HBM/L2 latency and transaction rate; there is
// no contraction in it and therefore no MFMA.
//
// Kernels
//   glibc_windows_kernel   per-lane jump-ahead into glibc's rand() stream
//   uniform_glibc_kernel   p == q == 1, reference-exact draws, one walker chain per lane
//   uniform_walker_kernel  p == q == 1, per-walker Philox stream
//   csr_rows_sorted_kernel decides whether the membership test may bisect
//   biased_walker_kernel   node2vec rejection sampling, per-walker Philox stream
//   biased_glibc_seq_kernel node2vec, reference-exact serial replay (one lane)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/trw.h"
#include "glibc_jump.hpp"

namespace {

using trw::kGlDeg;
using trw::kGlSep;
using trw::kJumpDigits;
using trw::kJumpRadix;
using trw::kJumpRow;

constexpr int kBlock = 256;          // 4 waves of 64
constexpr int kStatusBytes = 256;    // status block at the head of the workspace
constexpr uint32_t kTrialCap = 1u << 20;
constexpr int64_t kStTrialCap = TRW_ST_TRIAL_CAP;

// x^(d*256^k) mod (x^31 - x^28 - 1), see glibc_jump.hpp.  Uploaded once per device.
__device__ uint32_t g_jump_tab[kJumpDigits * kJumpRadix * kJumpRow];

struct SeedWindow {
    uint32_t w[kGlDeg];
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

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&o)[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
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
uniform_glibc_kernel(const int64_t *__restrict__ row_ptr, const int64_t *__restrict__ col_idx,
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
__device__ __forceinline__ bool row_contains(const int64_t *__restrict__ col_idx, int64_t start, int64_t deg,
                                             int64_t x, bool sorted)
{
    if (sorted) {
        int64_t lo = 0, n = deg;
        while (n > 0) {
            const int64_t half = n >> 1;
            const int64_t v = col_idx[start + lo + half];
            if (v < x) {
                lo += half + 1;
                n -= half + 1;
            } else {
                n = half;
            }
        }
        return lo < deg && col_idx[start + lo] == x;
    }
    for (int64_t e = start; e < start + deg; ++e)
        if (col_idx[e] == x) return true;
    return false;
}

// ---------------------------------------------------------------------------
// Biased (node2vec) walk on the per-walker stream (biased_walk, rw_cpu.cpp:109-201)
// ---------------------------------------------------------------------------
// One walker per lane.  Draw 0 picks the first neighbour; trial k >= 0 of the
// rejection loop consumes draws 2k+1 (candidate) and 2k+2 (acceptance), both
// always, in that order (rw_cpu.cpp:168-169).  Lanes advance through their own
// steps independently, so a lane that accepts early does not wait for the
// wave's slowest trial chain at every step, only at the end of the walk.
__global__ void __launch_bounds__(kBlock)
biased_walker_kernel(const int64_t *__restrict__ row_ptr, const int64_t *__restrict__ col_idx,
                     const int64_t *__restrict__ targets, int64_t n_nodes, int64_t nnz, int64_t n_walkers,
                     int64_t walker_base, int32_t L, uint32_t seed, Thresholds th, int64_t *__restrict__ out,
                     int64_t *__restrict__ status)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i == 0) status[2] = TRW_RNG_WALKER;
    if (i >= n_walkers) return;
    const bool sorted = (status[1] == 0);
    uint32_t st = 0;

    int64_t *row_out = out + i * (int64_t)(L + 1);
    const int64_t target = targets[i];
    row_out[0] = target;
    if ((uint64_t)target >= (uint64_t)n_nodes) {
        for (int32_t c = 1; c <= L; ++c) row_out[c] = target;
        atomicOr((unsigned long long *)&status[0], (unsigned long long)TRW_ST_NODE_RANGE);
        return;
    }
    const uint64_t gw = (uint64_t)walker_base + (uint64_t)i;
    uint32_t blk[4];
    uint32_t bi = 0;
    walker_block(gw, bi, seed, blk);

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
        bool accept;
        if (st_trial) {  // dead end or corrupt row: take it, the status bit tells the caller
            st |= st_trial;
            accept = true;
        } else if (x == t) {
            accept = ru < th.t0;
        } else if (row_contains(col_idx, t_start, t_deg, x, sorted)) {
            accept = ru < th.t1;
        } else {
            accept = ru < th.t2;
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
    if (st) atomicOr((unsigned long long *)&status[0], (unsigned long long)st);
}

// ---------------------------------------------------------------------------
// Biased walk, reference-exact: the serial glibc chain replayed by one lane
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
biased_glibc_seq_kernel(const int64_t *__restrict__ row_ptr, const int64_t *__restrict__ col_idx,
                        const int64_t *__restrict__ targets, int64_t n_nodes, int64_t nnz, int64_t n_walkers,
                        int32_t L, SeedWindow sw, Thresholds th, int64_t *__restrict__ out,
                        int64_t *__restrict__ status)
{
    __shared__ uint32_t w[kGlDeg];
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    status[2] = TRW_RNG_GLIBC;
    const bool sorted = (status[1] == 0);
    for (int j = 0; j < kGlDeg; ++j) w[j] = sw.w[j];
    uint32_t slot = 0;
    auto next = [&]() -> uint32_t {
        uint32_t k = slot + (kGlDeg - kGlSep);
        if (k >= kGlDeg) k -= kGlDeg;
        const uint32_t v = w[slot] + w[k];
        w[slot] = v;
        slot = (slot + 1 == kGlDeg) ? 0 : slot + 1;
        return v >> 1;
    };
    uint32_t st = 0;
    for (int64_t i = 0; i < n_walkers; ++i) {
        int64_t *row_out = out + i * (int64_t)(L + 1);
        const int64_t target = targets[i];
        row_out[0] = target;
        if ((uint64_t)target >= (uint64_t)n_nodes) {
            for (int32_t c = 1; c <= L; ++c) row_out[c] = target;
            st |= TRW_ST_NODE_RANGE;
            break;  // the chain's draw positions are undefined past a fault
        }
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
                else if (row_contains(col_idx, t_start, t_deg, x, sorted)) accept = ru < th.t1;
                else accept = ru < th.t2;
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

// ---------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------

thread_local std::string t_last_error;

int fail(int code, const std::string &msg)
{
    t_last_error = msg;
    return code;
}

int fail_hip(hipError_t e, const char *what)
{
    return fail(TRW_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

const std::vector<uint32_t> &host_jump_table()
{
    static const std::vector<uint32_t> tab = trw::build_jump_table();
    return tab;
}

// Upload the jump table to the current device once.
int ensure_jump_table()
{
    static std::mutex mu;
    static bool done[64] = {false};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail_hip(e, "hipGetDevice");
    if (dev < 0 || dev >= 64) return fail(TRW_E_HIP, "device ordinal out of range");
    std::lock_guard<std::mutex> lock(mu);
    if (done[dev]) return TRW_OK;
    const auto &tab = host_jump_table();
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_jump_tab), tab.data(), tab.size() * sizeof(uint32_t), 0,
                          hipMemcpyHostToDevice);
    if (e != hipSuccess) return fail_hip(e, "hipMemcpyToSymbol(jump table)");
    done[dev] = true;
    return TRW_OK;
}

// Walkers chained per lane in the glibc uniform kernel: enough lanes to fill the
// chip several times over, few enough that the 124-byte window per lane is noise.
int32_t glibc_chain_length(int64_t n_walkers)
{
    if (const char *env = std::getenv("TRW_GLIBC_CHAIN")) {
        const long v = std::strtol(env, nullptr, 10);
        if (v >= 1 && v <= 64) return (int32_t)v;
    }
    const int64_t g = n_walkers >> 21;  // keep >= 2M lanes in flight-able supply
    return (int32_t)(g < 1 ? 1 : (g > 4 ? 4 : g));
}

int resolve_mode(int32_t rng_mode, bool uniform, int64_t n_walkers, int64_t walker_base, int32_t L)
{
    if (rng_mode == TRW_RNG_GLIBC || rng_mode == TRW_RNG_WALKER) return rng_mode;
    if (rng_mode != TRW_RNG_AUTO) return -1;
    if (uniform) return TRW_RNG_GLIBC;
    const int64_t grain = 32768 / ((int64_t)L + 1);  // rw_cpu.cpp:137
    return (walker_base == 0 && n_walkers <= grain) ? TRW_RNG_GLIBC : TRW_RNG_WALKER;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Optional instrumentation (trw_profile_enable): three events around the two
// phases of a call -- "prepare" (jump-ahead windows / row-order probe) and the
// walk kernel proper -- recorded on the caller's stream.
struct Profile {
    bool on = false;
    bool valid = false;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
};
Profile g_prof;

void prof_mark(int k, hipStream_t stream)
{
    if (!g_prof.on) return;
    if (!g_prof.ev[k] && hipEventCreate(&g_prof.ev[k]) != hipSuccess) return;
    (void)hipEventRecord(g_prof.ev[k], stream);
    if (k == 2) g_prof.valid = true;
}

}  // namespace

extern "C" {

int trw_version(void) { return TRW_VERSION; }

const char *trw_last_error(void) { return t_last_error.c_str(); }

uint32_t trw_accept_threshold(double prob)
{
    // (double)r / RAND_MAX is monotone in r: bisect for the first r that is NOT accepted.
    if (!(prob > 0.0)) return 0;
    uint64_t lo = 0, hi = (uint64_t)1 << 31;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if ((double)mid / 2147483647.0 < prob) lo = mid + 1;
        else hi = mid;
    }
    return (uint32_t)lo;
}

void trw_glibc_window_at(int32_t seed, uint64_t pos, uint32_t window[31])
{
    const auto &tab = host_jump_table();
    trw::seed_window(seed, window);
    for (int k = 0; k < kJumpDigits; ++k) {
        const uint32_t d = (uint32_t)(pos >> (8 * k)) & 255u;
        if (d) trw::window_apply(&tab[((size_t)k * kJumpRadix + d) * kJumpRow], window);
    }
}

size_t trw_walk_csr_workspace_bytes(int64_t n_nodes, int64_t nnz, int64_t n_walkers, int32_t walk_length,
                                    double p, double q, int32_t rng_mode)
{
    (void)n_nodes;
    (void)nnz;
    size_t bytes = kStatusBytes;
    const bool uniform = (p == 1.0 && q == 1.0);
    if (uniform && rng_mode != TRW_RNG_WALKER && n_walkers > 0 && walk_length > 0) {
        const int32_t G = glibc_chain_length(n_walkers);
        const int64_t nlanes = (n_walkers + G - 1) / G;
        bytes += align_up((size_t)nlanes * kGlDeg * sizeof(uint32_t), 256);
    }
    return bytes;
}

int trw_walk_csr_hip(const int64_t *row_ptr, int64_t n_nodes, const int64_t *col_idx, int64_t nnz,
                     const int64_t *targets, int64_t n_walkers, int64_t walker_base, double p, double q,
                     int32_t walk_length, int32_t seed, int32_t rng_mode, int64_t *out, void *workspace,
                     size_t workspace_bytes, void *hip_stream)
{
    t_last_error.clear();
    if (n_nodes < 0 || nnz < 0 || n_walkers < 0 || walker_base < 0)
        return fail(TRW_E_ARG, "negative size argument");
    if (walk_length < 1) return fail(TRW_E_ARG, "walk_length must be >= 1");
    if (!(p > 0.0) || !(q > 0.0) || !std::isfinite(p) || !std::isfinite(q))
        return fail(TRW_E_ARG, "p and q must be positive and finite");
    if (!row_ptr) return fail(TRW_E_ARG, "row_ptr is null");
    if (n_walkers > 0 && (!targets || !out)) return fail(TRW_E_ARG, "targets/out is null");
    if (nnz > 0 && !col_idx) return fail(TRW_E_ARG, "col_idx is null");
    if (n_walkers > 0 && n_nodes == 0) return fail(TRW_E_ARG, "walkers on an empty graph");
    if (!workspace || workspace_bytes < (size_t)kStatusBytes)
        return fail(TRW_E_WORKSPACE, "workspace missing or smaller than the status block");
    if (((uintptr_t)workspace & 15u) != 0) return fail(TRW_E_WORKSPACE, "workspace must be 16-byte aligned");

    const bool uniform = (p == 1.0 && q == 1.0);  // rw_cpu.cpp:220
    const int mode = resolve_mode(rng_mode, uniform, n_walkers, walker_base, walk_length);
    if (mode < 0) return fail(TRW_E_RNG_MODE, "unknown rng_mode");
    if (mode == TRW_RNG_GLIBC && !uniform && walker_base != 0)
        return fail(TRW_E_RNG_MODE, "TRW_RNG_GLIBC biased walks replay one serial stream: walker_base must be 0");
    const unsigned __int128 last_pos = ((unsigned __int128)walker_base + (unsigned __int128)n_walkers) *
                                       (unsigned __int128)walk_length;
    if (mode == TRW_RNG_GLIBC && uniform && last_pos >= ((unsigned __int128)1 << 48))
        return fail(TRW_E_RANGE, "glibc stream position beyond 2^48 draws");

    hipStream_t stream = (hipStream_t)hip_stream;
    int64_t *status = (int64_t *)workspace;
    hipError_t e = hipMemsetAsync(status, 0, kStatusBytes, stream);
    if (e != hipSuccess) return fail_hip(e, "hipMemsetAsync(status)");
    if (n_walkers == 0) return TRW_OK;

    const int64_t L = walk_length;
    if (uniform && mode == TRW_RNG_GLIBC)
        if (int rc = ensure_jump_table()) return rc;
    prof_mark(0, stream);
    if (uniform) {
        if (mode == TRW_RNG_GLIBC) {
            const int32_t G = glibc_chain_length(n_walkers);
            const int64_t nlanes = (n_walkers + G - 1) / G;
            const size_t need = kStatusBytes + align_up((size_t)nlanes * kGlDeg * sizeof(uint32_t), 256);
            if (workspace_bytes < need) return fail(TRW_E_WORKSPACE, "workspace too small for the RNG windows");
            uint32_t *win = (uint32_t *)((char *)workspace + kStatusBytes);
            SeedWindow sw;
            trw::seed_window(seed, sw.w);
            const unsigned grid = (unsigned)((nlanes + kBlock - 1) / kBlock);
            glibc_windows_kernel<<<grid, kBlock, 0, stream>>>(sw, (uint64_t)walker_base * (uint64_t)L,
                                                             (uint64_t)G * (uint64_t)L, nlanes, win);
            prof_mark(1, stream);
            uniform_glibc_kernel<<<grid, kBlock, 0, stream>>>(row_ptr, col_idx, targets, n_nodes, nnz, n_walkers,
                                                             walk_length, G, nlanes, win, out, status);
        } else {
            const unsigned grid = (unsigned)((n_walkers + kBlock - 1) / kBlock);
            prof_mark(1, stream);
            uniform_walker_kernel<<<grid, kBlock, 0, stream>>>(row_ptr, col_idx, targets, n_nodes, nnz, n_walkers,
                                                              walker_base, walk_length, (uint32_t)seed, out,
                                                              status);
        }
    } else {
        // rw_cpu.cpp:130-134, evaluated in the same order, then turned into integer thresholds
        const double max_prob_init = std::fmax(1.0 / p, 1);
        const double max_prob = std::fmax(max_prob_init, 1.0 / q);
        Thresholds th;
        th.t0 = trw_accept_threshold(1.0 / p / max_prob);
        th.t1 = trw_accept_threshold(1.0 / max_prob);
        th.t2 = trw_accept_threshold(1.0 / q / max_prob);
        if (n_nodes > 0) {
            int64_t blocks = (n_nodes + kBlock - 1) / kBlock;
            if (blocks > 256 * 16) blocks = 256 * 16;
            csr_rows_sorted_kernel<<<(unsigned)blocks, kBlock, 0, stream>>>(row_ptr, col_idx, n_nodes, nnz, status);
        }
        prof_mark(1, stream);
        if (mode == TRW_RNG_GLIBC) {
            SeedWindow sw;
            trw::seed_window(seed, sw.w);
            biased_glibc_seq_kernel<<<1, 64, 0, stream>>>(row_ptr, col_idx, targets, n_nodes, nnz, n_walkers,
                                                         walk_length, sw, th, out, status);
        } else {
            const unsigned grid = (unsigned)((n_walkers + kBlock - 1) / kBlock);
            biased_walker_kernel<<<grid, kBlock, 0, stream>>>(row_ptr, col_idx, targets, n_nodes, nnz, n_walkers,
                                                             walker_base, walk_length, (uint32_t)seed, th, out,
                                                             status);
        }
    }
    prof_mark(2, stream);
    e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "kernel launch");
    return TRW_OK;
}

int trw_profile_enable(int on)
{
    g_prof.on = (on != 0);
    g_prof.valid = false;
    return TRW_OK;
}

int trw_profile_last(double *prepare_ms, double *walk_ms)
{
    t_last_error.clear();
    if (!g_prof.on || !g_prof.valid) return fail(TRW_E_ARG, "no profiled walk recorded");
    hipError_t e = hipEventSynchronize(g_prof.ev[2]);
    if (e != hipSuccess) return fail_hip(e, "hipEventSynchronize");
    float a = 0.f, b = 0.f;
    e = hipEventElapsedTime(&a, g_prof.ev[0], g_prof.ev[1]);
    if (e != hipSuccess) return fail_hip(e, "hipEventElapsedTime");
    e = hipEventElapsedTime(&b, g_prof.ev[1], g_prof.ev[2]);
    if (e != hipSuccess) return fail_hip(e, "hipEventElapsedTime");
    if (prepare_ms) *prepare_ms = a;
    if (walk_ms) *walk_ms = b;
    return TRW_OK;
}

int trw_walk_status(const void *workspace, void *hip_stream, int64_t status_out[4])
{
    t_last_error.clear();
    if (!workspace || !status_out) return fail(TRW_E_ARG, "null argument");
    hipStream_t stream = (hipStream_t)hip_stream;
    int64_t host[4] = {0, 0, 0, 0};
    hipError_t e = hipMemcpyAsync(host, workspace, sizeof host, hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) return fail_hip(e, "hipMemcpyAsync(status)");
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return fail_hip(e, "hipStreamSynchronize");
    for (int k = 0; k < 4; ++k) status_out[k] = host[k];
    return TRW_OK;
}

}  // extern "C"
