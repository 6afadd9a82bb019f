// trw_cpu.cpp -- the host leg of the CSR walk: trw_walk_csr_cpu (include/trw.h).
//
// The reference's dispatcher sends CPU tensors to walk_cpu (csrc/rw_init.cpp:20-24 ->
// csrc/cpu/rw_cpu.cpp:203-226); its README and its own tests call the walk that way.  This is that
// leg of the drop-in: product code, compiled into libtrw_hip.so next to the kernels, sharing with them
// the stream arithmetic of glibc_jump.hpp and nothing with the test oracle.
//
// What it computes is what csrc/cpu computes when it runs on one thread (the only regime in which the
// reference's output is defined): every draw comes from the ONE glibc rand() stream that
// srand(seed) starts, consumed walker-major.  Uniform walks use exactly one draw per step, so a thread
// that owns walkers [lo, hi) jumps straight to draw (walker_base+lo)*L and the result does not depend on
// the thread count; a biased walk's draw positions depend on every earlier walker's rejections, so its
// glibc form is one serial chain on one thread.  TRW_RNG_WALKER runs the per-walker Philox stream of the
// HIP kernels on the host threads (same rows as the GPU, any thread count).
#include <atomic>
#include <cmath>
#include <cstdint>
#include <string>
#include <thread>
#include <vector>

#include "../../include/trw.h"
#include "glibc_jump.hpp"

namespace trw_cpu {

// error text of the calling thread lives in trw_hip.hip (trw_last_error); this sets it
void set_last_error(const std::string &msg);

namespace {

constexpr uint32_t kRandMax = 2147483647u;
constexpr uint32_t kTrialLimit = 1u << 20;  // same cap as the kernels (TRW_ST_TRIAL_CAP)

struct Csr {
    const int64_t *row_ptr;
    const int64_t *col_idx;
    int64_t n_nodes, nnz;
};

// Outcome of a batch of walkers: 0 or a TRW_ST_* bit describing the first data fault met.
struct Fault {
    std::atomic<int> bits{0};
    void raise(int b) { bits.fetch_or(b, std::memory_order_relaxed); }
    bool any() const { return bits.load(std::memory_order_relaxed) != 0; }
};

// glibc TYPE_3: r[k] = r[k-31] + r[k-3] over a 31-word ring, value r[k] >> 1.
class GlibcRand {
public:
    GlibcRand(int32_t seed, uint64_t first_draw)
    {
        trw_glibc_window_at(seed, first_draw, ring_);
        at_ = 0;
    }
    uint32_t operator()()
    {
        const int lag = at_ + (trw::kGlDeg - trw::kGlSep);
        const uint32_t v = ring_[at_] + ring_[lag >= trw::kGlDeg ? lag - trw::kGlDeg : lag];
        ring_[at_] = v;
        at_ = at_ + 1 == trw::kGlDeg ? 0 : at_ + 1;
        return v >> 1;
    }

private:
    uint32_t ring_[trw::kGlDeg];
    int at_;
};

// Philox4x32-10, key (seed, 0), counter (walker_lo, walker_hi, block, 0); draw d = word[d % 4] >> 1 of block d / 4.
class WalkerRand {
public:
    WalkerRand(uint32_t seed, uint64_t walker) : seed_(seed), walker_(walker), next_(0) {}
    uint32_t operator()()
    {
        const uint32_t k = (uint32_t)(next_ & 3u);
        if (k == 0) refill((uint32_t)(next_ >> 2));
        ++next_;
        return word_[k] >> 1;
    }

private:
    void refill(uint32_t block)
    {
        uint32_t c0 = (uint32_t)walker_, c1 = (uint32_t)(walker_ >> 32), c2 = block, c3 = 0u;
        uint32_t k0 = seed_, k1 = 0u;
        for (int round = 0; round < 10; ++round) {
            const uint64_t a = (uint64_t)0xD2511F53u * c0, b = (uint64_t)0xCD9E8D57u * c2;
            const uint32_t n0 = (uint32_t)(b >> 32) ^ c1 ^ k0, n2 = (uint32_t)(a >> 32) ^ c3 ^ k1;
            c1 = (uint32_t)b;
            c3 = (uint32_t)a;
            c0 = n0;
            c2 = n2;
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        word_[0] = c0;
        word_[1] = c1;
        word_[2] = c2;
        word_[3] = c3;
    }
    uint32_t seed_;
    uint64_t walker_;
    uint64_t next_;
    uint32_t word_[4];
};

// The row of `v`; false (and a fault) when row_ptr does not describe one inside col_idx or it is empty --
// the reference computes rand() % 0 there (rw_cpu.cpp:20).
inline bool row_of(const Csr &g, int64_t v, int64_t &first, int64_t &count, Fault &fault)
{
    first = g.row_ptr[v];
    const int64_t end = g.row_ptr[v + 1];
    count = end - first;
    if (first < 0 || end < first || end > g.nnz) {
        fault.raise(TRW_ST_BAD_CSR);
        return false;
    }
    if (count == 0) {
        fault.raise(TRW_ST_DEGREE0);
        return false;
    }
    return true;
}

// sample_neighbor (rw_cpu.cpp:7-30): first + r % count, with the int64 promotion of rand().
inline int64_t pick(const Csr &g, int64_t first, int64_t count, uint32_t r, Fault &fault, bool &ok)
{
    const int64_t x = g.col_idx[first + (int64_t)r % count];
    if ((uint64_t)x >= (uint64_t)g.n_nodes) {
        fault.raise(TRW_ST_NODE_RANGE);
        ok = false;
    }
    return x;
}

// is_neighbor(x, t) (rw_cpu.cpp:32-56): a scan of t's row with early exit.
inline bool row_has(const Csr &g, int64_t first, int64_t count, int64_t x)
{
    for (const int64_t *e = g.col_idx + first, *end = e + count; e != end; ++e)
        if (*e == x) return true;
    return false;
}

struct Bias {
    double accept_back, accept_near, accept_far;  // prob_0, prob_1, prob_2 of rw_cpu.cpp:130-134
    double band_lo, band_hi;                      // min/max of (near, far): is_neighbor only matters in between
};

Bias make_bias(double p, double q)
{
    const double top0 = std::fmax(1.0 / p, 1);
    const double top = std::fmax(top0, 1.0 / q);
    Bias b;
    b.accept_back = 1.0 / p / top;
    b.accept_near = 1.0 / top;
    b.accept_far = 1.0 / q / top;
    b.band_lo = std::fmin(b.accept_near, b.accept_far);
    b.band_hi = std::fmax(b.accept_near, b.accept_far);
    return b;
}

// One uniform walk (rw_cpu.cpp:83-105): L draws.
template <class Rng>
bool uniform_row(const Csr &g, int64_t start, int32_t L, Rng &rng, int64_t *row, Fault &fault)
{
    row[0] = start;
    if ((uint64_t)start >= (uint64_t)g.n_nodes) {
        fault.raise(TRW_ST_NODE_RANGE);
        return false;
    }
    int64_t at = start;
    for (int32_t s = 1; s <= L; ++s) {
        int64_t first, count;
        bool ok = row_of(g, at, first, count, fault);
        if (ok) at = pick(g, first, count, rng(), fault, ok);
        if (!ok) return false;
        row[s] = at;
    }
    return true;
}

// One node2vec walk (rw_cpu.cpp:147-199): draw A picks the candidate, draw B decides, both always taken.
template <class Rng>
bool biased_row(const Csr &g, int64_t start, int32_t L, const Bias &bias, Rng &rng, int64_t *row, Fault &fault)
{
    row[0] = start;
    if ((uint64_t)start >= (uint64_t)g.n_nodes) {
        fault.raise(TRW_ST_NODE_RANGE);
        return false;
    }
    int64_t back_first, back_count;  // the row of the node two steps back (`t`)
    bool ok = row_of(g, start, back_first, back_count, fault);
    if (!ok) return false;
    int64_t back = start;
    int64_t here = pick(g, back_first, back_count, rng(), fault, ok);
    if (!ok) return false;
    row[1] = here;
    for (int32_t s = 2; s <= L; ++s) {
        int64_t first, count;
        if (!row_of(g, here, first, count, fault)) return false;
        int64_t x = here;
        for (uint32_t tries = 0;; ++tries) {
            const uint32_t ra = rng();
            const double u = (double)rng() / (double)kRandMax;
            x = pick(g, first, count, ra, fault, ok);
            if (!ok) return false;
            bool take;
            if (x == back) take = u < bias.accept_back;
            else if (u < bias.band_lo) take = true;     // below both thresholds: either class is accepted
            else if (!(u < bias.band_hi)) take = false;  // above both: either class is rejected
            else take = u < (row_has(g, back_first, back_count, x) ? bias.accept_near : bias.accept_far);
            if (take) break;
            if (tries + 1 >= kTrialLimit) {
                fault.raise(TRW_ST_TRIAL_CAP);
                return false;
            }
        }
        row[s] = x;
        back = here;
        back_first = first;
        back_count = count;
        here = x;
    }
    return true;
}

// Runs body(lo, hi) over [0, n) on up to `threads` host threads (contiguous blocks).
template <class Body>
void in_blocks(int64_t n, int threads, Body body)
{
    if (threads < 1) threads = 1;
    const int64_t min_block = 256;  // not worth a thread below this
    int64_t blocks = (n + min_block - 1) / min_block;
    if (blocks > threads) blocks = threads;
    if (blocks <= 1) {
        body((int64_t)0, n);
        return;
    }
    const int64_t per = (n + blocks - 1) / blocks;
    std::vector<std::thread> pool;
    for (int64_t b = 1; b < blocks; ++b) {
        const int64_t lo = b * per, hi = lo + per < n ? lo + per : n;
        if (lo < hi) pool.emplace_back([=]() { body(lo, hi); });
    }
    body((int64_t)0, per < n ? per : n);
    for (auto &t : pool) t.join();
}

std::string fault_text(int bits)
{
    std::string s;
    if (bits & TRW_ST_DEGREE0) s += "a walker reached a node with no out-edges (the reference divides by zero there); ";
    if (bits & TRW_ST_NODE_RANGE) s += "a start node or col_idx entry is outside [0, num_nodes); ";
    if (bits & TRW_ST_BAD_CSR) s += "row_ptr is not a monotone sequence within [0, len(col_idx)]; ";
    if (bits & TRW_ST_TRIAL_CAP) s += "a rejection loop exceeded 2^20 trials at one step; ";
    return s + "status bits " + std::to_string(bits);
}

int fail(int code, const std::string &msg)
{
    set_last_error(msg);
    return code;
}

}  // namespace
}  // namespace trw_cpu

extern "C" int trw_walk_csr_cpu(const int64_t *row_ptr, int64_t n_nodes, const int64_t *col_idx, int64_t nnz,
                                const int64_t *targets, int64_t n_walkers, int64_t walker_base, double p, double q,
                                int32_t walk_length, int32_t seed, int32_t rng_mode, int64_t *out, int32_t n_threads)
{
    using namespace trw_cpu;
    set_last_error("");
    if (n_nodes < 0 || nnz < 0 || n_walkers < 0 || walker_base < 0) return fail(TRW_E_ARG, "negative size argument");
    if (walk_length < 1) return fail(TRW_E_ARG, "walk_length must be >= 1");
    if (!(p > 0.0) || !(q > 0.0) || !std::isfinite(p) || !std::isfinite(q))
        return fail(TRW_E_ARG, "p and q must be positive and finite");
    if (!row_ptr) return fail(TRW_E_ARG, "row_ptr is null");
    if (n_walkers > 0 && (!targets || !out)) return fail(TRW_E_ARG, "targets/out is null");
    if (nnz > 0 && !col_idx) return fail(TRW_E_ARG, "col_idx is null");
    if (n_walkers > 0 && n_nodes == 0) return fail(TRW_E_ARG, "walkers on an empty graph");
    const bool uniform = (p == 1.0 && q == 1.0);  // rw_cpu.cpp:220
    // On the host AUTO is the reference's own stream for both walks: that IS what csrc/cpu computes on one thread.
    int mode = rng_mode == TRW_RNG_AUTO ? TRW_RNG_GLIBC : rng_mode;
    if (mode != TRW_RNG_GLIBC && mode != TRW_RNG_WALKER) return fail(TRW_E_RNG_MODE, "unknown rng_mode");
    if (mode == TRW_RNG_GLIBC && !uniform && walker_base != 0)
        return fail(TRW_E_RNG_MODE, "TRW_RNG_GLIBC biased walks replay one serial stream: walker_base must be 0");
    const int64_t L = walk_length;
    if (mode == TRW_RNG_GLIBC && uniform &&
        ((unsigned __int128)walker_base + (unsigned __int128)n_walkers) * (unsigned __int128)L >= ((unsigned __int128)1 << 48))
        return fail(TRW_E_RANGE, "glibc stream position beyond 2^48 draws");
    if (n_walkers == 0) return TRW_OK;

    const Csr g = {row_ptr, col_idx, n_nodes, nnz};
    const int64_t cols = L + 1;
    Fault fault;
    if (uniform && mode == TRW_RNG_GLIBC) {
        in_blocks(n_walkers, n_threads, [&](int64_t lo, int64_t hi) {
            GlibcRand rng(seed, (uint64_t)(walker_base + lo) * (uint64_t)L);
            for (int64_t i = lo; i < hi && !fault.any(); ++i)
                uniform_row(g, targets[i], walk_length, rng, out + i * cols, fault);
        });
    } else if (mode == TRW_RNG_GLIBC) {
        const Bias bias = make_bias(p, q);
        GlibcRand rng(seed, 0);
        for (int64_t i = 0; i < n_walkers && !fault.any(); ++i)
            biased_row(g, targets[i], walk_length, bias, rng, out + i * cols, fault);
    } else {
        const Bias bias = make_bias(p, q);
        in_blocks(n_walkers, n_threads, [&](int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi && !fault.any(); ++i) {
                WalkerRand rng((uint32_t)seed, (uint64_t)walker_base + (uint64_t)i);
                if (uniform) uniform_row(g, targets[i], walk_length, rng, out + i * cols, fault);
                else biased_row(g, targets[i], walk_length, bias, rng, out + i * cols, fault);
            }
        });
    }
    if (fault.any()) return fail(TRW_E_DATA, fault_text(fault.bits.load()));
    return TRW_OK;
}
