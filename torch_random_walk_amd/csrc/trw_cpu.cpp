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

// ---- host legs of the rows next to the CSR walk -------------------------------------------------------------
// What the reference's dispatcher does with CPU tensors for its other six entry points (csrc/rw_init.cpp:27-131):
// the csrc/cpu algorithms on ONE thread, i.e. on the one glibc rand() stream srand(seed) starts, consumed in the
// reference's order.  Written from the semantics this library's own kernels implement (trw_kernels_*.hpp cite the
// reference lines); nothing here is shared with the test oracle.
namespace trw_cpu {
namespace {

// sample_int(lo, hi) of the reference (csrc/cpu/cpu_utils.cpp:3-10): inclusive range, NO draw for a single value
inline int64_t sample_int(GlibcRand &rng, int64_t lo, int64_t hi)
{
    if (lo == hi) return lo;
    return lo + (int64_t)rng() % (hi - lo + 1);
}

struct Ranges {  // an inclusive (first, last) index per node, (-1, -1) for a node without rows
    const int64_t *index;
    int64_t n_nodes, n_rows;
    bool of(int64_t node, int64_t &first, int64_t &last, Fault &fault) const
    {
        first = last = -1;
        if ((uint64_t)node >= (uint64_t)n_nodes) return false;
        const int64_t a = index[2 * node], b = index[2 * node + 1];
        if (a == -1 || b == -1) return false;
        if (a < 0 || b < a || b >= n_rows) {
            fault.raise(TRW_ST_BAD_CSR);
            return false;
        }
        first = a;
        last = b;
        return true;
    }
};

}  // namespace
}  // namespace trw_cpu

extern "C" int trw_walk_edge_list_cpu(const int64_t *edges, int64_t n_edges, const int64_t *node_edge_index,
                                      int64_t n_nodes, const int64_t *targets, int64_t n_walkers, double p, double q,
                                      int32_t walk_length, int32_t seed, int64_t padding_idx, int32_t restart,
                                      int64_t *out)
{
    using namespace trw_cpu;
    set_last_error("");
    if (n_edges < 0 || n_nodes < 0 || n_walkers < 0) return fail(TRW_E_ARG, "negative size argument");
    if (walk_length < 1) return fail(TRW_E_ARG, "walk_length must be >= 1");
    if (!(p > 0.0) || !(q > 0.0) || !std::isfinite(p) || !std::isfinite(q))
        return fail(TRW_E_ARG, "p and q must be positive and finite");
    if ((n_edges > 0 && !edges) || (n_nodes > 0 && !node_edge_index)) return fail(TRW_E_ARG, "edge list / index is null");
    if (n_walkers > 0 && (!targets || !out)) return fail(TRW_E_ARG, "targets/out is null");
    const bool uniform = (p == 1.0 && q == 1.0);  // rw_cpu_edge_list.cpp:260
    const Bias bias = make_bias(p, q);
    const Ranges ranges = {node_edge_index, n_nodes, n_edges};
    const int64_t pad = padding_idx, cols = (int64_t)walk_length + 1;
    Fault fault;
    GlibcRand rng(seed, 0);
    auto tail_of = [&](int64_t e) {
        const int64_t t = edges[2 * e + 1];
        if (t != pad && (uint64_t)t >= (uint64_t)n_nodes) fault.raise(TRW_ST_NODE_RANGE);
        return t;
    };
    for (int64_t i = 0; i < n_walkers && !fault.any(); ++i) {
        int64_t *row = out + i * cols;
        const int64_t start = targets[i];
        row[0] = start;
        if (start != pad && (uint64_t)start >= (uint64_t)n_nodes) {
            fault.raise(TRW_ST_NODE_RANGE);
            break;
        }
        const int64_t jump = restart ? start : pad;  // rw_cpu_edge_list.cpp:103-108
        // sample_neighbor (:8-35): from the padding id to the jump node; a dead end steps onto the padding id
        auto sample = [&](int64_t node, int64_t first, int64_t last) {
            if (node == pad) return jump;
            if (first < 0) return pad;
            return tail_of(sample_int(rng, first, last));
        };
        int64_t here = start, first, last;
        if (uniform) {
            for (int32_t s = 1; s <= walk_length; ++s) {
                ranges.of(here, first, last, fault);
                here = sample(here, first, last);
                row[s] = here;
            }
            continue;
        }
        int64_t back = start, back_first, back_last;
        ranges.of(back, back_first, back_last, fault);
        here = sample(back, back_first, back_last);  // :188
        row[1] = here;
        for (int32_t s = 2; s <= walk_length && !fault.any(); ++s) {
            ranges.of(here, first, last, fault);
            int64_t taken = here;
            for (uint32_t tries = 0;; ++tries) {
                const int64_t x = sample(here, first, last);
                const double u = (double)rng() / (double)kRandMax;  // :198, always drawn
                bool take = false;
                taken = x;
                if (x == back && u < bias.accept_back) {  // :204-209
                    take = true;
                } else if (x == pad) {  // :212-217
                    take = u < bias.accept_back;
                    taken = jump;
                } else {  // :220-231 -- is_neighbor scans back's edges [first, last): never the last one (:51)
                    bool near = false;
                    if (u < bias.band_lo) {
                        take = true;
                    } else if (u < bias.band_hi) {
                        for (int64_t e = back_first; back_first >= 0 && e < back_last; ++e)
                            if (edges[2 * e + 1] == x) {
                                near = true;
                                break;
                            }
                        take = u < (near ? bias.accept_near : bias.accept_far);
                    }
                }
                if (take) break;
                if (tries + 1 >= kTrialLimit) {
                    fault.raise(TRW_ST_TRIAL_CAP);
                    break;
                }
            }
            row[s] = taken;
            back = here;
            back_first = first;
            back_last = last;
            here = taken;
        }
    }
    if (fault.any()) return fail(TRW_E_DATA, fault_text(fault.bits.load()));
    return TRW_OK;
}

extern "C" int trw_walk_triples_cpu(const int64_t *triples, int64_t n_triples, const int64_t *rt_index,
                                    int64_t n_entities, const int64_t *targets, int64_t n_walkers, int32_t walk_length,
                                    int64_t padding_idx, int32_t seed, int64_t *out)
{
    using namespace trw_cpu;
    set_last_error("");
    if (n_triples < 0 || n_entities < 0 || n_walkers < 0) return fail(TRW_E_ARG, "negative size argument");
    if (walk_length < 1) return fail(TRW_E_ARG, "walk_length must be >= 1");
    if ((n_triples > 0 && !triples) || (n_entities > 0 && !rt_index)) return fail(TRW_E_ARG, "triples / index is null");
    if (n_walkers > 0 && (!targets || !out)) return fail(TRW_E_ARG, "targets/out is null");
    const Ranges ranges = {rt_index, n_entities, n_triples};
    const int64_t pad = padding_idx, cols = 2 * (int64_t)walk_length + 1;
    Fault fault;
    GlibcRand rng(seed, 0);
    for (int64_t i = 0; i < n_walkers && !fault.any(); ++i) {
        int64_t *row = out + i * cols;
        int64_t node = targets[i];
        row[0] = node;
        for (int32_t h = 0; h < walk_length; ++h) {  // rw_cpu_triples.cpp:11-46: (relation, tail) of a uniform out-triple
            int64_t rel = pad, tail = pad, first, last;
            if (node != pad) {
                if ((uint64_t)node >= (uint64_t)n_entities) {
                    fault.raise(TRW_ST_NODE_RANGE);
                    break;
                }
                if (ranges.of(node, first, last, fault)) {
                    const int64_t e = sample_int(rng, first, last);
                    rel = triples[3 * e + 1];
                    tail = triples[3 * e + 2];
                }
            }
            row[2 * h + 1] = rel;
            row[2 * h + 2] = tail;
            node = tail;
        }
    }
    if (fault.any()) return fail(TRW_E_DATA, fault_text(fault.bits.load()));
    return TRW_OK;
}

namespace trw_cpu {
namespace {
int windows_args(const int64_t *walks, int64_t num_walks, int64_t cols, int32_t window_size)
{
    if (num_walks < 0) return fail(TRW_E_ARG, "negative size argument");
    if (cols < 1 || window_size < 1) return fail(TRW_E_ARG, "window_size and the walk width must be >= 1");
    if (num_walks > 0 && !walks) return fail(TRW_E_ARG, "walks is null");
    return TRW_OK;
}

// centre and rest of every window of every walk (windows_cpu.cpp:38-60): mid = window_size / 2
void gather_windows(const int64_t *walks, int64_t num_walks, int64_t cols, int32_t window_size, int64_t *centre,
                    int64_t *rest)
{
    const int64_t per_walk = cols - window_size + 1, width = window_size - 1;
    const int32_t mid = window_size / 2;
    for (int64_t w = 0; w < num_walks * per_walk; ++w) {
        const int64_t *win = walks + (w / per_walk) * cols + w % per_walk;
        centre[w] = win[mid];
        for (int32_t k = 0; k < width; ++k) rest[w * width + k] = win[k < mid ? k : k + 1];
    }
}

// target triple and the 2*window_size rows around it (windows_cpu.cpp:190-276), with the reference's walk[rel] in
// the head slot of left rows (:221) kept as it is
void gather_triple_windows(const int64_t *walks, int64_t num_walks, int64_t cols, int32_t window_size, int64_t pad,
                           int64_t *targets3, int64_t *rows)
{
    const int64_t per_walk = (cols - 1) / 2, slots = 2 * (int64_t)window_size;
    for (int64_t w = 0; w < num_walks * per_walk; ++w) {
        const int64_t *walk = walks + (w / per_walk) * cols;
        const int64_t tri = 2 * (w % per_walk) + 1;
        for (int c = 0; c < 3; ++c) targets3[w * 3 + c] = walk[tri - 1 + c];
        int64_t *win = rows + w * slots * 3;
        for (int64_t hop = 0; hop < window_size; ++hop) {
            const int64_t rel = tri - (hop + 1) * 2;
            win[hop * 3 + 0] = rel - 1 >= 0 ? walk[rel] : pad;
            win[hop * 3 + 1] = rel >= 0 ? walk[rel] : pad;
            win[hop * 3 + 2] = rel + 1 >= 0 ? walk[rel + 1] : pad;
            const int64_t rr = tri + (hop + 1) * 2;
            int64_t *o = win + (hop + window_size) * 3;
            o[0] = rr - 1 < cols ? walk[rr - 1] : pad;
            o[1] = rr < cols ? walk[rr] : pad;
            o[2] = rr + 1 < cols ? walk[rr + 1] : pad;
        }
    }
}
}  // namespace
}  // namespace trw_cpu

extern "C" int trw_to_windows_cpu(const int64_t *walks, int64_t num_walks, int64_t walk_cols, int32_t window_size,
                                  int64_t num_nodes, int32_t seed, int32_t cbow, int64_t *centre, int64_t *rest,
                                  int64_t *neg)
{
    using namespace trw_cpu;
    set_last_error("");
    if (int rc = windows_args(walks, num_walks, walk_cols, window_size)) return rc;
    if (window_size > walk_cols) return fail(TRW_E_ARG, "window_size must be between 1 and the walk width");
    if (num_nodes < 1) return fail(TRW_E_ARG, "num_nodes must be >= 1 (the reference computes rand() % num_nodes)");
    const int64_t total = num_walks * (walk_cols - window_size + 1), width = window_size - 1;
    if (total == 0) return TRW_OK;
    if (!centre || (cbow && !neg) || (width > 0 && (!rest || !neg))) return fail(TRW_E_ARG, "output pointer is null");
    gather_windows(walks, num_walks, walk_cols, window_size, centre, rest);
    GlibcRand rng(seed, 0);
    if (!cbow) {  // window_size-1 negatives per window, each rand() % num_nodes (windows_cpu.cpp:62-68)
        for (int64_t k = 0; k < total * width; ++k) neg[k] = (int64_t)rng() % num_nodes;
    } else {  // one negative per window, redrawn while it equals the positive node, 101 redraws at most (:129-135)
        for (int64_t w = 0; w < total; ++w) {
            int64_t x = sample_int(rng, 0, num_nodes - 1);
            for (int checks = 0; x == centre[w] && checks <= 100; ++checks) x = sample_int(rng, 0, num_nodes - 1);
            neg[w] = x;
        }
    }
    return TRW_OK;
}

extern "C" int trw_to_windows_triples_cpu(const int64_t *walks, int64_t num_walks, int64_t walk_cols,
                                          int32_t window_size, int64_t padding_idx, const int64_t *triples,
                                          int64_t num_triples, int32_t seed, int32_t cbow, int64_t *target_triples,
                                          int64_t *pos_windows, int64_t *neg)
{
    using namespace trw_cpu;
    set_last_error("");
    if (int rc = windows_args(walks, num_walks, walk_cols, window_size)) return rc;
    if (num_triples < 1 || !triples) return fail(TRW_E_ARG, "triples must hold at least one row");
    const int64_t total = num_walks * ((walk_cols - 1) / 2), slots = 2 * (int64_t)window_size;
    if (total == 0) return TRW_OK;
    if (!target_triples || !pos_windows || !neg) return fail(TRW_E_ARG, "output pointer is null");
    gather_triple_windows(walks, num_walks, walk_cols, window_size, padding_idx, target_triples, pos_windows);
    GlibcRand rng(seed, 0);
    auto copy_row = [&](int64_t *dst, int64_t r) {
        dst[0] = triples[3 * r];
        dst[1] = triples[3 * r + 1];
        dst[2] = triples[3 * r + 2];
    };
    if (!cbow) {  // 2*window_size drawn rows of the pool per window (windows_cpu.cpp:279-290)
        for (int64_t k = 0; k < total * slots; ++k) copy_row(neg + 3 * k, sample_int(rng, 0, num_triples - 1));
    } else {  // one drawn row, redrawn while it equals the positive triple (:374-389)
        for (int64_t w = 0; w < total; ++w) {
            const int64_t *pt = target_triples + 3 * w;
            int64_t r = sample_int(rng, 0, num_triples - 1);
            for (int checks = 0; triples[3 * r] == pt[0] && triples[3 * r + 1] == pt[1] && triples[3 * r + 2] == pt[2] &&
                                 checks <= 100; ++checks)
                r = sample_int(rng, 0, num_triples - 1);
            copy_row(neg + 3 * w, r);
        }
    }
    return TRW_OK;
}

// ---- wire format of the multi-GPU gather, host legs (CPU tensors through dist.walk_sharded) ------------------
extern "C" int trw_rows_narrow_cpu(const int64_t *rows, int64_t count, uint32_t *wire, int64_t *overflow)
{
    using namespace trw_cpu;
    set_last_error("");
    if (count < 0 || (count > 0 && (!rows || !wire))) return fail(TRW_E_ARG, "null pointer or negative count");
    uint64_t high = 0;
    for (int64_t i = 0; i < count; ++i) {
        const uint64_t v = (uint64_t)rows[i];
        high |= v >> 32;
        wire[i] = (uint32_t)v;
    }
    if (overflow && high) *overflow |= 1;
    return TRW_OK;
}

extern "C" int trw_rows_widen_cpu(const uint32_t *wire, int64_t count, int64_t *rows)
{
    using namespace trw_cpu;
    set_last_error("");
    if (count < 0 || (count > 0 && (!rows || !wire))) return fail(TRW_E_ARG, "null pointer or negative count");
    for (int64_t i = 0; i < count; ++i) rows[i] = (int64_t)wire[i];
    return TRW_OK;
}
