// trw_hip.hip -- MI355X (gfx950) random-walk library: host side and C ABI (include/trw.h).
//
// The hot path of Sachinx0e/torch_random_walk, written from scratch for CDNA4: the CSR walker
// behind torch_rw.rw.walk (reference csrc/cpu/rw_cpu.cpp is the behaviour, csrc/cuda/rw_cuda.cu
// the shape of the GPU entry point), and the rows next to it (triples walk, edge-list walk,
// window builders).  This is integer gather work bounded by the memory system's random-access
// rate; there is no contraction in it and therefore no MFMA.
//
// A CSR walk is two phases on the caller's stream:
//   prepare  csr_shadow_kernel (int64 CSR -> validated uint32 shadow); glibc_windows_kernel (per-lane
//            jump-ahead into rand()'s stream, uniform walks); csr_ext_kernel / csr_ext_rpos_kernel (fat
//            entries, large graphs); edge_table_build_kernel (biased walks, large batches)
//   walk     uniform_glibc_rows_kernel | biased_hash_kernel | biased_walker_kernel -- each writes the
//            caller's [N, L+1] int64 rows itself (there is no finish pass)
// Graphs beyond 2^31 nodes / 2^32 entries, the uniform walk on the per-walker stream and the serial
// reference-exact replay of a biased walk run the int64 kernels of trw_kernels_wide.hpp.  make_plan()
// decides which of these a call shape gets and where their buffers live in the caller's workspace.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/trw.h"
#include "glibc_jump.hpp"
#include "trw_kernels_wide.hpp"
#include "trw_kernels_narrow.hpp"
#include "trw_kernels_marks.hpp"
#include "trw_kernels_triples.hpp"
#include "trw_kernels_edgelist.hpp"
#include "trw_kernels_windows.hpp"
#include "trw_kernels_wire.hpp"

namespace {

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

}  // namespace

namespace trw_cpu {
// the host leg (trw_cpu.cpp) reports through the same thread-local text
void set_last_error(const std::string &msg) { t_last_error = msg; }
}  // namespace trw_cpu

namespace {

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

long env_long(const char *name, long lo, long hi, long dflt)
{
    if (const char *env = std::getenv(name)) {
        char *end = nullptr;
        const long v = std::strtol(env, &end, 10);
        if (end != env && v >= lo && v <= hi) return v;
    }
    return dflt;
}

// Walkers chained per lane in the glibc uniform kernel: enough lanes to fill the
// chip several times over, few enough that the 124-byte window per lane is noise.
int32_t glibc_chain_length(int64_t n_walkers)
{
    const long forced = env_long("TRW_GLIBC_CHAIN", 1, 64, 0);
    if (forced) return (int32_t)forced;
    const int64_t g = n_walkers >> 18;  // measured: 4 beats 1 from ~1M walkers up, no effect on large graphs
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

// ---- the prepared graph --------------------------------------------------------------------------
// Everything a walk derives from (row_ptr, col_idx) alone -- validation, the uint32 shadow, fat entries,
// the edge table -- lives in ONE blob whose layout is a function of (n_nodes, nnz, what).  A one-shot call
// builds it inside its workspace; a caller that walks a graph more than once builds it once
// (trw_prepare_csr_hip) and hands it to trw_walk_csr_prepared_hip.
struct PrepLayout {
    bool narrow = false;   // uint32 shadow kernels (n_nodes < 2^31, nnz < 2^32)
    bool ext16 = false;    // uniform: fat entries {v, start(v), deg(v)}, one gather per step
    bool biased = false;   // edge table + 32-byte fat entries
    bool marks = false;    // ... with the neighbourhood masks worked out for long rows too
    bool marks_full = false;  // ... for rows up to 1024 entries (TRW_PREP_MARKS_FULL) instead of 256
    uint32_t table_blocks = 0;  // the edge set: blocks of four 8-byte keys
    uint32_t heavy_room = 0, task_room = 0, big_task_room = 0;  // items; tasks of short tile rows / of long ones
    size_t off_row32 = 0, off_col32 = 0, off_ext16 = 0, off_table = 0, off_ext32 = 0, total = 0;
    // what a walk reads ends at `persistent`; behind it, scratch of the marking passes (offsets from the scratch's start)
    size_t persistent = 0, scratch = 0, off_heavy = 0;
    size_t off_incount = 0, off_inoff = 0, off_tasks = 0, off_tasks_big = 0;  // the tile pass: per-node in-list counters / segments, tasks
};

bool graph_fits_narrow(int64_t n_nodes, int64_t nnz)
{
    return n_nodes < ((int64_t)1 << 31) && nnz < ((int64_t)1 << 32) - 64 && env_long("TRW_FORCE_WIDE", 0, 1, 0) == 0;
}

// Blocks (of four 8-byte keys) of the edge set: `TRW_HASH_SLACK` TENTHS of a place per edge (default 25: load factor
// 0.4, 20 bytes per edge -- with 20 the whole prepared graph is 58 bytes per edge and the prepare pass 4 % slower, with
// 30 it is 66 and 4 % faster: misses, nine probes in ten of the prepare pass, end at the first block that has a free
// place), any number of blocks -- a block is picked by mulhi(hash, blocks).
uint32_t edge_table_blocks(int64_t nnz)
{
    const uint64_t tenths = (uint64_t)env_long("TRW_HASH_SLACK", 12, 160, 25);
    uint64_t blocks = (tenths * (uint64_t)nnz + 39) / 40;
    if (blocks < 256) blocks = 256;
    return blocks > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)blocks;  // (nnz < 2^32 and 3 places per edge stay below)
}

PrepLayout make_prep_layout(int64_t n_nodes, int64_t nnz, int32_t what)
{
    PrepLayout pl;
    size_t off = kStatusBytes;  // the header: validation bits and the row-order counters, written by the shadow pass
    pl.narrow = graph_fits_narrow(n_nodes, nnz) && what >= 0;
    if (pl.narrow) {
        pl.off_row32 = off;
        off += align_up((size_t)(n_nodes + 1 + 4) * sizeof(uint32_t), 256);  // +4: 16-byte gathers at the last node
        pl.off_col32 = off;
        off += align_up((size_t)nnz * sizeof(uint32_t) + 64, 256);
        if (what & TRW_PREP_UNIFORM) {
            pl.ext16 = true;
            pl.off_ext16 = off;
            off += align_up((size_t)nnz * 16 + 64, 256);
        }
        if (what & (TRW_PREP_BIASED | TRW_PREP_MARKS | TRW_PREP_MARKS_FULL)) {
            pl.biased = true;
            pl.marks = (what & (TRW_PREP_MARKS | TRW_PREP_MARKS_FULL)) != 0;
            pl.marks_full = (what & TRW_PREP_MARKS_FULL) != 0;
            pl.table_blocks = edge_table_blocks(nnz);
            pl.off_table = off;
            off += align_up((size_t)pl.table_blocks * 32 + 64, 256);
            pl.off_ext32 = off;
            off += align_up((size_t)nnz * 32 + 64, 256);
            if (pl.marks) {
                // Scratch of the marking passes (needed until trw_prepare_csr_hip's kernels have run, not by any walk).
                // Items {entry, source node} of the in-lists: the entries u -> v whose row of u is the shorter or equal one
                // and longer than the inline cap -- about a third of the entries on a power-law graph, half on a regular one,
                // ALL of them on a graph whose edges all point from shorter rows to longer ones.  Room for all of them: nothing
                // is ever dropped (round 4 had room for 3/5 and left the rest their all-ones masks without a word).
                size_t sc = 0;
                const uint64_t room = (uint64_t)nnz + 64;
                pl.heavy_room = room > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)room;
                pl.off_heavy = sc;
                sc += align_up((size_t)pl.heavy_room * 8, 256);
                // the tile pass: in-list size and segment per node, and its 8-byte task records
                pl.off_incount = sc;
                sc += align_up((size_t)(n_nodes + 1) * 4, 256);
                pl.off_inoff = sc;
                sc += align_up((size_t)(n_nodes + 1) * 4, 256);
                // Task records (64 entries of the tile row / 64 items of its in-list each), worst case.  Rows of 9 .. 255
                // entries: at most nnz/64 + nnz/64 + two per such row, of which there are fewer than nnz/9: 0.2535 nnz.
                // Longer rows: nnz/64 + nnz/64 + two per such row, fewer than nnz/256 of them: 0.0391 nnz.
                const uint64_t troom = (uint64_t)nnz / 4 + (uint64_t)nnz / 128 + 4096;
                pl.task_room = troom > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)troom;
                pl.off_tasks = sc;
                sc += align_up((size_t)pl.task_room * 8, 256);
                pl.big_task_room = (uint32_t)((uint64_t)nnz / 32 + (uint64_t)nnz / 128 + 4096);
                pl.off_tasks_big = sc;
                sc += align_up((size_t)pl.big_task_room * 8, 256);
                pl.scratch = sc;
            }
        }
    }
    pl.persistent = off;
    pl.total = off + pl.scratch;
    return pl;
}

// What one call of this shape would build for itself (each part costs a pass over the graph):
//   uniform: fat entries when the shadow does not fit the L2s and the batch takes more steps than the graph
//            has entries -- measured: 15.1 vs 22.6 ms on a 480 MB shadow (R-MAT 10M/110M), 1.14 vs 0.82 ms on
//            a 4.8 MB one (ER 100k/1.1M, cache resident); 1M walkers 3.1 ms compact / 3.2 ms fat, 2.75M 7.6 / 6.1;
//   biased:  the table + 32-byte entries from about nnz/2 steps (R-MAT 10M/110M, L=80: 344k walkers 18.5 ms
//            without / 23.7 ms with, 1M walkers 38.4 / 32.9), the bisection kernel below that.
int32_t plan_what(int64_t n_nodes, int64_t nnz, int64_t n_walkers, int32_t L, bool uniform, int32_t rng_mode)
{
    if (n_walkers <= 0 || L < 1) return TRW_PREP_NOTHING;
    // the kernels that index the caller's int64 arrays directly want no shadow: the uniform walk on the per-walker
    // stream nothing at all, the biased ones (serial replay, graphs beyond 2^31/2^32) the row order
    if (uniform && rng_mode == TRW_RNG_WALKER) return TRW_PREP_NOTHING;
    if (!graph_fits_narrow(n_nodes, nnz)) return uniform ? TRW_PREP_NOTHING : TRW_PREP_ORDER_ONLY;
    if (!uniform && rng_mode == TRW_RNG_GLIBC) {
        // The serial replay: one lane over the int64 arrays costs about 10 us per step on a large graph (dependent
        // gathers and a bisection per trial); the wave-cooperative replay over a prepared graph about 1.5 us, and
        // the prepare pass about a third of a nanosecond per entry -- worth it from nnz/16384 steps.
        const long forced = env_long("TRW_HASH", 0, 1, -1);
        const bool worth = (__int128)n_walkers * L * 16384 >= (__int128)nnz;
        // (with the masks of rows up to 256 entries: cheaper to build than the entries alone, see below, and fewer of the
        // replay's trials are left to a probe of the edge set)
        return (forced == 1 || (forced == -1 && worth)) ? (TRW_PREP_BIASED | TRW_PREP_MARKS) : TRW_PREP_ORDER_ONLY;
    }
    if (uniform) {
        const long forced = env_long("TRW_EXT", 0, 1, -1);
        const bool big = (size_t)(n_nodes + nnz) * 4 > ((size_t)32 << 20);  // the XCDs' aggregate L2
        const bool worth = big && (__int128)n_walkers * L >= (__int128)nnz;
        return (forced == 1 || (forced == -1 && worth)) ? TRW_PREP_UNIFORM : 0;
    }
    const long forced = env_long("TRW_HASH", 0, 1, -1);
    const __int128 steps = (__int128)n_walkers * L;
    const bool worth = steps >= (__int128)nnz / 2;
    if (!(forced == 1 || (forced == -1 && worth))) return 0;
    // The masks of longer rows.  Measured on R-MAT 10M/110M (round 5, prepare ms -> walk ms of 7.3 / 2.2 steps per entry):
    // entries + edge set alone, the owning thread marking rows up to 32 through the edge set: 42 -> 80 / 27; with the tile
    // pass for rows up to 256 (TRW_PREP_MARKS): 37 -> 46 / 16 -- cheaper to BUILD than without it, because the owning thread
    // then keeps only rows up to 8 --; up to 1 024 (TRW_PREP_MARKS_FULL): 46.5 -> 37.6 / 13.5.  So a call that is worth
    // the entries at all gets the first level with them, and the second from 9 steps per entry on.
    int32_t what = TRW_PREP_BIASED;
    if (steps >= (__int128)nnz * env_long("TRW_PLAN_MARKS_AT", 0, 1 << 20, 0)) what |= TRW_PREP_MARKS;
    if (steps >= (__int128)nnz * env_long("TRW_PLAN_FULL_AT", 0, 1 << 20, 9)) what |= TRW_PREP_MARKS | TRW_PREP_MARKS_FULL;
    return what;
}

// Per-call scratch of a walk over a prepared graph: the status block and the uniform walk's RNG windows.
struct CallLayout {
    bool windows = false;
    int32_t G = 1;  // walkers chained per lane (uniform glibc)
    int64_t nlanes = 0;
    size_t off_win = 0, total = 0;
};

CallLayout make_call_layout(int64_t n_walkers, int32_t L, bool uniform, int32_t rng_mode)
{
    CallLayout cl;
    size_t off = kStatusBytes;
    if (n_walkers > 0 && L >= 1) {
        cl.G = uniform ? glibc_chain_length(n_walkers) : 1;
        cl.nlanes = (n_walkers + cl.G - 1) / cl.G;
        if (uniform && rng_mode != TRW_RNG_WALKER) {
            cl.windows = true;
            cl.off_win = off;
            off += align_up((size_t)cl.nlanes * kGlDeg * sizeof(uint32_t), 256);
        }
    }
    cl.total = off;
    return cl;
}

// Optional instrumentation (trw_profile_enable): events around the three
// phases of a call, recorded on the caller's stream.
struct Profile {
    bool on = false;
    bool count = false;
    bool valid = false;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t prep[2] = {nullptr, nullptr};  // around the last trw_prepare_csr_hip
    bool prep_valid = false;
};
Profile g_prof;

void prof_mark(int k, hipStream_t stream)
{
    if (!g_prof.on) return;
    if (!g_prof.ev[k] && hipEventCreate(&g_prof.ev[k]) != hipSuccess) return;
    (void)hipEventRecord(g_prof.ev[k], stream);
    if (k == 3) g_prof.valid = true;
}

int device_cu_count()
{
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cus[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cus[dev] = v;
    }
    return cus[dev];
}

unsigned stream_grid(int64_t items)
{
    int64_t blocks = (items + kBlock - 1) / kBlock;
    const int64_t cap = 256 * 32;  // 256 CUs x 8 blocks, x4 for balance; the rest is grid-stride
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

// Buffers are cleared by a kernel of this library, never by hipMemsetAsync: a memset node captured into a
// HIP graph runs on the first launch of the graph only (ROCm 7.2, observed; tests/test_gpu_parity.py
// test_walk_is_graph_capturable), and a call has to be replayable.  `bytes` and `dst` are multiples of 16.
__global__ void __launch_bounds__(kBlock) fill_kernel(uint4 *__restrict__ dst, uint32_t word, size_t n16)
{
    const uint4 v = {word, word, word, word};
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n16; i += stride) dst[i] = v;
}

// The status block of a call over a prepared graph starts as a copy of the prepared header's findings
// (validation bits in word 0, row order in status[1] / the descent counters); everything else zero.
__global__ void status_seed_kernel(const int64_t *__restrict__ header, int64_t *__restrict__ status)
{
    const int i = threadIdx.x;
    if (i < kStatusBytes / 8) status[i] = (i == 0 || i == 1 || i == kStDescents || i == kStBoundary) ? header[i] : 0;
}

void device_fill(void *dst, uint32_t word, size_t bytes, hipStream_t stream)
{
    const size_t n16 = bytes / 16;
    fill_kernel<<<stream_grid((int64_t)n16), kBlock, 0, stream>>>((uint4 *)dst, word, n16);
}

}  // namespace

namespace {
struct TriplesPlan {
    bool narrow = false;
    size_t off_rt32 = 0, off_ix32 = 0, total = 0;
};

TriplesPlan make_triples_plan(int64_t n_triples, int64_t n_entities, int64_t n_walkers, int32_t L,
                              int64_t padding_idx, int32_t rng_mode)
{
    TriplesPlan pl;
    size_t off = kStatusBytes;
    const bool fits = n_entities < ((int64_t)1 << 31) && n_triples < ((int64_t)1 << 32) - 64 && padding_idx >= 0 &&
                      padding_idx < ((int64_t)1 << 31) && env_long("TRW_FORCE_WIDE", 0, 1, 0) == 0;
    // (L < 2^21: triples_walker_kernel addresses a wave's 64 rows with a 32-bit byte offset)
    pl.narrow = fits && rng_mode != TRW_RNG_GLIBC && n_walkers > 0 && L >= 1 && L < (1 << 21);
    if (pl.narrow) {
        pl.off_rt32 = off;
        off += align_up((size_t)n_triples * 8 + 64, 256);
        pl.off_ix32 = off;
        off += align_up((size_t)n_entities * 8 + 64, 256);
    }
    pl.total = off;
    return pl;
}

int resolve_triples_mode(int32_t rng_mode, int64_t n_walkers, int64_t walker_base, int32_t L)
{
    if (rng_mode == TRW_RNG_GLIBC || rng_mode == TRW_RNG_WALKER) return rng_mode;
    if (rng_mode != TRW_RNG_AUTO) return -1;
    // the reference is deterministic up to its parallel_for grain of 32768 walkers
    // (rw_cpu_triples.cpp:67); the serial replay is one lane, so it is also kept to small jobs
    return (walker_base == 0 && n_walkers <= 32768 && n_walkers * (int64_t)L <= ((int64_t)1 << 18)) ? TRW_RNG_GLIBC
                                                                                                    : TRW_RNG_WALKER;
}
}  // namespace

namespace {
// The prepared edge list (cf. PrepLayout): shadow, edge table, 64-byte entries -- a function of (sizes, padding id, what).
struct EdgeListLayout {
    bool narrow = false;
    bool hash = false;
    uint32_t hash_mask = 0;
    bool ent = false;  // 64-byte per-edge entries: the walker-pool kernel
    size_t off_tail32 = 0, off_ix32 = 0, off_table = 0, off_ent = 0, off_nd = 0, total = 0;
};

bool elist_fits_narrow(int64_t n_edges, int64_t n_nodes, int64_t padding_idx)
{
    return n_nodes < ((int64_t)1 << 31) && n_edges < ((int64_t)1 << 32) - 64 && padding_idx >= 0 &&
           padding_idx < ((int64_t)1 << 31) && env_long("TRW_FORCE_WIDE", 0, 1, 0) == 0;
}

EdgeListLayout make_elist_layout(int64_t n_edges, int64_t n_nodes, int64_t padding_idx, int32_t what)
{
    EdgeListLayout pl;
    size_t off = kStatusBytes;
    pl.narrow = what >= 0 && elist_fits_narrow(n_edges, n_nodes, padding_idx);
    if (pl.narrow) {
        pl.off_tail32 = off;
        off += align_up((size_t)n_edges * 4 + 64, 256);
        pl.off_ix32 = off;
        off += align_up((size_t)n_nodes * 8 + 64, 256);
        uint64_t cap = 1024;
        while (cap < 3 * (uint64_t)n_edges) cap <<= 1;
        if ((what & (TRW_ELPREP_TABLE | TRW_ELPREP_ENTRIES)) && cap <= ((uint64_t)1 << 31)) {
            pl.hash = true;
            pl.hash_mask = (uint32_t)(cap - 1);
            pl.off_table = off;
            off += (size_t)cap * 16;
            if (what & TRW_ELPREP_ENTRIES) {
                pl.ent = true;
                pl.off_ent = off;
                off += align_up((size_t)n_edges * 64 + 64, 256);
                pl.off_nd = off;
                off += align_up((size_t)n_nodes * 16 + 64, 256);
            }
        }
    }
    pl.total = off;
    return pl;
}

// What one call of this shape is worth preparing (`mode`: resolved, or the caller's for the upper bound).
int32_t elist_plan_what(int64_t n_edges, int64_t n_nodes, int64_t n_walkers, int32_t L, int64_t padding_idx,
                        int32_t mode, bool uniform)
{
    // (L < 2^22: elist_walk_kernel addresses a wave's 64 rows with a 32-bit byte offset)
    if (n_walkers <= 0 || L < 1 || L >= (1 << 22) || !elist_fits_narrow(n_edges, n_nodes, padding_idx)) return TRW_PREP_NOTHING;
    if (uniform) return mode == TRW_RNG_GLIBC ? TRW_PREP_NOTHING : 0;
    // Build the edge table unless the batch is tiny next to the graph: without it is_neighbor is the reference's
    // linear scan of an unsorted edge range (hubs: on R-MAT 1M/10M, 80 steps, 16 walkers took 168 ms that way, 256
    // walkers 500 ms; 5.6 ms with the table, which costs ~0.15 ns per edge).  The serial replay wants it too.
    const long forced = env_long("TRW_HASH", 0, 1, -1);
    const bool worth = (__int128)n_walkers * L * 65536 >= (__int128)n_edges;
    if (!(forced == 1 || (forced == -1 && worth))) return 0;
    // the entries cost a pass with about 20 table probes per edge (13 ms on R-MAT 1M/10M): from n_edges/4 steps on
    const long pforced = env_long("TRW_ELPOOL", 0, 1, -1);
    const bool pworth = (__int128)n_walkers * L * 4 >= (__int128)n_edges;
    if (mode != TRW_RNG_GLIBC && n_walkers < ((int64_t)1 << 32) && (pforced == 1 || (pforced == -1 && pworth)))
        return TRW_ELPREP_TABLE | TRW_ELPREP_ENTRIES;
    return TRW_ELPREP_TABLE;
}

int resolve_elist_mode(int32_t rng_mode, bool uniform, int64_t n_walkers, int64_t walker_base, int32_t L)
{
    if (rng_mode == TRW_RNG_GLIBC || rng_mode == TRW_RNG_WALKER) return rng_mode;
    if (rng_mode != TRW_RNG_AUTO) return -1;
    // rw_cpu_edge_list.cpp:83 / :149: the reference is deterministic up to its parallel_for grain; the
    // serial replay is one lane, so it is also kept to small jobs
    const int64_t grain = uniform ? 32768 : 32768 / ((int64_t)L + 1);
    return (walker_base == 0 && n_walkers <= grain && n_walkers * (int64_t)L <= ((int64_t)1 << 18)) ? TRW_RNG_GLIBC
                                                                                                   : TRW_RNG_WALKER;
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

int32_t trw_prepare_plan(int64_t n_nodes, int64_t nnz, int64_t n_walkers, int32_t walk_length, double p, double q,
                         int32_t rng_mode)
{
    if (n_nodes < 0 || nnz < 0) return 0;
    const bool uniform = (p == 1.0 && q == 1.0);
    const int mode = resolve_mode(rng_mode, uniform, n_walkers, 0, walk_length);
    return plan_what(n_nodes, nnz, n_walkers, walk_length, uniform, mode < 0 ? TRW_RNG_AUTO : mode);
}

size_t trw_prepare_csr_bytes(int64_t n_nodes, int64_t nnz, int32_t what)
{
    if (n_nodes < 0 || nnz < 0) return kStatusBytes;
    return make_prep_layout(n_nodes, nnz, what).total;
}

int trw_prepare_csr_region(int64_t n_nodes, int64_t nnz, int32_t what, int32_t which, size_t *offset, size_t *bytes)
{
    if (n_nodes < 0 || nnz < 0 || !offset || !bytes) return fail(TRW_E_ARG, "bad argument");
    const PrepLayout pl = make_prep_layout(n_nodes, nnz, what);
    *offset = 0;
    *bytes = 0;
    switch (which) {
    case TRW_REGION_HEADER: *bytes = kStatusBytes; break;
    case TRW_REGION_ROW32: if (pl.narrow) { *offset = pl.off_row32; *bytes = (size_t)(n_nodes + 1) * 4; } break;
    case TRW_REGION_COL32: if (pl.narrow) { *offset = pl.off_col32; *bytes = (size_t)nnz * 4; } break;
    case TRW_REGION_EXT16: if (pl.ext16) { *offset = pl.off_ext16; *bytes = (size_t)nnz * 16; } break;
    case TRW_REGION_TABLE: if (pl.biased) { *offset = pl.off_table; *bytes = (size_t)pl.table_blocks * 32; } break;
    case TRW_REGION_EXT32: if (pl.biased) { *offset = pl.off_ext32; *bytes = (size_t)nnz * 32; } break;
    case TRW_REGION_TASKS: if (pl.marks) { *offset = pl.persistent + pl.off_tasks; *bytes = pl.off_tasks_big + (size_t)pl.big_task_room * 8 - pl.off_tasks; } break;
    default: return fail(TRW_E_ARG, "unknown region");
    }
    return TRW_OK;
}

size_t trw_prepare_csr_persistent_bytes(int64_t n_nodes, int64_t nnz, int32_t what)
{
    if (n_nodes < 0 || nnz < 0) return kStatusBytes;
    return make_prep_layout(n_nodes, nnz, what).persistent;
}

size_t trw_prepare_csr_scratch_bytes(int64_t n_nodes, int64_t nnz, int32_t what)
{
    if (n_nodes < 0 || nnz < 0) return 0;
    return make_prep_layout(n_nodes, nnz, what).scratch;
}

int trw_prepare_csr_hip(const int64_t *row_ptr, int64_t n_nodes, const int64_t *col_idx, int64_t nnz, int32_t what,
                        void *prepared, size_t prepared_bytes, void *hip_stream)
{
    return trw_prepare_csr_scratch_hip(row_ptr, n_nodes, col_idx, nnz, what, prepared, prepared_bytes, nullptr, 0, hip_stream);
}

int trw_prepare_csr_scratch_hip(const int64_t *row_ptr, int64_t n_nodes, const int64_t *col_idx, int64_t nnz, int32_t what,
                                void *prepared, size_t prepared_bytes, void *scratch, size_t scratch_bytes, void *hip_stream)
{
    t_last_error.clear();
    if (n_nodes < 0 || nnz < 0) return fail(TRW_E_ARG, "negative size argument");
    if (!row_ptr) return fail(TRW_E_ARG, "row_ptr is null");
    if (nnz > 0 && !col_idx) return fail(TRW_E_ARG, "col_idx is null");
    if (!prepared || ((uintptr_t)prepared & 15u) != 0) return fail(TRW_E_WORKSPACE, "prepared buffer missing or not 16-byte aligned");
    const PrepLayout pl = make_prep_layout(n_nodes, nnz, what);
    // the marking passes' scratch: the caller's, or the tail of the prepared buffer
    if (scratch) {
        if (((uintptr_t)scratch & 15u) != 0 || scratch_bytes < pl.scratch) return fail(TRW_E_WORKSPACE, "scratch misaligned or too small for this graph");
        if (prepared_bytes < pl.persistent) return fail(TRW_E_WORKSPACE, "prepared buffer too small for this graph");
    } else if (prepared_bytes < pl.total) {
        return fail(TRW_E_WORKSPACE, "prepared buffer too small for this graph (it holds the marking passes' scratch too)");
    }
    hipStream_t stream = (hipStream_t)hip_stream;
    char *ws = (char *)prepared;
    char *sc = scratch ? (char *)scratch : ws + pl.persistent;
    int64_t *header = (int64_t *)ws;
    if (g_prof.on && (g_prof.prep[0] || hipEventCreate(&g_prof.prep[0]) == hipSuccess))
        (void)hipEventRecord(g_prof.prep[0], stream);
    device_fill(header, 0u, kStatusBytes, stream);
    if (!pl.narrow) {  // int64 kernels: only the row order is worth knowing ahead (may is_neighbor bisect?)
        if (n_nodes > 0 && what != TRW_PREP_NOTHING) {
            int64_t blocks = (n_nodes + kBlock - 1) / kBlock;
            if (blocks > 256 * 16) blocks = 256 * 16;
            csr_rows_sorted_kernel<<<(unsigned)blocks, kBlock, 0, stream>>>(row_ptr, col_idx, n_nodes, nnz, header);
        }
    } else {
        uint32_t *row32 = (uint32_t *)(ws + pl.off_row32);
        uint32_t *col32 = (uint32_t *)(ws + pl.off_col32);
        const unsigned grid = stream_grid(nnz > n_nodes ? nnz : n_nodes + 1);
        csr_shadow_kernel<true><<<grid, kBlock, 0, stream>>>(row_ptr, col_idx, n_nodes, nnz, row32, col32, header);
        if (pl.ext16)
            csr_ext_kernel<<<stream_grid(nnz), kBlock, 0, stream>>>(row32, col32, nnz, (u32x4_a16 *)(ws + pl.off_ext16));
        if (pl.biased) {
            uint32_t *table = (uint32_t *)(ws + pl.off_table);
            device_fill(table, 0xFFFFFFFFu, (size_t)pl.table_blocks * 32 + 64, stream);
            edge_table_build_kernel<<<stream_grid((nnz + 7) / 8), kBlock, 0, stream>>>(row32, col32, n_nodes, nnz, table,
                                                                                     pl.table_blocks, header);
            // The entries and their masks (trw_kernels_marks.hpp).  The owning thread settles the entries of rows of u of up
            // to 8 entries (32 when no tile pass follows, and then also those whose row of v is that short); the tile pass -- a
            // Bloom filter in front of the bisection -- everything else whose shorter row has up to 256 / 1 024 entries.
            // Measured on R-MAT 10M/110M, full prepare: round 4's form (table probes for shorter rows <= 8, every element of
            // the longer ones bisected into an 8 192-entry tile) 68.1 ms; with the filter and rows of v of any length riding
            // along in u's tile 55.0 (profiles/r05_prepare_ab.txt).
            const uint32_t inline_cap = (uint32_t)env_long("TRW_COMMON_MAX", 0, 4096, pl.marks ? 8 : 32);
            const uint32_t heavy_cap =
                pl.marks ? (uint32_t)env_long("TRW_HEAVY_MAX", 0, 1 << 20, pl.marks_full ? 1024 : 256) : 0u;
            const bool tiles = pl.marks && heavy_cap > inline_cap;
            uint32_t *in_count = pl.marks ? (uint32_t *)(sc + pl.off_incount) : nullptr;
            uint32_t *in_off = pl.marks ? (uint32_t *)(sc + pl.off_inoff) : nullptr;
            u32x4_a16 *ext32 = (u32x4_a16 *)(ws + pl.off_ext32);
            if (tiles) device_fill(in_count, 0u, align_up((size_t)(n_nodes + 1) * 4, 256), stream);
            entries_kernel<<<stream_grid((nnz + 7) / 8), kBlock, 0, stream>>>(row32, col32, n_nodes, nnz, table, pl.table_blocks,
                                                                            inline_cap, heavy_cap, tiles ? 1u : 0u, ext32, in_count,
                                                                            header);
            if (tiles) {
                TileTask *tasks = (TileTask *)(sc + pl.off_tasks), *tasks_big = (TileTask *)(sc + pl.off_tasks_big);
                HeavyItem *items = (HeavyItem *)(sc + pl.off_heavy);
                marks_plan_kernel<<<stream_grid(n_nodes), kBlock, 0, stream>>>(row32, n_nodes, in_count, in_off, tasks, pl.task_room,
                                                                               tasks_big, pl.big_task_room, pl.heavy_room, inline_cap,
                                                                               header);
                marks_scatter_kernel<<<stream_grid(n_nodes), kBlock, 0, stream>>>(row32, ext32, n_nodes, inline_cap, heavy_cap, in_off,
                                                                                  items, header);
                // long tile rows: 2 048 entries of tile + 64 000 bits of filter + 3 KB of tables and survivor stack per wave:
                // 40.9 KB, FOUR workgroups of 8 waves per CU (with a 4 096-entry tile it was three: 53.7 -> 47.5 ms for the
                // whole pass on one box -- the pass waits for memory and for its own barriers, resident waves are what it
                // is short of); short ones (up to 255 entries): a workgroup of ONE wave, 5.2 KB
                tile_bloom_kernel<TRW_TILE_CAP, TRW_BLOOM_WORDS, 512, 8, TRW_BIG_GRAB, kStBigCount, kStBigNext, false>
                    <<<(unsigned)device_cu_count() * 4, 512, 0, stream>>>(row32, col32, ext32, items, tasks_big, pl.big_task_room,
                                                                          in_count, in_off, inline_cap, heavy_cap, header);
                tile_bloom_kernel<256, 256, 64, 8, TRW_SMALL_GRAB, kStTaskCount, kStTaskNext, true>
                    <<<(unsigned)device_cu_count() * 32, 64, 0, stream>>>(row32, col32, ext32, items, tasks, pl.task_room, in_count,
                                                                          in_off, inline_cap, heavy_cap, header);
            }
        }
    }
    if (g_prof.on && g_prof.prep[0] && (g_prof.prep[1] || hipEventCreate(&g_prof.prep[1]) == hipSuccess)) {
        (void)hipEventRecord(g_prof.prep[1], stream);
        g_prof.prep_valid = true;
    }
#ifdef TRW_TILE_CLOCKS
    {
        (void)hipStreamSynchronize(stream);
        long long h[32];
        (void)hipMemcpy(h, prepared, sizeof(h), hipMemcpyDeviceToHost);
        double tot = 0;
        for (int k = 24; k < 31; ++k) tot += (double)h[k];
        fprintf(stderr, "[TRW_TILE_CLOCKS] wave cycles: grab %.1f%% examine %.1f%% tile change %.1f%% fetch %.1f%% write-out %.1f%% filter+push %.1f%% search %.1f%% (total %.3e over all waves of the clocked tile kernel; tasks %lld short rows + %lld long rows, in-list items %lld, dropped %lld)\n",
                100.0 * h[24] / tot, 100.0 * h[25] / tot, 100.0 * h[26] / tot, 100.0 * h[27] / tot, 100.0 * h[28] / tot, 100.0 * h[29] / tot, 100.0 * h[30] / tot, tot, h[16], h[21], h[20], h[19]);
        unsigned long long d[16] = {0};
        (void)hipMemcpyFromSymbol(d, HIP_SYMBOL(g_tile_dbg), sizeof(d));
        fprintf(stderr, "[TRW_TILE_CLOCKS] searches: whole-row tiles %llu (%llu survivors, %.0f cycles each), sampled tiles %llu (%llu survivors, %.0f cycles each); elements against whole-row tiles %.3e, against sampled ones %.3e; batches %.3e in %.3e wave-tasks\n",
                d[0], d[2], d[0] ? (double)d[4] / d[0] : 0.0, d[1], d[3], d[1] ? (double)d[5] / d[1] : 0.0, (double)d[6], (double)d[7], (double)d[8], (double)d[9]);
    }
#endif
#ifdef TRW_PREP_COUNT
    {
        (void)hipStreamSynchronize(stream);
        unsigned long long a = 0, b = 0, c = 0;
        (void)hipMemcpyFromSymbol(&a, HIP_SYMBOL(g_cnt_hits_u), 8);
        (void)hipMemcpyFromSymbol(&b, HIP_SYMBOL(g_cnt_rpos), 8);
        (void)hipMemcpyFromSymbol(&c, HIP_SYMBOL(g_cnt_probes_u), 8);
        fprintf(stderr, "[TRW_PREP_COUNT] mask hits walking u's row %llu of %llu probes, reverse edges found %llu\n", a, c, b);
    }
#endif
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "kernel launch");
    return TRW_OK;
}

size_t trw_walk_csr_prepared_workspace_bytes(int64_t n_walkers, int32_t walk_length, double p, double q, int32_t rng_mode)
{
    return make_call_layout(n_walkers, walk_length, p == 1.0 && q == 1.0, rng_mode).total;
}

// The fused walk + to_windows (trw_walk_windows_csr_prepared_hip): where the uniform walk kernel puts the windows.
struct FusedWindows {
    int32_t window_size = 0;
    int64_t *centre = nullptr, *pos = nullptr;
};

static int walk_prepared(const void *prepared, size_t prepared_bytes, int32_t what, const int64_t *row_ptr,
                         int64_t n_nodes, const int64_t *col_idx, int64_t nnz, const int64_t *targets,
                         int64_t n_walkers, int64_t walker_base, double p, double q, int32_t walk_length, int32_t seed,
                         int32_t rng_mode, int64_t *out, void *workspace, size_t workspace_bytes, void *hip_stream,
                         bool mark_prepare, const FusedWindows *fused = nullptr, uint32_t *out32 = nullptr);

static int check_walk_args(const int64_t *row_ptr, int64_t n_nodes, const int64_t *col_idx, int64_t nnz,
                           const int64_t *targets, int64_t n_walkers, int64_t walker_base, double p, double q,
                           int32_t walk_length, const int64_t *out, const void *workspace, size_t workspace_bytes,
                           bool out_optional = false)
{
    if (n_nodes < 0 || nnz < 0 || n_walkers < 0 || walker_base < 0)
        return fail(TRW_E_ARG, "negative size argument");
    if (walk_length < 1) return fail(TRW_E_ARG, "walk_length must be >= 1");
    if (!(p > 0.0) || !(q > 0.0) || !std::isfinite(p) || !std::isfinite(q))
        return fail(TRW_E_ARG, "p and q must be positive and finite");
    if (!row_ptr) return fail(TRW_E_ARG, "row_ptr is null");
    if (n_walkers > 0 && (!targets || (!out && !out_optional))) return fail(TRW_E_ARG, "targets/out is null");
    if (nnz > 0 && !col_idx) return fail(TRW_E_ARG, "col_idx is null");
    if (n_walkers > 0 && n_nodes == 0) return fail(TRW_E_ARG, "walkers on an empty graph");
    if (!workspace || workspace_bytes < (size_t)kStatusBytes)
        return fail(TRW_E_WORKSPACE, "workspace missing or smaller than the status block");
    if (((uintptr_t)workspace & 15u) != 0) return fail(TRW_E_WORKSPACE, "workspace must be 16-byte aligned");
    return TRW_OK;
}

int trw_walk_csr_prepared_hip(const void *prepared, size_t prepared_bytes, int32_t what, const int64_t *row_ptr,
                              int64_t n_nodes, const int64_t *col_idx, int64_t nnz, const int64_t *targets,
                              int64_t n_walkers, int64_t walker_base, double p, double q, int32_t walk_length,
                              int32_t seed, int32_t rng_mode, int64_t *out, void *workspace, size_t workspace_bytes,
                              void *hip_stream)
{
    return walk_prepared(prepared, prepared_bytes, what, row_ptr, n_nodes, col_idx, nnz, targets, n_walkers,
                         walker_base, p, q, walk_length, seed, rng_mode, out, workspace, workspace_bytes, hip_stream,
                         true);
}

int trw_walk_csr_prepared_u32_hip(const void *prepared, size_t prepared_bytes, int32_t what, const int64_t *row_ptr,
                                  int64_t n_nodes, const int64_t *col_idx, int64_t nnz, const int64_t *targets,
                                  int64_t n_walkers, int64_t walker_base, double p, double q, int32_t walk_length,
                                  int32_t seed, int32_t rng_mode, uint32_t *out32, void *workspace, size_t workspace_bytes,
                                  void *hip_stream)
{
    if (n_walkers > 0 && !out32) return fail(TRW_E_ARG, "out32 is null");
    if (n_nodes > ((int64_t)1 << 32)) return fail(TRW_E_UNSUPPORTED, "narrow rows: node ids beyond 2^32");
    return walk_prepared(prepared, prepared_bytes, what, row_ptr, n_nodes, col_idx, nnz, targets, n_walkers,
                         walker_base, p, q, walk_length, seed, rng_mode, nullptr, workspace, workspace_bytes, hip_stream,
                         true, nullptr, n_walkers > 0 ? out32 : (uint32_t *)workspace);
}

static int walk_prepared(const void *prepared, size_t prepared_bytes, int32_t what, const int64_t *row_ptr,
                         int64_t n_nodes, const int64_t *col_idx, int64_t nnz, const int64_t *targets,
                         int64_t n_walkers, int64_t walker_base, double p, double q, int32_t walk_length, int32_t seed,
                         int32_t rng_mode, int64_t *out, void *workspace, size_t workspace_bytes, void *hip_stream,
                         bool mark_prepare, const FusedWindows *fused, uint32_t *out32)
{
    t_last_error.clear();
    if (int rc = check_walk_args(row_ptr, n_nodes, col_idx, nnz, targets, n_walkers, walker_base, p, q, walk_length, out,
                                 workspace, workspace_bytes, fused != nullptr || out32 != nullptr))
        return rc;
    if (out32 && (out || fused)) return fail(TRW_E_ARG, "narrow rows: one output only");
    const PrepLayout pl = make_prep_layout(n_nodes, nnz, what);
    if (!prepared || prepared_bytes < pl.persistent || ((uintptr_t)prepared & 15u) != 0)
        return fail(TRW_E_WORKSPACE, "prepared graph missing, misaligned or smaller than its layout");

    const bool uniform = (p == 1.0 && q == 1.0);  // rw_cpu.cpp:220
    const int mode = resolve_mode(rng_mode, uniform, n_walkers, walker_base, walk_length);
    if (mode < 0) return fail(TRW_E_RNG_MODE, "unknown rng_mode");
    if (mode == TRW_RNG_GLIBC && !uniform && walker_base != 0)
        return fail(TRW_E_RNG_MODE, "TRW_RNG_GLIBC biased walks replay one serial stream: walker_base must be 0");
    if (!uniform && what == TRW_PREP_NOTHING && n_walkers > 0)
        return fail(TRW_E_ARG, "a biased walk needs the row order: prepare with TRW_PREP_ORDER_ONLY at least");
    const unsigned __int128 last_pos = ((unsigned __int128)walker_base + (unsigned __int128)n_walkers) *
                                       (unsigned __int128)walk_length;
    if (mode == TRW_RNG_GLIBC && uniform && last_pos >= ((unsigned __int128)1 << 48))
        return fail(TRW_E_RANGE, "glibc stream position beyond 2^48 draws");

    hipStream_t stream = (hipStream_t)hip_stream;
    int64_t *status = (int64_t *)workspace;
    const int64_t *header = (const int64_t *)prepared;
    hipError_t e = hipSuccess;
    if (mark_prepare) prof_mark(0, stream);  // (a one-shot call marked it before its prepare pass)
    // the call's status block starts from what the prepare pass found (validation bits, row order)
    status_seed_kernel<<<1, 64, 0, stream>>>(header, status);
    if (n_walkers == 0) return TRW_OK;

    const CallLayout cl = make_call_layout(n_walkers, walk_length, uniform, mode);
    if (workspace_bytes < cl.total) return fail(TRW_E_WORKSPACE, "workspace too small for this call shape");
    const char *pw = (const char *)prepared;
    char *ws = (char *)workspace;
    uint32_t *win = (uint32_t *)(ws + cl.off_win);
    const uint32_t *row32 = (const uint32_t *)(pw + pl.off_row32);
    const uint32_t *col32 = (const uint32_t *)(pw + pl.off_col32);
    const uint32_t *table = (const uint32_t *)(pw + pl.off_table);
    const int64_t L = walk_length;
    const unsigned walker_grid = (unsigned)((n_walkers + kBlock - 1) / kBlock);
    // (uniform_glibc_rows_kernel addresses a wave's 64 rows with a 32-bit byte offset)
    const bool rows_fit = (int64_t)64 * cl.G * ((int64_t)L + 1) < ((int64_t)1 << 29);

    if (uniform && mode == TRW_RNG_GLIBC)
        if (int rc = ensure_jump_table()) return rc;

    prof_mark(1, stream);
    const unsigned lane_grid = (unsigned)((cl.nlanes + kBlock - 1) / kBlock);
    Thresholds th = {0, 0, 0};
    BiasedParams bp = {0, 0, 0, 0, 0};
    if (!uniform) {
        // rw_cpu.cpp:130-134, evaluated in the same order, then turned into integer thresholds
        const double max_prob_init = std::fmax(1.0 / p, 1);
        const double max_prob = std::fmax(max_prob_init, 1.0 / q);
        th.t0 = trw_accept_threshold(1.0 / p / max_prob);
        th.t1 = trw_accept_threshold(1.0 / max_prob);
        th.t2 = trw_accept_threshold(1.0 / q / max_prob);
        bp.t0 = th.t0;
        bp.t1 = th.t1;
        bp.t2 = th.t2;
        bp.lo = th.t1 < th.t2 ? th.t1 : th.t2;
        bp.hi = th.t1 < th.t2 ? th.t2 : th.t1;
    }
    if (uniform) {
        if (mode == TRW_RNG_GLIBC) {
            SeedWindow sw;
            trw::seed_window(seed, sw.w);
            glibc_windows_kernel<<<lane_grid, kBlock, 0, stream>>>(sw, (uint64_t)walker_base * (uint64_t)L,
                                                                  (uint64_t)cl.G * (uint64_t)L, cl.nlanes, win);
            // BUILDING the fat entries pays from about nnz steps (plan_what: in one batch; the torch layer's cache: over
            // the calls made so far).  Once they are there they are used by every batch that is not small next to the
            // graph -- one gather per step instead of two (a shard of BASELINE configs[3], 0.65 nnz steps: 41.7 -> 30.4
            // ms); a small batch revisits few rows and is better off with the compact shadow.
            const long ext_forced = env_long("TRW_EXT", 0, 1, -1);
            const bool use_fat = pl.ext16 && (ext_forced == 1 || (ext_forced == -1 && (__int128)n_walkers * L * 4 >= (__int128)nnz));
            if (fused) {  // (the caller made sure of the shape: walk_windows_fusable)
                if (use_fat)
                    uniform_glibc_rows_kernel<true, true><<<lane_grid, kBlock, 0, stream>>>(
                        row32, col32, (const u32x4_a16 *)(pw + pl.off_ext16), targets, n_nodes, n_walkers, walk_length,
                        cl.G, cl.nlanes, win, out, status, fused->window_size, fused->centre, fused->pos);
                else
                    uniform_glibc_rows_kernel<false, true><<<lane_grid, kBlock, 0, stream>>>(
                        row32, col32, nullptr, targets, n_nodes, n_walkers, walk_length, cl.G, cl.nlanes, win, out,
                        status, fused->window_size, fused->centre, fused->pos);
            } else if (out32) {  // narrow rows straight from the walk: the row-tile kernels only
                if (!(pl.narrow && rows_fit)) return fail(TRW_E_UNSUPPORTED, "narrow rows: this graph runs the int64 kernels");
                if (use_fat)
                    uniform_glibc_rows_kernel<true, false, uint32_t><<<lane_grid, kBlock, 0, stream>>>(
                        row32, col32, (const u32x4_a16 *)(pw + pl.off_ext16), targets, n_nodes, n_walkers, walk_length,
                        cl.G, cl.nlanes, win, out32, status, 0, nullptr, nullptr);
                else
                    uniform_glibc_rows_kernel<false, false, uint32_t><<<lane_grid, kBlock, 0, stream>>>(
                        row32, col32, nullptr, targets, n_nodes, n_walkers, walk_length, cl.G, cl.nlanes, win, out32, status,
                        0, nullptr, nullptr);
            } else if (pl.narrow && rows_fit && use_fat)
                uniform_glibc_rows_kernel<true, false><<<lane_grid, kBlock, 0, stream>>>(
                    row32, col32, (const u32x4_a16 *)(pw + pl.off_ext16), targets, n_nodes, n_walkers, walk_length,
                    cl.G, cl.nlanes, win, out, status, 0, nullptr, nullptr);
            else if (pl.narrow && rows_fit)
                uniform_glibc_rows_kernel<false, false><<<lane_grid, kBlock, 0, stream>>>(
                    row32, col32, nullptr, targets, n_nodes, n_walkers, walk_length, cl.G, cl.nlanes, win, out, status, 0,
                    nullptr, nullptr);
            else
                uniform_glibc_wide_kernel<<<lane_grid, kBlock, 0, stream>>>(row_ptr, col_idx, targets, n_nodes, nnz,
                                                                           n_walkers, walk_length, cl.G, cl.nlanes,
                                                                           win, out, status);
        } else {
            if (out32) return fail(TRW_E_UNSUPPORTED, "narrow rows: not from the uniform walk on the per-walker stream");
            uniform_walker_kernel<<<walker_grid, kBlock, 0, stream>>>(row_ptr, col_idx, targets, n_nodes, nnz,
                                                                     n_walkers, walker_base, walk_length,
                                                                     (uint32_t)seed, out, status);
        }
    } else if (mode == TRW_RNG_GLIBC) {
        if (out32) return fail(TRW_E_UNSUPPORTED, "narrow rows: not from the serial replay");
        SeedWindow sw;
        trw::seed_window(seed, sw.w);
        // sorted graph with a table and fat entries: the whole wave replays the chain (64 speculative trials at a
        // time); anything else, one lane over the int64 arrays.  Each kernel returns at once when the graph is not
        // its kind (the row order is known on the device only).
        const bool wave_replay = pl.narrow && pl.biased && env_long("TRW_WAVE_REPLAY", 0, 1, 1) == 1;
        if (wave_replay)
            biased_glibc_wave_kernel<<<1, 64, 0, stream>>>(row32, col32, (const u32x4_a16 *)(pw + pl.off_ext32), table,
                                                          pl.table_blocks, targets, n_nodes, n_walkers, walk_length, sw,
                                                          bp, out, status);
        biased_glibc_seq_kernel<<<1, 64, 0, stream>>>(row_ptr, col_idx, targets, n_nodes, nnz, n_walkers,
                                                     walk_length, sw, th, wave_replay ? 1 : 0, out, status);
    } else if (pl.narrow && pl.biased) {
        // persistent lanes: walkers are dealt out by a device counter (status[kStNextWalker])
        int64_t blocks = (n_walkers + kBlock - 1) / kBlock;
        // Workgroups per CU (TRW_RESIDENT_BLOCKS; 8 = every wave slot)
        const int64_t resident = (int64_t)device_cu_count() * env_long("TRW_RESIDENT_BLOCKS", 1, 8, 5);
        if (blocks > resident) blocks = resident;
        const u32x4_a16 *ext32 = (const u32x4_a16 *)(pw + pl.off_ext32);
        const bool mask_kernel = env_long("TRW_MASK_KERNEL", 0, 1, 1) == 1;
        // sorted rows: the mask kernel; unsorted rows: the state-machine kernel (each returns at once when the
        // graph is not its kind -- the row order is known on the device only)
        // walker pool (lanes as executors over walker state in LDS) unless switched off or the batch does not fit its
        // 32-bit walker numbers; TRW_POOL_KERNEL=0 runs the lane-per-walker mask kernel instead
        const bool pool_kernel = mask_kernel && n_walkers < ((int64_t)1 << 32) && env_long("TRW_POOL_KERNEL", 0, 1, 1) == 1;
        if (out32 && !pool_kernel) return fail(TRW_E_UNSUPPORTED, "narrow rows: only the walker-pool kernel writes them");
        if (pool_kernel) {
            // 4 workgroups per CU: a wave's pool is 9.8 KB of LDS (kPoolSlots = 122 walkers x 20 words), 40.6 KB per workgroup
            // (a batch that does not even give every CU one workgroup of 64-walker waves is bound by latency: its grid is
            // sized for 64 walkers per wave instead of full pools -- 405 / 4 096 / 50 000 walkers: 2.0 / 2.1 / 2.3 ->
            // 1.7 / 1.8 / 2.0 ms; from 200 000 walkers on half-filled pools lose, 2.5 -> 2.8 ms)
            const bool small = n_walkers <= (int64_t)device_cu_count() * kPoolWaves * 64;
            const int64_t per_block = (int64_t)kPoolWaves * (small ? 64 : kPoolSlots);
            int64_t pblocks = (n_walkers + per_block - 1) / per_block;
            // TRW_POOL_SPARE_BLOCKS: workgroups to leave out of the persistent grid.  With every CU holding three of them
            // a kernel on another stream (an RCCL send/recv group) does not get on the chip until the walk is over
            // (scripts/overlap_probe.py); a few CUs with only two leave it room.  bench.py sets it for N > 1.
            int64_t president = (int64_t)device_cu_count() * env_long("TRW_POOL_BLOCKS", 1, 8, 4) -
                                env_long("TRW_POOL_SPARE_BLOCKS", 0, 512, 0);
            if (president < 1) president = 1;
            if (pblocks > president) pblocks = president;
            if (out32)
                biased_pool_kernel<false, uint32_t><<<(unsigned)pblocks, kBlock, 0, stream>>>(
                    row32, col32, ext32, table, pl.table_blocks, targets, n_nodes, n_walkers, walker_base, walk_length,
                    (uint32_t)seed, bp, out32, status);
            else if (g_prof.count)
                biased_pool_kernel<true><<<(unsigned)pblocks, kBlock, 0, stream>>>(
                    row32, col32, ext32, table, pl.table_blocks, targets, n_nodes, n_walkers, walker_base, walk_length,
                    (uint32_t)seed, bp, out, status);
            else
                biased_pool_kernel<false><<<(unsigned)pblocks, kBlock, 0, stream>>>(
                    row32, col32, ext32, table, pl.table_blocks, targets, n_nodes, n_walkers, walker_base, walk_length,
                    (uint32_t)seed, bp, out, status);
        } else if (mask_kernel) {
            if (g_prof.count)
                biased_mask_kernel<true><<<(unsigned)blocks, kBlock, 0, stream>>>(
                    row32, col32, ext32, table, pl.table_blocks, targets, n_nodes, n_walkers, walker_base, walk_length,
                    (uint32_t)seed, bp, out, status);
            else
                biased_mask_kernel<false><<<(unsigned)blocks, kBlock, 0, stream>>>(
                    row32, col32, ext32, table, pl.table_blocks, targets, n_nodes, n_walkers, walker_base, walk_length,
                    (uint32_t)seed, bp, out, status);
        }
        const int32_t want_pos = (bp.t0 > bp.hi && env_long("TRW_NO_POS", 0, 1, 0) == 0) ? 1 : 0;
        if (out32)
            biased_hash_kernel<false, uint32_t><<<(unsigned)blocks, kBlock, 0, stream>>>(
                row32, col32, (const uint32_t *)ext32, table, pl.table_blocks, targets, n_nodes, n_walkers, walker_base,
                walk_length, (uint32_t)seed, bp, want_pos, mask_kernel ? 1 : 0, out32, status);
        else if (g_prof.count)
            biased_hash_kernel<true><<<(unsigned)blocks, kBlock, 0, stream>>>(
                row32, col32, (const uint32_t *)ext32, table, pl.table_blocks, targets, n_nodes, n_walkers, walker_base,
                walk_length, (uint32_t)seed, bp, want_pos, mask_kernel ? 1 : 0, out, status);
        else
            biased_hash_kernel<false><<<(unsigned)blocks, kBlock, 0, stream>>>(
                row32, col32, (const uint32_t *)ext32, table, pl.table_blocks, targets, n_nodes, n_walkers, walker_base,
                walk_length, (uint32_t)seed, bp, want_pos, mask_kernel ? 1 : 0, out, status);
    } else if (out32) {
        return fail(TRW_E_UNSUPPORTED, "narrow rows: a biased walk writes them over a graph prepared with TRW_PREP_BIASED only");
    } else if (pl.narrow) {
        // Locating t inside prev's row once per step (a bisection here) pays when at least half of the trials
        // draw above `hi` and can then be decided from the candidate's offset alone.
        const bool use_pos = bp.t0 > bp.hi && bp.hi <= (1u << 30) && env_long("TRW_NO_POS", 0, 1, 0) == 0;
#define TRW_LAUNCH_BIASED(POS, COUNT)                                                                          \
    biased_walker_kernel<POS, COUNT><<<walker_grid, kBlock, 0, stream>>>(                                      \
        row32, col32, targets, n_nodes, n_walkers, walker_base, walk_length, (uint32_t)seed, bp, out, status)
        if (use_pos) {
            if (g_prof.count) TRW_LAUNCH_BIASED(true, true);
            else TRW_LAUNCH_BIASED(true, false);
        } else {
            if (g_prof.count) TRW_LAUNCH_BIASED(false, true);
            else TRW_LAUNCH_BIASED(false, false);
        }
#undef TRW_LAUNCH_BIASED
    } else {
        if (g_prof.count)
            biased_walker_wide_kernel<true><<<walker_grid, kBlock, 0, stream>>>(
                row_ptr, col_idx, targets, n_nodes, nnz, n_walkers, walker_base, walk_length, (uint32_t)seed, th,
                out, status);
        else
            biased_walker_wide_kernel<false><<<walker_grid, kBlock, 0, stream>>>(
                row_ptr, col_idx, targets, n_nodes, nnz, n_walkers, walker_base, walk_length, (uint32_t)seed, th,
                out, status);
    }
    prof_mark(2, stream);
    prof_mark(3, stream);  // every walk kernel writes the caller's rows itself: nothing to finish
    e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "kernel launch");
    return TRW_OK;
}

// One-shot form: prepares what this call shape is worth (plan_what) inside the workspace, then walks.
size_t trw_walk_csr_workspace_bytes(int64_t n_nodes, int64_t nnz, int64_t n_walkers, int32_t walk_length,
                                    double p, double q, int32_t rng_mode)
{
    if (n_nodes < 0 || nnz < 0) return kStatusBytes;
    const bool uniform = (p == 1.0 && q == 1.0);
    const int32_t what = plan_what(n_nodes, nnz, n_walkers, walk_length, uniform, rng_mode);
    return align_up(make_call_layout(n_walkers, walk_length, uniform, rng_mode).total, 256) +
           make_prep_layout(n_nodes, nnz, what).total;
}

int trw_walk_csr_hip(const int64_t *row_ptr, int64_t n_nodes, const int64_t *col_idx, int64_t nnz,
                     const int64_t *targets, int64_t n_walkers, int64_t walker_base, double p, double q,
                     int32_t walk_length, int32_t seed, int32_t rng_mode, int64_t *out, void *workspace,
                     size_t workspace_bytes, void *hip_stream)
{
    t_last_error.clear();
    if (int rc = check_walk_args(row_ptr, n_nodes, col_idx, nnz, targets, n_walkers, walker_base, p, q, walk_length, out,
                                 workspace, workspace_bytes))
        return rc;
    const bool uniform = (p == 1.0 && q == 1.0);
    const int mode = resolve_mode(rng_mode, uniform, n_walkers, walker_base, walk_length);
    if (mode < 0) return fail(TRW_E_RNG_MODE, "unknown rng_mode");
    if (mode == TRW_RNG_GLIBC && !uniform && walker_base != 0)
        return fail(TRW_E_RNG_MODE, "TRW_RNG_GLIBC biased walks replay one serial stream: walker_base must be 0");
    if (n_walkers == 0) {
        device_fill(workspace, 0u, kStatusBytes, (hipStream_t)hip_stream);
        return TRW_OK;
    }
    // planned with the resolved mode: never larger than what trw_walk_csr_workspace_bytes sized with the caller's
    const int32_t what = plan_what(n_nodes, nnz, n_walkers, walk_length, uniform, mode);
    const size_t call_bytes = align_up(make_call_layout(n_walkers, walk_length, uniform, mode).total, 256);
    const size_t prep_bytes = make_prep_layout(n_nodes, nnz, what).total;
    if (workspace_bytes < call_bytes + prep_bytes) return fail(TRW_E_WORKSPACE, "workspace too small for this call shape");
    void *prepared = (char *)workspace + call_bytes;
    prof_mark(0, (hipStream_t)hip_stream);
    if (int rc = trw_prepare_csr_hip(row_ptr, n_nodes, col_idx, nnz, what, prepared, prep_bytes, hip_stream)) return rc;
    return walk_prepared(prepared, prep_bytes, what, row_ptr, n_nodes, col_idx, nnz, targets, n_walkers, walker_base, p,
                         q, walk_length, seed, mode, out, workspace, call_bytes, hip_stream, false);
}

// ---- triples (knowledge-graph) walk, reference csrc/cpu/rw_cpu_triples.cpp -----------------------

size_t trw_walk_triples_workspace_bytes(int64_t n_triples, int64_t n_entities, int64_t n_walkers,
                                        int32_t walk_length, int64_t padding_idx, int32_t rng_mode)
{
    if (n_triples < 0 || n_entities < 0) return kStatusBytes;
    return make_triples_plan(n_triples, n_entities, n_walkers, walk_length, padding_idx, rng_mode).total;
}

int trw_walk_triples_hip(const int64_t *triples, int64_t n_triples, const int64_t *rt_index, int64_t n_entities,
                         const int64_t *targets, int64_t n_walkers, int64_t walker_base, int32_t walk_length,
                         int64_t padding_idx, int32_t seed, int32_t rng_mode, int64_t *out, void *workspace,
                         size_t workspace_bytes, void *hip_stream)
{
    t_last_error.clear();
    if (n_triples < 0 || n_entities < 0 || n_walkers < 0 || walker_base < 0)
        return fail(TRW_E_ARG, "negative size argument");
    if (walk_length < 1) return fail(TRW_E_ARG, "walk_length must be >= 1");
    if (n_triples > 0 && !triples) return fail(TRW_E_ARG, "triples is null");
    if (n_entities > 0 && !rt_index) return fail(TRW_E_ARG, "relation_tail_index is null");
    if (n_walkers > 0 && (!targets || !out)) return fail(TRW_E_ARG, "targets/out is null");
    if (!workspace || workspace_bytes < (size_t)kStatusBytes)
        return fail(TRW_E_WORKSPACE, "workspace missing or smaller than the status block");
    if (((uintptr_t)workspace & 15u) != 0) return fail(TRW_E_WORKSPACE, "workspace must be 16-byte aligned");
    const int mode = resolve_triples_mode(rng_mode, n_walkers, walker_base, walk_length);
    if (mode < 0) return fail(TRW_E_RNG_MODE, "unknown rng_mode");
    if (mode == TRW_RNG_GLIBC && walker_base != 0)
        return fail(TRW_E_RNG_MODE, "TRW_RNG_GLIBC triples walks replay one serial stream: walker_base must be 0");

    hipStream_t stream = (hipStream_t)hip_stream;
    int64_t *status = (int64_t *)workspace;
    hipError_t e = hipSuccess;
    device_fill(status, 0u, kStatusBytes, stream);
    if (n_walkers == 0) return TRW_OK;

    const TriplesPlan pl = make_triples_plan(n_triples, n_entities, n_walkers, walk_length, padding_idx, mode);
    if (workspace_bytes < pl.total) return fail(TRW_E_WORKSPACE, "workspace too small for this call shape");
    char *ws = (char *)workspace;
    const unsigned walker_grid = (unsigned)((n_walkers + kBlock - 1) / kBlock);

    prof_mark(0, stream);
    if (pl.narrow)
        triples_shadow_kernel<<<stream_grid(n_triples > n_entities ? n_triples : n_entities), kBlock, 0, stream>>>(
            triples, n_triples, rt_index, n_entities, padding_idx, (u32x2_a8 *)(ws + pl.off_rt32),
            (u32x2_a8 *)(ws + pl.off_ix32), status);
    prof_mark(1, stream);
    if (mode == TRW_RNG_GLIBC) {
        SeedWindow sw;
        trw::seed_window(seed, sw.w);
        triples_glibc_seq_kernel<<<1, 64, 0, stream>>>(triples, n_triples, rt_index, targets, n_entities, n_walkers,
                                                      walk_length, padding_idx, sw, out, status);
    } else if (pl.narrow) {
        triples_walker_kernel<<<walker_grid, kBlock, 0, stream>>>(
            (const u32x2_a8 *)(ws + pl.off_rt32), (const u32x2_a8 *)(ws + pl.off_ix32), targets, n_entities,
            n_walkers, walker_base, walk_length, (uint32_t)padding_idx, (uint32_t)seed, out, status);
    } else {
        triples_walker_wide_kernel<<<walker_grid, kBlock, 0, stream>>>(triples, n_triples, rt_index, targets,
                                                                      n_entities, n_walkers, walker_base, walk_length,
                                                                      padding_idx, (uint32_t)seed, out, status);
    }
    prof_mark(2, stream);
    prof_mark(3, stream);  // triples_walker_kernel writes the rows itself: nothing to finish
    e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "kernel launch");
    return TRW_OK;
}

// ---- edge-list walk, reference csrc/cpu/rw_cpu_edge_list.cpp -------------------------------------

int32_t trw_prepare_edge_list_plan(int64_t n_edges, int64_t n_nodes, int64_t n_walkers, int32_t walk_length, double p,
                                   double q, int64_t padding_idx, int32_t rng_mode)
{
    if (n_edges < 0 || n_nodes < 0) return TRW_PREP_NOTHING;
    const bool uniform = (p == 1.0 && q == 1.0);
    const int mode = resolve_elist_mode(rng_mode, uniform, n_walkers, 0, walk_length);
    return elist_plan_what(n_edges, n_nodes, n_walkers, walk_length, padding_idx, mode < 0 ? TRW_RNG_AUTO : mode, uniform);
}

size_t trw_prepare_edge_list_bytes(int64_t n_edges, int64_t n_nodes, int64_t padding_idx, int32_t what)
{
    if (n_edges < 0 || n_nodes < 0) return kStatusBytes;
    return make_elist_layout(n_edges, n_nodes, padding_idx, what).total;
}

int trw_prepare_edge_list_hip(const int64_t *edges, int64_t n_edges, const int64_t *node_edge_index, int64_t n_nodes,
                              int64_t padding_idx, int32_t what, void *prepared, size_t prepared_bytes, void *hip_stream)
{
    t_last_error.clear();
    if (n_edges < 0 || n_nodes < 0) return fail(TRW_E_ARG, "negative size argument");
    if (n_edges > 0 && !edges) return fail(TRW_E_ARG, "edge_list_indexed is null");
    if (n_nodes > 0 && !node_edge_index) return fail(TRW_E_ARG, "node_edge_index is null");
    if (!prepared || ((uintptr_t)prepared & 15u) != 0) return fail(TRW_E_WORKSPACE, "prepared buffer missing or not 16-byte aligned");
    const EdgeListLayout pl = make_elist_layout(n_edges, n_nodes, padding_idx, what);
    if (prepared_bytes < pl.total) return fail(TRW_E_WORKSPACE, "prepared buffer too small for this edge list");
    hipStream_t stream = (hipStream_t)hip_stream;
    char *ws = (char *)prepared;
    int64_t *header = (int64_t *)ws;
    if (g_prof.on && (g_prof.prep[0] || hipEventCreate(&g_prof.prep[0]) == hipSuccess))
        (void)hipEventRecord(g_prof.prep[0], stream);
    device_fill(header, 0u, kStatusBytes, stream);
    if (pl.narrow) {
        elist_shadow_kernel<<<stream_grid(n_edges > n_nodes ? n_edges : n_nodes), kBlock, 0, stream>>>(
            edges, n_edges, node_edge_index, n_nodes, padding_idx, (uint32_t *)(ws + pl.off_tail32),
            (u32x2_a8 *)(ws + pl.off_ix32), header);
        if (pl.hash) {
            uint32_t *el_table = (uint32_t *)(ws + pl.off_table);
            device_fill(el_table, 0xFFFFFFFFu, ((size_t)pl.hash_mask + 1) * 16, stream);
            elist_table_build_kernel<<<stream_grid(n_edges), kBlock, 0, stream>>>(
                edges, n_edges, n_nodes, (const uint32_t *)(ws + pl.off_tail32),
                (const u32x2_a8 *)(ws + pl.off_ix32), el_table, pl.hash_mask);
            if (pl.ent) {
                elist_ext_kernel<<<stream_grid(n_edges), kBlock, 0, stream>>>(
                    edges, n_edges, n_nodes, (uint32_t)padding_idx, (const uint32_t *)(ws + pl.off_tail32),
                    (const u32x2_a8 *)(ws + pl.off_ix32), el_table, pl.hash_mask, (u32x4_a16 *)(ws + pl.off_ent));
                elist_node_kernel<<<stream_grid(n_nodes), kBlock, 0, stream>>>(
                    n_nodes, (uint32_t)padding_idx, (const uint32_t *)(ws + pl.off_tail32),
                    (const u32x2_a8 *)(ws + pl.off_ix32), el_table, pl.hash_mask, (u32x4_a16 *)(ws + pl.off_nd));
            }
        }
    }
    if (g_prof.on && g_prof.prep[0] && (g_prof.prep[1] || hipEventCreate(&g_prof.prep[1]) == hipSuccess)) {
        (void)hipEventRecord(g_prof.prep[1], stream);
        g_prof.prep_valid = true;
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "kernel launch");
    return TRW_OK;
}

static int walk_edge_list_prepared(const void *prepared, size_t prepared_bytes, int32_t what, const int64_t *edges,
                                   int64_t n_edges, const int64_t *node_edge_index, int64_t n_nodes,
                                   const int64_t *targets, int64_t n_walkers, int64_t walker_base, double p, double q,
                                   int32_t walk_length, int32_t seed, int64_t padding_idx, int32_t restart,
                                   int32_t rng_mode, int64_t *out, void *workspace, size_t workspace_bytes,
                                   void *hip_stream, bool mark_prepare)
{
    t_last_error.clear();
    if (n_edges < 0 || n_nodes < 0 || n_walkers < 0 || walker_base < 0)
        return fail(TRW_E_ARG, "negative size argument");
    if (walk_length < 1) return fail(TRW_E_ARG, "walk_length must be >= 1");
    if (!(p > 0.0) || !(q > 0.0) || !std::isfinite(p) || !std::isfinite(q))
        return fail(TRW_E_ARG, "p and q must be positive and finite");
    if (n_edges > 0 && !edges) return fail(TRW_E_ARG, "edge_list_indexed is null");
    if (n_nodes > 0 && !node_edge_index) return fail(TRW_E_ARG, "node_edge_index is null");
    if (n_walkers > 0 && (!targets || !out)) return fail(TRW_E_ARG, "targets/out is null");
    if (!workspace || workspace_bytes < (size_t)kStatusBytes)
        return fail(TRW_E_WORKSPACE, "workspace missing or smaller than the status block");
    if (((uintptr_t)workspace & 15u) != 0) return fail(TRW_E_WORKSPACE, "workspace must be 16-byte aligned");
    const bool uniform = (p == 1.0 && q == 1.0);  // rw_cpu_edge_list.cpp:260
    const int mode = resolve_elist_mode(rng_mode, uniform, n_walkers, walker_base, walk_length);
    if (mode < 0) return fail(TRW_E_RNG_MODE, "unknown rng_mode");
    if (mode == TRW_RNG_GLIBC && walker_base != 0)
        return fail(TRW_E_RNG_MODE, "TRW_RNG_GLIBC edge-list walks replay one serial stream: walker_base must be 0");
    const EdgeListLayout pl = make_elist_layout(n_edges, n_nodes, padding_idx, what);
    if (!prepared || prepared_bytes < pl.total || ((uintptr_t)prepared & 15u) != 0)
        return fail(TRW_E_WORKSPACE, "prepared edge list missing, misaligned or smaller than its layout");
    if (walk_length >= (1 << 22) && pl.narrow) return fail(TRW_E_ARG, "walk_length too large for the uint32 kernels");

    hipStream_t stream = (hipStream_t)hip_stream;
    int64_t *status = (int64_t *)workspace;
    hipError_t e = hipSuccess;
    if (mark_prepare) prof_mark(0, stream);
    status_seed_kernel<<<1, 64, 0, stream>>>((const int64_t *)prepared, status);
    if (n_walkers == 0) return TRW_OK;
    const char *ws = (const char *)prepared;
    const unsigned walker_grid = (unsigned)((n_walkers + kBlock - 1) / kBlock);

    BiasedParams bp = {0, 0, 0, 0, 0};
    {
        const double max_prob_init = std::fmax(1.0 / p, 1);  // rw_cpu_edge_list.cpp:158-162
        const double max_prob = std::fmax(max_prob_init, 1.0 / q);
        bp.t0 = trw_accept_threshold(1.0 / p / max_prob);
        bp.t1 = trw_accept_threshold(1.0 / max_prob);
        bp.t2 = trw_accept_threshold(1.0 / q / max_prob);
        bp.lo = bp.t1 < bp.t2 ? bp.t1 : bp.t2;
        bp.hi = bp.t1 < bp.t2 ? bp.t2 : bp.t1;
    }
    EdgeListView<false> wide = {edges, node_edge_index, nullptr, nullptr, n_edges, n_nodes};
    EdgeListView<true> narrow = {nullptr, nullptr, (const uint32_t *)(ws + pl.off_tail32),
                                 (const u32x2_a8 *)(ws + pl.off_ix32), n_edges, n_nodes};
    const uint32_t *el_table = pl.hash ? (const uint32_t *)(ws + pl.off_table) : nullptr;

    prof_mark(1, stream);
    if (mode == TRW_RNG_GLIBC) {
        SeedWindow sw;
        trw::seed_window(seed, sw.w);
        elist_glibc_seq_kernel<<<1, 64, 0, stream>>>(wide, targets, n_walkers, walk_length, padding_idx, restart,
                                                    uniform ? 0 : 1, sw, bp, el_table, pl.hash_mask, out, status);
    } else if (pl.narrow) {
        if (uniform)
            elist_walk_kernel<false, true><<<walker_grid, kBlock, 0, stream>>>(
                narrow, targets, n_walkers, walker_base, walk_length, padding_idx, restart, (uint32_t)seed, bp,
                el_table, pl.hash_mask, out, status);
        else if (pl.ent && n_walkers < ((int64_t)1 << 32) && env_long("TRW_ELPOOL", 0, 1, 1) == 1) {
            // walker pool over the prepared entries: 3 workgroups per CU (a wave's pool: 124 walkers x 22 words + windows)
            // (small batches: 64 walkers per wave, cf. the CSR pool)
            const int64_t el_per_block = (int64_t)kPoolWaves * (n_walkers <= (int64_t)device_cu_count() * kPoolWaves * 64 ? 64 : kElSlots);
            int64_t pblocks = (n_walkers + el_per_block - 1) / el_per_block;
            const int64_t president = (int64_t)device_cu_count() * env_long("TRW_ELPOOL_BLOCKS", 1, 8, 3);
            if (pblocks > president) pblocks = president;
            if (g_prof.count)
                elist_pool_kernel<true><<<(unsigned)pblocks, kBlock, 0, stream>>>(
                    (const u32x4_a16 *)(ws + pl.off_nd), (const u32x4_a16 *)(ws + pl.off_ent), el_table, pl.hash_mask,
                    targets, n_nodes, n_walkers, walker_base, walk_length, (uint32_t)padding_idx, restart, (uint32_t)seed,
                    bp, out, status);
            else
                elist_pool_kernel<false><<<(unsigned)pblocks, kBlock, 0, stream>>>(
                    (const u32x4_a16 *)(ws + pl.off_nd), (const u32x4_a16 *)(ws + pl.off_ent), el_table, pl.hash_mask,
                    targets, n_nodes, n_walkers, walker_base, walk_length, (uint32_t)padding_idx, restart, (uint32_t)seed,
                    bp, out, status);
        } else
            elist_walk_kernel<true, true><<<walker_grid, kBlock, 0, stream>>>(
                narrow, targets, n_walkers, walker_base, walk_length, padding_idx, restart, (uint32_t)seed, bp,
                el_table, pl.hash_mask, out, status);
    } else {
        if (uniform)
            elist_walk_kernel<false, false><<<walker_grid, kBlock, 0, stream>>>(
                wide, targets, n_walkers, walker_base, walk_length, padding_idx, restart, (uint32_t)seed, bp, nullptr,
                0u, out, status);
        else
            elist_walk_kernel<true, false><<<walker_grid, kBlock, 0, stream>>>(
                wide, targets, n_walkers, walker_base, walk_length, padding_idx, restart, (uint32_t)seed, bp, nullptr,
                0u, out, status);
    }
    prof_mark(2, stream);
    prof_mark(3, stream);  // the walk kernels write the rows themselves: nothing to finish
    e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "kernel launch");
    return TRW_OK;
}

int trw_walk_edge_list_prepared_hip(const void *prepared, size_t prepared_bytes, int32_t what, const int64_t *edges,
                                    int64_t n_edges, const int64_t *node_edge_index, int64_t n_nodes,
                                    const int64_t *targets, int64_t n_walkers, int64_t walker_base, double p, double q,
                                    int32_t walk_length, int32_t seed, int64_t padding_idx, int32_t restart,
                                    int32_t rng_mode, int64_t *out, void *workspace, size_t workspace_bytes,
                                    void *hip_stream)
{
    return walk_edge_list_prepared(prepared, prepared_bytes, what, edges, n_edges, node_edge_index, n_nodes, targets,
                                   n_walkers, walker_base, p, q, walk_length, seed, padding_idx, restart, rng_mode, out,
                                   workspace, workspace_bytes, hip_stream, true);
}

// One-shot form: prepares what this call shape is worth inside the workspace, then walks.
size_t trw_walk_edge_list_workspace_bytes(int64_t n_edges, int64_t n_nodes, int64_t n_walkers, int32_t walk_length,
                                          int64_t padding_idx, int32_t rng_mode)
{
    if (n_edges < 0 || n_nodes < 0) return kStatusBytes;
    // sized for the biased walk (the larger of the two shapes): the query does not know p and q
    const int32_t what = elist_plan_what(n_edges, n_nodes, n_walkers, walk_length, padding_idx, rng_mode, false);
    return (size_t)kStatusBytes + make_elist_layout(n_edges, n_nodes, padding_idx, what).total;
}

int trw_walk_edge_list_hip(const int64_t *edges, int64_t n_edges, const int64_t *node_edge_index, int64_t n_nodes,
                           const int64_t *targets, int64_t n_walkers, int64_t walker_base, double p, double q,
                           int32_t walk_length, int32_t seed, int64_t padding_idx, int32_t restart, int32_t rng_mode,
                           int64_t *out, void *workspace, size_t workspace_bytes, void *hip_stream)
{
    t_last_error.clear();
    if (n_edges < 0 || n_nodes < 0 || n_walkers < 0 || walker_base < 0)
        return fail(TRW_E_ARG, "negative size argument");
    if (walk_length < 1) return fail(TRW_E_ARG, "walk_length must be >= 1");
    if (!workspace || workspace_bytes < (size_t)kStatusBytes)
        return fail(TRW_E_WORKSPACE, "workspace missing or smaller than the status block");
    if (((uintptr_t)workspace & 15u) != 0) return fail(TRW_E_WORKSPACE, "workspace must be 16-byte aligned");
    const bool uniform = (p == 1.0 && q == 1.0);
    const int mode = resolve_elist_mode(rng_mode, uniform, n_walkers, walker_base, walk_length);
    if (mode < 0) return fail(TRW_E_RNG_MODE, "unknown rng_mode");
    if (n_walkers == 0) {
        device_fill(workspace, 0u, kStatusBytes, (hipStream_t)hip_stream);
        return TRW_OK;
    }
    const int32_t what = elist_plan_what(n_edges, n_nodes, n_walkers, walk_length, padding_idx, mode, uniform);
    const size_t prep_bytes = make_elist_layout(n_edges, n_nodes, padding_idx, what).total;
    if (workspace_bytes < (size_t)kStatusBytes + prep_bytes)
        return fail(TRW_E_WORKSPACE, "workspace too small for this call shape");
    void *prepared = (char *)workspace + kStatusBytes;
    prof_mark(0, (hipStream_t)hip_stream);
    if (int rc = trw_prepare_edge_list_hip(edges, n_edges, node_edge_index, n_nodes, padding_idx, what, prepared,
                                           prep_bytes, hip_stream))
        return rc;
    return walk_edge_list_prepared(prepared, prep_bytes, what, edges, n_edges, node_edge_index, n_nodes, targets,
                                   n_walkers, walker_base, p, q, walk_length, seed, padding_idx, restart, mode, out,
                                   workspace, kStatusBytes, hip_stream, false);
}

// ---- window builders, reference csrc/cpu/windows_cpu.cpp ------------------------------------------

// A second stream for work that does not depend on the call's main kernel (the negatives of the window builders:
// glibc's stream modulo num_nodes, write-bound) so that it runs BESIDE it (the walk: bound by gather latency; the window
// gather: its own store stream): side_begin() makes the side stream wait for everything queued on the caller's stream up
// to side_mark(), side_join() makes the caller's stream wait for the side work.  Event-based fork / join is what a stream
// capture records as graph dependencies, so a captured call replays with the same overlap.
//
// Lanes.  A side stream, its fork event and its join event form a LANE; a device keeps a small pool of
// them, made once and reused (no stream or event is created or destroyed per call).  A call holds its lane from
// side_mark() to side_join(), so two host threads never share one.  A stream that a capture has pulled in stays in that
// capture until it ends -- so a lane is only handed to a call whose own stream is in the SAME capture state (both not
// capturing, or both in the capture with the same id): a thread that is not capturing never launches into somebody
// else's graph.  A capturing call takes a lane of its own capture or an idle one that is in no capture (it enters the
// caller's capture through the fork event and leaves it with the join: the captured graph holds the fork, a replay
// runs with the same overlap); only when the pool has neither does it stay in line (the work is queued on the caller's
// stream: same result, no overlap) rather than create a stream in the middle of a capture.
struct SideLane {
    hipStream_t stream = nullptr;
    hipEvent_t fork_ev = nullptr, join_ev = nullptr;
    bool held = false;  // between side_mark and side_join of one call
};
struct SideStream {
    SideLane *lane = nullptr;
    hipStream_t stream = nullptr;
    bool ok = false;
};

static std::mutex g_side_mu;
static std::vector<SideLane *> g_side_lanes[64];

static unsigned long long capture_id_of(hipStream_t st, bool *capturing)
{
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    if (hipStreamGetCaptureInfo(st, &cap, &id) != hipSuccess) {
        (void)hipGetLastError();
        *capturing = false;
        return 0;
    }
    *capturing = cap != hipStreamCaptureStatusNone;
    return *capturing ? id : 0;
}

static std::atomic<long long> g_side_acquired{0}, g_side_acquired_capturing{0};  // (trw_side_lane_counts: tests)

static SideLane *side_lane_acquire(hipStream_t main)
{
    int dev = 0;
    if (env_long("TRW_SIDE_STREAM", 0, 1, 1) == 0 || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    bool main_capturing = false;
    const unsigned long long main_id = capture_id_of(main, &main_capturing);
    std::lock_guard<std::mutex> lock(g_side_mu);
    // A lane of the caller's own capture first (it is in that graph already), then an idle one that is in no capture: it
    // joins the caller's through the fork event's hipStreamWaitEvent and leaves it with the join.  Never a lane that
    // sits in somebody else's capture, and a call that is not capturing never takes a capturing lane.
    SideLane *idle = nullptr;
    for (SideLane *l : g_side_lanes[dev]) {
        if (l->held) continue;
        bool lane_capturing = false;
        const unsigned long long lane_id = capture_id_of(l->stream, &lane_capturing);
        if (lane_capturing) {
            if (!main_capturing || lane_id != main_id) continue;
            idle = l;
            break;
        }
        if (!idle) idle = l;
    }
    if (idle) {
        idle->held = true;
        ++g_side_acquired;
        if (main_capturing) ++g_side_acquired_capturing;
        return idle;
    }
    if (main_capturing || g_side_lanes[dev].size() >= 16) return nullptr;  // never make a stream inside a capture
    SideLane *l = new SideLane;
    if (hipStreamCreateWithFlags(&l->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&l->fork_ev, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&l->join_ev, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        if (l->join_ev) (void)hipEventDestroy(l->join_ev);
        if (l->fork_ev) (void)hipEventDestroy(l->fork_ev);
        if (l->stream) (void)hipStreamDestroy(l->stream);
        delete l;
        return nullptr;
    }
    l->held = true;
    g_side_lanes[dev].push_back(l);
    ++g_side_acquired;
    return l;
}

static void side_lane_release(SideLane *l)
{
    if (!l) return;
    std::lock_guard<std::mutex> lock(g_side_mu);
    l->held = false;
}

// Marks the fork point on the caller's stream (everything queued so far); !ok: no side stream, the work stays in line.
static SideStream side_mark(hipStream_t main)
{
    SideStream s;
    s.lane = side_lane_acquire(main);
    if (!s.lane) return s;
    s.stream = s.lane->stream;
    if (hipEventRecord(s.lane->fork_ev, main) != hipSuccess) {
        (void)hipGetLastError();
        side_lane_release(s.lane);
        s.lane = nullptr;
        return s;
    }
    s.ok = true;
    return s;
}

// The stream the side work goes to: the side stream, made to wait for the fork point -- or the caller's own.
static hipStream_t side_begin(hipStream_t main, SideStream &s)
{
    if (s.ok && hipStreamWaitEvent(s.stream, s.lane->fork_ev, 0) == hipSuccess) return s.stream;
    s.ok = false;
    return main;
}

// The caller's stream waits for the side work; the lane goes back to the pool.
static void side_join(hipStream_t main, SideStream &s)
{
    if (s.ok) {
        if (hipEventRecord(s.lane->join_ev, s.stream) != hipSuccess || hipStreamWaitEvent(main, s.lane->join_ev, 0) != hipSuccess)
            (void)hipStreamSynchronize(s.stream);  // (never expected; keeps the ordering the caller relies on)
    }
    side_lane_release(s.lane);
    s.lane = nullptr;
    s.ok = false;
}

// centre / pos of every window: blocks of whole walks through LDS, or -- walks too long for the tile -- in place
static void launch_windows_gather(const int64_t *walks, int64_t num_walks, int64_t cols, int32_t window_size,
                                  int64_t *centre, int64_t *pos, hipStream_t stream)
{
    const int64_t step_end = cols - window_size + 1, per_walk = step_end * (window_size > 1 ? window_size - 1 : 1);
    int64_t wpb = kWinTileWords / cols;
    const int64_t by_out = (((int64_t)1 << 22) - 1) / per_walk;
    if (wpb > by_out) wpb = by_out;
    // enough blocks to fill the chip a few times over, each with a few thousand values to write
    while (wpb > 1 && (num_walks + wpb - 1) / wpb < 4096 && wpb * per_walk > 2048) wpb /= 2;
    if (wpb >= 1 && num_walks > 0 && (num_walks + wpb - 1) / wpb < ((int64_t)1 << 31))
        windows_gather_kernel<<<(unsigned)((num_walks + wpb - 1) / wpb), kBlock, 0, stream>>>(
            walks, num_walks, cols, window_size, (int32_t)wpb, centre, pos);
    else
        windows_gather_long_kernel<<<stream_grid(num_walks * step_end), kBlock, 0, stream>>>(walks, num_walks, cols,
                                                                                           window_size, centre, pos);
}

// Can this call shape run as ONE walk kernel that emits the windows from its LDS tile?  (Reference-exact uniform walk over
// the narrow layout; the tile's ring of 16 columns holds a window that started window_size + 6 columns ago.)
static bool walk_windows_fusable(int64_t n_nodes, int64_t nnz, int64_t n_walkers, int32_t walk_length, int32_t window_size,
                                 int32_t what)
{
    if (n_walkers <= 0 || walk_length < 1 || window_size < 1 || window_size > 9 || window_size > walk_length + 1) return false;
    if (what < 0 || !make_prep_layout(n_nodes, nnz, what).narrow) return false;
    const CallLayout cl = make_call_layout(n_walkers, walk_length, true, TRW_RNG_GLIBC);
    // (a wave's 64 rows of windows are addressed with 32-bit element offsets)
    return (int64_t)64 * cl.G * ((int64_t)walk_length + 1) < ((int64_t)1 << 29);
}

int trw_walk_windows_fused(int64_t n_nodes, int64_t nnz, int64_t n_walkers, int32_t walk_length, int32_t window_size,
                           int32_t what)
{
    return walk_windows_fusable(n_nodes, nnz, n_walkers, walk_length, window_size, what) ? 1 : 0;
}

size_t trw_walk_windows_workspace_bytes(int64_t n_walkers, int32_t walk_length, int32_t window_size)
{
    const size_t walk_bytes = align_up(make_call_layout(n_walkers, walk_length, true, TRW_RNG_GLIBC).total, 256);
    return walk_bytes + trw_to_windows_workspace_bytes(n_walkers, (int64_t)walk_length + 1, window_size);
}

int trw_walk_windows_csr_prepared_hip(const void *prepared, size_t prepared_bytes, int32_t what, const int64_t *row_ptr,
                                      int64_t n_nodes, const int64_t *col_idx, int64_t nnz, const int64_t *targets,
                                      int64_t n_walkers, int64_t walker_base, int32_t walk_length, int32_t seed,
                                      int32_t window_size, int64_t num_nodes, int32_t window_seed, int64_t *walks,
                                      int64_t *target_nodes, int64_t *pos_windows, int64_t *neg_windows, void *workspace,
                                      size_t workspace_bytes, void *hip_stream)
{
    t_last_error.clear();
    const int64_t cols = (int64_t)walk_length + 1;
    if (walk_length < 1 || window_size < 1 || window_size > cols)
        return fail(TRW_E_ARG, "window_size must be between 1 and walk_length + 1");
    if (num_nodes < 1) return fail(TRW_E_ARG, "num_nodes must be >= 1 (the reference computes rand() % num_nodes)");
    if (n_walkers < 0 || walker_base < 0) return fail(TRW_E_ARG, "negative size argument");
    if (!workspace || workspace_bytes < trw_walk_windows_workspace_bytes(n_walkers, walk_length, window_size))
        return fail(TRW_E_WORKSPACE, "workspace too small for this call shape");
    const int64_t step_end = cols - window_size + 1, width = window_size - 1;
    const int64_t total_windows = n_walkers * step_end, draws = total_windows * width;
    if (total_windows > 0 && (!target_nodes || (width > 0 && (!pos_windows || !neg_windows))))
        return fail(TRW_E_ARG, "output pointer is null");
    const bool fused = walk_windows_fusable(n_nodes, nnz, n_walkers, walk_length, window_size, what);
    if (!fused && !walks && n_walkers > 0)
        return fail(TRW_E_ARG, "this call shape is not fused (trw_walk_windows_fused): it needs the walks buffer");
    const unsigned __int128 last = ((unsigned __int128)walker_base + n_walkers) * step_end * width;
    if (last >= ((unsigned __int128)1 << 48)) return fail(TRW_E_RANGE, "glibc stream position beyond 2^48 draws");
    hipStream_t stream = (hipStream_t)hip_stream;
    const size_t walk_bytes = align_up(make_call_layout(n_walkers, walk_length, true, TRW_RNG_GLIBC).total, 256);
    // The negatives do not depend on the walks (glibc's stream modulo num_nodes, as trw_to_windows_hip fills it): they are
    // written on the side stream, beside the walk.  The fork point is here, in front of the walk, but the walk is queued
    // FIRST: a walk that is already resident leaves the fill the CU share it needs (one workgroup per CU beside the
    // walk's five), a fill that starts first takes the LDS of every CU and the walk waits for it to finish.
    SideStream side;
    if (draws > 0) {
        if (int rc = ensure_jump_table()) return rc;
        side = side_mark(stream);
    }
    FusedWindows fw;
    fw.window_size = window_size;
    fw.centre = target_nodes;
    fw.pos = pos_windows;
    // the walk (its status block is the call's), with or without the windows coming out of the same kernel
    const int rc_walk = walk_prepared(prepared, prepared_bytes, what, row_ptr, n_nodes, col_idx, nnz, targets, n_walkers,
                                      walker_base, 1.0, 1.0, walk_length, seed, TRW_RNG_GLIBC, walks, workspace, walk_bytes,
                                      hip_stream, true, fused ? &fw : nullptr);
    if (draws > 0 && rc_walk == TRW_OK) {
        hipStream_t ss = side_begin(stream, side);
        uint32_t *win = (uint32_t *)((char *)workspace + walk_bytes + kStatusBytes);
        const int64_t nlanes = (draws + fill_run(draws) - 1) / fill_run(draws);
        SeedWindow sw;
        trw::seed_window(window_seed, sw.w);
        glibc_windows_kernel<<<(unsigned)((nlanes + kBlock - 1) / kBlock), kBlock, 0, ss>>>(
            sw, (uint64_t)walker_base * (uint64_t)step_end * (uint64_t)width, (uint64_t)fill_run(draws), nlanes, win);
        glibc_fill_mod_kernel<false><<<(unsigned)((nlanes + kBlock - 1) / kBlock), kBlock, 0, ss>>>(
            win, nlanes, draws, num_nodes, nullptr, neg_windows);
    }
    if (rc_walk == TRW_OK && total_windows > 0 && !fused)
        launch_windows_gather(walks, n_walkers, cols, window_size, target_nodes, pos_windows, stream);
    side_join(stream, side);
    if (rc_walk != TRW_OK) return rc_walk;
    if (total_windows == 0) return TRW_OK;
    prof_mark(3, stream);  // (the profile's "finish" phase: everything after the walk kernel)
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "kernel launch");
    return TRW_OK;
}

static int windows_check(const int64_t *walks, int64_t num_walks, int64_t walk_cols, int32_t window_size,
                         int64_t num_nodes, int64_t base, void *workspace, size_t workspace_bytes)
{
    if (num_walks < 0 || base < 0) return fail(TRW_E_ARG, "negative size argument");
    if (walk_cols < 1 || window_size < 1 || window_size > walk_cols)
        return fail(TRW_E_ARG, "window_size must be between 1 and the walk width");
    if (num_nodes < 1) return fail(TRW_E_ARG, "num_nodes must be >= 1 (the reference computes rand() % num_nodes)");
    if (num_walks > 0 && !walks) return fail(TRW_E_ARG, "walks is null");
    if (!workspace || workspace_bytes < (size_t)kStatusBytes)
        return fail(TRW_E_WORKSPACE, "workspace missing or smaller than the status block");
    if (((uintptr_t)workspace & 15u) != 0) return fail(TRW_E_WORKSPACE, "workspace must be 16-byte aligned");
    return TRW_OK;
}

size_t trw_to_windows_workspace_bytes(int64_t num_walks, int64_t walk_cols, int32_t window_size)
{
    if (num_walks <= 0 || walk_cols < 1 || window_size < 1 || window_size > walk_cols) return kStatusBytes;
    const int64_t draws = num_walks * (walk_cols - window_size + 1) * (int64_t)(window_size - 1);
    const int64_t nlanes = (draws + fill_run(draws) - 1) / fill_run(draws);
    return kStatusBytes + align_up((size_t)nlanes * kGlDeg * sizeof(uint32_t), 256);
}

int trw_to_windows_hip(const int64_t *walks, int64_t num_walks, int64_t walk_cols, int32_t window_size,
                       int64_t num_nodes, int32_t seed, int64_t walk_base, int64_t *target_nodes,
                       int64_t *pos_windows, int64_t *neg_windows, void *workspace, size_t workspace_bytes,
                       void *hip_stream)
{
    t_last_error.clear();
    if (int rc = windows_check(walks, num_walks, walk_cols, window_size, num_nodes, walk_base, workspace,
                               workspace_bytes))
        return rc;
    hipStream_t stream = (hipStream_t)hip_stream;
    int64_t *status = (int64_t *)workspace;
    hipError_t e = hipSuccess;
    device_fill(status, 0u, kStatusBytes, stream);
    const int64_t step_end = walk_cols - window_size + 1;
    const int64_t total_windows = num_walks * step_end;
    if (total_windows == 0) return TRW_OK;
    if (!target_nodes || (window_size > 1 && (!pos_windows || !neg_windows)))
        return fail(TRW_E_ARG, "output pointer is null");
    const int64_t width = window_size - 1;
    const int64_t draws = total_windows * width;
    const unsigned __int128 last = ((unsigned __int128)walk_base + num_walks) * step_end * width;
    if (last >= ((unsigned __int128)1 << 48)) return fail(TRW_E_RANGE, "glibc stream position beyond 2^48 draws");
    if (workspace_bytes < trw_to_windows_workspace_bytes(num_walks, walk_cols, window_size))
        return fail(TRW_E_WORKSPACE, "workspace too small for this call shape");
    if (int rc = ensure_jump_table()) return rc;

    prof_mark(0, stream);
    uint32_t *win = (uint32_t *)((char *)workspace + kStatusBytes);
    const int64_t nlanes = (draws + fill_run(draws) - 1) / fill_run(draws);
    prof_mark(1, stream);
    // two independent store streams, side by side: the negatives (the stream modulo num_nodes) on the side stream, the
    // windows' own nodes on the caller's
    SideStream side;
    if (draws > 0) {
        side = side_mark(stream);
        hipStream_t ss = side_begin(stream, side);
        SeedWindow sw;
        trw::seed_window(seed, sw.w);
        glibc_windows_kernel<<<(unsigned)((nlanes + kBlock - 1) / kBlock), kBlock, 0, ss>>>(
            sw, (uint64_t)walk_base * (uint64_t)step_end * (uint64_t)width, (uint64_t)fill_run(draws), nlanes, win);
        glibc_fill_mod_kernel<false><<<(unsigned)((nlanes + kBlock - 1) / kBlock), kBlock, 0, ss>>>(
            win, nlanes, draws, num_nodes, nullptr, neg_windows);
    }
    launch_windows_gather(walks, num_walks, walk_cols, window_size, target_nodes, pos_windows, stream);
    side_join(stream, side);
    prof_mark(2, stream);
    prof_mark(3, stream);
    e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "kernel launch");
    return TRW_OK;
}

int trw_to_windows_cbow_hip(const int64_t *walks, int64_t num_walks, int64_t walk_cols, int32_t window_size,
                            int64_t num_nodes, int32_t seed, int64_t window_base, int32_t rng_mode,
                            int64_t *pos_nodes, int64_t *neg_nodes, int64_t *windows, void *workspace,
                            size_t workspace_bytes, void *hip_stream)
{
    t_last_error.clear();
    if (int rc = windows_check(walks, num_walks, walk_cols, window_size, num_nodes, window_base, workspace,
                               workspace_bytes))
        return rc;
    const int64_t total_windows = num_walks * (walk_cols - window_size + 1);
    int mode = rng_mode;
    if (mode == TRW_RNG_AUTO)  // serial replay (one lane) while the job is small, one stream per window beyond
        mode = (window_base == 0 && total_windows <= ((int64_t)1 << 18)) ? TRW_RNG_GLIBC : TRW_RNG_WALKER;
    if (mode != TRW_RNG_GLIBC && mode != TRW_RNG_WALKER) return fail(TRW_E_RNG_MODE, "unknown rng_mode");
    if (mode == TRW_RNG_GLIBC && window_base != 0)
        return fail(TRW_E_RNG_MODE, "TRW_RNG_GLIBC replays one serial stream: window_base must be 0");
    hipStream_t stream = (hipStream_t)hip_stream;
    int64_t *status = (int64_t *)workspace;
    hipError_t e = hipSuccess;
    device_fill(status, 0u, kStatusBytes, stream);
    if (total_windows == 0) return TRW_OK;
    if (!pos_nodes || !neg_nodes || (window_size > 1 && !windows)) return fail(TRW_E_ARG, "output pointer is null");

    prof_mark(0, stream);
    prof_mark(1, stream);
    launch_windows_gather(walks, num_walks, walk_cols, window_size, pos_nodes, windows, stream);
    if (mode == TRW_RNG_GLIBC) {
        SeedWindow sw;
        trw::seed_window(seed, sw.w);
        cbow_negatives_glibc_seq_kernel<<<1, 64, 0, stream>>>(pos_nodes, total_windows, num_nodes, sw, neg_nodes,
                                                             status);
    } else {
        cbow_negatives_walker_kernel<<<(unsigned)((total_windows + kBlock - 1) / kBlock), kBlock, 0, stream>>>(
            pos_nodes, total_windows, num_nodes, window_base, (uint32_t)seed, neg_nodes, status);
    }
    prof_mark(2, stream);
    prof_mark(3, stream);
    e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "kernel launch");
    return TRW_OK;
}

static int triples_windows_check(const int64_t *walks, int64_t num_walks, int64_t walk_cols, int32_t window_size,
                                 const int64_t *triples, int64_t num_triples, int64_t base, void *workspace,
                                 size_t workspace_bytes)
{
    if (num_walks < 0 || base < 0) return fail(TRW_E_ARG, "negative size argument");
    if (walk_cols < 1 || window_size < 1) return fail(TRW_E_ARG, "window_size and the walk width must be >= 1");
    if (num_triples < 1 || !triples) return fail(TRW_E_ARG, "triples must hold at least one row");
    if (num_walks > 0 && !walks) return fail(TRW_E_ARG, "walks is null");
    if (!workspace || workspace_bytes < (size_t)kStatusBytes)
        return fail(TRW_E_WORKSPACE, "workspace missing or smaller than the status block");
    if (((uintptr_t)workspace & 15u) != 0) return fail(TRW_E_WORKSPACE, "workspace must be 16-byte aligned");
    return TRW_OK;
}

namespace {
// target triples + positive windows of the two triples window builders
void launch_triples_gather(const int64_t *walks, int64_t num_walks, int64_t walk_cols, int32_t window_size,
                           int64_t padding_idx, int64_t *target_triples, int64_t *pos_windows, hipStream_t stream)
{
    const int64_t per_walk = (walk_cols - 1) / 2;
    const int64_t per_walk_elems = per_walk * 2 * (int64_t)window_size * 3;
    if (num_walks <= 0 || per_walk <= 0) return;
    if (per_walk_elems < ((int64_t)1 << 22) && num_walks < ((int64_t)1 << 31)) {
        int64_t wpb = 32768 / per_walk_elems;  // ~32K elements (256 KB) per block
        wpb = wpb < 1 ? 1 : wpb;
        const int64_t blocks = (num_walks + wpb - 1) / wpb;
        triples_windows_gather_kernel<<<(unsigned)blocks, kBlock, 0, stream>>>(
            walks, num_walks, walk_cols, window_size, padding_idx, (int32_t)wpb, target_triples, pos_windows);
    } else {
        triples_windows_gather_long_kernel<<<stream_grid(num_walks * per_walk), kBlock, 0, stream>>>(
            walks, num_walks, walk_cols, window_size, padding_idx, target_triples, pos_windows);
    }
}
}  // namespace

size_t trw_to_windows_triples_workspace_bytes(int64_t num_walks, int64_t walk_cols, int32_t window_size)
{
    if (num_walks <= 0 || walk_cols < 1 || window_size < 1) return kStatusBytes;
    const int64_t draws = num_walks * ((walk_cols - 1) / 2) * 2 * (int64_t)window_size;
    const int64_t nlanes = (draws + fill_run(draws) - 1) / fill_run(draws);
    return kStatusBytes + align_up((size_t)nlanes * kGlDeg * sizeof(uint32_t), 256);
}

int trw_to_windows_triples_hip(const int64_t *walks, int64_t num_walks, int64_t walk_cols, int32_t window_size,
                               int64_t padding_idx, const int64_t *triples, int64_t num_triples, int32_t seed,
                               int64_t walk_base, int64_t *target_triples, int64_t *pos_windows,
                               int64_t *neg_windows, void *workspace, size_t workspace_bytes, void *hip_stream)
{
    t_last_error.clear();
    if (int rc = triples_windows_check(walks, num_walks, walk_cols, window_size, triples, num_triples, walk_base,
                                       workspace, workspace_bytes))
        return rc;
    hipStream_t stream = (hipStream_t)hip_stream;
    hipError_t e = hipSuccess;
    device_fill(workspace, 0u, kStatusBytes, stream);
    const int64_t per_walk = (walk_cols - 1) / 2;
    const int64_t total_windows = num_walks * per_walk;
    if (total_windows == 0) return TRW_OK;
    if (!target_triples || !pos_windows || !neg_windows) return fail(TRW_E_ARG, "output pointer is null");
    const int64_t slots = 2 * (int64_t)window_size;
    // sample_int(0, num_triples-1) draws nothing when there is a single triple (cpu_utils.cpp:4-5)
    const int64_t draws = total_windows * slots;
    const unsigned __int128 last = ((unsigned __int128)walk_base + num_walks) * per_walk * slots;
    if (last >= ((unsigned __int128)1 << 48)) return fail(TRW_E_RANGE, "glibc stream position beyond 2^48 draws");
    if (workspace_bytes < trw_to_windows_triples_workspace_bytes(num_walks, walk_cols, window_size))
        return fail(TRW_E_WORKSPACE, "workspace too small for this call shape");
    if (int rc = ensure_jump_table()) return rc;
    prof_mark(0, stream);
    uint32_t *win = (uint32_t *)((char *)workspace + kStatusBytes);
    const int64_t nlanes = (draws + fill_run(draws) - 1) / fill_run(draws);
    SeedWindow sw;
    trw::seed_window(seed, sw.w);
    glibc_windows_kernel<<<(unsigned)((nlanes + kBlock - 1) / kBlock), kBlock, 0, stream>>>(
        sw, num_triples == 1 ? 0 : (uint64_t)walk_base * (uint64_t)per_walk * (uint64_t)slots, (uint64_t)fill_run(draws),
        nlanes, win);
    prof_mark(1, stream);
    launch_triples_gather(walks, num_walks, walk_cols, window_size, padding_idx, target_triples, pos_windows, stream);
    glibc_fill_mod_kernel<true><<<(unsigned)((nlanes + kBlock - 1) / kBlock), kBlock, 0, stream>>>(
        win, nlanes, draws, num_triples, triples, neg_windows);
    prof_mark(2, stream);
    prof_mark(3, stream);
    e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "kernel launch");
    return TRW_OK;
}

int trw_to_windows_triples_cbow_hip(const int64_t *walks, int64_t num_walks, int64_t walk_cols, int32_t window_size,
                                    int64_t padding_idx, const int64_t *triples, int64_t num_triples, int32_t seed,
                                    int64_t window_base, int32_t rng_mode, int64_t *pos_triples,
                                    int64_t *neg_triples, int64_t *pos_windows, void *workspace,
                                    size_t workspace_bytes, void *hip_stream)
{
    t_last_error.clear();
    if (int rc = triples_windows_check(walks, num_walks, walk_cols, window_size, triples, num_triples, window_base,
                                       workspace, workspace_bytes))
        return rc;
    const int64_t total_windows = num_walks * ((walk_cols - 1) / 2);
    int mode = rng_mode;
    if (mode == TRW_RNG_AUTO)
        mode = (window_base == 0 && total_windows <= ((int64_t)1 << 18)) ? TRW_RNG_GLIBC : TRW_RNG_WALKER;
    if (mode != TRW_RNG_GLIBC && mode != TRW_RNG_WALKER) return fail(TRW_E_RNG_MODE, "unknown rng_mode");
    if (mode == TRW_RNG_GLIBC && window_base != 0)
        return fail(TRW_E_RNG_MODE, "TRW_RNG_GLIBC replays one serial stream: window_base must be 0");
    hipStream_t stream = (hipStream_t)hip_stream;
    int64_t *status = (int64_t *)workspace;
    hipError_t e = hipSuccess;
    device_fill(status, 0u, kStatusBytes, stream);
    if (total_windows == 0) return TRW_OK;
    if (!pos_triples || !neg_triples || !pos_windows) return fail(TRW_E_ARG, "output pointer is null");
    prof_mark(0, stream);
    prof_mark(1, stream);
    launch_triples_gather(walks, num_walks, walk_cols, window_size, padding_idx, pos_triples, pos_windows, stream);
    SeedWindow sw;
    trw::seed_window(seed, sw.w);
    if (mode == TRW_RNG_GLIBC)
        cbow_negative_triples_kernel<true><<<1, 64, 0, stream>>>(pos_triples, total_windows, triples, num_triples, 0,
                                                                (uint32_t)seed, sw, neg_triples, status);
    else
        cbow_negative_triples_kernel<false><<<(unsigned)((total_windows + kBlock - 1) / kBlock), kBlock, 0, stream>>>(
            pos_triples, total_windows, triples, num_triples, window_base, (uint32_t)seed, sw, neg_triples, status);
    prof_mark(2, stream);
    prof_mark(3, stream);
    e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "kernel launch");
    return TRW_OK;
}

void trw_side_lane_counts(int64_t counts[2])
{
    if (!counts) return;
    counts[0] = g_side_acquired.load();
    counts[1] = g_side_acquired_capturing.load();
}

int trw_profile_enable(int flags)
{
    g_prof.on = (flags & 1) != 0;
    g_prof.count = (flags & 2) != 0;
    g_prof.valid = false;
    g_prof.prep_valid = false;
    return TRW_OK;
}

int trw_profile_prepare_last(double *prepare_ms)
{
    t_last_error.clear();
    if (!g_prof.on || !g_prof.prep_valid || !prepare_ms) return fail(TRW_E_ARG, "no profiled prepare recorded");
    hipError_t e = hipEventSynchronize(g_prof.prep[1]);
    if (e != hipSuccess) return fail_hip(e, "hipEventSynchronize");
    float ms = 0.f;
    e = hipEventElapsedTime(&ms, g_prof.prep[0], g_prof.prep[1]);
    if (e != hipSuccess) return fail_hip(e, "hipEventElapsedTime");
    *prepare_ms = ms;
    g_prof.prep_valid = false;  // reported once
    return TRW_OK;
}

int trw_profile_last(double *prepare_ms, double *walk_ms, double *finish_ms)
{
    t_last_error.clear();
    if (!g_prof.on || !g_prof.valid) return fail(TRW_E_ARG, "no profiled walk recorded");
    hipError_t e = hipEventSynchronize(g_prof.ev[3]);
    if (e != hipSuccess) return fail_hip(e, "hipEventSynchronize");
    double *outs[3] = {prepare_ms, walk_ms, finish_ms};
    for (int k = 0; k < 3; ++k) {
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, g_prof.ev[k], g_prof.ev[k + 1]);
        if (e != hipSuccess) return fail_hip(e, "hipEventElapsedTime");
        if (outs[k]) *outs[k] = ms;
    }
    return TRW_OK;
}

int trw_walk_status(const void *workspace, void *hip_stream, int64_t status_out[4])
{
    t_last_error.clear();
    if (!workspace || !status_out) return fail(TRW_E_ARG, "null argument");
    hipStream_t stream = (hipStream_t)hip_stream;
    int64_t host[16] = {0};
    hipError_t e = hipMemcpyAsync(host, workspace, sizeof host, hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) return fail_hip(e, "hipMemcpyAsync(status)");
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return fail_hip(e, "hipStreamSynchronize");
    status_out[0] = host[0];
    status_out[1] = (host[1] != 0 || host[kStDescents] != host[kStBoundary]) ? 1 : 0;
    status_out[2] = host[2];
    status_out[3] = host[3];
    return TRW_OK;
}

int trw_selftest_hip(void *scratch, void *hip_stream, uint64_t *failures)
{
    t_last_error.clear();
    if (!scratch || !failures) return fail(TRW_E_ARG, "null argument");
    hipStream_t stream = (hipStream_t)hip_stream;
    device_fill(scratch, 0u, 16, stream);
    fastmod_selftest_kernel<<<(unsigned)(device_cu_count() * 8), kBlock, 0, stream>>>(1u, 0x80000000u,
                                                                                       (unsigned long long *)scratch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "fastmod_selftest_kernel");
    unsigned long long host = 0;
    e = hipMemcpyAsync(&host, scratch, sizeof host, hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) return fail_hip(e, "hipMemcpyAsync(selftest)");
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return fail_hip(e, "hipStreamSynchronize");
    *failures = host;
    return TRW_OK;
}

// ---- deferred status: asynchronous read-backs of status blocks -----------------------------------
namespace {
struct StatusSlot {
    int64_t *host = nullptr;  // pinned, 16 words
    hipEvent_t ev = nullptr;
    int device = -1;          // the device `ev` was created on: an event is only ever recorded on a stream of its own device
    bool busy = false;
};
constexpr int kStatusSlots = 64;
std::mutex g_status_mu;
StatusSlot g_status_slots[kStatusSlots];
int64_t g_status_carry = 0;  // bits of read-backs that were finished to make room, not yet collected

// ORs the bits of every finished read-back into *bits and frees its slot; returns how many are still in flight.
int status_sweep(bool wait, int64_t *bits)
{
    int pending = 0;
    for (auto &sl : g_status_slots) {
        if (!sl.busy) continue;
        hipError_t e = wait ? hipEventSynchronize(sl.ev) : hipEventQuery(sl.ev);
        if (e == hipErrorNotReady) {
            ++pending;
            continue;
        }
        sl.busy = false;  // finished (or failed: nothing more to learn from it)
        if (e == hipSuccess && bits) *bits |= sl.host[0];
    }
    return pending;
}
}  // namespace

// ---- small stream helpers for hosts that keep prepared graphs across streams ---------------------------
int trw_stream_is_capturing(void *hip_stream)
{
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)hip_stream, &cap) != hipSuccess) return 0;
    return cap != hipStreamCaptureStatusNone ? 1 : 0;
}

void *trw_event_record(void *hip_stream)
{
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return nullptr;
    if (hipEventRecord(ev, (hipStream_t)hip_stream) != hipSuccess) {
        (void)hipEventDestroy(ev);
        return nullptr;
    }
    return (void *)ev;
}

int trw_event_wait(void *event, void *hip_stream)
{
    if (!event) return TRW_OK;
    const hipError_t e = hipStreamWaitEvent((hipStream_t)hip_stream, (hipEvent_t)event, 0);
    return e == hipSuccess ? TRW_OK : fail_hip(e, "hipStreamWaitEvent");
}

void trw_event_destroy(void *event)
{
    if (event) (void)hipEventDestroy((hipEvent_t)event);
}

int trw_status_post(const void *workspace, void *hip_stream)
{
    t_last_error.clear();
    if (!workspace) return fail(TRW_E_ARG, "null argument");
    hipStream_t stream = (hipStream_t)hip_stream;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) return TRW_OK;
    int device = 0;
    hipError_t e = hipGetDevice(&device);  // (the caller's device guard made the stream's device current)
    if (e != hipSuccess) return fail_hip(e, "hipGetDevice");
    std::lock_guard<std::mutex> lock(g_status_mu);
    // A slot is reused on the device that made its event only: free slots of this device first, then
    // one that has no event yet, then -- every slot taken by other devices -- a free foreign one, re-made here.
    StatusSlot *slot = nullptr;
    for (int round = 0; round < 2 && !slot; ++round) {
        StatusSlot *fresh = nullptr, *foreign = nullptr;
        for (auto &sl : g_status_slots) {
            if (sl.busy) continue;
            if (sl.device == device) {
                slot = &sl;
                break;
            }
            if (sl.device < 0) {
                if (!fresh) fresh = &sl;
            } else if (!foreign) {
                foreign = &sl;
            }
        }
        if (!slot) slot = fresh ? fresh : foreign;
        if (!slot) status_sweep(true, &g_status_carry);  // every slot in flight: finish them, keep their bits
    }
    if (!slot) return fail(TRW_E_HIP, "no status slot free");
    if (slot->device != device && slot->ev) {
        (void)hipEventDestroy(slot->ev);
        slot->ev = nullptr;
    }
    if (!slot->host) {
        e = hipHostMalloc((void **)&slot->host, 16 * sizeof(int64_t), hipHostMallocPortable);
        if (e != hipSuccess) return fail_hip(e, "hipHostMalloc(status slot)");
    }
    if (!slot->ev) {
        slot->device = -1;
        e = hipEventCreateWithFlags(&slot->ev, hipEventDisableTiming);
        if (e != hipSuccess) return fail_hip(e, "hipEventCreate(status slot)");
        slot->device = device;
    }
    slot->host[0] = 0;
    e = hipMemcpyAsync(slot->host, workspace, 16 * sizeof(int64_t), hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) return fail_hip(e, "hipMemcpyAsync(status)");
    e = hipEventRecord(slot->ev, stream);
    if (e != hipSuccess) {
        // the copy is in flight into a slot nobody will wait for: let it land before the slot can be handed out again
        (void)hipStreamSynchronize(stream);
        return fail_hip(e, "hipEventRecord(status)");
    }
    slot->busy = true;
    return TRW_OK;
}

int trw_status_collect(int wait, int64_t *bits)
{
    t_last_error.clear();
    if (!bits) return fail(TRW_E_ARG, "null argument");
    std::lock_guard<std::mutex> lock(g_status_mu);
    *bits = g_status_carry;
    g_status_carry = 0;
    return status_sweep(wait != 0, bits);
}

// ---- wire format of the multi-GPU gather (trw_kernels_wire.hpp) -----------------------------------------
int trw_rows_narrow_hip(const int64_t *rows, int64_t count, uint32_t *wire, int64_t *overflow, void *hip_stream)
{
    t_last_error.clear();
    if (count < 0 || (count > 0 && (!rows || !wire))) return fail(TRW_E_ARG, "null pointer or negative count");
    if (count == 0) return TRW_OK;
    hipStream_t stream = (hipStream_t)hip_stream;
    const bool aligned = (((uintptr_t)rows | (uintptr_t)wire) & 15u) == 0 && count / 4 / 256 < ((int64_t)1 << 31) - 1;
    if (aligned)  // one trip per thread: a grid of count / 4 threads (+1 block: the last count % 4 values)
        rows_narrow_kernel<<<(unsigned)(count / 4 / 256 + 1), 256, 0, stream>>>(rows, count, wire,
                                                                               (unsigned long long *)overflow);
    else
        rows_narrow_scalar_kernel<<<stream_grid(count), 256, 0, stream>>>(rows, count, wire,
                                                                          (unsigned long long *)overflow);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? TRW_OK : fail_hip(e, "rows_narrow_kernel");
}

int trw_rows_widen_hip(const uint32_t *wire, int64_t count, int64_t *rows, void *hip_stream)
{
    t_last_error.clear();
    if (count < 0 || (count > 0 && (!rows || !wire))) return fail(TRW_E_ARG, "null pointer or negative count");
    if (count == 0) return TRW_OK;
    hipStream_t stream = (hipStream_t)hip_stream;
    const bool aligned = (((uintptr_t)rows | (uintptr_t)wire) & 15u) == 0 && count / 4 / 256 < ((int64_t)1 << 31) - 1;
    if (aligned)
        rows_widen_kernel<<<(unsigned)(count / 4 / 256 + 1), 256, 0, stream>>>(wire, count, rows);
    else
        rows_widen_scalar_kernel<<<stream_grid(count), 256, 0, stream>>>(wire, count, rows);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? TRW_OK : fail_hip(e, "rows_widen_kernel");
}

int trw_walk_counters(const void *workspace, void *hip_stream, uint64_t counters_out[4])
{
    t_last_error.clear();
    if (!workspace || !counters_out) return fail(TRW_E_ARG, "null argument");
    hipStream_t stream = (hipStream_t)hip_stream;
    uint64_t host[4] = {0, 0, 0, 0};
    hipError_t e = hipMemcpyAsync(host, (const char *)workspace + 4 * sizeof(int64_t), sizeof host,
                                  hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) return fail_hip(e, "hipMemcpyAsync(counters)");
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return fail_hip(e, "hipStreamSynchronize");
    for (int k = 0; k < 4; ++k) counters_out[k] = host[k];
    return TRW_OK;
}

}  // extern "C"
