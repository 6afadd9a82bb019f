// trw_kernels_edgelist.hpp -- edge-list walk, SURVEY 8(f) row f-2.
//
// Reference: walk_edge_list_cpu (csrc/cpu/rw_cpu_edge_list.cpp:240-265) ->
// uniform_walk_edge_list (:64-126) | biased_walk_edge_list (:128-238),
// sample_neighbor (:8-35), is_neighbor (:37-62), sample_int (cpu_utils.cpp:3-10).
// The graph is an edge list [E,2] sorted by head plus an inclusive (first,last)
// index per node; dead ends step to the padding id and a step from the padding id
// goes to the jump node (start node if `restart`, else padding) without a draw.
// Kept quirks: is_neighbor never looks at a head's last edge; in the biased loop a
// rejected return to t is tested again as neighbour / non-neighbour.
//
//   elist_shadow_kernel     int64 edges/index -> uint32 tails + (first,last) pairs, validated
//   elist_walk_kernel<B,N>  one walker per lane on the per-walker stream (B: biased, N: uint32 shadow
//                           + rows through WaveRows, else int64 in place)
//   elist_glibc_seq_kernel  the reference's serial glibc stream replayed by one lane
// Included by trw_hip.hip only.
#pragma once

namespace {

__global__ void __launch_bounds__(kBlock)
elist_shadow_kernel(const int64_t *__restrict__ edges, int64_t n_edges, const int64_t *__restrict__ index,
                    int64_t n_nodes, int64_t padding_idx, uint32_t *__restrict__ tail32,
                    u32x2_a8 *__restrict__ ix32, int64_t *__restrict__ status)
{
    const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    uint32_t st = 0;
    for (int64_t e = tid; e < n_edges; e += stride) {
        int64_t tail = edges[2 * e + 1];
        if (tail != padding_idx && (uint64_t)tail >= (uint64_t)n_nodes) {
            st |= TRW_ST_NODE_RANGE;
            tail = padding_idx;
        }
        tail32[e] = (uint32_t)tail;
    }
    for (int64_t v = tid; v < n_nodes; v += stride) {
        const int64_t a = index[2 * v], b = index[2 * v + 1];
        u32x2_a8 o = {kNoEdge, kNoEdge};
        if (!(a == -1 || b == -1)) {  // rw_cpu_edge_list.cpp:21
            if (a < 0 || b < a || b >= n_edges) st |= TRW_ST_BAD_CSR;
            else {
                o.x = (uint32_t)a;
                o.y = (uint32_t)b;
            }
        }
        ix32[v] = o;
    }
    if (st) atomicOr((unsigned long long *)&status[0], (unsigned long long)st);
}

// The edge table of the edge-list walk: one slot {u, v, ~posmask, where} per distinct (head, tail) pair, every edge
// included (all bytes 0xFF when empty).  Tails are in input order inside a head's range, not sorted, so "where is v in
// u's range" is a set of offsets:
//   posmask bit b   (kept inverted: inserts only clear bits) some occurrence sits at an offset == b (mod 32)
//   where >> 2      the FIRST such offset (atomicMin over the occurrences' (offset << 2 | last << 1))
//   where bit 1     0 when some occurrence is not u's last edge: is_neighbor(x, t) looks at t's edges [first, last) --
//                   never the last one (rw_cpu_edge_list.cpp:51) -- so THIS bit is the reference's membership test
//                   (the minimum is the first occurrence, which is the last edge only when it is the only one)
//   where bit 0     1 when the pair occurs more than once: every occurrence that finds the slot taken sets it AFTER its
//                   atomicMin, so whichever minimum stands last carries it
// With one occurrence the first offset is where the pair is, exactly: a filter over a hub's range then has 128
// distinct bits instead of the 32 of the modulo mask (which round 2 used alone: half of all steps looked a candidate up).
__global__ void __launch_bounds__(kBlock)
elist_table_build_kernel(const int64_t *__restrict__ edges, int64_t n_edges, int64_t n_nodes,
                         const uint32_t *__restrict__ tail32, const u32x2_a8 *__restrict__ ix32,
                         uint32_t *__restrict__ table, uint32_t mask)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n_edges; e += stride) {
        const int64_t head = edges[2 * e];
        if ((uint64_t)head >= (uint64_t)n_nodes) continue;  // never indexed by a walk
        const u32x2_a8 ix = ix32[head];
        if (ix.x == kNoEdge || (uint32_t)e < ix.x || (uint32_t)e > ix.y) continue;  // outside the head's range
        const uint32_t u = (uint32_t)head, v = tail32[e];
        const unsigned long long key = ((unsigned long long)v << 32) | u;
        uint32_t slot = edge_hash(u, v) & mask;
        while (true) {
            uint32_t *ent = table + 4 * (size_t)slot;
            const unsigned long long old = atomicCAS((unsigned long long *)ent, ~0ull, key);
            if (old == ~0ull || old == key) {
                const uint32_t off = (uint32_t)e - ix.x;
                atomicAnd(ent + 2, ~(1u << (off & 31u)));
                const uint32_t before = atomicMin(ent + 3, (off << 2) | ((uint32_t)e == ix.y ? 2u : 0u));
                if (before != 0xFFFFFFFFu) atomicOr(ent + 3, 1u);  // not the first to arrive: the pair is a multi-edge
                break;
            }
            slot = (slot + 1) & mask;
        }
    }
}

// the slot of pair (u, v): false when absent; `first` = first offset of v in u's range (where it is, when `posmask`
// comes back 0: a single occurrence), `posmask` = the offsets modulo 32 of a pair that occurs more than once
__device__ __forceinline__ bool elist_table_find(const uint32_t *__restrict__ table, uint32_t mask, uint32_t u,
                                                 uint32_t v, uint32_t &first, uint32_t &posmask, bool &in_range)
{
    uint32_t slot = edge_hash(u, v) & mask;
    while (true) {
        const u32x4_a16 kv = *reinterpret_cast<const u32x4_a16 *>(table + 4 * (size_t)slot);
        if (kv.x == 0xFFFFFFFFu && kv.y == 0xFFFFFFFFu) return false;
        if (kv.x == u && kv.y == v) {
            first = kv.w >> 2;
            posmask = (kv.w & 1u) ? ~kv.z : 0u;
            in_range = (kv.w & 2u) == 0u;
            return true;
        }
        slot = (slot + 1) & mask;
    }
}

// is_neighbor(x, t) of the reference: is v a tail of one of u's edges other than its last?
__device__ __forceinline__ bool elist_table_contains(const uint32_t *__restrict__ table, uint32_t mask, uint32_t u,
                                                     uint32_t v)
{
    uint32_t first, pm;
    bool in_range = false;
    return elist_table_find(table, mask, u, v, first, pm, in_range) && in_range;
}

// Uniform views of the two layouts: node ids as int64, "no edges" as first < 0.
template <bool kNarrow>
struct EdgeListView {
    const int64_t *edges;   // wide
    const int64_t *index;   // wide
    const uint32_t *tail32; // narrow
    const u32x2_a8 *ix32;   // narrow
    int64_t n_edges, n_nodes;

    // (first, last) of `node`; first = -1 when it has no out-edges.  Reports what it cannot trust.
    __device__ __forceinline__ void range(int64_t node, int64_t &first, int64_t &last, uint32_t &st) const
    {
        if (kNarrow) {
            const u32x2_a8 ix = ix32[node];
            first = ix.x == kNoEdge ? -1 : (int64_t)ix.x;
            last = (int64_t)ix.y;
        } else {
            const int64_t a = index[2 * node], b = index[2 * node + 1];
            first = -1;
            last = -1;
            if (!(a == -1 || b == -1)) {
                if (a < 0 || b < a || b >= n_edges) st |= TRW_ST_BAD_CSR;
                else {
                    first = a;
                    last = b;
                }
            }
        }
    }
    __device__ __forceinline__ int64_t tail(int64_t e, int64_t pad, uint32_t &st) const
    {
        if (kNarrow) return (int64_t)tail32[e];
        const int64_t t = edges[2 * e + 1];
        if (t != pad && (uint64_t)t >= (uint64_t)n_nodes) {
            st |= TRW_ST_NODE_RANGE;
            return pad;
        }
        return t;
    }
};

// The walker's own draw stream: draw k is the k-th draw it actually consumes.
// (The block's words are four named registers, not an array: indexed by d & 3 an array is turned into a lookup
// in scratch memory -- 40 bytes per lane and a scratch_load per draw in three instantiations of elist_walk_kernel.)
struct WalkerDraws {
    uint64_t gw;
    uint32_t seed, d;
    uint32_t b0, b1, b2, b3;
    __device__ __forceinline__ uint32_t next()
    {
        if ((d & 3u) == 0) {
            uint32_t blk[4];
            walker_block(gw, d >> 2, seed, blk);
            b0 = blk[0];
            b1 = blk[1];
            b2 = blk[2];
            b3 = blk[3];
        }
        const uint32_t k = d & 3u;
        ++d;
        const uint32_t lo = (k & 1u) ? b1 : b0, hi = (k & 1u) ? b3 : b2;
        return ((k & 2u) ? hi : lo) >> 1;
    }
};

// Uniform walk: the lanes of a wave step in lockstep; kNarrow: their columns reach the caller's rows through
// the wave's LDS ring, one aligned 64-byte line per row at a time (WaveRows, trw_kernels_narrow.hpp), so
// every lane of the wave stays to the end.
// Biased walk: ONE flat loop, one rejection trial per lane and iteration, every lane at its own step -- a
// wave then runs for the longest total of trials among its lanes instead of the sum over the steps of the
// longest rejection loop (34.8 -> 22.8 ms on R-MAT 1M/10M); rows go out four columns at a time (RowCombiner).
template <bool kBiased, bool kNarrow>
__global__ void __launch_bounds__(kBlock)
elist_walk_kernel(EdgeListView<kNarrow> g, const int64_t *__restrict__ targets, int64_t n_walkers,
                  int64_t walker_base, int32_t L, int64_t pad, int32_t restart, uint32_t seed, BiasedParams bp,
                  const uint32_t *__restrict__ table, uint32_t mask, int64_t *__restrict__ out,
                  int64_t *__restrict__ status)
{
    constexpr bool kRing = kNarrow && !kBiased;  // rows through the wave's LDS ring
    __shared__ uint32_t tiles[kRing ? (kBlock / 64) * 64 * kTileStride : 1];
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i == 0) {
        status[2] = TRW_RNG_WALKER;
        status[3] = kNarrow ? 1 : 0;
    }
    const bool exists = i < n_walkers;
    if (!kRing && !exists) return;
    WaveRows rows = {tiles + (kRing ? (threadIdx.x >> 6) * 64 * kTileStride : 0), (int)(threadIdx.x & 63)};
    RowCombiner row = {out + i * (int64_t)(L + 1), 0u, 0u, 0u, 0u};
    const int64_t wave_walker0 = (int64_t)blockIdx.x * kBlock + (__builtin_amdgcn_readfirstlane(threadIdx.x) & ~63u);
    uint32_t st = 0;
    const int64_t target = exists ? targets[i] : pad;
    const int64_t jump = restart ? target : pad;  // rw_cpu_edge_list.cpp:103-108
    // a start node that is neither a node nor the padding id is reported and its row filled with it
    const bool bad_start = exists && target != pad && (uint64_t)target >= (uint64_t)g.n_nodes;
    const bool active = exists && !bad_start;
    if (bad_start) {
        st |= TRW_ST_NODE_RANGE;
        if (kNarrow)
            for (int32_t c = 0; c <= L; ++c) out[i * (int64_t)(L + 1) + c] = target;
    }
    auto emit = [&](int32_t col, int64_t v) {  // kRing: called by all lanes of the wave together
        if (kRing) {
            rows.put(col, active ? (uint32_t)v : kTileSkip);
            if ((col & 7) == 7 || col == L) rows.flush(wave_walker0, 1, n_walkers, L + 1, col, out);
        } else if (kNarrow) {
            if (active) row.emit((uint32_t)v, col, L);
        } else {
            out[i * (int64_t)(L + 1) + col] = v;
        }
    };
    emit(0, target);
    WalkerDraws rng;
    rng.gw = (uint64_t)walker_base + (uint64_t)i;
    rng.seed = seed;
    rng.d = 0;
    rng.b0 = rng.b1 = rng.b2 = rng.b3 = 0u;

    // the (first, last) edge range of a node the way the reference's accessors would see it; a walker
    // standing on the padding id has a range only if the padding id is itself a row of the index
    auto range_of = [&](int64_t node, int64_t &first, int64_t &last) {
        first = -1;
        last = -1;
        if ((uint64_t)node < (uint64_t)g.n_nodes) g.range(node, first, last, st);
    };
    // sample_neighbor (:8-35) given the range of `node`
    auto sample = [&](int64_t node, int64_t first, int64_t last) -> int64_t {
        if (node == pad) return jump;
        if (first < 0) return pad;
        int64_t e = first;
        if (first != last) e += (int64_t)draw_mod(rng.next(), last + 1 - first);
        return g.tail(e, pad, st);
    };

    int64_t prev = target, p_first = -1, p_last = -1;
    if (!kBiased) {
        for (int32_t s = 1; s <= L; ++s) {
            if (active) {
                range_of(prev, p_first, p_last);
                prev = sample(prev, p_first, p_last);
            }
            emit(s, prev);
        }
    } else {
        int64_t t = prev, t_first = -1, t_last = -1;
        if (active) {
            range_of(t, t_first, t_last);
            prev = sample(t, t_first, t_last);  // step 1 (:188)
        }
        emit(1, prev);
        if (active) range_of(prev, p_first, p_last);
        int32_t s = active ? 2 : L + 1;  // a bad start's row is already written
        if (!kNarrow && bad_start)
            for (int32_t c = 2; c <= L; ++c) emit(c, target);
        uint32_t rejected = 0;
        while (s <= L) {  // one trial per iteration
            const int64_t x = sample(prev, p_first, p_last);   // 0 or 1 draw
            const uint32_t ru = rng.next();                    // always 1 draw (:198)
            bool accept = false;
            int64_t selected = x;
            if (x == t && ru < bp.t0) {                        // :204-209
                accept = true;
            } else if (x == pad) {                             // :212-217
                accept = ru < bp.t0;
                selected = jump;
            } else if (ru < bp.lo) {                           // below both thresholds: class is irrelevant
                accept = true;
            } else if (ru < bp.hi) {                           // is_neighbor decides (:220-231)
                bool nb = false;
                if (t_first >= 0 && t_last > t_first) {
                    if (table) {
                        nb = elist_table_contains(table, mask, (uint32_t)t, (uint32_t)x);
                    } else {
                        for (int64_t e = t_first; e < t_last; ++e)  // [first, last): never the last edge
                            if (g.tail(e, pad, st) == x) {
                                nb = true;
                                break;
                            }
                    }
                }
                accept = ru < (nb ? bp.t1 : bp.t2);
            }
            if (!accept && ++rejected >= kTrialCap) {
                st |= (uint32_t)kStTrialCap;
                accept = true;
            }
            if (accept) {
                emit(s, selected);
                t = prev;
                t_first = p_first;
                t_last = p_last;
                prev = selected;
                rejected = 0;
                if (++s <= L) range_of(prev, p_first, p_last);
            }
        }
    }
    if (st) atomicOr((unsigned long long *)&status[0], (unsigned long long)st);
}

// The reference's single serial glibc stream, replayed in walker order by one lane (int64 in place).
// One lane cannot afford the reference's linear is_neighbor scan on hub ranges (100 walkers x 80 steps on
// R-MAT 1M/10M: 10.8 s): the scan is only run for draws that it can decide, and through the edge table when
// the call built one (same answers, same draws).
__global__ void __launch_bounds__(64)
elist_glibc_seq_kernel(EdgeListView<false> g, const int64_t *__restrict__ targets, int64_t n_walkers, int32_t L,
                       int64_t pad, int32_t restart, int32_t biased, SeedWindow sw, BiasedParams bp,
                       const uint32_t *__restrict__ table, uint32_t mask, int64_t *__restrict__ out,
                       int64_t *__restrict__ status)
{
    __shared__ uint32_t w[kGlDeg];
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    status[2] = TRW_RNG_GLIBC;
    GlibcSerial glibc;
    glibc.init(w, sw);
    auto next = [&]() -> uint32_t { return glibc.next(); };
    uint32_t st = 0;
    for (int64_t i = 0; i < n_walkers && !st; ++i) {
        int64_t *row = out + i * (int64_t)(L + 1);
        const int64_t target = targets[i];
        const int64_t jump = restart ? target : pad;
        row[0] = target;
        if (target != pad && (uint64_t)target >= (uint64_t)g.n_nodes) {
            st |= TRW_ST_NODE_RANGE;  // the chain's draw positions are undefined past a fault
            for (int32_t c = 1; c <= L; ++c) row[c] = target;
            break;
        }
        auto range_of = [&](int64_t node, int64_t &first, int64_t &last) {
            first = -1;
            last = -1;
            if ((uint64_t)node < (uint64_t)g.n_nodes) g.range(node, first, last, st);
        };
        auto sample = [&](int64_t node, int64_t first, int64_t last) -> int64_t {
            if (node == pad) return jump;
            if (first < 0) return pad;
            int64_t e = first;
            if (first != last) e += (int64_t)draw_mod(next(), last + 1 - first);
            return g.tail(e, pad, st);
        };
        int64_t prev = target, p_first, p_last;
        if (!biased) {
            for (int32_t s = 1; s <= L; ++s) {
                range_of(prev, p_first, p_last);
                prev = sample(prev, p_first, p_last);
                row[s] = prev;
            }
        } else {
            int64_t t = prev, t_first, t_last;
            range_of(t, t_first, t_last);
            prev = sample(t, t_first, t_last);
            row[1] = prev;
            for (int32_t s = 2; s <= L && !st; ++s) {
                range_of(prev, p_first, p_last);
                int64_t selected;
                uint32_t rejected = 0;
                while (true) {
                    const int64_t x = sample(prev, p_first, p_last);
                    const uint32_t ru = next();
                    bool accept = false;
                    selected = x;
                    if (x == t && ru < bp.t0) accept = true;
                    else if (x == pad) {
                        accept = ru < bp.t0;
                        selected = jump;
                    } else if (ru < bp.lo) {  // below both thresholds: class is irrelevant
                        accept = true;
                    } else if (ru < bp.hi) {  // is_neighbor decides
                        bool nb = false;
                        if (t_first >= 0 && t_last > t_first) {
                            if (table) {
                                nb = elist_table_contains(table, mask, (uint32_t)t, (uint32_t)x);
                            } else {
                                for (int64_t e = t_first; e < t_last; ++e)
                                    if (g.tail(e, pad, st) == x) {
                                        nb = true;
                                        break;
                                    }
                            }
                        }
                        accept = ru < (nb ? bp.t1 : bp.t2);
                    }
                    if (accept || st) break;
                    if (++rejected >= kTrialCap) {
                        st |= (uint32_t)kStTrialCap;
                        break;
                    }
                }
                row[s] = selected;
                t = prev;
                t_first = p_first;
                t_last = p_last;
                prev = selected;
            }
        }
    }
    if (st) atomicOr((unsigned long long *)&status[0], (unsigned long long)st);
}

// ===========================================================================
// Biased edge-list walk over prepared entries: walker pool (cf. biased_pool_kernel, trw_kernels_narrow.hpp)
// ===========================================================================
// One 64-byte entry per edge e = (u -> v), everything the step after u -> v needs (t = u, prev = v):
//     word 0..3    v, first(v), cnt(v) (edges of v; 0: none), flags
//     word 4..7    nb mask: bit (j & 127) = "the tail at offset j of v's range is in u's edges [first, last)"
//                  -- the reference's is_neighbor(x, t), rw_cpu_edge_list.cpp:37-62, last edge excluded
//     word 8       where t = u sits in v's range -- tails keep their input order, so that is a set of offsets:
//                      cnt(v) <= 32            bit j = "offset j holds u", exact
//                      kElTmOffset             the ONE offset that holds u (0xFFFFFFFF: none), exact
//                      kElTmLook               bit (j & 31) = "some offset == j (mod 32) may hold u": load it and see
//     word 9       the only tail of v when cnt(v) == 1 (that step has a fixed candidate and draws nothing for it)
// The nb mask is exact for cnt(v) <= 128 (the range is scanned); for longer ranges it is a 32-bit filter replicated
// four times, built from the table's position masks by walking u's edges (kElFilter: clear = certainly not,
// set = load the candidate and see); a range that holds the padding id as a tail is all "see" (kElHasPad).
// (One word for t instead of a second 128-bit mask: the walker pool keeps a copy per walker in LDS, and three
// words less per walker are a third workgroup per CU.)
constexpr uint32_t kElFilter = 1u, kElHasPad = 2u, kElTmOffset = 4u, kElTmLook = 8u;
#ifndef TRW_EL_FILTER_WALK
#define TRW_EL_FILTER_WALK 256
#endif
constexpr uint32_t kElFilterWalk = TRW_EL_FILTER_WALK;  // longest u range walked for a filter mask; all-ones beyond

// Per node, 16 bytes: {first, last, flags, only tail}: what a walker needs when it reaches a node WITHOUT an entry
// (from the padding id, by restart): its edge range, whether that range holds the padding id as a tail
// (kElHasPad), and the tail of its single edge when it has exactly one.
__global__ void __launch_bounds__(kBlock)
elist_node_kernel(int64_t n_nodes, uint32_t pad, const uint32_t *__restrict__ tail32,
                  const u32x2_a8 *__restrict__ ix32, const uint32_t *__restrict__ table, uint32_t mask,
                  u32x4_a16 *__restrict__ nd)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < n_nodes; v += stride) {
        const u32x2_a8 ix = ix32[v];
        uint32_t flags = 0, only = 0;
        if (ix.x != kNoEdge) {
            const uint32_t cnt = ix.y - ix.x + 1u;
            if (cnt == 1u) only = tail32[ix.x];
            if (cnt <= 128u) {
                for (uint32_t j = 0; j < cnt; ++j)
                    if (tail32[ix.x + j] == pad) flags |= kElHasPad;
            } else {
                uint32_t at0, pm;
                bool in_range;
                if (elist_table_find(table, mask, (uint32_t)v, pad, at0, pm, in_range)) flags |= kElHasPad;
            }
        }
        u32x4_a16 o = {ix.x, ix.y, flags, only};
        nd[v] = o;
    }
}

__global__ void __launch_bounds__(kBlock)
elist_ext_kernel(const int64_t *__restrict__ edges, int64_t n_edges, int64_t n_nodes, uint32_t pad,
                 const uint32_t *__restrict__ tail32, const u32x2_a8 *__restrict__ ix32,
                 const uint32_t *__restrict__ table, uint32_t mask, u32x4_a16 *__restrict__ ent)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n_edges; e += stride) {
        const int64_t head = edges[2 * e];
        const uint32_t v = tail32[e];
        uint32_t first = kNoEdge, cnt = 0, flags = 0, only = 0, tm = 0;
        Mask128 nb;
        nb.clear();
        if (v != pad && v < (uint32_t)n_nodes && (uint64_t)head < (uint64_t)n_nodes) {
            const uint32_t u = (uint32_t)head;
            const u32x2_a8 ix = ix32[v];
            if (ix.x != kNoEdge) {
                first = ix.x;
                cnt = ix.y - ix.x + 1u;
            }
            if (cnt == 1u) only = tail32[first];
            uint32_t at0, pm;
            bool in_range;
            if (cnt >= 1u && cnt <= 128u) {
                uint32_t occ = 0, at = 0xFFFFFFFFu, m32 = 0;
                for (uint32_t j = 0; j < cnt; ++j) {
                    const uint32_t x = tail32[first + j];
                    if (x == pad) flags |= kElHasPad;
                    if (x == u) {
                        if (occ++ == 0) at = j;
                        m32 |= 1u << (j & 31u);
                    }
                    if (elist_table_contains(table, mask, u, x)) nb.set(j);
                }
                if (cnt <= 32u) {
                    tm = m32;
                } else if (occ <= 1u) {
                    tm = at;
                    flags |= kElTmOffset;
                } else {
                    tm = m32;
                    flags |= kElTmLook;
                }
            } else if (cnt > 128u) {
                // a range too long to scan: the masks are FILTERS over offsets modulo 128, built from the table -- the
                // offset itself for a pair that occurs once, its offsets modulo 32 (in all four words) for a multi-edge
                flags |= kElFilter;
                if (elist_table_find(table, mask, v, pad, at0, pm, in_range)) flags |= kElHasPad;
                if (elist_table_find(table, mask, v, u, at0, pm, in_range)) {  // t is in v's range
                    if (pm == 0u) {
                        tm = at0;
                        flags |= kElTmOffset;  // ... once: that offset, exactly
                    } else {
                        tm = pm;
                        flags |= kElTmLook;
                    }
                } else {
                    tm = 0u;
                    flags |= kElTmLook;  // (a plain mask with no bit set: no offset holds t)
                }
                nb.fill();
                const u32x2_a8 ux = ix32[u];
                if (ux.x != kNoEdge && ux.y - ux.x <= kElFilterWalk) {
                    nb.clear();
                    for (uint32_t k = ux.x; k < ux.y; ++k) {  // u's edges [first, last): the last one never counts
                        if (elist_table_find(table, mask, v, tail32[k], at0, pm, in_range)) {
                            if (pm == 0u) nb.set(at0);
                            else
                                for (int w = 0; w < 4; ++w) nb.w[w] |= pm;
                        }
                    }
                }
            }
            if (flags & kElHasPad) {  // any candidate may be the padding id: every trial looks
                flags = (flags & ~kElTmOffset) | kElFilter | kElTmLook;
                nb.fill();
                tm = 0xFFFFFFFFu;
            }
        }
        u32x4_a16 a = {v, first, cnt, flags}, b = {nb.w[0], nb.w[1], nb.w[2], nb.w[3]}, c = {tm, only, 0u, 0u},
                  d = {0u, 0u, 0u, 0u};
        ent[4 * (size_t)e] = a;
        ent[4 * (size_t)e + 1] = b;
        ent[4 * (size_t)e + 2] = c;
        ent[4 * (size_t)e + 3] = d;
    }
}

#ifndef TRW_ELPOOL_SLOTS
#define TRW_ELPOOL_SLOTS 124
#endif
constexpr int kElSlots = TRW_ELPOOL_SLOTS;  // walkers resident per wave
constexpr int kElWin = 12;                  // draws a burst can see: the slot's current Philox block + the next two
enum : uint32_t { EK_RANGE = 0, EK_ENTRY, EK_TEST, EK_MEMBER, EK_FIXNB, EK_PEEK, EK_NOENTRY };
// 22 words per walker + 2 KB of draw window per wave: 13.3 KB per wave, 53.4 KB per workgroup, THREE workgroups per CU
// (26 words and a 3 KB window fitted two: 10.6 -> 9.1 ms on the R-MAT 1M/10M list)
enum : int {
    EF_T = 0, EF_PREV, EF_FIRST, EF_CNT, EF_Z, EF_NB0, EF_NB1, EF_NB2, EF_NB3,
    EF_TM,     // where t sits in PREV's range: word 8 of the entry, read as flag bits 2..3 say
    EF_FLAGS,  // bits 0..4 kEf*, bits 5..7 EK_* of the queued / in-flight gather, bits 8..28 rejections at this step,
               // bits 29..31 how far the probe chain of the current table lookup has gone
    EF_S, EF_WALKER, EF_C0, EF_C1, EF_C2, EF_C3,
    EF_D,      // draws the walker has consumed; C0..C3 is the Philox block of the LAST one, block (D - 1) / 4
    EF_X,      // fixed step: the node an accepted trial selects; table probes: the node asked about
    EF_AT,     // entry to gather (kNoEdge: the selected node is reached without an entry -- padding, restart)
    EF_AUX,    // fixed step: accept iff draw < AUX; candidate under test: its acceptance draw
    EF_JUMP,   // where a walk goes from the padding id: its start node (restart) or the padding id
    EF_COUNT
};
// slot flags: kEfFilter / kEfTmOffset / kEfTmLook are the entry's kElFilter / kElTmOffset / kElTmLook;
// kEfFixed: this step has a FIXED candidate; kEfTNoNb: t has no edge that is_neighbor would look at
constexpr uint32_t kEfFilter = kElFilter, kEfFixed = 2u, kEfTmOffset = kElTmOffset, kEfTmLook = kElTmLook, kEfTNoNb = 16u;
constexpr uint32_t kEfLow = 31u, kEfKindShift = 5, kEfKind = 7u << 5, kEfRejShift = 8, kEfRej = 0x1FFFFFu << 8,
                   kEfProbeShift = 29, kEfProbe = 7u << 29;
static_assert(kTrialCap <= (kEfRej >> kEfRejShift), "the rejection count must fit its field");
// what of an entry's flag word a slot keeps
constexpr uint32_t kEfFromEntry = kElFilter | kElTmOffset | kElTmLook;

// (a macro, not a lambda: called from two places a lambda was not inlined and its captures went to scratch memory)
#define TRW_EL_ISSUE_PASS() \
    do { \
                uint32_t n, slot; \
                issue_q.pop(lane, n, slot); \
                if ((uint32_t)lane < n) { \
                    const uint32_t flags = P[EF_FLAGS][slot]; \
                    const uint32_t kind = (flags & kEfKind) >> kEfKindShift; \
                    const uint32_t at = P[EF_AT][slot], prev = P[EF_PREV][slot], t = P[EF_T][slot], x = P[EF_X][slot]; \
                    const bool from_ent = kind == EK_ENTRY || kind == EK_TEST || kind == EK_PEEK; \
                    const bool from_tab = kind == EK_MEMBER || kind == EK_FIXNB; \
                    const uint32_t tslot = (edge_hash(t, x) + (flags >> kEfProbeShift)) & mask; \
                    const char *base = from_ent ? reinterpret_cast<const char *>(ent) \
                                                : (from_tab ? reinterpret_cast<const char *>(table) : reinterpret_cast<const char *>(nd)); \
                    const uint64_t off = from_ent ? (uint64_t)at << 6 : (uint64_t)(from_tab ? tslot : prev) << 4; \
                    const uint32_t *addr = reinterpret_cast<const uint32_t *>(base + off); \
                    if (kind != EK_NOENTRY) e0 = *reinterpret_cast<const u32x4_a4 *>(addr); \
                    if (kind == EK_ENTRY || kind == EK_TEST) { \
                        e1 = *reinterpret_cast<const u32x4_a4 *>(addr + 4); \
                        e2 = *reinterpret_cast<const u32x2_a4 *>(addr + 8); \
                    } \
                    hold = true; \
                    hs = slot; \
                    hkind = kind; \
                } \
    } while (0)

template <bool kCount>
__global__ void __launch_bounds__(kBlock)
elist_pool_kernel(const u32x4_a16 *__restrict__ nd, const u32x4_a16 *__restrict__ ent,
                  const uint32_t *__restrict__ table, uint32_t mask, const int64_t *__restrict__ targets,
                  int64_t n_nodes, int64_t n_walkers, int64_t walker_base, int32_t L, uint32_t pad, int32_t restart,
                  uint32_t seed, BiasedParams bp, int64_t *__restrict__ out, int64_t *__restrict__ status)
{
    static_assert(kElSlots >= 64 && kElSlots <= kQueueCap, "pool size must be 64..128 slots");
    __shared__ uint32_t pool_mem[kPoolWaves][EF_COUNT][kElSlots];
#ifndef TRW_EL_REFILL_MIN
#define TRW_EL_REFILL_MIN 8  // free slots that trigger a claim of new walkers (32: 6.86, 16: 6.70, 8: 6.65 ms)
#endif
#ifndef TRW_EL_ISSUE_MIN
#define TRW_EL_ISSUE_MIN TRW_POOL_ISSUE_MIN
#endif
#ifndef TRW_EL_TRIAL_MIN
#define TRW_EL_TRIAL_MIN TRW_POOL_TRIAL_MIN
#endif
#ifndef TRW_EL_LATE_ISSUE
#define TRW_EL_LATE_ISSUE 2  // 1: a second issue pass after the burst when no gather is held; 2: only once the batch has run
                             // out (pools that drain: 6.83 -> 6.72 ms; always: 6.98)
#endif
#ifndef TRW_EL_SCALE_NUM
#define TRW_EL_SCALE_NUM 0  // > 0: a pass waits for this many eighths of the RESIDENT walkers (a pool that drains)
#endif
    constexpr int kWinOff = 4;  // window positions below this are read from the slot's C0..C3, not from win_mem
    __shared__ uint32_t win_mem[kPoolWaves][kElWin - kWinOff][64];
    __shared__ uint8_t queue_mem[kPoolWaves][3][kQueueCap];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        status[2] = TRW_RNG_WALKER;
        status[3] = 5;
    }
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t(*P)[kElSlots] = pool_mem[wv];
    uint32_t(*W)[64] = win_mem[wv];
    PoolQueue trial_q = {queue_mem[wv][0], 0u, 0u}, issue_q = {queue_mem[wv][1], 0u, 0u}, free_q = {queue_mem[wv][2], 0u, 0u};
    for (int i = lane; i < kElSlots; i += 64) free_q.q[i] = (uint8_t)i;
    free_q.count = kElSlots;
    pool_fence();
    const bool band_takes_near = bp.t1 > bp.t2;
    const uint32_t v_clear = band_takes_near ? 0u : 1u, v_set = band_takes_near ? 1u : 0u;
    const int32_t cols = L + 1;
    // the Philox key schedule in vector registers (see philox4x32_10_keyed)
    PhiloxKeysV key_r;
    philox_keys_v(seed, key_r);

    uint32_t st = 0;
    uint32_t n_trials = 0, n_fetch = 0, n_tests = 0, n_probes = 0;
    bool hold = false;
    uint32_t hs = 0, hkind = 0;
    u32x4_a4 e0 = {0u, 0u, 0u, 0u}, e1 = {0u, 0u, 0u, 0u};
    // (only the words of an entry that are used are loaded: a load's unused destination registers get other values
    // assigned to them, and the first write to one waits for the load -- a memory latency at the head of the burst)
    u32x2_a4 e2 = {0u, 0u};
    bool more_walkers = true;

    // counting variant only: shader-clock cycles per phase, burst passes / lanes / trial repetitions (status[20..27])
    uint64_t w_cycles[4] = {0, 0, 0, 0}, w_bursts = 0, w_burst_lanes = 0, w_reps = 0, w_rep_lanes = 0;
    uint64_t w_wait = 0;  // cycles the adopt pass waited for its gathers (status[28])
    uint64_t w_kind[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // (kept in registers: an atomic per pass made the counting run 17x slower)
    uint64_t t_mark = kCount ? clock64() : 0;
    auto lap = [&](int phase) {
        if (kCount) {
            const uint64_t now = clock64();
            w_cycles[phase] += now - t_mark;
            t_mark = now;
        }
    };

    auto row_of = [&](uint32_t slot) { return out + (int64_t)P[EF_WALKER][slot] * cols; };
    // new low flag bits and gather kind for a slot; the rejection count stays, the probe count restarts
    auto set_kind = [&](uint32_t slot, uint32_t low, uint32_t kind) {
        P[EF_FLAGS][slot] = (P[EF_FLAGS][slot] & kEfRej) | (low & kEfLow) | (kind << kEfKindShift);
    };
    // one more rejection at this step; true when that was the last one allowed
    auto reject = [&](uint32_t slot) {
        const uint32_t f = P[EF_FLAGS][slot] + (1u << kEfRejShift);
        P[EF_FLAGS][slot] = f;
        return ((f & kEfRej) >> kEfRejShift) >= kTrialCap;
    };
    // accept threshold of a FIXED candidate x for a walker whose t is `t` (rw_cpu_edge_list.cpp:204-231 with x constant):
    //   x is the padding id -> t0;  else max(x == t ? t0 : 0, is_neighbor ? t1 : t2)
    auto fixed_threshold = [&](uint32_t x, uint32_t t, bool nb) {
        if (x == pad) return bp.t0;
        const uint32_t cls = nb ? bp.t1 : bp.t2;
        return (x == t && bp.t0 > cls) ? bp.t0 : cls;
    };
    // The walker of `slot` now stands on PREV without an entry describing it (it came through the padding id or a
    // restart): decide what its next step looks like.  Returns 0 -> trial queue, 1 -> issue queue.
    auto arrive_bare = [&](uint32_t slot) -> int {
        const uint32_t prev = P[EF_PREV][slot], jump = P[EF_JUMP][slot];
        const uint32_t tn = P[EF_FLAGS][slot] & kEfTNoNb;
        if (prev == pad) {  // the candidate is the jump node, no draw for it (rw_cpu_edge_list.cpp:14-16)
            P[EF_X][slot] = jump;
            P[EF_AT][slot] = kNoEdge;
            if (jump == pad) {
                P[EF_AUX][slot] = bp.t0;
                set_kind(slot, kEfFixed | tn, 0u);
                return 0;
            }
            if (tn) {  // t is a dead end (the usual way onto the padding id): the jump node is no neighbour of it
                P[EF_AUX][slot] = fixed_threshold(jump, P[EF_T][slot], false);
                set_kind(slot, kEfFixed | tn, 0u);
                return 0;
            }
            set_kind(slot, kEfFixed, EK_FIXNB);  // is the jump node a neighbour of t?
            return 1;
        }
        set_kind(slot, tn, EK_RANGE);  // a real node: its edge range first
        return 1;
    };
    // ... with its range and masks in the slot (FIRST, CNT, masks, their flag bits): fixed or multi-candidate step
    auto classify = [&](uint32_t slot, uint32_t only_tail, bool have_only) -> int {
        const uint32_t cnt = P[EF_CNT][slot], t = P[EF_T][slot], jump = P[EF_JUMP][slot];
        const uint32_t fbits = P[EF_FLAGS][slot] & (kEfFilter | kEfTmOffset | kEfTmLook | kEfTNoNb);
        if (cnt == 0u) {  // dead end: the candidate is the padding id
            P[EF_X][slot] = (pad == t) ? pad : jump;
            P[EF_AT][slot] = kNoEdge;
            P[EF_AUX][slot] = bp.t0;
            set_kind(slot, fbits | kEfFixed, 0u);
            return 0;
        }
        if (cnt == 1u) {  // one edge: the candidate is its tail, no draw for it (cpu_utils.cpp:4-5)
            if (!have_only) {
                P[EF_AT][slot] = P[EF_FIRST][slot];
                set_kind(slot, fbits | kEfFixed, EK_PEEK);
                return 1;
            }
            if (only_tail == pad) {
                P[EF_X][slot] = (pad == t) ? pad : jump;
                P[EF_AT][slot] = kNoEdge;
                P[EF_AUX][slot] = bp.t0;
                set_kind(slot, fbits | kEfFixed, 0u);
                return 0;
            }
            P[EF_X][slot] = only_tail;
            P[EF_AT][slot] = P[EF_FIRST][slot];
            if (fbits & kEfFilter) {  // masks say nothing about this single edge (bare arrival): ask the table
                if (fbits & kEfTNoNb) {
                    P[EF_AUX][slot] = fixed_threshold(only_tail, t, false);
                    set_kind(slot, fbits | kEfFixed, 0u);
                    return 0;
                }
                set_kind(slot, fbits | kEfFixed, EK_FIXNB);
                return 1;
            }
            // (a one-edge range is described exactly: TM is a plain bit mask for ranges up to 32)
            const bool is_t = P[EF_TM][slot] & 1u, nb = P[EF_NB0][slot] & 1u;
            const uint32_t cls = nb ? bp.t1 : bp.t2;
            P[EF_AUX][slot] = (is_t && bp.t0 > cls) ? bp.t0 : cls;
            set_kind(slot, fbits | kEfFixed, 0u);
            return 0;
        }
        FastMod fm;
        fm.init_exact(cnt);
        P[EF_Z][slot] = fm.z;
        set_kind(slot, fbits, 0u);
        return 0;
    };
    // The walker of `slot` takes `node` as column S: onto the entry in e0..e2 (`with_entry`) or without one (padding id,
    // jump, restart).  Returns -1 freed, 0 trial queue, 1 issue queue, 2 "classify(slot, only_tail) is still to be done"
    // -- the callers run ONE copy of classify for every way a walker can arrive (the lambdas are inlined where they are
    // called: the adopt pass used to execute three copies of this and two of classify in every pass).
    auto step_to = [&](uint32_t slot, uint32_t node, bool with_entry, bool &to_free, uint32_t &only_tail) -> int {
        int32_t s = (int32_t)P[EF_S][slot];
        row_of(slot)[s] = (int64_t)node;
        ++s;
        P[EF_S][slot] = (uint32_t)s;
        const uint32_t left = P[EF_PREV][slot];
        // the node being left becomes t: does is_neighbor(., t) have anything to look at?  (edges [first, last))
        const bool tnonb = left == pad ? pad >= (uint32_t)n_nodes : P[EF_CNT][slot] <= 1u;
        P[EF_T][slot] = left;
        P[EF_PREV][slot] = node;
        if (s > L) {
            to_free = true;
            return -1;
        }
        if (!with_entry || node == pad) {
            P[EF_FLAGS][slot] = tnonb ? kEfTNoNb : 0u;  // (no rejections yet at the new step)
            return arrive_bare(slot);
        }
        P[EF_FIRST][slot] = e0.y;
        P[EF_CNT][slot] = e0.z;
        P[EF_NB0][slot] = e1.x;
        P[EF_NB1][slot] = e1.y;
        P[EF_NB2][slot] = e1.z;
        P[EF_NB3][slot] = e1.w;
        P[EF_TM][slot] = e2.x;
        P[EF_FLAGS][slot] = (e0.w & kEfFromEntry) | (tnonb ? kEfTNoNb : 0u);
        only_tail = e2.y;
        return 2;
    };

    while (true) {
        // ================= adopt =================
#if TRW_POOL_PRIO
        __builtin_amdgcn_s_setprio(TRW_POOL_PRIO == 1 ? 3 : 0);  // (cf. biased_pool_kernel)
#endif
        {
            bool to_trial = false, to_issue = false, to_free = false;
            // The gathers are waited for HERE, once: left to the compiler, every branch below that reads e0..e2 got its
            // own `s_waitcnt vmcnt(0)`, behind the column stores of the branches before it -- a store's round trip
            // exposed two or three times per pass.  (An s_waitcnt the compiler can see: it knows e0..e2 are there.)
            const uint64_t t_wait = kCount ? clock64() : 0;
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
            if (kCount) {  // how many gathers of each kind were consumed (status[12 + kind]) and in how many passes ([19])
                w_wait += clock64() - t_wait;
#pragma unroll
                for (uint32_t k = 0; k < 7u; ++k) w_kind[k] += (uint64_t)__popcll(__ballot(hold && hkind == k));
                ++w_kind[7];
            }
            if (hold) {
                const uint32_t slot = hs;
                int where = -1;  // 0 trial queue, 1 issue queue, -1 neither (freed), 2 classify pending
                // what the kind's branch decided: 0 nothing more, 1 step onto the entry in e0..e2, 2 step to `bare` without one
                int act = 0;
                uint32_t bare = 0u, only_tail = 0u;
                const uint32_t flags = P[EF_FLAGS][slot];
                if (hkind == EK_RANGE) {  // e0 = {first, last, flags, only tail} of PREV
                    const uint32_t first = e0.x, cnt = first == kNoEdge ? 0u : e0.y - e0.x + 1u;
                    if (P[EF_S][slot] == 1u) {  // the first step samples a neighbour of the start node outright (:188)
                        P[EF_CNT][slot] = cnt;  // (the start node becomes t: step_* looks at this)
                        if (cnt == 0u) {
                            act = 2;
                            bare = pad;
                        } else {
                            uint32_t idx = 0u;
                            if (cnt > 1u) {  // draw 0 (a single edge is taken without a draw, cpu_utils.cpp:4-5)
                                P[EF_D][slot] = 1u;
                                idx = (P[EF_C0][slot] >> 1) % cnt;
                            }
                            P[EF_AT][slot] = first + idx;
                            set_kind(slot, 0u, EK_ENTRY);
                            if (kCount) ++n_fetch;
                            where = 1;
                        }
                    } else {
                        // A node reached without an entry: it was jumped to from the padding id, so t IS the padding id.
                        // When the padding id is no node of the graph nothing is its neighbour and only a tail equal
                        // to it "is t": a range without such tails has all-zero masks, exactly.  Otherwise every
                        // candidate is looked at.
                        const bool plain = P[EF_T][slot] == pad && pad >= (uint32_t)n_nodes && !(e0.z & kElHasPad);
                        const uint32_t fill = plain ? 0u : 0xFFFFFFFFu;
                        P[EF_FIRST][slot] = first;
                        P[EF_CNT][slot] = cnt;
                        P[EF_NB0][slot] = P[EF_NB1][slot] = P[EF_NB2][slot] = P[EF_NB3][slot] = fill;
                        P[EF_TM][slot] = fill;
                        P[EF_FLAGS][slot] = (flags & (kEfRej | kEfTNoNb)) | (plain ? 0u : (kEfFilter | kEfTmLook));
                        only_tail = e0.w;
                        where = 2;
                    }
                } else if (hkind == EK_ENTRY) {
                    act = 1;
                } else if (hkind == EK_NOENTRY) {
                    act = 2;
                    bare = P[EF_X][slot];
                } else if (hkind == EK_PEEK) {  // e0.x = the only tail of PREV
                    const uint32_t x = e0.x, t = P[EF_T][slot];
                    if (x == pad) {
                        P[EF_X][slot] = (pad == t) ? pad : P[EF_JUMP][slot];
                        P[EF_AT][slot] = kNoEdge;
                        P[EF_AUX][slot] = bp.t0;
                        set_kind(slot, flags, 0u);
                        where = 0;
                    } else {
                        P[EF_X][slot] = x;
                        set_kind(slot, flags, EK_FIXNB);
                        where = 1;
                    }
                } else if (hkind == EK_TEST) {  // e0.. = a candidate the masks could not decide: the reference's own tests
                    if (kCount) ++n_tests;
                    const uint32_t x = e0.x, t = P[EF_T][slot], ru = P[EF_AUX][slot];
                    const uint32_t idx = P[EF_AT][slot] - P[EF_FIRST][slot];
                    const uint32_t k = (idx >> 5) & 3u;
                    const uint32_t word = P[EF_NB0 + (int)k][slot];
                    const bool nb_bit = (word >> (idx & 31u)) & 1u;
                    int verdict;  // 0 reject, 1 accept the entry, 2 accept the jump node, 3 ask the table
                    if (x == t && ru < bp.t0) verdict = 1;
                    else if (x == pad) verdict = ru < bp.t0 ? 2 : 0;
                    else if (ru < bp.lo) verdict = 1;
                    else if (ru >= bp.hi) verdict = 0;
                    else if (!nb_bit || (flags & kEfTNoNb)) verdict = band_takes_near ? 0 : 1;
                    else verdict = 3;
                    if (verdict == 1) {
                        act = 1;
                    } else if (verdict == 2) {
                        act = 2;
                        bare = P[EF_JUMP][slot];
                    } else if (verdict == 3) {
                        P[EF_X][slot] = x;
                        set_kind(slot, flags, EK_MEMBER);
                        where = 1;
                    } else if (reject(slot)) {
                        st |= (uint32_t)kStTrialCap;
                        act = 1;
                    } else {
                        set_kind(slot, flags, 0u);
                        where = 0;
                    }
                } else {  // EK_MEMBER / EK_FIXNB: e0 = one table slot {u, v, ~posmask, ~flags}
                    if (kCount) ++n_probes;
                    const uint32_t t = P[EF_T][slot], x = P[EF_X][slot];
                    bool empty = e0.x == 0xFFFFFFFFu;
                    bool match = e0.x == t && e0.y == x;
                    if (!empty && !match && (flags & kEfProbe) == kEfProbe) {
                        // the chain is longer than the probe field counts: this lane walks the rest of it here
                        uint32_t sl = (edge_hash(t, x) + 8u) & mask;
                        while (true) {
                            const u32x4_a16 g = *reinterpret_cast<const u32x4_a16 *>(table + 4 * (size_t)sl);
                            if (kCount) ++n_probes;
                            empty = g.x == 0xFFFFFFFFu;
                            match = g.x == t && g.y == x;
                            if (empty || match) {
                                e0.w = g.w;
                                break;
                            }
                            sl = (sl + 1u) & mask;
                        }
                    }
                    if (!empty && !match) {
                        P[EF_FLAGS][slot] = flags + (1u << kEfProbeShift);  // next slot of the probe chain
                        where = 1;
                    } else {
                        const bool nb = match && (e0.w & 2u) == 0u;  // (the slot's "some occurrence is not t's last edge" bit)
                        if (hkind == EK_FIXNB) {
                            P[EF_AUX][slot] = fixed_threshold(x, t, nb);
                            set_kind(slot, flags, 0u);
                            where = 0;
                        } else if (nb == band_takes_near) {  // the draw sat inside the band: accepted, gather the entry again
                            set_kind(slot, flags, EK_ENTRY);
                            if (kCount) ++n_fetch;
                            where = 1;
                        } else if (reject(slot)) {
                            st |= (uint32_t)kStTrialCap;
                            set_kind(slot, flags, EK_ENTRY);
                            where = 1;
                        } else {
                            set_kind(slot, flags, 0u);
                            where = 0;
                        }
                    }
                }
                if (act) where = step_to(slot, act == 1 ? e0.x : bare, act == 1, to_free, only_tail);
                if (where == 2) where = classify(slot, only_tail, true);
                to_trial = where == 0;
                to_issue = where == 1;
            }
            trial_q.push(to_trial, hs);
            issue_q.push(to_issue, hs);
            free_q.push(to_free, hs);
            hold = false;
            pool_fence();
        }
        lap(0);

        // ================= refill =================
        if (more_walkers && (free_q.count >= (uint32_t)TRW_EL_REFILL_MIN || trial_q.count + issue_q.count < 64u) && free_q.count > 0u) {
            uint32_t n, slot;
            const uint32_t avail = free_q.count < 64u ? free_q.count : 64u;
            long long base = 0;
            if (lane == 0) base = (long long)atomicAdd((unsigned long long *)&status[kStNextWalker], (unsigned long long)avail);
            base = __shfl(base, 0);
            int64_t left = n_walkers - (int64_t)base;
            left = left < 0 ? 0 : left;
            const uint32_t take = left < (int64_t)avail ? (uint32_t)left : avail;
            if (take < avail) more_walkers = false;
            const uint32_t keep_count = free_q.count;
            free_q.pop(lane, n, slot);
            if (take < n) {
                free_q.head = (free_q.head + (uint32_t)kQueueCap - (n - take)) & (uint32_t)(kQueueCap - 1);
                free_q.count = keep_count - take;
            }
            bool to_issue = false, back = false;
            if ((uint32_t)lane < take) {
                const int64_t walker = (int64_t)base + lane;
                const int64_t tg = targets[walker];
                if (tg != (int64_t)pad && (uint64_t)tg >= (uint64_t)n_nodes) {
                    st |= TRW_ST_NODE_RANGE;  // reported; the row is filled with the bad id
                    int64_t *rowp = out + walker * cols;
                    for (int32_t c = 0; c <= L; ++c) rowp[c] = tg;
                    back = true;
                } else {
                    P[EF_WALKER][slot] = (uint32_t)walker;
                    out[walker * cols] = tg;  // column 0
                    P[EF_JUMP][slot] = restart ? (uint32_t)tg : pad;  // rw_cpu_edge_list.cpp:103-108
                    P[EF_PREV][slot] = (uint32_t)tg;
                    P[EF_T][slot] = (uint32_t)tg;
                    P[EF_S][slot] = 1u;
                    P[EF_D][slot] = 0u;
                    uint32_t blk[4];
                    walker_block((uint64_t)walker_base + (uint64_t)walker, 0u, seed, blk);
                    P[EF_C0][slot] = blk[0];
                    P[EF_C1][slot] = blk[1];
                    P[EF_C2][slot] = blk[2];
                    P[EF_C3][slot] = blk[3];
                    if ((uint32_t)tg == pad) {  // step 1 from the padding id goes to the jump node (= the padding id)
                        P[EF_X][slot] = pad;
                        P[EF_FLAGS][slot] = EK_NOENTRY << kEfKindShift;
                    } else {
                        P[EF_FLAGS][slot] = EK_RANGE << kEfKindShift;
                    }
                    to_issue = true;
                }
            }
            issue_q.push(to_issue, slot);
            free_q.push(back, slot);
            pool_fence();
        }
        lap(1);
        if (!more_walkers && free_q.count == (uint32_t)kElSlots) break;

        // ================= issue =================
        uint32_t issue_min = (uint32_t)TRW_EL_ISSUE_MIN, trial_min = (uint32_t)TRW_EL_TRIAL_MIN;
        if (TRW_EL_SCALE_NUM > 0) {
            const uint32_t resident = (uint32_t)kElSlots - free_q.count;
            uint32_t cap = (resident * (uint32_t)TRW_EL_SCALE_NUM) >> 3;
            cap = cap < 4u ? 4u : cap;
            issue_min = issue_min < cap ? issue_min : cap;
            trial_min = trial_min < cap ? trial_min : cap;
        }
        // ONE address and ONE load instruction for all kinds of gather: loads of the three kinds issued from three
        // branches wrote the same registers under different lane masks, and the compiler put a full
        // `s_waitcnt vmcnt(0)` between them -- two memory latencies exposed in every iteration.
        if (issue_q.count >= issue_min || (issue_q.count > 0u && trial_q.count < trial_min)) TRW_EL_ISSUE_PASS();

        lap(2);
        // ================= burst =================
#if TRW_POOL_PRIO
        __builtin_amdgcn_s_setprio(TRW_POOL_PRIO == 1 ? 0 : 3);
#endif
        if (trial_q.count >= trial_min || (trial_q.count > 0u && issue_q.count < issue_min)) {
            uint32_t n, slot;
            trial_q.pop(lane, n, slot);
            if (kCount) {
                ++w_bursts;
                w_burst_lanes += n;
            }
            const bool busy = (uint32_t)lane < n;
            // Straight-line form (round 3, cf. the burst of biased_pool_kernel): the slot's state, the eight draws the four
            // trials of a visit can consume and the four mask words they index are each read from LDS in ONE batch, the four
            // trials are evaluated side by side, and the first one that ends the visit is picked afterwards -- three LDS round
            // trips per visit where the loop made three per TRIAL, and no exec-mask regions.
            const uint32_t first = P[EF_FIRST][slot], cnt = P[EF_CNT][slot];
            FastMod fm;
            fm.d = busy && cnt ? cnt : 1u;
            fm.z = P[EF_Z][slot];
            const uint32_t tm = P[EF_TM][slot];
            const uint32_t flags = P[EF_FLAGS][slot];
            const bool fixed = (flags & kEfFixed) != 0u, filter = (flags & kEfFilter) != 0u;
            const bool tm_offset = (flags & kEfTmOffset) != 0u, tm_look = (flags & kEfTmLook) != 0u;
            const bool tnonb = (flags & kEfTNoNb) != 0u;  // nothing is a neighbour of t: every nb bit reads as clear
            const uint32_t aux = P[EF_AUX][slot];
            uint32_t rejected = (flags & kEfRej) >> kEfRejShift;
            const uint32_t d0 = busy ? P[EF_D][slot] : 0u;
            const uint64_t gw = (uint64_t)walker_base + (uint64_t)P[EF_WALKER][slot];
            // the window of draws: positions 0..3 are the slot's block (C0..C3, the block of the last draw consumed),
            // 4..11 the next two blocks, worked out here
            const uint32_t cb = d0 ? (d0 - 1u) >> 2 : 0u;
            {
                uint32_t nbk[4];
                philox4x32_10_keyed((uint32_t)gw, (uint32_t)(gw >> 32), cb + 1u, 0u, key_r, nbk);
                W[0][lane] = nbk[0];
                W[1][lane] = nbk[1];
                W[2][lane] = nbk[2];
                W[3][lane] = nbk[3];
                philox4x32_10_keyed((uint32_t)gw, (uint32_t)(gw >> 32), cb + 2u, 0u, key_r, nbk);
                W[4][lane] = nbk[0];
                W[5][lane] = nbk[1];
                W[6][lane] = nbk[2];
                W[7][lane] = nbk[3];
            }
            const uint32_t pos0 = d0 - 4u * cb;  // next draw inside the window (0..4)
            uint32_t dr[8];  // draws pos0 .. pos0 + 7 of the window
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t w = pos0 + (uint32_t)j;
                const uint32_t *src = (j >= 4 || w >= 4u) ? &W[w - 4u][lane] : &P[EF_C0 + (int)w][slot];
                dr[j] = *src >> 1;
            }
            const uint32_t need = fixed ? 1u : 2u;
            uint32_t vk[4], idxk[4], nbwk[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                idxk[k] = fm.mod31(dr[2 * k]);
                nbwk[k] = P[EF_NB0 + (int)((idxk[k] >> 5) & 3u)][slot];  // (by address: cf. biased_pool_kernel)
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t ru = dr[2 * k + 1], idx = idxk[k], sh = idx & 31u;
                const bool nbb = !tnonb && ((nbwk[k] >> sh) & 1u);
                const bool tmb = tm_offset ? idx == tm : ((tm >> sh) & 1u) != 0u;  // (a plain mask only describes ranges <= 32)
                const uint32_t v_band = nbb ? (filter ? 2u : v_set) : v_clear;
                const uint32_t v_cls = ru < bp.lo ? 1u : (ru >= bp.hi ? 0u : v_band);
                // the offset may hold t: exact -> the reference's first test, then its class; otherwise look
                const uint32_t v_t = tm_look ? 2u : ((ru < bp.t0) ? 1u : v_cls);
                const uint32_t v_free = tmb ? v_t : v_cls;
                const uint32_t v_fixed = dr[k] < aux ? 1u : 0u;  // a fixed candidate consumes ONE draw per trial
                vk[k] = fixed ? v_fixed : v_free;
            }
            uint32_t verdict = 0u, at = 0u, ru_keep = 0u, used = 0u;
            bool done = !busy;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool take = !done;
                const uint32_t rj = vk[k] == 0u ? rejected + 1u : rejected;
                const bool capped = rj >= kTrialCap;
                if (take && capped) st |= (uint32_t)kStTrialCap;
                const uint32_t v = capped ? 1u : vk[k];
                rejected = take ? rj : rejected;
                verdict = take ? v : verdict;
                at = take ? first + idxk[k] : at;
                ru_keep = take ? dr[2 * k + 1] : ru_keep;
                used = take ? (uint32_t)(k + 1) : used;
                done = done || v != 0u;
                if (kCount) {
                    const unsigned long long ran = __ballot(take);
                    w_reps += ran != 0ull ? 1u : 0u;
                    w_rep_lanes += (uint64_t)__popcll(ran);
                    if (take) ++n_trials;
                }
            }
            const uint32_t pos = pos0 + need * used;
            bool to_issue = false, to_trial = false, to_free = false;
            if (busy) {
                // the block of the last draw consumed becomes the slot's block
                const uint32_t d1 = d0 + (pos - pos0);
                const uint32_t shift = (d1 ? (d1 - 1u) >> 2 : 0u) - cb;  // 0, 1 or 2 blocks on
                if (shift) {
                    P[EF_C0][slot] = W[4 * shift - kWinOff][lane];
                    P[EF_C1][slot] = W[4 * shift + 1 - kWinOff][lane];
                    P[EF_C2][slot] = W[4 * shift + 2 - kWinOff][lane];
                    P[EF_C3][slot] = W[4 * shift + 3 - kWinOff][lane];
                }
                P[EF_D][slot] = d1;
                const uint32_t low = flags & kEfLow;
                P[EF_FLAGS][slot] = low | (rejected << kEfRejShift);
                if (verdict == 0u) {
                    to_trial = true;
                } else if (fixed && P[EF_AT][slot] == kNoEdge) {  // taken without an entry (padding / jump): no gather
                    uint32_t unused_tail = 0u;
                    const int where = step_to(slot, P[EF_X][slot], false, to_free, unused_tail);
                    to_trial = where == 0;
                    to_issue = where == 1;
                } else if (fixed) {  // the fixed candidate is taken through its entry
                    set_kind(slot, low, EK_ENTRY);
                    if (kCount) ++n_fetch;
                    to_issue = true;
                } else {
                    P[EF_AT][slot] = at;
                    P[EF_AUX][slot] = ru_keep;
                    set_kind(slot, low, verdict == 1u ? EK_ENTRY : EK_TEST);
                    if (kCount) ++n_fetch;
                    to_issue = true;
                }
            }
            issue_q.push(to_issue, slot);
            trial_q.push(to_trial, slot);
            free_q.push(to_free, slot);
            pool_fence();
        }
#if TRW_EL_LATE_ISSUE
        // nothing in flight and walkers waiting for a gather (a pool that drains, or one whose issue queue was short
        // before the burst filled it): they leave now instead of an iteration later
        if (__ballot(hold) == 0ull && issue_q.count > 0u && (TRW_EL_LATE_ISSUE == 1 || !more_walkers)) TRW_EL_ISSUE_PASS();
#endif
        lap(3);
    }

    if (st) atomicOr((unsigned long long *)&status[0], (unsigned long long)st);
    if (kCount) {
        const uint64_t a = wave_sum(n_trials), b = wave_sum(n_fetch), c = wave_sum(n_tests), d = wave_sum(n_probes);
        if (lane == 0) {
            const uint64_t w[8] = {w_cycles[0], w_cycles[1], w_cycles[2], w_cycles[3], w_bursts, w_burst_lanes, w_reps, w_rep_lanes};
            for (int k = 0; k < 8; ++k) atomicAdd((unsigned long long *)&status[20 + k], (unsigned long long)w[k]);
            for (int k = 0; k < 8; ++k) atomicAdd((unsigned long long *)&status[12 + k], (unsigned long long)w_kind[k]);
            atomicAdd((unsigned long long *)&status[28], (unsigned long long)w_wait);
            atomicAdd((unsigned long long *)&status[4], (unsigned long long)a);
            atomicAdd((unsigned long long *)&status[5], (unsigned long long)b);
            atomicAdd((unsigned long long *)&status[6], (unsigned long long)c);
            atomicAdd((unsigned long long *)&status[7], (unsigned long long)d);
        }
    }
}

#undef TRW_EL_ISSUE_PASS

}  // namespace
