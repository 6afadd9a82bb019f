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

// is_neighbor(x, t) looks at t's edges [first, last) -- never the last one (rw_cpu_edge_list.cpp:51).  That
// exact set, {(head, tail) of every edge that is not its head's last}, goes into the same open-addressing
// table the CSR walk uses (slots of {u, v, -, -}, all 0xFF when empty), so the test is one probe instead
// of a scan of t's edges.
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
        if (ix.x == kNoEdge || (uint32_t)e < ix.x || (uint32_t)e >= ix.y) continue;  // outside [first, last)
        const uint32_t u = (uint32_t)head, v = tail32[e];
        const unsigned long long key = ((unsigned long long)v << 32) | u;
        uint32_t slot = edge_hash(u, v) & mask;
        while (true) {
            const unsigned long long old = atomicCAS((unsigned long long *)(table + 4 * (size_t)slot), ~0ull, key);
            if (old == ~0ull || old == key) break;
            slot = (slot + 1) & mask;
        }
    }
}

__device__ __forceinline__ bool elist_table_contains(const uint32_t *__restrict__ table, uint32_t mask, uint32_t u,
                                                     uint32_t v)
{
    uint32_t slot = edge_hash(u, v) & mask;
    while (true) {
        const u32x2_a8 kv = *reinterpret_cast<const u32x2_a8 *>(table + 4 * (size_t)slot);
        if (kv.x == 0xFFFFFFFFu) return false;
        if (kv.x == u && kv.y == v) return true;
        slot = (slot + 1) & mask;
    }
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
struct WalkerDraws {
    uint64_t gw;
    uint32_t seed, d;
    uint32_t blk[4];
    __device__ __forceinline__ uint32_t next()
    {
        if ((d & 3u) == 0) walker_block(gw, d >> 2, seed, blk);
        const uint32_t k = d & 3u;
        ++d;
        return (k == 0 ? blk[0] : k == 1 ? blk[1] : k == 2 ? blk[2] : blk[3]) >> 1;
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

}  // namespace
