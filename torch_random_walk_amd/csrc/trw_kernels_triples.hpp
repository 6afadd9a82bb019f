// trw_kernels_triples.hpp -- knowledge-graph (triples) walk, SURVEY 8(f) row f-1.
//
// Reference: triples::walk_triples_cpu (csrc/cpu/rw_cpu_triples.cpp:105-127) ->
// uniform_walk_triples (:48-103) -> triples::sample_neighbor (:11-46) ->
// sample_int (csrc/cpu/cpu_utils.cpp:3-10).  A uniform walk that emits the
// (relation, tail) of every hop: out[i] = start, r1, t1, r2, t2, ...  A hop from
// the padding id or from an entity without out-edges yields (padding, padding)
// and consumes no draw; neither does a hop from a single-edge entity.
//
//   triples_shadow_kernel   int64 triples [T,3] / index [n,2] -> uint32 pairs
//                           (8 bytes per triple instead of 24), validated
//   triples_walker_kernel   one walker per lane on the per-walker stream; per hop
//                           one 8-byte gather of the index pair and one 8-byte
//                           gather of (relation, tail); rows written through WaveRows
//   triples_glibc_seq_kernel the reference's serial glibc stream replayed by one lane
// Included by trw_hip.hip only.
#pragma once

namespace {

constexpr uint32_t kNoEdge = 0xFFFFFFFFu;

typedef uint32_t u32x2_a8 __attribute__((ext_vector_type(2)));

// rt32[e] = {relation, tail};  ix32[v] = {first, last} or {kNoEdge, kNoEdge}.
// A tail that is neither an entity nor the padding id is reported and turned into padding.
__global__ void __launch_bounds__(kBlock)
triples_shadow_kernel(const int64_t *__restrict__ triples, int64_t n_triples, const int64_t *__restrict__ rt_index,
                      int64_t n_entities, int64_t padding_idx, u32x2_a8 *__restrict__ rt32,
                      u32x2_a8 *__restrict__ ix32, int64_t *__restrict__ status)
{
    const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    uint32_t st = 0;
    for (int64_t e = tid; e < n_triples; e += stride) {
        const int64_t rel = triples[3 * e + 1];
        int64_t tail = triples[3 * e + 2];
        if ((uint64_t)rel >= ((uint64_t)1 << 31)) st |= TRW_ST_NODE_RANGE;
        if (tail != padding_idx && (uint64_t)tail >= (uint64_t)n_entities) {
            st |= TRW_ST_NODE_RANGE;
            tail = padding_idx;
        }
        u32x2_a8 v = {(uint32_t)rel, (uint32_t)tail};
        rt32[e] = v;
    }
    for (int64_t v = tid; v < n_entities; v += stride) {
        const int64_t a = rt_index[2 * v], b = rt_index[2 * v + 1];
        u32x2_a8 o = {kNoEdge, kNoEdge};
        if (!(a == -1 || b == -1)) {  // rw_cpu_triples.cpp:26
            if (a < 0 || b < a || b >= n_triples) st |= TRW_ST_BAD_CSR;
            else {
                o.x = (uint32_t)a;
                o.y = (uint32_t)b;
            }
        }
        ix32[v] = o;
    }
    if (st) atomicOr((unsigned long long *)&status[0], (unsigned long long)st);
}

// The lanes of a wave hop in lockstep; their columns reach the caller's [N, 2L+1] rows through the wave's
// LDS ring, one aligned 64-byte line per row at a time (WaveRows, trw_kernels_narrow.hpp).
__global__ void __launch_bounds__(kBlock)
triples_walker_kernel(const u32x2_a8 *__restrict__ rt32, const u32x2_a8 *__restrict__ ix32,
                      const int64_t *__restrict__ targets, int64_t n_entities, int64_t n_walkers,
                      int64_t walker_base, int32_t L, uint32_t pad, uint32_t seed, int64_t *__restrict__ out,
                      int64_t *__restrict__ status)
{
    __shared__ uint32_t tiles[(kBlock / 64) * 64 * kTileStride];
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i == 0) {
        status[2] = TRW_RNG_WALKER;
        status[3] = 1;
    }
    WaveRows rows = {tiles + (threadIdx.x >> 6) * 64 * kTileStride, (int)(threadIdx.x & 63)};
    const int64_t wave_walker0 = (int64_t)blockIdx.x * kBlock + (__builtin_amdgcn_readfirstlane(threadIdx.x) & ~63u);
    const int32_t cols = 2 * L + 1;
    bool live = i < n_walkers;  // the whole wave stays: every lane helps writing the wave's rows
    uint32_t node = pad;
    if (live) {
        const int64_t tg = targets[i];
        node = (uint32_t)tg;
        if (tg != (int64_t)pad && (uint64_t)tg >= (uint64_t)n_entities) {
            atomicOr((unsigned long long *)&status[0], (unsigned long long)TRW_ST_NODE_RANGE);
            node = pad;
            // the start node is the only column that may be anything: a row whose start fits no id at
            // all is filled with it
            if ((uint64_t)tg >= (uint64_t)(pad >= (uint64_t)n_entities ? (int64_t)pad + 1 : n_entities)) {
                live = false;
                for (int32_t c = 0; c < cols; ++c) out[i * (int64_t)cols + c] = tg;
            }
        }
        if (live) rows.put(0, (uint32_t)tg);
    }
    if (!live) rows.put(0, kTileSkip);
    const uint64_t gw = (uint64_t)walker_base + (uint64_t)i;
    uint32_t blk[4] = {0, 0, 0, 0};
    uint32_t d = 0;  // draws this walker has consumed
    // A walker that has reached the padding id stays there: every later column of its row is the padding id.  Left to
    // the ring those columns leave as one 64-byte piece per row and flush, scattered over the wave's 64 rows -- on a
    // graph with sinks (the bench's: 12 live hops of 80 on average) they are most of what the kernel writes, at the
    // memory system's RANDOM-access rate.  Instead the wave writes the rest of such a row at once, 512 contiguous bytes
    // per store instruction, and the ring skips the row from then on.
    bool filled = !live;  // rows that take nothing more from the ring
    int64_t *const row0 = out + wave_walker0 * (int64_t)cols;
    const int ln = (int)(threadIdx.x & 63);
    for (int32_t h = 0; h < L; ++h) {
        uint32_t rel = pad, tail = pad;
        if (live && node != pad) {
            const u32x2_a8 ix = ix32[node];
            if (ix.x != kNoEdge) {
                uint32_t e = ix.x;
                if (ix.y != ix.x) {  // sample_int draws only when there is a choice (cpu_utils.cpp:4-5)
                    if ((d & 3u) == 0) walker_block(gw, d >> 2, seed, blk);
                    const uint32_t k = d & 3u;
                    const uint32_t r = (k == 0 ? blk[0] : k == 1 ? blk[1] : k == 2 ? blk[2] : blk[3]) >> 1;
                    ++d;
                    e += r % (ix.y + 1u - ix.x);
                }
                const u32x2_a8 rt = rt32[e];
                rel = rt.x;
                tail = rt.y;
            }
        }
        const int32_t c = 2 * h + 1;
        rows.put(c, !filled ? rel : kTileSkip);
        if ((c & 7) == 7) rows.flush(wave_walker0, 1, n_walkers, cols, c, out);
        rows.put(c + 1, !filled ? tail : kTileSkip);
        if (((c + 1) & 7) == 7 || c + 1 == cols - 1) rows.flush(wave_walker0, 1, n_walkers, cols, c + 1, out);
        node = tail;
        // rows whose walker has just arrived on the padding id: columns c + 2 .. cols - 1, by the whole wave
        unsigned long long fresh = __ballot(!filled && tail == pad && c + 2 < cols);
        filled = filled || tail == pad;
        while (fresh) {
            const int r = (int)__ffsll((long long)fresh) - 1;
            fresh &= fresh - 1ull;
            int64_t *rowp = row0 + (int64_t)r * cols;
            for (int32_t k = c + 2 + ln; k < cols; k += 64) rowp[k] = (int64_t)pad;
        }
    }
}

// int64 in, int64 out, one walker per lane: graphs whose ids do not fit the uint32 shadow.
__global__ void __launch_bounds__(kBlock)
triples_walker_wide_kernel(const int64_t *__restrict__ triples, int64_t n_triples,
                           const int64_t *__restrict__ rt_index, const int64_t *__restrict__ targets,
                           int64_t n_entities, int64_t n_walkers, int64_t walker_base, int32_t L, int64_t pad,
                           uint32_t seed, int64_t *__restrict__ out, int64_t *__restrict__ status)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i == 0) status[2] = TRW_RNG_WALKER;
    if (i >= n_walkers) return;
    int64_t *row = out + i * (2 * (int64_t)L + 1);
    int64_t node = targets[i];
    row[0] = node;
    uint32_t st = 0;
    const uint64_t gw = (uint64_t)walker_base + (uint64_t)i;
    uint32_t blk[4] = {0, 0, 0, 0};
    uint32_t d = 0;
    for (int32_t h = 0; h < L; ++h) {
        int64_t rel = pad, tail = pad;
        if (node != pad) {
            if ((uint64_t)node >= (uint64_t)n_entities) {
                st |= TRW_ST_NODE_RANGE;
            } else {
                const int64_t a = rt_index[2 * node], b = rt_index[2 * node + 1];
                if (!(a == -1 || b == -1)) {
                    if (a < 0 || b < a || b >= n_triples) {
                        st |= TRW_ST_BAD_CSR;
                    } else {
                        int64_t e = a;
                        if (a != b) {
                            if ((d & 3u) == 0) walker_block(gw, d >> 2, seed, blk);
                            const uint32_t k = d & 3u;
                            const uint32_t r = (k == 0 ? blk[0] : k == 1 ? blk[1] : k == 2 ? blk[2] : blk[3]) >> 1;
                            ++d;
                            e += (int64_t)draw_mod(r, b + 1 - a);
                        }
                        rel = triples[3 * e + 1];
                        tail = triples[3 * e + 2];
                    }
                }
            }
        }
        row[2 * h + 1] = rel;
        row[2 * h + 2] = tail;
        node = tail;
    }
    if (st) atomicOr((unsigned long long *)&status[0], (unsigned long long)st);
}

// The reference's single serial glibc stream, replayed in walker order by one lane.
__global__ void __launch_bounds__(64)
triples_glibc_seq_kernel(const int64_t *__restrict__ triples, int64_t n_triples,
                         const int64_t *__restrict__ rt_index, const int64_t *__restrict__ targets,
                         int64_t n_entities, int64_t n_walkers, int32_t L, int64_t pad, SeedWindow sw,
                         int64_t *__restrict__ out, int64_t *__restrict__ status)
{
    __shared__ uint32_t w[kGlDeg];
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    status[2] = TRW_RNG_GLIBC;
    GlibcSerial glibc;
    glibc.init(w, sw);
    auto next = [&]() -> uint32_t { return glibc.next(); };
    uint32_t st = 0;
    for (int64_t i = 0; i < n_walkers && !st; ++i) {
        int64_t *row = out + i * (2 * (int64_t)L + 1);
        int64_t node = targets[i];
        row[0] = node;
        for (int32_t h = 0; h < L; ++h) {
            int64_t rel = pad, tail = pad;
            if (node != pad) {
                if ((uint64_t)node >= (uint64_t)n_entities) {
                    st |= TRW_ST_NODE_RANGE;  // the chain's draw positions are undefined past a fault
                } else {
                    const int64_t a = rt_index[2 * node], b = rt_index[2 * node + 1];
                    if (!(a == -1 || b == -1)) {
                        if (a < 0 || b < a || b >= n_triples) {
                            st |= TRW_ST_BAD_CSR;
                        } else {
                            int64_t e = a;
                            if (a != b) e += (int64_t)draw_mod(next(), b + 1 - a);
                            rel = triples[3 * e + 1];
                            tail = triples[3 * e + 2];
                        }
                    }
                }
            }
            row[2 * h + 1] = rel;
            row[2 * h + 2] = tail;
            node = tail;
        }
    }
    if (st) atomicOr((unsigned long long *)&status[0], (unsigned long long)st);
}

}  // namespace
