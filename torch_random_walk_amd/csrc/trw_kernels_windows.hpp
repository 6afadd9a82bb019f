// trw_kernels_windows.hpp -- skip-gram / CBOW window builders, SURVEY 8(f) row f-3 (first half).
//
// Reference: to_windows_cpu (csrc/cpu/windows_cpu.cpp:5-77), to_windows_cbow_cpu (:80-158).
// Every walk [cols] yields cols-window_size+1 windows; a window gives its middle node, the other
// window_size-1 nodes, and negatives drawn uniformly from [0, num_nodes).
//   * to_windows draws exactly window_size-1 negatives per window, so negative k of the whole output is
//     draw k of glibc's stream: neg_windows (row-major) IS the stream modulo num_nodes.  It is filled by
//     lanes that jump straight to their run of the stream (the uniform walk's jump-ahead) -- reference
//     exact at any size, in parallel.
//   * to_windows_cbow redraws its one negative while it equals the positive node, a data-dependent number
//     of draws: serial replay of the reference's stream (one lane) or one Philox stream per window.
// Included by trw_hip.hip only.
#pragma once

namespace {

typedef int64_t i64x2_a8 __attribute__((ext_vector_type(2), aligned(8)));  // two values, stored as one 16-byte piece

// Draws per lane of glibc_fill_mod_kernel: a multiple of the generator order, 4 turns for small jobs and up
// to 64 for large ones (every lane costs one jump-ahead, ~6 31x31 polynomial products; 2^18 lanes fill the chip
// -- at 1 M lanes the jump-ahead pass was 135 us in front of a 730 us fill).
__host__ __device__ inline int64_t fill_run(int64_t draws)
{
    int64_t turns = draws / ((int64_t)kGlDeg << 18);   // about 2^18 lanes: every lane costs one jump-ahead
    turns = turns < 4 ? 4 : (turns > 64 ? 64 : turns);
    turns += turns & 1;  // even: the fill leaves two turns of a lane (62 draws, 496 bytes) as one contiguous piece
    return turns * kGlDeg;
}

// n / d and n % d for n < 2^22 and one small d: one multiply by the rounded reciprocal, one correction.
struct SmallDiv {
    float rcp;
    uint32_t d;
    __device__ __forceinline__ void init(uint32_t den)
    {
        d = den;
        rcp = 1.0f / (float)den;
    }
    __device__ __forceinline__ void divmod(uint32_t n, uint32_t &q, uint32_t &r) const
    {
        q = (uint32_t)((float)n * rcp);
        r = n - q * d;
        if ((int32_t)r < 0) {
            --q;
            r += d;
        } else if (r >= d) {
            ++q;
            r -= d;
        }
    }
};

// centre[w] = middle node of window w; pos[w][*] = the window without its middle.
// A block owns `walks_per_block` whole walks: their rows come into LDS with one coalesced read, their windows are
// one contiguous piece of both outputs, written 16 bytes per thread (two consecutive values) -- a write-only
// stream reaches 6.9 TB/s on this part that way, 6.1 with 8 bytes per thread, and a grid of one trip per thread
// runs a third faster than a persistent grid-stride one (scripts/machine_probe.hip).  All index arithmetic is
// 32-bit with reciprocal multiplies (the round-1 form divided 64-bit indices twice per element).
// (Host: walks_per_block * cols <= kWinTileWords, walks_per_block * step_end * width < 2^22.)
constexpr int kWinTileWords = 4096;  // int64 walk entries a block keeps in LDS (32 KB)

__global__ void __launch_bounds__(kBlock)
windows_gather_kernel(const int64_t *__restrict__ walks, int64_t num_walks, int64_t cols, int32_t window_size,
                      int32_t walks_per_block, int64_t *__restrict__ centre, int64_t *__restrict__ pos)
{
    __shared__ int64_t tile[kWinTileWords];
    const uint32_t step_end = (uint32_t)(cols - window_size + 1);
    const uint32_t mid = (uint32_t)window_size / 2u, width = (uint32_t)window_size - 1u;
    const int64_t walk0 = (int64_t)blockIdx.x * walks_per_block;
    const int64_t left = num_walks - walk0;
    const uint32_t nw = left < walks_per_block ? (uint32_t)left : (uint32_t)walks_per_block;
    const uint32_t n_in = nw * (uint32_t)cols;
    const int64_t *src = walks + walk0 * cols;
    for (uint32_t i = threadIdx.x; i < n_in; i += kBlock) tile[i] = src[i];
    __syncthreads();
    SmallDiv by_walk_c, by_walk_p, by_width;
    by_walk_c.init(step_end);
    // ---- centres: nw * step_end consecutive values
    {
        int64_t *dst = centre + walk0 * step_end;
        const uint32_t n = nw * step_end;
        for (uint32_t o = 2u * threadIdx.x; o < n; o += 2u * kBlock) {
            int64_t v[2];
#pragma unroll
            for (uint32_t e = 0; e < 2u; ++e) {
                uint32_t wl, st;
                by_walk_c.divmod(o + e < n ? o + e : o, wl, st);
                v[e] = tile[wl * (uint32_t)cols + st + mid];
            }
            if (o + 1u < n)
                *reinterpret_cast<i64x2_a8 *>(dst + o) = i64x2_a8{v[0], v[1]};
            else
                dst[o] = v[0];
        }
    }
    if (width == 0u) return;
    // ---- the rest of each window: nw * step_end * width consecutive values
    by_walk_p.init(step_end * width);
    by_width.init(width);
    int64_t *dst = pos + walk0 * step_end * width;
    const uint32_t n = nw * step_end * width;
    for (uint32_t o = 2u * threadIdx.x; o < n; o += 2u * kBlock) {
        int64_t v[2];
#pragma unroll
        for (uint32_t e = 0; e < 2u; ++e) {
            uint32_t wl, rem, st, i;
            by_walk_p.divmod(o + e < n ? o + e : o, wl, rem);
            by_width.divmod(rem, st, i);
            v[e] = tile[wl * (uint32_t)cols + st + i + (i >= mid ? 1u : 0u)];
        }
        if (o + 1u < n)
            *reinterpret_cast<i64x2_a8 *>(dst + o) = i64x2_a8{v[0], v[1]};
        else
            dst[o] = v[0];
    }
}

// The same, one thread per output element over the caller's rows in place: walks too long for the tile above.
__global__ void __launch_bounds__(kBlock)
windows_gather_long_kernel(const int64_t *__restrict__ walks, int64_t num_walks, int64_t cols, int32_t window_size,
                           int64_t *__restrict__ centre, int64_t *__restrict__ pos)
{
    const int64_t step_end = cols - window_size + 1;
    const int64_t total = num_walks * step_end;
    const int32_t mid = window_size / 2;
    const int32_t width = window_size - 1;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    for (int64_t w = tid; w < total; w += stride) {
        const int64_t walk_idx = w / step_end;
        centre[w] = walks[walk_idx * cols + (w - walk_idx * step_end) + mid];
    }
    const int64_t elems = total * width;
    for (int64_t f = tid; f < elems; f += stride) {
        const int64_t w = f / width;
        const int32_t i = (int32_t)(f - w * width);
        const int64_t walk_idx = w / step_end;
        pos[f] = walks[walk_idx * cols + (w - walk_idx * step_end) + (i < mid ? i : i + 1)];
    }
}

// out[k] = (draw k of the stream) % num_nodes for k in [0, total); lane l produces draws
// [l*run, (l+1)*run), run = fill_run(total), from the window glibc_windows_kernel left in win[.][l].
// With `rows` set, draw k selects a row of that [num_nodes][3] table instead and out[k][0..2] gets the row
// (the negatives of to_windows_triples, windows_cpu.cpp:279-290).
// A lane's draws are consecutive outputs, so lanes writing their own values would scatter 8-byte stores
// ~2 KB apart: two turns of the generator (62 draws per lane) go through a per-wave LDS tile instead and
// leave as 496-byte contiguous pieces, two per wave store, 16 bytes per thread.
template <bool kRows>
__global__ void __launch_bounds__(kBlock)
glibc_fill_mod_kernel(const uint32_t *__restrict__ win, int64_t nlanes, int64_t total, int64_t num_nodes,
                      const int64_t *__restrict__ rows, int64_t *__restrict__ out)
{
    constexpr int kPiece = 2 * kGlDeg;   // draws of one lane per trip through the tile: two turns of the generator
    constexpr int kStride = kPiece + 1;  // 63: odd, conflict-free lane-per-row writes
    __shared__ uint32_t tiles[(kBlock / 64) * 64 * kStride];
    uint32_t *t = tiles + (threadIdx.x >> 6) * 64 * kStride;
    const int ln = threadIdx.x & 63;
    const int64_t lane = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t wave_lane0 = lane - ln;
    const bool in_range = lane < nlanes;  // the whole wave stays: every lane helps writing the wave's runs
    uint32_t w[kGlDeg];
#pragma unroll
    for (int j = 0; j < kGlDeg; ++j) w[j] = in_range ? win[(int64_t)j * nlanes + lane] : 0u;
    const bool small = num_nodes < ((int64_t)1 << 32);
    FastMod fm;
    fm.init(small ? (uint32_t)num_nodes : 1u);
    const int64_t run = fill_run(total);  // (an even number of turns)
    for (int rep = 0; rep < (int)(run / kPiece); ++rep) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int j = 0; j < kGlDeg; ++j) {
                const uint32_t v = w[j] + w[(j + kGlDeg - kGlSep) % kGlDeg];
                w[j] = v;
                const uint32_t r = v >> 1;
                t[ln * kStride + half * kGlDeg + j] = small ? fm.mod(r) : r;  // r < 2^31 <= num_nodes otherwise
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (kRows) {
            // half a wave per producing lane and turn: 31 of its 32 threads carry one draw (a 24-byte row) each
            const int j = ln & 31;
#pragma unroll 4
            for (int pass = 0; pass < 64; ++pass) {
                const int src = pass & 63, half = ln >> 5;
                const int64_t k = (wave_lane0 + src) * run + (int64_t)rep * kPiece + half * kGlDeg + j;
                if (j < kGlDeg && k < total) {
                    const int64_t *row = rows + (int64_t)t[src * kStride + half * kGlDeg + j] * 3;
                    int64_t *dst = out + k * 3;
                    dst[0] = row[0];
                    dst[1] = row[1];
                    dst[2] = row[2];
                }
            }
        } else {
            // half a wave per producing lane: 31 of its 32 threads carry two draws, 16 bytes, each -- the lane's 496
            // bytes leave as one contiguous piece (8-byte aligned: the store is split by the hardware where it must)
            const int j2 = 2 * (ln & 31);
#pragma unroll 4
            for (int pass = 0; pass < 32; ++pass) {
                const int src = 2 * pass + (ln >> 5);
                const int64_t k = (wave_lane0 + src) * run + (int64_t)rep * kPiece + j2;
                if (j2 < kPiece && k < total) {
                    const int64_t a = (int64_t)t[src * kStride + j2], b = (int64_t)t[src * kStride + j2 + 1];
                    if (k + 1 < total)
                        *reinterpret_cast<i64x2_a8 *>(out + k) = i64x2_a8{a, b};
                    else
                        out[k] = a;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// Triples windows (windows_cpu.cpp:160-474): the target triple and the 2*window_size positive rows around
// it, with the reference's off-by-one left half and its walk[rel] in the head slot of left rows kept as
// they are.  One thread per OUTPUT element: a block owns `walks_per_block` whole walks, whose windows are
// one contiguous piece of both outputs, and writes it 2 KB per block store; the walk entries are re-read
// from L1/L2.  (Host: walks_per_block * per_walk * 2*window_size*3 < 2^22.)
__global__ void __launch_bounds__(kBlock)
triples_windows_gather_kernel(const int64_t *__restrict__ walks, int64_t num_walks, int64_t cols,
                              int32_t window_size, int64_t pad, int32_t walks_per_block,
                              int64_t *__restrict__ target_triples, int64_t *__restrict__ pos_windows)
{
    const uint32_t per_walk = (uint32_t)((cols - 1) / 2);
    const uint32_t elems = 2u * (uint32_t)window_size * 3u;  // per window
    const int64_t walk0 = (int64_t)blockIdx.x * walks_per_block;
    const int64_t left = num_walks - walk0;
    const uint32_t nw = left < walks_per_block ? (uint32_t)left : (uint32_t)walks_per_block;
    SmallDiv by_elems, by_walk, by3;
    by_elems.init(elems);
    by_walk.init(per_walk);
    by3.init(3u);
    const int64_t *walk_base = walks + walk0 * cols;
    int64_t *pos = pos_windows + walk0 * (int64_t)per_walk * elems;
    const uint32_t n_pos = nw * per_walk * elems;
    for (uint32_t o = threadIdx.x; o < n_pos; o += kBlock) {
        uint32_t wl, rem, walk_local, win, slot, c;
        by_elems.divmod(o, wl, rem);
        by_walk.divmod(wl, walk_local, win);
        by3.divmod(rem, slot, c);
        const int64_t *walk = walk_base + (int64_t)walk_local * cols;
        const int64_t tri = 2 * (int64_t)win + 1;  // index of the target relation
        int64_t v = pad;
        if (slot < (uint32_t)window_size) {  // left rows (the reference also writes row window_size, overwritten by the right half)
            const int64_t rel = tri - ((int64_t)slot + 1) * 2;
            if (c == 0) {
                if (rel - 1 >= 0) v = walk[rel];  // sic: walk[rel], windows_cpu.cpp:221
            } else if (c == 1) {
                if (rel >= 0) v = walk[rel];
            } else {
                if (rel + 1 >= 0) v = walk[rel + 1];
            }
        } else {
            const int64_t idx = tri + ((int64_t)slot - window_size + 1) * 2 - 1 + c;
            if (idx < cols) v = walk[idx];
        }
        pos[o] = v;
    }
    int64_t *tgt = target_triples + walk0 * (int64_t)per_walk * 3;
    const uint32_t n_tgt = nw * per_walk * 3u;
    for (uint32_t o = threadIdx.x; o < n_tgt; o += kBlock) {
        uint32_t wl, c, walk_local, win;
        by3.divmod(o, wl, c);
        by_walk.divmod(wl, walk_local, win);
        tgt[o] = walk_base[(int64_t)walk_local * cols + 2 * (int64_t)win + c];
    }
}

// The same, one thread per window: walks so long that one walk's windows overflow the 32-bit element
// index of the kernel above.
__global__ void __launch_bounds__(kBlock)
triples_windows_gather_long_kernel(const int64_t *__restrict__ walks, int64_t num_walks, int64_t cols,
                              int32_t window_size, int64_t pad, int64_t *__restrict__ target_triples,
                              int64_t *__restrict__ pos_windows)
{
    const int64_t per_walk = (cols - 1) / 2;
    const int64_t total = num_walks * per_walk;
    const int64_t slots = 2 * (int64_t)window_size;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t w = (int64_t)blockIdx.x * kBlock + threadIdx.x; w < total; w += stride) {
        const int64_t walk_idx = w / per_walk;
        const int64_t tri = 2 * (w - walk_idx * per_walk) + 1;  // index of the target relation
        const int64_t *walk = walks + walk_idx * cols;
        target_triples[w * 3 + 0] = walk[tri - 1];
        target_triples[w * 3 + 1] = walk[tri];
        target_triples[w * 3 + 2] = walk[tri + 1];
        int64_t *win = pos_windows + w * slots * 3;
        // left rows 0..window_size-1 (the reference also writes row window_size, which the right half overwrites)
        for (int64_t hop = 0; hop < window_size; ++hop) {
            const int64_t rel = tri - (hop + 1) * 2;
            win[hop * 3 + 0] = rel - 1 >= 0 ? walk[rel] : pad;  // sic: walk[rel], windows_cpu.cpp:221
            win[hop * 3 + 1] = rel >= 0 ? walk[rel] : pad;
            win[hop * 3 + 2] = rel + 1 >= 0 ? walk[rel + 1] : pad;
        }
        for (int64_t hop = 0; hop < window_size; ++hop) {
            const int64_t rel = tri + (hop + 1) * 2;
            int64_t *o = win + (hop + window_size) * 3;
            o[0] = rel - 1 < cols ? walk[rel - 1] : pad;
            o[1] = rel < cols ? walk[rel] : pad;
            o[2] = rel + 1 < cols ? walk[rel + 1] : pad;
        }
    }
}

// CBOW negative triple: one drawn row of `triples`, redrawn while it equals the positive triple
// (windows_cpu.cpp:374-389); kSerial replays the reference's one glibc stream with a single lane.
template <bool kSerial>
__global__ void __launch_bounds__(kSerial ? 64 : kBlock)
cbow_negative_triples_kernel(const int64_t *__restrict__ pos_triples, int64_t total,
                             const int64_t *__restrict__ triples, int64_t num_triples, int64_t window_base,
                             uint32_t seed, SeedWindow sw, int64_t *__restrict__ neg_triples,
                             int64_t *__restrict__ status)
{
    __shared__ uint32_t gw[kGlDeg];
    GlibcSerial glibc;
    glibc.w = gw;
    glibc.slot = 0;
    auto gnext = [&]() -> uint32_t { return glibc.next(); };
    int64_t lo, hi;
    if (kSerial) {
        if (threadIdx.x != 0 || blockIdx.x != 0) return;
        status[2] = TRW_RNG_GLIBC;
        glibc.init(gw, sw);
        lo = 0;
        hi = total;
    } else {
        lo = (int64_t)blockIdx.x * kBlock + threadIdx.x;
        if (lo == 0) status[2] = TRW_RNG_WALKER;
        hi = lo < total ? lo + 1 : lo;
    }
    for (int64_t w = lo; w < hi; ++w) {
        const int64_t ph = pos_triples[w * 3], pr = pos_triples[w * 3 + 1], pt = pos_triples[w * 3 + 2];
        WalkerDraws rng;
        rng.gw = (uint64_t)window_base + (uint64_t)w;
        rng.seed = seed;
        rng.d = 0;
        auto sample = [&]() -> int64_t {
            if (num_triples == 1) return 0;
            return (int64_t)draw_mod(kSerial ? gnext() : rng.next(), num_triples);
        };
        int64_t idx = sample();
        int checks = 0;
        while (triples[idx * 3] == ph && triples[idx * 3 + 1] == pr && triples[idx * 3 + 2] == pt && checks <= 100) {
            idx = sample();
            ++checks;
        }
        neg_triples[w * 3 + 0] = triples[idx * 3];
        neg_triples[w * 3 + 1] = triples[idx * 3 + 1];
        neg_triples[w * 3 + 2] = triples[idx * 3 + 2];
    }
}

// One negative per window on the window's own stream (key window_base + w), redrawn while it equals
// the positive node, at most 101 times (windows_cpu.cpp:129-135).
__global__ void __launch_bounds__(kBlock)
cbow_negatives_walker_kernel(const int64_t *__restrict__ pos_nodes, int64_t total, int64_t num_nodes,
                             int64_t window_base, uint32_t seed, int64_t *__restrict__ neg_nodes,
                             int64_t *__restrict__ status)
{
    const int64_t w = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (w == 0) status[2] = TRW_RNG_WALKER;
    if (w >= total) return;
    const int64_t pos_node = pos_nodes[w];
    WalkerDraws rng;
    rng.gw = (uint64_t)window_base + (uint64_t)w;
    rng.seed = seed;
    rng.d = 0;
    auto sample = [&]() -> int64_t {  // sample_int(0, num_nodes-1): no draw for a single-value range
        return num_nodes == 1 ? 0 : (int64_t)draw_mod(rng.next(), num_nodes);
    };
    int64_t neg = sample();
    int checks = 0;
    while (neg == pos_node && checks <= 100) {
        neg = sample();
        ++checks;
    }
    neg_nodes[w] = neg;
}

// The same on the reference's single serial glibc stream: one lane, windows in order.
__global__ void __launch_bounds__(64)
cbow_negatives_glibc_seq_kernel(const int64_t *__restrict__ pos_nodes, int64_t total, int64_t num_nodes,
                                SeedWindow sw, int64_t *__restrict__ neg_nodes, int64_t *__restrict__ status)
{
    __shared__ uint32_t w[kGlDeg];
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    status[2] = TRW_RNG_GLIBC;
    GlibcSerial glibc;
    glibc.init(w, sw);
    auto next = [&]() -> uint32_t { return glibc.next(); };
    for (int64_t i = 0; i < total; ++i) {
        const int64_t pos_node = pos_nodes[i];
        int64_t neg = num_nodes == 1 ? 0 : (int64_t)draw_mod(next(), num_nodes);
        int checks = 0;
        while (neg == pos_node && checks <= 100) {
            neg = num_nodes == 1 ? 0 : (int64_t)draw_mod(next(), num_nodes);
            ++checks;
        }
        neg_nodes[i] = neg;
    }
}

}  // namespace
