// trw_kernels_narrow.hpp -- the production kernels: 32-bit shadow CSR, rows written line by line.
//
// What the profile of the int64 kernels showed (profiles/r01_*): both walks are
// bound by the number of cache lines that miss L2 (~5e10 64-byte requests/s, the
// memory system's random-access ceiling, profiles/r01_machine_probe.txt), and every
// 8-byte scattered store cost 32 bytes of HBM write.  So these kernels are built to
// touch fewer lines:
//
//   csr_shadow_kernel         one streaming pass: int64 CSR -> uint32 shadow (half the
//                             footprint, twice the entries per line), validates it and
//                             counts order descents so the walk knows whether rows are sorted
//   csr_ext_kernel            fat entries {v, start(v), deg(v)}: one gather per step
//   uniform_glibc_rows_kernel reference-exact uniform walk; the wave writes its 64 rows one
//                             aligned 64-byte line at a time through an LDS ring (WaveRows)
//   edge_table_build_kernel   open-addressing set of the edges {u, v, first offset, multiplicity}
//   (trw_kernels_marks.hpp)   32-byte entries for the biased walk: the neighbour's row, where the node
//                             being left sits in it, a 128-bit mask answering is_neighbor per offset
//   biased_pool_kernel        node2vec rejection sampling on the per-walker stream, the default: a pool of
//                             walkers per wave in LDS, lanes as executors (sorted rows, prepared entries)
//   biased_mask_kernel        the same over the same entries, one walker per lane (>= 2^32 walkers)
//   biased_hash_kernel        the same for graphs with unsorted rows: edge table + per-lane state machine
//   biased_walker_kernel      the same walk for small batches (bisection instead of the table)
//   biased_glibc_wave_kernel  the reference's ONE serial glibc stream, replayed by a whole wave
//
// Included by trw_hip.hip only.  Valid for n_nodes < 2^31 and nnz < 2^32.
#pragma once

namespace {


// ---------------------------------------------------------------------------
// Shadow CSR
// ---------------------------------------------------------------------------
// col32[e] = col_idx[e], row32[v] = row_ptr[v]; out-of-contract values are
// clamped (so nothing downstream can index outside the arrays) and reported.
// Rows are all sorted  <=>  every descent col[e-1] > col[e] sits on a row
// boundary  <=>  status[kStDescents] == status[kStBoundary].
template <bool kOrder>
__global__ void __launch_bounds__(kBlock)
csr_shadow_kernel(const int64_t *__restrict__ row_ptr, const int64_t *__restrict__ col_idx, int64_t n_nodes,
                  int64_t nnz, uint32_t *__restrict__ row32, uint32_t *__restrict__ col32,
                  int64_t *__restrict__ status)
{
    const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    uint32_t st = 0;
    uint32_t desc = 0, bdesc = 0;
    for (int64_t e = tid; e < nnz; e += stride) {
        const int64_t c = col_idx[e];
        const bool ok = (uint64_t)c < (uint64_t)n_nodes;
        if (!ok) st |= TRW_ST_NODE_RANGE;
        col32[e] = ok ? (uint32_t)c : 0u;
        if (kOrder && e + 1 < nnz) desc += (c > col_idx[e + 1]) ? 1u : 0u;
    }
    for (int64_t v = tid; v <= n_nodes; v += stride) {
        int64_t r = row_ptr[v];
        bool bad = r < 0 || r > nnz;
        if (v < n_nodes && row_ptr[v + 1] < r) bad = true;
        // a single row of 2^31 entries or more (17 GB of parallel edges) would alias the flag bit the 32-byte entries
        // keep in the top of their degree word (kMaskIsFilter): outside what the uint32 layout describes, reported
        if (v < n_nodes && row_ptr[v + 1] - r >= ((int64_t)1 << 31)) bad = true;
        if (bad) st |= TRW_ST_BAD_CSR;
        r = r < 0 ? 0 : (r > nnz ? nnz : r);
        row32[v] = (uint32_t)r;
        if (kOrder && v >= 1 && r > 0 && r < nnz && row_ptr[v - 1] < r)  // first node whose row starts at r
            bdesc += (col_idx[r - 1] > col_idx[r]) ? 1u : 0u;
    }
    if (st) atomicOr((unsigned long long *)&status[0], (unsigned long long)st);
    if (kOrder) {
        const uint64_t a = wave_sum(desc), b = wave_sum(bdesc);
        if ((threadIdx.x & 63) == 0) {
            if (a) atomicAdd((unsigned long long *)&status[kStDescents], (unsigned long long)a);
            if (b) atomicAdd((unsigned long long *)&status[kStBoundary], (unsigned long long)b);
        }
    }
}

// The row of `node` in the shadow.  row32 is clamped to [0, nnz]; a non-monotone
// pair (already reported by csr_shadow_kernel) reads as an empty row.
// Both offsets come back from ONE 8-byte gather (4-byte aligned global_load_dwordx2):
// a divergent load costs the memory pipeline per lane-request, not per byte.
typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef uint32_t u32x4_a16 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void load_row32(const uint32_t *__restrict__ row32, uint32_t node, uint32_t &start,
                                           uint32_t &deg)
{
    const u32x2_a4 ab = *reinterpret_cast<const u32x2_a4 *>(row32 + node);
    start = ab.x;
    deg = ab.y >= ab.x ? ab.y - ab.x : 0u;
}

// ---------------------------------------------------------------------------
// Uniform walk over "fat" entries: one gather per step
// ---------------------------------------------------------------------------
// ext[e] = {v, row32[v], deg(v), 0} for v = col32[e]: the entry a walker picks already carries the row of
// the node it lands on, so a step is ONE 16-byte gather instead of the dependent pair row32 -> col32.
// Costs one pass over the entries per call (random 8-byte reads of row32, which is cache resident) and
// 16 bytes per entry of workspace.
__global__ void __launch_bounds__(kBlock)
csr_ext_kernel(const uint32_t *__restrict__ row32, const uint32_t *__restrict__ col32, int64_t nnz,
               u32x4_a16 *__restrict__ ext)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < nnz; e += stride) {
        const uint32_t v = col32[e];
        uint32_t start, deg;
        load_row32(row32, v, start, deg);
        u32x4_a16 o = {v, start, deg, 0u};
        ext[e] = o;
    }
}

// ---------------------------------------------------------------------------
// Lane-per-walker columns -> line-aligned int64 row stores, inside the walk kernel
// ---------------------------------------------------------------------------
// The lanes of a wave advance in lockstep (same column of 64 different rows).  Each lane parks its
// columns in a 16-slot ring of a per-wave LDS tile; every 8 columns the wave writes, for each of its 64
// rows, the one 64-byte line of the caller's [N, L+1] int64 tensor that has just become complete (8 lanes
// per row, 8 rows per pass) -- rows start at any multiple of 8 bytes, so which columns make up a line
// differs from row to row (`phase`).  Only full, aligned lines reach HBM (a row's first and last line
// excepted); 8 bytes of HBM traffic per column, no staging buffer and no transpose pass.
constexpr int kTileCols = 16;
constexpr int kTileStride = kTileCols + 1;  // odd stride: conflict-free lane-per-row writes
constexpr uint32_t kTileSkip = 0xFFFFFFFFu;  // not a node id (ids < 2^31): nothing to write for this row

struct WaveRows {
    uint32_t *t;  // this wave's tile, [64][kTileStride]
    int lane;     // lane within the wave
    __device__ __forceinline__ void put(int32_t col, uint32_t v) { t[lane * kTileStride + (col & (kTileCols - 1))] = v; }

    // Call after column c of every row is in the tile, when (c & 7) == 7 or c is the last column.
    // Rows: walker `walker0 + r * walker_stride` for wave lane r.  Everything but the lane's offset inside
    // the wave's block of rows is wave-uniform and lives in scalar registers (the host guarantees
    // 64 * walker_stride * cols * 8 < 2^32, so that offset is a 32-bit byte offset).
    // OutT: int64_t (the reference's rows) or uint32_t (narrow rows for the multi-GPU gather: node ids < 2^32); a "line"
    // is 8 columns either way -- 64 bytes of int64, an aligned 32-byte half line of uint32.
    template <class OutT>
    __device__ __forceinline__ void flush(int64_t walker0, int32_t walker_stride, int64_t n_walkers, int32_t cols,
                                          int32_t c, OutT *__restrict__ out) const
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int64_t w0 = walker0;  // wave-uniform by contract
        int64_t left = n_walkers - w0;  // rows of this wave that exist: those with r * stride < left
        left = left < 0 ? 0 : (left + walker_stride - 1) / walker_stride;
        const int nrows = left > 64 ? 64 : (int)left;
        OutT *base = out + w0 * cols;
        const uint32_t phase0 = (uint32_t)((uintptr_t)base / sizeof(OutT));  // low 3 bits: where row 0 starts in its line
        const uint32_t row_elems = (uint32_t)walker_stride * (uint32_t)cols;
        // (opaque copy: keeps the compiler from hoisting every pass's per-lane offsets out of the walk loop
        // into registers that the 31-word generator window needs)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int e = ln & 7;
        const bool last = c == cols - 1;
        // group j of a row = the columns of its j-th line: [8j - phase, 8j - phase + 8)
        const int j0 = last ? (cols - 1) >> 3 : c >> 3;  // the groups completed before were written before
        const int j1 = last ? j0 + 1 : j0;
        for (int j = j0; j <= j1; ++j) {
#pragma unroll 2
            for (int pass = 0; pass < 8; ++pass) {
                const int row = 8 * pass + (ln >> 3);
                const uint32_t off0 = (uint32_t)row * row_elems;
                const int32_t col = 8 * j - (int32_t)((phase0 + off0) & 7u) + e;
                if (col >= 0 && col <= c && row < nrows) {
                    const uint32_t v = t[row * kTileStride + (col & (kTileCols - 1))];
                    if (v != kTileSkip) base[off0 + (uint32_t)col] = (OutT)v;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
};

// The skip-gram windows of the wave's rows, straight from the tile (rw.walk_windows: the walk fused with
// to_windows, windows_cpu.cpp:5-77, so that the [N, L+1] rows need not make the round trip through HBM).
// Call right after flush() for the same column c: the windows that column c (and the columns since the previous
// flush) completed -- at most 8 per row -- go out: centre[w] = middle node, pos[w][0..ws-2] = the others.  A row's
// new windows are ONE contiguous piece of `pos` (up to 8 * (ws-1) values), written by `lpr` adjacent lanes.
// The ring holds 16 columns and a window started up to ws + 6 columns ago: window_size <= 9.
#ifndef TRW_WIN_STORE16
#define TRW_WIN_STORE16 1
#endif
typedef int64_t i64x2_a16 __attribute__((ext_vector_type(2)));

struct WaveWindows {
    int32_t ws, mid, width, step_end;
    uint32_t lpr_shift;   // lanes per row = 1 << lpr_shift >= 8 * width
    uint32_t inv_width;   // ceil(65536 / width): f / width == (f * inv_width) >> 16 for f < 64
    int64_t *centre, *pos;
    __device__ __forceinline__ void init(int32_t window_size, int32_t cols, int64_t *centre_, int64_t *pos_)
    {
        ws = window_size;
        mid = window_size / 2;
        width = window_size - 1;
        step_end = cols - window_size + 1;
        lpr_shift = width <= 1 ? 3u : (width <= 2 ? 4u : (width <= 4 ? 5u : 6u));
        inv_width = width > 0 ? (65536u + (uint32_t)width - 1u) / (uint32_t)width : 0u;
        centre = centre_;
        pos = pos_;
    }
    // the windows [w_lo, w_hi) that became complete with column c
    __device__ __forceinline__ void range(int32_t cols, int32_t c, int32_t &w_lo, int32_t &w_hi) const
    {
        const bool last = c == cols - 1;
        const int32_t c_prev = (last && (c & 7) != 7) ? (c & ~7) - 1 : c - 8;  // the flush before this one (< 0: none)
        w_lo = c_prev - ws + 2;
        w_lo = w_lo < 0 ? 0 : w_lo;
        w_hi = c - ws + 2;
        w_hi = w_hi < 0 ? 0 : w_hi;
    }
    __device__ __forceinline__ void flush(const WaveRows &rows, int64_t walker0, int32_t walker_stride, int64_t n_walkers,
                                          int32_t cols, int32_t c) const
    {
        int32_t w_lo, w_hi;
        range(cols, c, w_lo, w_hi);
        const int32_t nwin = w_hi - w_lo;
        if (nwin <= 0) return;
        int64_t left = n_walkers - walker0;
        left = left < 0 ? 0 : (left + walker_stride - 1) / walker_stride;
        const int nrows = left > 64 ? 64 : (int)left;
        int ln = rows.lane;
        asm volatile("" : "+v"(ln));
        const uint32_t row_windows = (uint32_t)walker_stride * (uint32_t)step_end;  // windows between the wave's rows
        // ---- centres: 8 lanes per row, 8 rows per pass
        int64_t *cbase = centre + walker0 * step_end + w_lo;
        {
            const int k = ln & 7;
#pragma unroll 2
            for (int pass = 0; pass < 8; ++pass) {
                const int row = 8 * pass + (ln >> 3);
                if (k < nwin && row < nrows) {
                    const uint32_t v = rows.t[row * kTileStride + ((w_lo + k + mid) & (kTileCols - 1))];
                    if (v != kTileSkip) cbase[(uint32_t)row * row_windows + (uint32_t)k] = (int64_t)v;
                }
            }
        }
        if (width == 0) return;
        // ---- the rest of each window: a row's piece is nwin * width consecutive values
        int64_t *pbase = pos + (walker0 * step_end + w_lo) * width;
        const uint32_t lpr = 1u << lpr_shift;
#if TRW_WIN_STORE16
        // An even width: a row's piece starts on a 16-byte boundary and holds an even number of values, so a lane
        // takes TWO of them and stores 16 bytes -- half the store instructions per flush (the stores are what the
        // fused kernel adds to the walk).
        if ((width & 1) == 0 && ((uintptr_t)pos & 15u) == 0u) {
            const uint32_t lpr2 = lpr >> 1, rows_per_pass = 128u >> lpr_shift;
            const uint32_t q = (uint32_t)ln & (lpr2 - 1u), sub = (uint32_t)ln >> (lpr_shift - 1u);
            const uint32_t f0 = 2u * q, f1 = f0 + 1u;
            const uint32_t k0 = (f0 * inv_width) >> 16, i0 = f0 - k0 * (uint32_t)width;
            const uint32_t k1 = (f1 * inv_width) >> 16, i1 = f1 - k1 * (uint32_t)width;
            const uint32_t col0 = (uint32_t)w_lo + k0 + i0 + (i0 >= (uint32_t)mid ? 1u : 0u);
            const uint32_t col1 = (uint32_t)w_lo + k1 + i1 + (i1 >= (uint32_t)mid ? 1u : 0u);
            const bool mine = f0 < (uint32_t)nwin * (uint32_t)width;
            for (uint32_t pass = 0; pass < lpr2; ++pass) {
                const uint32_t row = pass * rows_per_pass + sub;
                if (mine && row < (uint32_t)nrows) {
                    const uint32_t v0 = rows.t[row * kTileStride + (col0 & (kTileCols - 1))];
                    const uint32_t v1 = rows.t[row * kTileStride + (col1 & (kTileCols - 1))];
                    if (v0 != kTileSkip)  // (a row is skipped as a whole)
                        *reinterpret_cast<i64x2_a16 *>(pbase + row * row_windows * (uint32_t)width + f0) =
                            i64x2_a16{(int64_t)v0, (int64_t)v1};
                }
            }
            return;
        }
#endif
        const uint32_t rows_per_pass = 64u >> lpr_shift;
        const uint32_t f = (uint32_t)ln & (lpr - 1u), sub = (uint32_t)ln >> lpr_shift;
        const uint32_t k = (f * inv_width) >> 16, i = f - k * (uint32_t)width;
        const uint32_t col = (uint32_t)w_lo + k + i + (i >= (uint32_t)mid ? 1u : 0u);
        const bool mine = f < (uint32_t)nwin * (uint32_t)width;
        for (uint32_t pass = 0; pass < lpr; ++pass) {
            const uint32_t row = pass * rows_per_pass + sub;
            if (mine && row < (uint32_t)nrows) {
                const uint32_t v = rows.t[row * kTileStride + (col & (kTileCols - 1))];
                if (v != kTileSkip) pbase[row * row_windows * (uint32_t)width + f] = (int64_t)v;
            }
        }
    }
};

// For kernels whose lanes advance at their own pace (the biased walks): the last (up to) four columns of
// the lane's row wait in registers until they fill a 32-byte-aligned group of the caller's row, the
// granule a partial HBM write costs anyway (one write request per four columns instead of one per column).
typedef uint32_t u32x4_o16 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_a8 __attribute__((ext_vector_type(2)));
template <class OutT>
struct RowCombinerT {
    OutT *sp;  // where the next column goes
    uint32_t wb0, wb1, wb2, wb3;
    __device__ __forceinline__ void emit(uint32_t v, int32_t col, int32_t last_col)
    {
        const uint32_t k = (uint32_t)((uintptr_t)sp / sizeof(OutT)) & 3u;
        wb0 = k == 0u ? v : wb0;
        wb1 = k == 1u ? v : wb1;
        wb2 = k == 2u ? v : wb2;
        wb3 = k == 3u ? v : wb3;
        ++sp;
        if (k == 3u || col == last_col) {
            const uint32_t have = min(k + 1u, (uint32_t)col + 1u);  // columns of THIS row in the group
            OutT *grp = sp - (k + 1u);
            if (have == 4u) {
                if (sizeof(OutT) == 8) {
                    const i64x2_a16 lo2 = {(int64_t)wb0, (int64_t)wb1}, hi2 = {(int64_t)wb2, (int64_t)wb3};
                    *reinterpret_cast<i64x2_a16 *>(grp) = lo2;
                    *reinterpret_cast<i64x2_a16 *>(grp + 2) = hi2;
                } else {
                    *reinterpret_cast<u32x4_o16 *>(grp) = u32x4_o16{wb0, wb1, wb2, wb3};  // (16-byte aligned: k == 3)
                }
            } else {
                const uint32_t first = k + 1u - have;
                if (first == 0u) grp[0] = (OutT)wb0;
                if (first <= 1u && k >= 1u) grp[1] = (OutT)wb1;
                if (first <= 2u && k >= 2u) grp[2] = (OutT)wb2;
                if (k >= 3u) grp[3] = (OutT)wb3;
            }
        }
    }
};
typedef RowCombinerT<int64_t> RowCombiner;

// ---------------------------------------------------------------------------
// Uniform walk, reference-exact draws (uniform_walk, rw_cpu.cpp:58-107)
// ---------------------------------------------------------------------------
// Lane l owns walkers [l*G, l*G+G): G*L consecutive words of the glibc stream, produced in registers: one
// add per draw on a 31-word window that is indexed statically, four draws at a time through a 31-way
// wave-uniform switch (so the step, with its row-flush code, exists four times in the kernel instead of
// once per window position).  kFat = false: a step is two dependent gathers, row32[v],row32[v+1] then
// col32[...]; kFat = true: one 16-byte gather of a fat entry.  Rows are written through WaveRows.
#define TRW_GEN4(J)                                                                     \
    case J: {                                                                           \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                              \
            constexpr int base_ = J;                                                    \
            const int jj_ = (base_ + i_) % kGlDeg;                                      \
            const uint32_t v_ = w[jj_] + w[(jj_ + kGlDeg - kGlSep) % kGlDeg];           \
            w[jj_] = v_;                                                                \
            d[i_] = v_ >> 1;                                                            \
        }                                                                               \
    } break;

// Register budget: the fat-entry walk is bound by the memory system's request rate and runs faster with 5
// waves per SIMD than with 8 (measured 16.3 vs 17.3 ms on R-MAT 10M/110M, and no better with 2-4; the compact walk
// does not care).
// kWin: the fused walk + to_windows (WaveWindows above); `out` may then be null (the rows themselves are not wanted).
template <bool kFat, bool kWin, class OutT = int64_t>
__global__ void __launch_bounds__(kBlock, (kFat || kWin) ? 4 : 8)
uniform_glibc_rows_kernel(const uint32_t *__restrict__ row32, const uint32_t *__restrict__ col32,
                          const u32x4_a16 *__restrict__ ext, const int64_t *__restrict__ targets, int64_t n_nodes,
                          int64_t n_walkers, int32_t L, int32_t G, int64_t nlanes, const uint32_t *__restrict__ win,
                          OutT *__restrict__ out, int64_t *__restrict__ status, int32_t window_size,
                          int64_t *__restrict__ centre, int64_t *__restrict__ pos)
{
    __shared__ uint32_t tiles[(kBlock / 64) * 64 * kTileStride];
    const int64_t lane = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (lane == 0) {
        status[2] = TRW_RNG_GLIBC;
        status[3] = kFat ? 3 : 1;
    }
    const bool in_range = lane < nlanes;  // the whole wave stays: every lane helps writing the wave's rows
    WaveRows rows = {tiles + (threadIdx.x >> 6) * 64 * kTileStride, (int)(threadIdx.x & 63)};
    WaveWindows wins;
    if (kWin) wins.init(window_size, L + 1, centre, pos);
    // first walker of the wave's lane 0: wave-uniform, kept in scalar registers
    const int64_t wave_walker0 = ((int64_t)blockIdx.x * kBlock + (__builtin_amdgcn_readfirstlane(threadIdx.x) & ~63u)) * G;

    uint32_t w[kGlDeg];
#pragma unroll
    for (int j = 0; j < kGlDeg; ++j) w[j] = in_range ? win[(int64_t)j * nlanes + lane] : 0u;

    uint32_t st = 0;
    int64_t walker = lane * G;
    int32_t s = 0;  // next column of the current walker; 0 = start a new walker
    int32_t g = 0;  // which of the lane's G walkers
    uint32_t cur = 0, start = 0, deg = 0;
    bool live = false;
    const int32_t total = G * L;

    auto step = [&](uint32_t r) {
        if (s == 0) {  // wave-uniform: every lane starts its next walker together
            live = walker < n_walkers;
            if (live) {
                const int64_t tg = targets[walker];
                if ((uint64_t)tg >= (uint64_t)n_nodes) {  // reported; the row is filled with the bad id
                    st |= TRW_ST_NODE_RANGE;
                    live = false;
                    if (out)
                        for (int32_t c = 0; c <= L; ++c) out[walker * (int64_t)(L + 1) + c] = (OutT)tg;
                    if (kWin) {  // ... and so are its windows
                        for (int64_t k = walker * wins.step_end; k < (walker + 1) * wins.step_end; ++k) {
                            centre[k] = tg;
                            for (int32_t i = 0; i < wins.width; ++i) pos[k * wins.width + i] = tg;
                        }
                    }
                }
                cur = (uint32_t)tg;
                if (kFat && live) load_row32(row32, cur, start, deg);  // the only row32 gather of the walk
            }
            rows.put(0, live ? cur : kTileSkip);
            s = 1;
        }
        if (live) {
            if (!kFat) load_row32(row32, cur, start, deg);
            if (deg == 0) {
                st |= TRW_ST_DEGREE0;  // rw_cpu.cpp:20 divides by zero here
            } else if (kFat) {
                const u32x4_a16 x = ext[start + r % deg];
                cur = x.x;
                start = x.y;
                deg = x.z;
            } else {
                cur = col32[start + r % deg];
            }
        }
        rows.put(s, live ? cur : kTileSkip);
        if ((s & 7) == 7 || s == L) {
            if (!kWin || out) rows.flush(wave_walker0 + g, G, n_walkers, L + 1, s, out);
            if (kWin) {
                if (!out) {  // (flush() fences the tile on both sides; without it, do so here)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                wins.flush(rows, wave_walker0 + g, G, n_walkers, L + 1, s);
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        if (++s > L) {
            s = 0;
            ++walker;
            ++g;
        }
    };

    int32_t j0 = 0;  // window position of the next draw
    for (int32_t base = 0; base < total; base += 4) {
        uint32_t d[4];
        switch (j0) {
            TRW_GEN4(0) TRW_GEN4(1) TRW_GEN4(2) TRW_GEN4(3) TRW_GEN4(4) TRW_GEN4(5) TRW_GEN4(6) TRW_GEN4(7)
            TRW_GEN4(8) TRW_GEN4(9) TRW_GEN4(10) TRW_GEN4(11) TRW_GEN4(12) TRW_GEN4(13) TRW_GEN4(14) TRW_GEN4(15)
            TRW_GEN4(16) TRW_GEN4(17) TRW_GEN4(18) TRW_GEN4(19) TRW_GEN4(20) TRW_GEN4(21) TRW_GEN4(22) TRW_GEN4(23)
            TRW_GEN4(24) TRW_GEN4(25) TRW_GEN4(26) TRW_GEN4(27) TRW_GEN4(28) TRW_GEN4(29) TRW_GEN4(30)
        default: d[0] = d[1] = d[2] = d[3] = 0; break;
        }
        j0 += 4;
        if (j0 >= kGlDeg) j0 -= kGlDeg;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (base + i < total) step(d[i]);  // wave-uniform
    }
    if (st) atomicOr((unsigned long long *)&status[0], (unsigned long long)st);
}
#undef TRW_GEN4

// ---------------------------------------------------------------------------
// Biased (node2vec) walk on the per-walker stream (biased_walk, rw_cpu.cpp:109-201)
// ---------------------------------------------------------------------------

// n % d for many n and one d: LLVM's exact 32-bit expansion with the
// reciprocal hoisted out (one Newton step on v_rcp_f32, two corrections).
struct FastMod {
    uint32_t d, z;
    __device__ __forceinline__ void init(uint32_t den)
    {
        d = den;
        z = (uint32_t)(__builtin_amdgcn_rcpf((float)den) * 4294966784.0f);  // 0x4f7ffffe
        z += __umulhi(z, (0u - den) * z);
    }
    __device__ __forceinline__ uint32_t mod(uint32_t x) const
    {
        uint32_t r = x - __umulhi(x, z) * d;
        if (r >= d) r -= d;
        if (r >= d) r -= d;
        return r;
    }
    // The walker pools run one r % d per rejection trial and are bound by instruction issue: with the reciprocal
    // made EXACT, z = floor((2^32 - 1) / d), one correction is enough for the 31-bit draws they reduce:
    //   delta = 2^32/d - z is in (0, 1], so x*z/2^32 = x/d - x*delta/2^32 > x/d - 1/2 for x < 2^31, i.e. the quotient
    //   estimate is floor(x/d) or one less, never more.
    // init() leaves z at most two below the exact value (never above: the estimate is built to underestimate), which
    // trw_selftest_hip checks for every divisor a row can have.
    __device__ __forceinline__ void init_exact(uint32_t den)
    {
        init(den);
        uint32_t rem = 0xFFFFFFFFu - z * den;  // z * den <= 2^32 - 1 because z <= floor((2^32 - 1) / den)
        if (rem >= den) {
            ++z;
            rem -= den;
        }
        if (rem >= den) ++z;
    }
    __device__ __forceinline__ uint32_t mod31(uint32_t x) const  // x < 2^31, z from init_exact
    {
        const uint32_t r = x - __umulhi(x, z) * d;
        const uint32_t r1 = r - d;
        return r1 < r ? r1 : r;  // (r >= d  <=>  r - d does not wrap)
    }
};

// Self-test of the arithmetic above (trw_selftest_hip): every divisor in [d_lo, d_hi) -- is init_exact's reciprocal
// exact, did init() stay within two below it, does mod31 agree with % on the corner draws and a few hashed ones.
__global__ void __launch_bounds__(kBlock)
fastmod_selftest_kernel(uint32_t d_lo, uint32_t d_hi, unsigned long long *__restrict__ failures)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    uint32_t bad = 0;
    for (uint64_t d64 = (uint64_t)d_lo + (uint64_t)blockIdx.x * kBlock + threadIdx.x; d64 < d_hi; d64 += stride) {
        const uint32_t d = (uint32_t)d64;
        FastMod a, b;
        a.init(d);
        b.init_exact(d);
        const uint32_t exact = 0xFFFFFFFFu / d;
        if (b.z != exact || a.z > exact || exact - a.z > 2u) ++bad;
        const uint32_t top = 0x7FFFFFFFu;
        uint32_t h = d * 0x9E3779B1u;
        const uint32_t xs[10] = {0u, d - 1u, d & top, top, top - (d - 1u), (top / d) * d, (top / d) * d - 1u,
                                 (h ^ (h >> 15)) & top, (h * 0x85EBCA77u) & top, ((h + 0x165667B1u) * 0xC2B2AE3Du) & top};
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const uint32_t x = xs[k] & top;
            if (b.mod31(x) != x % d || a.mod(x) != x % d) ++bad;
        }
    }
    if (bad) atomicAdd(failures, (unsigned long long)bad);
}

// First offset in [0, n) whose entry is >= x (n if none); rows sorted ascending.
__device__ __forceinline__ uint32_t lower_bound32(const uint32_t *__restrict__ col32, uint32_t start, uint32_t n,
                                                  uint32_t x, uint32_t &probes)
{
    uint32_t lo = 0;
    while (n > 0) {
        const uint32_t half = n >> 1;
        const uint32_t v = col32[start + lo + half];
        ++probes;
        if (v < x) {
            lo += half + 1;
            n -= half + 1;
        } else {
            n = half;
        }
    }
    return lo;
}

// is_neighbor(x, t) (rw_cpu.cpp:32-56): does t's row contain x?
template <bool kSorted>
__device__ __forceinline__ bool row_contains32(const uint32_t *__restrict__ col32, uint32_t start, uint32_t deg,
                                               uint32_t x, uint32_t &probes)
{
    if (kSorted) {
        const uint32_t lo = lower_bound32(col32, start, deg, x, probes);
        if (lo >= deg) return false;
        ++probes;
        return col32[start + lo] == x;
    }
    for (uint32_t e = 0; e < deg; ++e) {
        ++probes;
        if (col32[start + e] == x) return true;
    }
    return false;
}

struct BiasedParams {
    uint32_t t0, t1, t2;  // accept iff draw < t_k, k = distance class of the candidate (0: back to t,
                          // 1: neighbour of t, 2: neither) -- the reference's fp64 compares as integers
    uint32_t lo, hi;      // min(t1,t2), max(t1,t2)
};

// One walker per lane.  Draw 0 picks the first neighbour; trial k of the
// rejection loop consumes draws 2k+1 (candidate index) and 2k+2 (acceptance),
// both always and in that order (rw_cpu.cpp:168-169), so the outcome of every
// trial is fixed by the stream -- but most trials can be DECIDED without
// reading the graph:
//   * draw >= hi: no candidate other than t itself could be accepted.  With
//     sorted rows the offsets at which t sits in prev's row are known per step
//     (one bisection when the step starts, kPos), so the trial is decided from
//     the candidate's offset alone: no load.
//   * lo <= draw < hi is the only band in which is_neighbor() matters; below lo
//     every candidate is accepted, so the membership test is skipped there.
// Lanes advance through their own steps independently (a lane that accepts
// early starts its next step at once); columns go to stage[s*n_walkers + i].
template <bool kSorted, bool kPos, bool kCount>
__device__ __forceinline__ void biased_walk_lane(const uint32_t *__restrict__ row32,
                                                 const uint32_t *__restrict__ col32, uint32_t target, int64_t i,
                                                 int64_t n_walkers, uint64_t gw, int32_t L, uint32_t seed,
                                                 const BiasedParams &bp, RowCombiner &row, uint32_t &st,
                                                 uint32_t &n_trials, uint32_t &n_fetch, uint32_t &n_tests,
                                                 uint32_t &n_probes)
{
    uint32_t blk[4];
    uint32_t bi = 0;
    walker_block(gw, bi, seed, blk);

    // step 1: uniform neighbour of the start node (rw_cpu.cpp:159)
    uint32_t t = target, t_start, t_deg;
    load_row32(row32, t, t_start, t_deg);
    uint32_t prev = t;
    if (t_deg == 0) st |= TRW_ST_DEGREE0;
    else prev = col32[t_start + (blk[0] >> 1) % t_deg];
    row.emit(prev, 1, L);

    uint32_t p_start, p_deg;
    load_row32(row32, prev, p_start, p_deg);
    FastMod fm;
    uint32_t pos_lo = 0, pos_hi = 0;  // offsets of t inside prev's row: [pos_lo, pos_hi)
    auto enter_step = [&]() {
        if (p_deg == 0) return;
        fm.init(p_deg);
        if (kPos) {
            pos_lo = lower_bound32(col32, p_start, p_deg, t, n_probes);
            pos_hi = pos_lo;
            while (pos_hi < p_deg && (++n_probes, col32[p_start + pos_hi] == t)) ++pos_hi;
        }
    };
    enter_step();

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
        if (kCount) ++n_trials;

        bool accept;
        uint32_t x;
        if (p_deg == 0) {  // dead end: reported, the walker stays (the reference divides by zero)
            st |= TRW_ST_DEGREE0;
            x = prev;
            accept = true;
        } else {
            const uint32_t idx = fm.mod(rc);
            if (kPos && ru >= bp.hi) {
                x = t;
                accept = ru < bp.t0 && idx >= pos_lo && idx < pos_hi;
            } else {
                x = col32[p_start + idx];
                if (kCount) ++n_fetch;
                if (x == t) accept = ru < bp.t0;
                else if (ru < bp.lo) accept = true;
                else if (ru >= bp.hi) accept = false;
                else {
                    if (kCount) ++n_tests;
                    accept = ru < (row_contains32<kSorted>(col32, t_start, t_deg, x, n_probes) ? bp.t1 : bp.t2);
                }
            }
        }
        if (!accept && ++rejected >= kTrialCap) {
            st |= (uint32_t)kStTrialCap;
            if (kPos && ru >= bp.hi) x = col32[p_start + fm.mod(rc)];
            accept = true;
        }
        if (accept) {
            row.emit(x, s, L);
            t = prev;
            t_start = p_start;
            t_deg = p_deg;
            prev = x;
            ++s;
            rejected = 0;
            if (s <= L) {
                load_row32(row32, prev, p_start, p_deg);
                enter_step();
            }
        }
    }
}

template <bool kPos, bool kCount>
__global__ void __launch_bounds__(kBlock)
biased_walker_kernel(const uint32_t *__restrict__ row32, const uint32_t *__restrict__ col32,
                     const int64_t *__restrict__ targets, int64_t n_nodes, int64_t n_walkers, int64_t walker_base,
                     int32_t L, uint32_t seed, BiasedParams bp, int64_t *__restrict__ out,
                     int64_t *__restrict__ status)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i == 0) {
        status[2] = TRW_RNG_WALKER;
        status[3] = 1;
    }
    const bool sorted = status[kStDescents] == status[kStBoundary];
    uint32_t st = 0;
    uint32_t n_trials = 0, n_fetch = 0, n_tests = 0, n_probes = 0;
    if (i < n_walkers) {
        const int64_t tg = targets[i];
        RowCombiner row = {out + i * (int64_t)(L + 1), 0u, 0u, 0u, 0u};
        if ((uint64_t)tg >= (uint64_t)n_nodes) {
            st |= TRW_ST_NODE_RANGE;  // reported; the row is filled with the bad id
            for (int32_t c = 0; c <= L; ++c) row.sp[c] = tg;
        } else {
            row.emit((uint32_t)tg, 0, L);  // column 0
            const uint64_t gw = (uint64_t)walker_base + (uint64_t)i;
            if (sorted)
                biased_walk_lane<true, kPos, kCount>(row32, col32, (uint32_t)tg, i, n_walkers, gw, L, seed, bp, row,
                                                     st, n_trials, n_fetch, n_tests, n_probes);
            else
                biased_walk_lane<false, false, kCount>(row32, col32, (uint32_t)tg, i, n_walkers, gw, L, seed, bp,
                                                       row, st, n_trials, n_fetch, n_tests, n_probes);
        }
    }
    if (st) atomicOr((unsigned long long *)&status[0], (unsigned long long)st);
    if (kCount) {  // whole waves reach this point together
        const uint64_t a = wave_sum(n_trials), b = wave_sum(n_fetch), c = wave_sum(n_tests), d = wave_sum(n_probes);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd((unsigned long long *)&status[4], (unsigned long long)a);
            atomicAdd((unsigned long long *)&status[5], (unsigned long long)b);
            atomicAdd((unsigned long long *)&status[6], (unsigned long long)c);
            atomicAdd((unsigned long long *)&status[7], (unsigned long long)d);
        }
    }
}


// ===========================================================================
// Biased walk, large batches: edge hash table + per-lane state machine
// ===========================================================================
// Profiles of biased_walker_kernel (profiles/r01_*): ~40 dependent gathers per
// step (two bisections of ~10 probes each dominate), VALU lanes 19 % busy because
// a wave runs every branch of the trial for whichever lanes need it, and the
// whole wave idles behind its slowest lane.  This path removes all three:
//   * edge_table_build_kernel hashes every edge (u,v) once per call into an
//     open-addressing table {u, v, first offset of v in u's row, multiplicity}.
//     "is x a neighbour of t" and "where does t sit in prev's row" become ONE
//     16-byte gather each (expected 1.2 probes at load factor <= 0.5) instead of
//     a bisection.  It is an exact set, so every answer is the reference's.
//   * biased_hash_kernel runs each lane as a small state machine that issues at
//     most ONE 16-byte gather per loop iteration, whatever it is for (row
//     offsets, hash slot, candidate entry): all lanes of a wave keep a request
//     in flight every iteration instead of taking turns through nested loops.
//   * lanes that finish a walk fetch the next walker index from a device
//     counter, so no lane waits for the slowest walk of its wave.
// Results are identical to biased_walker_kernel (same stream, same decisions).

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

constexpr int kStNextWalker = 10;  // status slot: next unassigned walker (dynamic distribution)

__device__ __forceinline__ uint32_t edge_hash(uint32_t u, uint32_t v)
{
    uint32_t h = u * 0x9E3779B1u ^ (v * 0x85EBCA77u + 0x165667B1u);
    h ^= h >> 16;
    h *= 0x7FEB352Du;
    h ^= h >> 15;
    h *= 0x846CA68Bu;
    h ^= h >> 16;
    return h;
}

// The edge set (round 4: 16 bytes per edge where the table of {u, v, first offset, multiplicity} slots took 78).
// Keys of 8 bytes {u, v} (u == 0xFFFFFFFF: free) in BLOCKS of four -- 32 bytes, an aligned half of a cache line, what a
// probe reads anyway -- `nb` blocks, any number (2.5 places per edge by default: load 0.4; TRW_HASH_SLACK tenths of a
// place per edge, 12 at least; the block of an edge is mulhi(hash, nb), no power of two needed).  A key lives in the first free place of its home block or of the blocks
// after it, so places fill in order: a lookup that meets a free place has seen everything its key could be behind.
// Walkers only ever ask "is (a, b) an edge"; WHERE b sits in a's row -- the prepare pass wants it for rpos and for
// the masks, the walk for the rare run of parallel edges that does not fit rpos -- is a bisection of the sorted row
// on a hit (row_find32).
__device__ __forceinline__ uint32_t edge_block(uint32_t u, uint32_t v, uint32_t nb) { return __umulhi(edge_hash(u, v), nb); }

// One 16-byte half of a block, g = {u0, v0, u1, v1}: is (a, b) one of the two keys / is the chain over here?
template <class V4>
__device__ __forceinline__ bool half_has(const V4 &g, uint32_t a, uint32_t b)
{
    return (g.x == a && g.y == b) || (g.z == a && g.w == b);
}
template <class V4>
__device__ __forceinline__ bool half_ends(const V4 &g) { return g.z == 0xFFFFFFFFu; }  // (places fill in order)

// table: nb blocks of four keys, all bytes 0xFF when empty.
// Each thread owns 8 consecutive entries: one bisection of row32 finds the row of the first,
// the others follow by walking row32 forward.  With sorted rows (the shadow pass counted no
// descent inside a row) equal neighbours are adjacent and only the first entry of a run is inserted;
// otherwise every entry is, and a parallel edge finds its key there already.
__global__ void __launch_bounds__(kBlock)
edge_table_build_kernel(const uint32_t *__restrict__ row32, const uint32_t *__restrict__ col32, int64_t n_nodes,
                        int64_t nnz, uint32_t *__restrict__ table, uint32_t nb,
                        const int64_t *__restrict__ status)
{
    constexpr int kPer = 8;
    const bool sorted = status[kStDescents] == status[kStBoundary];
    const uint32_t e_first = row32[0], e_last = row32[n_nodes];  // entries outside belong to no row: not edges
    const int64_t chunks = (nnz + kPer - 1) / kPer;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x; c < chunks; c += stride) {
        const uint32_t e0 = (uint32_t)(c * kPer);
        // the row that holds entry e0: last u with row32[u] <= e0
        uint32_t u = 0, n = (uint32_t)n_nodes;
        while (n > 1) {
            const uint32_t half = n >> 1;
            if (row32[u + half] <= e0) {
                u += half;
                n -= half;
            } else {
                n = half;
            }
        }
        uint32_t row_lo = row32[u], row_hi = row32[u + 1];
        for (int k = 0; k < kPer; ++k) {
            const uint32_t e = e0 + k;
            if (e >= (uint32_t)nnz) break;
            while (e >= row_hi && u + 1 < (uint32_t)n_nodes) {  // next non-empty row
                ++u;
                row_lo = row_hi;
                row_hi = row32[u + 1];
            }
            if (e < e_first || e >= e_last) continue;
            const uint32_t v = col32[e];
            if (sorted && e > row_lo && col32[e - 1] == v) continue;  // not the first of its run: one key per distinct edge
            const unsigned long long key = ((unsigned long long)v << 32) | u;
            unsigned long long *keys = (unsigned long long *)table;
            uint32_t b = edge_block(u, v, nb);
            for (bool placed = false; !placed;) {
#pragma unroll
                for (int j = 0; j < 4 && !placed; ++j) {
                    const unsigned long long old = atomicCAS(keys + 4 * (size_t)b + j, ~0ull, key);
                    placed = old == ~0ull || old == key;  // (old == key: a parallel edge of an unsorted row)
                }
                b = b + 1u == nb ? 0u : b + 1u;
            }
        }
    }
}

// Fat entries for the biased walk, 32 bytes per CSR entry e = (u -> v):
//     word 0..3   v, start(v), deg(v) | kMaskIsFilter, rpos
//     word 4..7   a 128-bit neighbourhood mask of the step that follows u -> v
// A walker that took u -> v stands on prev = v with t = u and draws candidates from v's row by OFFSET
// (idx = draw % deg(v)).  Everything the rejection trial has to know about offset idx is here:
//   * rpos says where u itself sits in v's row ((multiplicity << 28) | first offset when multiplicity <= 15
//     and offset < 2^28; 0 = u is not in v's row; 1 = look it up in the edge table) -- the "x == t" case;
//   * mask bit (idx & 127) says whether the node at offset idx of v's row is in u's row -- is_neighbor(x, t)
//     of the reference (rw_cpu.cpp:32-56), precomputed.  For deg(v) <= 128 the bit is the exact answer.
//     For longer rows offsets alias modulo 128 and the bit is the OR over its aliases: a clear bit still
//     means "not a neighbour" for certain, a set bit means "ask the edge table" (kMaskIsFilter set).
// So a trial is decided from registers, and the graph is read once per STEP: the 32-byte entry of the
// accepted offset, which carries the next step's row, rpos and mask.
//
// Cost of the mask: the common neighbours of u and v.  Who works them out for which entry -- the thread that owns
// the entry, or the tile pass -- is in trw_kernels_marks.hpp; beyond `heavy_cap` entries in the shorter row, and for
// graphs whose rows are not sorted, the mask is all ones + filter (every band trial asks the edge set, as the walk
// did before there were masks).
constexpr uint32_t kRposAbsent = 0u, kRposLookup = 1u;
constexpr uint32_t kMaskIsFilter = 0x80000000u;  // top bit of the deg word (rows have fewer than 2^31 entries)
constexpr uint32_t kNoCommon = kMaskIsFilter;    // (name used by the unsorted-rows kernel when it strips the bit)
constexpr int kStHeavyCount = 11;                // status slot of the PREPARED header: entries queued for the heavy pass

// is (a -> b) an edge?  (one probe chain of the edge set, a block -- both halves, one line -- at a time)
__device__ __forceinline__ bool edge_table_find(const uint32_t *__restrict__ table, uint32_t nb, uint32_t a, uint32_t b)
{
    uint32_t blk = edge_block(a, b, nb);
    while (true) {
        const u32x4_a16 g0 = *reinterpret_cast<const u32x4_a16 *>(table + 8 * (size_t)blk);
        const u32x4_a16 g1 = *reinterpret_cast<const u32x4_a16 *>(table + 8 * (size_t)blk + 4);
        if (half_has(g0, a, b) || half_has(g1, a, b)) return true;
        if (half_ends(g0) || half_ends(g1)) return false;
        blk = blk + 1u == nb ? 0u : blk + 1u;
    }
}

struct Mask128 {
    uint32_t w[4];
    __device__ __forceinline__ void clear() { w[0] = w[1] = w[2] = w[3] = 0u; }
    __device__ __forceinline__ void fill() { w[0] = w[1] = w[2] = w[3] = 0xFFFFFFFFu; }
    __device__ __forceinline__ void set(uint32_t off)
    {
        const uint32_t b = 1u << (off & 31u), k = (off >> 5) & 3u;
        w[0] |= k == 0u ? b : 0u;
        w[1] |= k == 1u ? b : 0u;
        w[2] |= k == 2u ? b : 0u;
        w[3] |= k == 3u ? b : 0u;
    }
    __device__ __forceinline__ void set_run(uint32_t lo, uint32_t cnt)
    {
        if (cnt >= 128u) {
            fill();
            return;
        }
        for (uint32_t k = 0; k < cnt; ++k) set(lo + k);
    }
};

// First offset in sorted row [start, start+n) whose entry is >= x (n if none); `run`: how many entries equal x.
__device__ __forceinline__ uint32_t row_find32(const uint32_t *__restrict__ col32, uint32_t start, uint32_t n, uint32_t x,
                                               uint32_t &run)
{
    uint32_t lo = 0, len = n;
    while (len > 0) {
        const uint32_t half = len >> 1;
        if (col32[start + lo + half] < x) {
            lo += half + 1;
            len -= half + 1;
        } else {
            len = half;
        }
    }
    run = 0;
    while (lo + run < n && col32[start + lo + run] == x) ++run;
    return lo;
}

// row_find32 for the prepare pass's hits (a common neighbour / a reverse edge found in the edge set: 6 M of them on
// R-MAT 10M/110M, 29 per wave of the entries kernel -- and a lane that bisects holds its whole wave for ten dependent
// round trips): FOUR pivots per round, loaded together, so a round trip divides the range by five -- four rounds
// for a row of a thousand entries instead of ten.
__device__ __forceinline__ uint32_t row_find32_wide(const uint32_t *__restrict__ col32, uint32_t start, uint32_t n, uint32_t x,
                                                    uint32_t &run)
{
    uint32_t lo = 0, len = n;  // the answer (first offset whose entry is >= x) lies in [lo, lo + len]
    while (len > 8u) {
        const uint32_t step = len / 5u;
        const uint32_t *p = col32 + start + lo - 1u;
        const uint32_t a = p[step], b = p[2u * step], c = p[3u * step], d = p[4u * step];  // (independent loads: one round trip)
        const uint32_t k = (a < x ? 1u : 0u) + (b < x ? 1u : 0u) + (c < x ? 1u : 0u) + (d < x ? 1u : 0u);
        lo += k * step;            // entries lo .. lo + k*step - 1 are < x
        len = k == 4u ? len - 4u * step : step - 1u;  // (k < 4: the entry at lo + step - 1 is >= x)
    }
    uint32_t v[8];
#pragma unroll
    for (uint32_t i = 0; i < 8u; ++i) v[i] = i < len ? col32[start + lo + i] : 0xFFFFFFFFu;
    uint32_t k = 0;
#pragma unroll
    for (uint32_t i = 0; i < 8u; ++i) k += v[i] < x ? 1u : 0u;
    lo += k;
    run = 0;
    while (lo + run < n && col32[start + lo + run] == x) ++run;
    return lo;
}

// The two rare detours of the walker pool that read memory on the spot -- a run of parallel edges too long for rpos, a
// probe chain longer than the slot's counter -- as REAL calls: inlined, their loads shared registers with the pool's
// gathers in flight, and the compiler put a full `s_waitcnt vmcnt(0)` in front of the issue pass's loads of every
// iteration (walk 36.3 -> 40.4 ms on R-MAT 10M/110M).
// (Both return their two results packed in one uint64: through a pointer the caller's variable lived in scratch memory,
// and the private segment was set up for every wave of the kernel for it -- 16 bytes a lane, round 4.)
__device__ __attribute__((noinline)) uint64_t row_find32_cold(const uint32_t *col32, uint32_t start, uint32_t n, uint32_t x)
{
    uint32_t run;
    const uint32_t lo = row_find32(col32, start, n, x, run);
    return ((uint64_t)run << 32) | lo;  // first offset | how many entries equal x
}
__device__ __attribute__((noinline)) uint64_t edge_chain_cold(const uint32_t *table, uint32_t nb, uint32_t a, uint32_t b, uint32_t skip)
{
    uint32_t blk = edge_block(a, b, nb), probes = 0;
    for (uint32_t k = 0; k < skip; ++k) blk = blk + 1u == nb ? 0u : blk + 1u;
    while (true) {
        const u32x4_a16 g0 = *reinterpret_cast<const u32x4_a16 *>(table + 8 * (size_t)blk);
        const u32x4_a16 g1 = *reinterpret_cast<const u32x4_a16 *>(table + 8 * (size_t)blk + 4);
        ++probes;
        if (half_has(g0, a, b) || half_has(g1, a, b)) return ((uint64_t)probes << 32) | 1u;  // found | blocks looked at
        if (half_ends(g0) || half_ends(g1)) return (uint64_t)probes << 32;
        blk = blk + 1u == nb ? 0u : blk + 1u;
    }
}

// Mask of the step after u -> v: offsets of v's row [v_start, v_start+v_deg) whose node is in u's row
// [u_lo, u_hi), by walking the shorter row with one probe of the edge set per entry: independent gathers, what a single
// thread does best (the owning thread's way for short rows; measured against bisections into the longer row: 36 vs 62 ms
// for the whole pass with rows up to 32, round 2).
#ifdef TRW_PREP_COUNT
static __device__ unsigned long long g_cnt_hits_u = 0, g_cnt_rpos = 0, g_cnt_probes_u = 0;  // (debug counters)
#endif
__device__ __forceinline__ void common_mask_table(const uint32_t *__restrict__ col32, const uint32_t *__restrict__ table,
                                                  uint32_t nb, uint32_t u, uint32_t u_lo, uint32_t u_hi, uint32_t v,
                                                  uint32_t v_start, uint32_t v_deg, Mask128 &m)
{
    const uint32_t u_deg = u_hi - u_lo;
    if (u_deg <= v_deg) {  // walk u's row: each distinct w in it, looked up in v's row
        for (uint32_t i = 0; i < u_deg; ++i) {
            const uint32_t w = col32[u_lo + i];
            if (i > 0 && col32[u_lo + i - 1] == w) continue;  // sorted rows: a run of equal entries is one node
#ifdef TRW_PREP_COUNT
            atomicAdd(&g_cnt_probes_u, 1ull);
#endif
            if (edge_table_find(table, nb, v, w)) {  // a common neighbour: where it sits in v's row
#ifdef TRW_PREP_COUNT
                atomicAdd(&g_cnt_hits_u, 1ull);
#endif
                uint32_t run;
                const uint32_t lo = row_find32_wide(col32, v_start, v_deg, w, run);
                m.set_run(lo, run);
            }
        }
    } else {  // walk v's row: the node at each offset, looked up in u's row
        for (uint32_t j = 0; j < v_deg; ++j)
            if (edge_table_find(table, nb, u, col32[v_start + j])) m.set(j);
    }
}

struct HeavyItem {
    uint32_t e, u;
};

// ---------------------------------------------------------------------------
// The tile pass (kernels: trw_kernels_marks.hpp): what its tasks and in-lists look like
// ---------------------------------------------------------------------------
// The mask of u -> v is the intersection of two sorted rows, written as positions of v's row.  The longer row is kept
// in LDS (a tile) and the shorter ones are streamed past it; the entries that share a longer row share the tile:
//   * longer row = u's: u's entries are consecutive in the CSR.  A task is (u, 128 or 64 of its entries); v's row is
//     streamed (coalesced), element j found in the tile sets bit j.
//   * longer row = v's: the entries that point AT v are scattered, so they are collected first: the entries kernel
//     counts them per v, the plan kernel hands every v a segment of the item array, the scatter kernel fills the
//     segments with (e, u).  A task is (v, 64 items of its in-list); u's row is streamed, element w found at position
//     lo of the tile sets bits lo .. lo+run-1 (run > 1: parallel edges).
#ifndef TRW_TASK_ENTRIES
#define TRW_TASK_ENTRIES 128
#endif
#ifndef TRW_TASK_ITEMS
#define TRW_TASK_ITEMS 64
#endif
constexpr uint32_t kTaskEntries = TRW_TASK_ENTRIES;   // long tile rows: entries of the tile row per task
constexpr uint32_t kTaskItems = TRW_TASK_ITEMS;      // in-list items per task
constexpr int kStTaskCount = 16, kStTaskNext = 18;  // header slots
constexpr uint32_t kNoSegment = 0xFFFFFFFFu;

struct TileTask {  // 8 bytes: what else a task needs (how many, of how many) the tile kernel has from row32 / the in-list arrays
    uint32_t node_mode;  // node | mode << 31; mode 0: up to kTaskEntries entries of node's row from `start` on,
    uint32_t start;      //                     mode 1: up to kTaskItems items of its in-list from items[start] on
};

// exclusive prefix sum of v over the wave's lanes; total = the sum
__device__ __forceinline__ uint32_t wave_prefix_excl(uint32_t v, int lane, uint32_t &total)
{
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = (uint32_t)__shfl_up((int)inc, d);
        if (lane >= d) inc += y;
    }
    total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);  // (a scalar: what is decided from it stays a scalar branch)
    return inc - v;
}

constexpr uint32_t kTileSentinel = 0x7FFFFFFFu;  // padding of a tile: above every node id (n_nodes < 2^31)
typedef __attribute__((address_space(3))) uint32_t lds_u32;

enum : uint32_t { LS_IDLE = 0, LS_START, LS_ROW, LS_POS, LS_TRIAL, LS_CAND, LS_MEMBER, LS_ACCEPT };

constexpr int kRing = 16;  // draws buffered per lane in LDS (4 Philox blocks)
// Rejection trials a lane may run back to back in one iteration while they need no memory.  The body of
// the iteration is executed by the wave for the union of its lanes' states (~250 VALU instructions), so
// an iteration spent on a memory-free trial alone is expensive.
#ifndef TRW_TRIALS_PER_ITER
#define TRW_TRIALS_PER_ITER 4
#endif
constexpr int kTrialsPerIter = TRW_TRIALS_PER_ITER;
// A state's block (probe, candidate, trials, accept) is skipped while fewer than TRW_VOTE lanes of the wave are
// waiting for it; every (TRW_VOTE_TICK+1)-th iteration runs everything.  Each pass through a block then serves
// more lanes: the wave executes the union of its lanes' blocks, and instruction issue -- not memory -- is what
// this kernel runs out of (measured on R-MAT 10M/110M: 128.7 ms with 0, 123-124 ms for 8..16 lanes and a tick
// of 4..16, 128.8 ms again at 32).  Same walks either way: a lane's state only waits.
#ifndef TRW_VOTE
#define TRW_VOTE 12
#endif
#ifndef TRW_VOTE_TICK
#define TRW_VOTE_TICK 7
#endif

template <bool kCount, class OutT = int64_t>
__global__ void __launch_bounds__(kBlock)
biased_hash_kernel(const uint32_t *__restrict__ row32, const uint32_t *__restrict__ col32,
                   const uint32_t *__restrict__ ext, const uint32_t *__restrict__ table, uint32_t tnb, const int64_t *__restrict__ targets,
                   int64_t n_nodes, int64_t n_walkers, int64_t walker_base, int32_t L, uint32_t seed,
                   BiasedParams bp, int32_t want_pos, int32_t only_unsorted, OutT *__restrict__ out,
                   int64_t *__restrict__ status)
{
    // Sorted graphs are walked by biased_mask_kernel; the host cannot see which kind it has without a
    // synchronisation, so it launches both and the one that does not apply returns here.
    if (only_unsorted && status[kStDescents] == status[kStBoundary]) return;
    // Per-lane ring of upcoming draws, draw d of the lane's walker at ring[d % kRing][tid] (slot-major: a
    // lane always hits its own bank).  The ring is topped up for ALL lanes of the wave at once whenever one
    // lane runs dry, so the Philox rounds execute with (nearly) every lane active instead of for the few
    // lanes that happen to need a block in a given iteration.
    __shared__ uint32_t ring[kRing * kBlock];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        status[2] = TRW_RNG_WALKER;
        status[3] = 2;
    }
    // Offsets of t inside prev's row: the edge set answers "is it an edge", not "where" (that was the 16-byte slots'
    // job until round 4), so this kernel -- the walk over UNSORTED rows, where a node's parallel edges are not one run
    // anyway -- always looks at its candidate.  (LS_POS and the code under use_pos stay for the day it is wanted.)
    const bool use_pos = false;
    (void)want_pos;
    const int lane = threadIdx.x & 63;
    uint32_t *my_ring = ring + threadIdx.x;
    // candidates come from the fat entries (32 bytes each; this kernel reads the first 16: {v, start(v), deg(v),
    // rpos}) when the call built them: the accepted candidate then brings its own row and the LS_ROW gather of
    // the next step is skipped
    const bool fat = ext != nullptr;
    const uint32_t *cand_base = fat ? ext : col32;
    const uint32_t cand_shift = fat ? 3u : 0u;

    uint32_t st = 0;
    uint32_t n_trials = 0, n_fetch = 0, n_tests = 0, n_probes = 0;

    uint32_t state = LS_START;
    int64_t walker = 0;
    uint64_t gw = 0;
    int32_t s = 0;
    uint32_t t = 0, prev = 0, t_start = 0, t_deg = 0, p_start = 0, p_deg = 0;
    uint32_t pos_lo = 0, pos_cnt = 0;
    FastMod fm;
    fm.d = 1;
    fm.z = 0;
    uint32_t d_cons = 0, d_prod = 0;  // draws consumed / produced for the current walker
    uint64_t slot = 0;
    uint32_t ku = 0, kv = 0, ru = 0, rejected = 0;
    uint32_t c_start = 0, c_deg = 0, c_rpos = kRposLookup;  // the candidate's row and where prev sits in it (fat entries)
    bool no_common = false;  // rows of t and prev share no node (kNoCommon of the entry the walker came through)
    const uint32_t *addr = row32;
    u32x4_a4 g = {0u, 0u, 0u, 0u};  // the lane's gather: issued when it enters a gather state, consumed by that state's block
    bool fresh = false;             // a gather state was entered (or its probe moved on): issue the load
    uint32_t acc_cand = 0, iter = 0;
    bool acc_known = false;
    RowCombinerT<OutT> row = {out, 0u, 0u, 0u, 0u};  // the caller's row, four columns per store

    while (true) {
        // ---- hand out walkers to the lanes that need one (one atomic per wave) ----
        const unsigned long long need = __ballot(state == LS_START);
        if (need) {
            const int leader = (int)__ffsll((long long)need) - 1;
            long long base = 0;
            if (lane == leader)
                base = (long long)atomicAdd((unsigned long long *)&status[kStNextWalker],
                                            (unsigned long long)__popcll(need));
            base = __shfl(base, leader);
            if (state == LS_START) {
                walker = base + __popcll(need & ((1ull << lane) - 1ull));
                if (walker >= n_walkers) {
                    state = LS_IDLE;
                } else {
                    const int64_t tg = targets[walker];
                    // lanes advance at their own pace, so nothing is gained by staging lane-contiguously:
                    // this kernel writes the caller's int64 row directly, four columns at a time (emit)
                    row.sp = out + walker * (int64_t)(L + 1);
                    if ((uint64_t)tg >= (uint64_t)n_nodes) {
                        st |= TRW_ST_NODE_RANGE;  // reported; the row is filled with the bad id
                        for (int32_t c = 0; c <= L; ++c) row.sp[c] = (OutT)tg;
                    } else {
                        row.emit((uint32_t)tg, 0, L);  // column 0
                        gw = (uint64_t)walker_base + (uint64_t)walker;
                        d_cons = 0;
                        d_prod = 0;
                        s = 1;
                        rejected = 0;
                        prev = (uint32_t)tg;  // step 1 leaves the start node
                        t = prev;
                        addr = row32 + prev;
                        fresh = true;
                        state = LS_ROW;
                    }
                }
            }
        }
        if (__ballot(state != LS_IDLE) == 0) break;

        // ---- the one gather of this iteration ----
        if (fresh) {
            g = *reinterpret_cast<const u32x4_a4 *>(addr);
            fresh = false;
        }
        // which blocks run this time
        bool run_probe = true, run_cand = true, run_trial = true, run_acc = true;
        if (TRW_VOTE > 0) {
            const int n_probe = __popcll(__ballot(state == LS_POS || state == LS_MEMBER));
            const int n_cand = __popcll(__ballot(state == LS_CAND));
            const int n_trial = __popcll(__ballot(state == LS_TRIAL));
            const int n_acc = __popcll(__ballot(state == LS_ACCEPT));
            const bool tick = (iter++ & (uint32_t)TRW_VOTE_TICK) == 0u;
            run_probe = tick || n_probe >= TRW_VOTE;
            run_cand = tick || n_cand >= TRW_VOTE;
            run_trial = tick || n_trial >= TRW_VOTE;
            run_acc = tick || n_acc >= TRW_VOTE;
            if (!(run_probe || run_cand || run_trial || run_acc)) run_probe = run_cand = run_trial = run_acc = true;
        }

        // ---- top up the draw rings when some lane is about to run dry ----
        const bool live = state != LS_IDLE && state != LS_START;
        if (__ballot(live && d_prod - d_cons < 2u * kTrialsPerIter)) {
            while (__ballot(live && d_prod - d_cons <= (uint32_t)(kRing - 4))) {
                if (live && d_prod - d_cons <= (uint32_t)(kRing - 4)) {
                    uint32_t blk[4];
                    walker_block(gw, d_prod >> 2, seed, blk);
#pragma unroll
                    for (int k = 0; k < 4; ++k) my_ring[((d_prod + k) & (kRing - 1)) * kBlock] = blk[k];
                    d_prod += 4;
                }
            }
        }

        // ---- consume the gather: every state reduces to "accept cand", "reject", "probe (ku,kv)" or nothing ----
        bool ev_accept = false, ev_known = false, ev_reject = false, ev_probe = false;
        uint32_t cand = 0, probe_state = LS_POS;
        if (state == LS_ROW) {  // g = row32[prev], row32[prev+1], ...
            p_start = g.x;
            p_deg = g.y >= g.x ? g.y - g.x : 0u;
            if (p_deg == 0) {  // dead end: reported, the walker stays (the reference divides by zero)
                st |= TRW_ST_DEGREE0;
                cand = prev;
                ev_accept = true;
            } else if (s == 1) {  // first step: plain uniform neighbour, draw 0 (rw_cpu.cpp:159)
                const uint32_t r0 = my_ring[0] >> 1;
                d_cons = 1;
                addr = cand_base + ((size_t)(p_start + r0 % p_deg) << cand_shift);
                fresh = true;
                ru = 0;
                state = LS_CAND;
            } else {
                fm.init(p_deg);
                if (use_pos) {
                    ku = prev;
                    kv = t;
                    ev_probe = true;
                } else {
                    state = LS_TRIAL;
                }
            }
        } else if (run_probe && (state == LS_POS || state == LS_MEMBER)) {  // g = one half (two keys) of a block of the edge set
            if (kCount) ++n_probes;
            const bool match = half_has(g, ku, kv);
            const bool empty = !match && half_ends(g);
            if (!empty && !match) {
                slot = slot + 1u == 2ull * tnb ? 0u : slot + 1u;
                addr = table + 4 * slot;
                fresh = true;
            } else if (state == LS_POS) {  // where t sits in prev's row
                pos_lo = match ? g.z : 0u;
                pos_cnt = match ? g.w + 1u : 0u;
                state = LS_TRIAL;
            } else {  // is x a neighbour of t?
                cand = kv;
                if (ru < (match ? bp.t1 : bp.t2)) ev_accept = true;
                else ev_reject = true;
            }
        } else if (run_cand && state == LS_CAND) {  // g.x = the candidate node (g.y, g.z = its row when fat)
            cand = g.x;
            c_start = g.y;
            c_deg = g.z;  // fat entries: kNoCommon rides in the top bit, taken off where the row is adopted
            c_rpos = g.w;
            if (s == 1) {
                ev_accept = true;
            } else {
                if (kCount) ++n_fetch;
                if (cand == t) {
                    if (ru < bp.t0) ev_accept = ev_known = true;
                    else ev_reject = true;
                } else if (ru < bp.lo) {
                    ev_accept = true;
                } else if (ru >= bp.hi) {
                    ev_reject = true;
                } else {
                    if (kCount) ++n_tests;
                    ku = t;
                    kv = cand;
                    ev_probe = true;
                    probe_state = LS_MEMBER;
                }
            }
        }
        if (ev_reject) {
            state = LS_TRIAL;
            if (++rejected >= kTrialCap) {
                st |= (uint32_t)kStTrialCap;
                ev_accept = true;
                ev_known = false;
            }
        }

        // ---- one rejection trial for every lane that is between gathers ----
        for (int rep = 0; run_trial && rep < kTrialsPerIter && state == LS_TRIAL && !ev_accept && d_prod - d_cons >= 2u; ++rep) {
            const uint32_t rc = my_ring[(d_cons & (kRing - 1)) * kBlock] >> 1;
            ru = my_ring[((d_cons + 1) & (kRing - 1)) * kBlock] >> 1;
            d_cons += 2;
            if (kCount) ++n_trials;
            const uint32_t idx = fm.mod(rc);
            if (use_pos && (ru >= bp.hi || (no_common && ru >= bp.lo))) {
                // What the candidate is does not matter beyond "is it t" -- t sits at offsets
                // [pos_lo, pos_lo+pos_cnt) -- because is_neighbor cannot change the answer: the draw is above
                // both thresholds, or t's row and prev's share no node, so every other candidate is a non-neighbour.
                const bool at_t = idx - pos_lo < pos_cnt;
                if (at_t && ru < bp.t0) {
                    cand = t;
                    ev_accept = ev_known = true;
                } else if (!at_t && ru < bp.t2) {  // a non-neighbour passes (t2 > t1 only): fetch it, it is accepted
                    addr = cand_base + ((size_t)(p_start + idx) << cand_shift);
                    fresh = true;
                    ru = 0;
                    state = LS_CAND;
                } else if (++rejected >= kTrialCap) {
                    st |= (uint32_t)kStTrialCap;
                    addr = cand_base + ((size_t)(p_start + idx) << cand_shift);  // take whatever this trial drew
                    fresh = true;
                    ru = 0;
                    rejected = 0;
                    state = LS_CAND;
                }
            } else {
                addr = cand_base + ((size_t)(p_start + idx) << cand_shift);
                fresh = true;
                state = LS_CAND;
            }
        }

        // ---- emit an accepted candidate and move to the next step ----
        if (TRW_VOTE > 0) {  // accepts wait for company too
            if (ev_accept) {
                state = LS_ACCEPT;
                acc_cand = cand;
                acc_known = ev_known;
            }
            ev_accept = run_acc && state == LS_ACCEPT;
            cand = acc_cand;
            ev_known = acc_known;
        }
        if (ev_accept) {
            row.emit(cand, s, L);
            ++s;
            rejected = 0;
            if (s > L) {
                state = LS_START;
            } else if (ev_known && t_deg != 0) {  // cand == t: its row is the one we came through
                const uint32_t ts = t_start, td = t_deg, tt = t;
                t_start = p_start;
                t_deg = p_deg;
                p_start = ts;
                p_deg = td;
                t = prev;
                prev = tt;
                fm.init(p_deg);
                if (use_pos) {
                    ku = prev;
                    kv = t;
                    ev_probe = true;
                    probe_state = LS_POS;
                } else {
                    state = LS_TRIAL;
                }
            } else {
                t = prev;
                t_start = p_start;
                t_deg = p_deg;
                prev = cand;
                no_common = false;
                if (fat && (c_deg & ~kNoCommon) != 0 && !(st & TRW_ST_DEGREE0)) {  // the candidate brought its row along
                    p_start = c_start;
                    p_deg = c_deg & ~kNoCommon;
                    no_common = false;  // (the top bit of the deg word marks a filter mask now; stripped above)
                    fm.init(p_deg);
                    if (use_pos && c_rpos == kRposLookup) {
                        ku = prev;
                        kv = t;
                        ev_probe = true;
                        probe_state = LS_POS;
                    } else {  // the entry already said where t sits in prev's row
                        pos_lo = c_rpos & 0x0FFFFFFFu;
                        pos_cnt = c_rpos >> 28;
                        state = LS_TRIAL;
                    }
                } else {
                    addr = row32 + prev;
                    fresh = true;
                    state = LS_ROW;
                }
            }
        }

        // ---- start a table probe ----
        if (ev_probe) {
            slot = 2ull * edge_block(ku, kv, tnb);
            addr = table + 4 * slot;
            fresh = true;
            state = probe_state;
        }
    }

    if (st) atomicOr((unsigned long long *)&status[0], (unsigned long long)st);
    if (kCount) {
        const uint64_t a = wave_sum(n_trials), b = wave_sum(n_fetch), c = wave_sum(n_tests), d = wave_sum(n_probes);
        if (lane == 0) {
            atomicAdd((unsigned long long *)&status[4], (unsigned long long)a);
            atomicAdd((unsigned long long *)&status[5], (unsigned long long)b);
            atomicAdd((unsigned long long *)&status[6], (unsigned long long)c);
            atomicAdd((unsigned long long *)&status[7], (unsigned long long)d);
        }
    }
}

// ===========================================================================
// Biased walk over the 32-byte fat entries: trials decided in registers
// ===========================================================================
// The kernel for sorted graphs with a prepared edge set + fat entries (entries_kernel).  Same stream, same
// decisions, same rows as biased_walker_kernel / biased_hash_kernel -- what changes is where the answers come
// from.  With t = the node two steps back and prev = the node the walker stands on, a rejection trial draws an
// OFFSET idx into prev's row and an acceptance value ru, and the reference (rw_cpu.cpp:166-195) accepts
//     x == t              iff ru < t0
//     x in t's row        iff ru < t1
//     otherwise           iff ru < t2.
// The entry the walker came through tells, per offset: is it t (rpos), is its node in t's row (mask bit).  So:
//     offset holds t                      -> decided (accept iff ru < t0)
//     ru below both / above both of t1,t2 -> decided
//     in between, mask bit clear          -> not a neighbour, decided
//     in between, bit set, exact mask     -> neighbour, decided
//     in between, bit set, filter mask    -> load the entry (the candidate), ask the edge table, then decide
// A lane runs its trials back to back out of its LDS draw ring and touches memory once per step: the 32-byte
// entry of the accepted offset, which is at once the next node, its row, rpos and the next step's mask.  That
// gather is issued from inside the trial burst and collected after it, so its latency hides behind the other
// lanes' trials; lanes take their next walker from a device counter as they finish.
enum : uint32_t { MS_IDLE = 0, MS_START, MS_ROW, MS_CAND, MS_MEMBER, MS_POS, MS_TRIAL };

#ifndef TRW_MASK_PAIRS
#define TRW_MASK_PAIRS 8
#endif
constexpr int kMaskPairs = TRW_MASK_PAIRS;  // Philox blocks (= pairs of trials) per lane between two visits of the gather/adopt code

template <bool kCount>
__global__ void __launch_bounds__(kBlock)
biased_mask_kernel(const uint32_t *__restrict__ row32, const uint32_t *__restrict__ col32, const u32x4_a16 *__restrict__ ext,
                   const uint32_t *__restrict__ table, uint32_t tnb, const int64_t *__restrict__ targets,
                   int64_t n_nodes, int64_t n_walkers, int64_t walker_base, int32_t L, uint32_t seed, BiasedParams bp,
                   int64_t *__restrict__ out, int64_t *__restrict__ status)
{
    // unsorted rows: t's offsets in prev's row are not one run, rpos cannot describe them -- biased_hash_kernel's job
    if (status[kStDescents] != status[kStBoundary]) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        status[2] = TRW_RNG_WALKER;
        status[3] = 4;
    }
    const int lane = threadIdx.x & 63;
    // in the band between the two thresholds: a neighbour is accepted iff t1 is the upper one
    const bool band_takes_near = bp.t1 > bp.t2;

    uint32_t st = 0;
    uint32_t n_trials = 0, n_fetch = 0, n_tests = 0, n_probes = 0;
    uint32_t w_iters = 0, w_reps = 0, w_refills = 0, w_adopts = 0, w_adopt_lanes = 0;  // wave-level (kCount)

    uint32_t state = MS_START;
    int64_t walker = 0;
    uint64_t gw = 0;
    int32_t s = 0;
    uint32_t t = 0, prev = 0, p_start = 0, p_deg = 0;
    uint32_t pos_lo = 0, pos_cnt = 0;
    uint32_t m0 = 0, m1 = 0, m2 = 0, m3 = 0;
    bool exact = false;
    FastMod fm;
    fm.d = 1;
    fm.z = 0;
    // The walker's draws, in registers: draw d is word d % 4 of Philox block d / 4 (>> 1).  Draw 0 takes the first
    // step; trial k uses draws 2k+1 and 2k+2 -- words 1,2 of a block when k is even, word 3 and word 0 of the NEXT
    // block when k is odd.  The lane keeps words 1..3 of block `blk_at` and whether its next trial is an odd one.
    uint32_t w1 = 0, w2 = 0, w3 = 0, blk_at = 0;
    bool odd = false;
    uint64_t slot = 0;
    uint32_t ru = 0, rejected = 0;
    bool taken = false;               // MS_CAND: the entry in flight is an accepted one (else: a candidate to test)
    // The ONE gather a lane has in flight lands in e0 (+ e1 for a 32-byte entry): a row32 pair, a table slot or
    // a fat entry.  One load site per iteration keeps the compiler's wait to one place (it orders every load
    // after whatever else might still be writing the same registers).
    u32x4_a4 e0 = {0u, 0u, 0u, 0u}, e1 = {0u, 0u, 0u, 0u};
    u32x4_a4 c0 = {0u, 0u, 0u, 0u}, c1 = {0u, 0u, 0u, 0u};  // the candidate entry, kept while the table is asked about it
    // what the lane wants fetched at the end of this iteration: 0 nothing, 1 the 32-byte entry `want_at`,
    // 2 the row32 pair of node `want_at`, 3 table slot `slot`
    uint32_t want = 0, want_at = 0;
    RowCombiner row = {out, 0u, 0u, 0u, 0u};
    // verdict of a band trial by (mask bit, exact): 0 reject, 1 accept, 2 load the candidate and ask the table
    const uint32_t v_clear = band_takes_near ? 0u : 1u, v_set_exact = band_takes_near ? 1u : 0u;

    while (true) {
        // ---- collect what the previous iteration asked for (the one wait of the loop) ----
        bool adopt = false;
        if (state == MS_ROW) {  // e0 = row32[start node], row32[start node + 1], ...
            p_start = e0.x;
            p_deg = e0.y >= e0.x ? e0.y - e0.x : 0u;
            if (p_deg == 0) {  // dead end at the start node: reported, the walker stays (the reference divides by zero)
                st |= TRW_ST_DEGREE0;
                for (; s <= L; ++s) row.emit(prev, s, L);
                state = MS_START;
            } else {  // first step: plain uniform neighbour, draw 0 (rw_cpu.cpp:159)
                uint32_t blk[4];
                walker_block(gw, 0u, seed, blk);
                const uint32_t r0 = blk[0] >> 1;
                w1 = blk[1];
                w2 = blk[2];
                w3 = blk[3];
                blk_at = 0u;
                odd = false;
                want = 1u;
                want_at = p_start + r0 % p_deg;
                if (kCount) ++n_fetch;
                taken = true;
                state = MS_CAND;
            }
        } else if (state == MS_CAND) {  // e0, e1 = the entry at the offset the trial drew
            c0 = e0;
            c1 = e1;
            if (taken) {
                adopt = true;
            } else {  // a candidate from a row whose mask is only a filter: is it in t's row?
                if (kCount) ++n_tests;
                slot = 2ull * edge_block(t, c0.x, tnb);
                want = 3u;
                state = MS_MEMBER;
            }
        } else if (state == MS_MEMBER) {  // e0 = one half (two keys) of a block of the edge set
            if (kCount) ++n_probes;
            const bool match = half_has(e0, t, c0.x);
            const bool empty = !match && half_ends(e0);
            if (!empty && !match) {
                slot = slot + 1u == 2ull * tnb ? 0u : slot + 1u;
                want = 3u;
            } else if (ru < (match ? bp.t1 : bp.t2)) {
                adopt = true;
            } else {
                state = MS_TRIAL;
                if (++rejected >= kTrialCap) {
                    st |= (uint32_t)kStTrialCap;
                    adopt = true;
                }
            }
        }
        if (kCount) {
            ++w_iters;
            const unsigned long long ab = __ballot(adopt);
            if (ab) ++w_adopts;
            w_adopt_lanes += (uint32_t)__popcll(ab);
        }
        if (adopt) {  // step onto the entry in c0/c1
            const uint32_t x = c0.x;
            row.emit(x, s, L);
            ++s;
            rejected = 0;
            t = prev;
            prev = x;
            p_start = c0.y;
            p_deg = c0.z & ~kMaskIsFilter;
            if (s > L) {
                state = MS_START;
            } else if (p_deg == 0) {  // dead end: reported, the walker stays
                st |= TRW_ST_DEGREE0;
                for (; s <= L; ++s) row.emit(prev, s, L);
                state = MS_START;
            } else {
                exact = (c0.z & kMaskIsFilter) == 0u;
                m0 = c1.x;
                m1 = c1.y;
                m2 = c1.z;
                m3 = c1.w;
                fm.init(p_deg);
                pos_lo = c0.w & 0x0FFFFFFFu;
                pos_cnt = c0.w >> 28;
                state = MS_TRIAL;
                if (c0.w == kRposLookup)  // too many parallel edges / too far into the row for the packed form: found here
                    pos_lo = row_find32(col32, p_start, p_deg, t, pos_cnt);
            }
        }

        // ---- hand out walkers to the lanes that need one (one atomic per wave) ----
        const unsigned long long need = __ballot(state == MS_START);
        if (need) {
            const int leader = (int)__ffsll((long long)need) - 1;
            long long base = 0;
            if (lane == leader)
                base = (long long)atomicAdd((unsigned long long *)&status[kStNextWalker],
                                            (unsigned long long)__popcll(need));
            base = __shfl(base, leader);
            if (state == MS_START) {
                walker = base + __popcll(need & ((1ull << lane) - 1ull));
                if (walker >= n_walkers) {
                    state = MS_IDLE;
                } else {
                    const int64_t tg = targets[walker];
                    row.sp = out + walker * (int64_t)(L + 1);
                    if ((uint64_t)tg >= (uint64_t)n_nodes) {
                        st |= TRW_ST_NODE_RANGE;  // reported; the row is filled with the bad id
                        for (int32_t c = 0; c <= L; ++c) row.sp[c] = tg;
                    } else {
                        row.emit((uint32_t)tg, 0, L);  // column 0
                        gw = (uint64_t)walker_base + (uint64_t)walker;
                        s = 1;
                        rejected = 0;
                        prev = (uint32_t)tg;  // step 1 leaves the start node
                        t = prev;
                        want = 2u;
                        want_at = prev;
                        state = MS_ROW;
                    }
                }
            }
        }
        if (__ballot(state != MS_IDLE) == 0) break;

        // ---- the trial burst: memory-free trials back to back, two per Philox block; a lane leaves at its first
        //      accepted (or to-be-tested) offset ----
        auto trial = [&](uint32_t rc, uint32_t ru_now) {
            ru = ru_now;
            if (kCount) ++n_trials;
            const uint32_t idx = fm.mod(rc);
            const uint32_t k = (idx >> 5) & 3u;
            const uint32_t word = k == 0u ? m0 : (k == 1u ? m1 : (k == 2u ? m2 : m3));
            const bool bit = (word >> (idx & 31u)) & 1u;
            const uint32_t v_band = bit ? (exact ? v_set_exact : 2u) : v_clear;
            const uint32_t v_other = ru_now < bp.lo ? 1u : (ru_now >= bp.hi ? 0u : v_band);
            uint32_t verdict = (idx - pos_lo < pos_cnt) ? (ru_now < bp.t0 ? 1u : 0u) : v_other;
            rejected = verdict == 0u ? rejected + 1u : rejected;
            if (rejected >= kTrialCap) {  // take whatever this trial drew
                st |= (uint32_t)kStTrialCap;
                verdict = 1u;
            }
            if (verdict != 0u) {
                want = 1u;
                want_at = p_start + idx;
                if (kCount) ++n_fetch;
                taken = verdict == 1u;
                state = MS_CAND;
            }
        };
#pragma unroll 1
        for (int rep = 0; rep < kMaskPairs; ++rep) {
            if (__ballot(state == MS_TRIAL) == 0) break;
            if (kCount) ++w_reps;
            if (state == MS_TRIAL) {
                uint32_t nb[4];
                walker_block(gw, blk_at + 1u, seed, nb);
                // first trial of the pair: an even one reads this block, an odd one straddles into the next
                const bool was_odd = odd;
                trial((was_odd ? w3 : w1) >> 1, (was_odd ? nb[0] : w2) >> 1);
                if (was_odd) {  // moved on to the next block (now an even trial on it)
                    w1 = nb[1];
                    w2 = nb[2];
                    w3 = nb[3];
                    ++blk_at;
                }
                odd = !was_odd;
                if (state == MS_TRIAL) {  // second trial of the pair
                    trial((was_odd ? w1 : w3) >> 1, (was_odd ? w2 : nb[0]) >> 1);
                    if (!was_odd) {
                        w1 = nb[1];
                        w2 = nb[2];
                        w3 = nb[3];
                        ++blk_at;
                    }
                    odd = was_odd;
                }
            }
        }

        // ---- the gather of this iteration, collected at the top of the next ----
        if (want != 0u) {
            const uint32_t *addr = want == 1u ? reinterpret_cast<const uint32_t *>(ext + 2 * (size_t)want_at)
                                              : (want == 2u ? row32 + want_at : table + 4 * slot);
            e0 = *reinterpret_cast<const u32x4_a4 *>(addr);
            if (want == 1u) e1 = *reinterpret_cast<const u32x4_a4 *>(addr + 4);
        }
        want = 0u;
    }

    if (st) atomicOr((unsigned long long *)&status[0], (unsigned long long)st);
    if (kCount) {
        const uint64_t a = wave_sum(n_trials), b = wave_sum(n_fetch), c = wave_sum(n_tests), d = wave_sum(n_probes);
        if (lane == 0) {
            atomicAdd((unsigned long long *)&status[4], (unsigned long long)a);
            atomicAdd((unsigned long long *)&status[5], (unsigned long long)b);
            atomicAdd((unsigned long long *)&status[6], (unsigned long long)c);
            atomicAdd((unsigned long long *)&status[7], (unsigned long long)d);
            // how the wave spent its passes: iterations, burst repetitions, refill passes, iterations with an adopt,
            // lanes adopting -- lane utilisation of each block = lane events / (passes x 64)
            atomicAdd((unsigned long long *)&status[12], (unsigned long long)w_iters);
            atomicAdd((unsigned long long *)&status[13], (unsigned long long)w_reps);
            atomicAdd((unsigned long long *)&status[14], (unsigned long long)w_refills);
            atomicAdd((unsigned long long *)&status[15], ((unsigned long long)w_adopts << 40) | (unsigned long long)w_adopt_lanes);
        }
    }
}

// ===========================================================================
// Biased walk, walker pool: walker state in LDS, lanes as executors (ballot / prefix-sum compaction)
// ===========================================================================
// biased_mask_kernel keeps one walker per lane, and a wave executes the union of what its lanes are doing: the
// counters say 35 of 64 lanes are active per instruction -- the Philox rounds and the trials run with the lanes
// that happen to be between gathers (45 %), the step-onto-the-next-entry code with the handful of lanes that
// accepted in the last burst (17 %).  Here a wave owns a POOL of kPoolSlots walkers whose state lives in LDS
// (struct of arrays, one word per field and slot) and three queues of slot numbers, and every phase of the
// loop takes a FULL wave of work items from its queue (ballot + mbcnt prefix sums compact the pushes):
//     adopt   the lanes that issued a gather in the previous iteration consume it for their slot: step onto the
//             entry (emit the column, write the next step's row / rpos / mask into the slot) -> trial queue;
//             first-step row, edge-table probes likewise
//     refill  free slots take new walkers from the wave's reserve (64 claimed from the device counter at a time,
//             start rows and first steps prefetched) -> issue queue
//     issue   pop up to 64 slots that want a gather, form the address, load; the data is collected by `adopt`
//             in the NEXT iteration, after the burst below -- the wave hides its own memory latency
//     burst   pop up to 64 slots that are between gathers, load their trial state, run up to 2*kPoolPairs
//             rejection trials each in registers (two per Philox block), write back; lanes with an accepted (or
//             to-be-tested) offset push their slot to the issue queue, the others back to the trial queue
// A phase waits until its queue holds most of a wave (or nothing else can run), so its instructions execute
// with most lanes active whatever the individual walkers are doing.  Same stream, same decisions, same rows as
// biased_mask_kernel.
// 1: columns leave the pool in aligned groups of FOUR (see the adopt pass) instead of pairs: one more word per slot, so
// 122-slot pools for the same four workgroups per CU (320 bytes a slot + 1 600: 40 640 of the 40 960 a workgroup may have).
// L2 misses per step 1.97 -> 1.71, bytes written to HBM 13.3 -> 7.0 GB, HBM traffic 87.8 -> 80.8 GB
// (profiles/r05_pool_quad_stores.txt).  While the burst's Philox blocks were 40 v_xor_b32 each this was 0.5 ms SLOWER (the
// kernel was bound by its vector instructions alone); with v_bitop3_b32 doing two of them at once it is 0.65 ms faster.
#ifndef TRW_POOL_QUAD_STORES
#define TRW_POOL_QUAD_STORES 1
#endif
#ifndef TRW_POOL_SLOTS
#define TRW_POOL_SLOTS (TRW_POOL_QUAD_STORES ? 122 : 128)
#endif
// Pairs of trials per visit of the burst: 2 while the pools stood a sixth empty (2 / 3: 41.7 / 41.7 ms), 3 since they
// are kept full (2 / 3 / 4: 38.3 / 37.6 / 38.6 ms).
#ifndef TRW_POOL_PAIRS
#define TRW_POOL_PAIRS 3
#endif
constexpr int kPoolSlots = TRW_POOL_SLOTS;  // walkers resident per wave (<= 128; about 2 per lane keeps every phase fed)
constexpr int kQueueCap = TRW_POOL_SLOTS > 128 ? 256 : 128;  // ring capacity of the slot queues (a power of two >= kPoolSlots)
constexpr int kPoolPairs = TRW_POOL_PAIRS;  // Philox blocks (pairs of trials) per visit of the burst
constexpr int kPoolWaves = kBlock / 64;
// A phase runs when its queue holds at least this many slots (or when nothing else can make progress)
// Wave priority by phase: high while a wave consumes / issues gathers (so its requests get out early), low in
// the arithmetic of the burst.  Measured 45.7 ms against 46.2 (the reverse: 46.9).  0: off.
#ifndef TRW_POOL_PRIO
#define TRW_POOL_PRIO 1
#endif
// The burst's pairs as straight-line code (no "does any lane still run" vote between them): the loop's exits cost
// about 40 scalar instructions per pair in lane-mask merges; 42.7 -> 41.8 ms.
#ifndef TRW_POOL_UNROLL
#define TRW_POOL_UNROLL 1
#endif
#ifndef TRW_POOL_ISSUE_MIN
#define TRW_POOL_ISSUE_MIN 48
#endif
#ifndef TRW_POOL_TRIAL_MIN
#define TRW_POOL_TRIAL_MIN 48
#endif
// Free slots are refilled when this many have gathered (a refill pass is an atomic on the device counter and one Philox
// block for the pass).
#ifndef TRW_POOL_PAIR_STORES
#define TRW_POOL_PAIR_STORES 1
#endif
#ifndef TRW_POOL_REFILL_MIN
#define TRW_POOL_REFILL_MIN 8
#endif
constexpr uint32_t kPoolIssueMin = TRW_POOL_ISSUE_MIN, kPoolTrialMin = TRW_POOL_TRIAL_MIN, kPoolRefillMin = TRW_POOL_REFILL_MIN;
enum : uint32_t { PK_ROW = 0, PK_ENTRY = 1, PK_TEST = 2, PK_MEMBER = 3 };  // what a slot's gather is for
// kPfExact: the slot's mask bits are exact answers (a filter otherwise); kPfOdd: the stream stands at word 3 of the
// current block; kPfWide: t's run in prev's row does not fit PF_RPOS
constexpr uint32_t kPfExact = 1u, kPfOdd = 2u, kPfWide = 8u;
constexpr uint32_t kPfKind = 7u << 4, kPfRejShift = 8, kPfRejMask = 0x1FFFFFu << 8, kPfProbeShift = 29, kPfProbe = 7u << 29;
static_assert(kTrialCap <= (kPfRejMask >> kPfRejShift), "the rejection count must fit its field");

struct PoolQueue {  // a ring of slot numbers (bytes) in LDS; head/count are wave-uniform
    uint8_t *q;
    uint32_t head, count;
    // pops up to 64 entries: lane i < n gets one
    __device__ __forceinline__ uint32_t pop(int lane, uint32_t &n, uint32_t &slot)
    {
        n = count < 64u ? count : 64u;
        slot = q[(head + (uint32_t)lane) & (uint32_t)(kQueueCap - 1)];
        head = (head + n) & (uint32_t)(kQueueCap - 1);
        count -= n;
        return n;
    }
    // lanes with `yes` append their slot (compacted: ballot + prefix count)
    __device__ __forceinline__ void push(bool yes, uint32_t slot)
    {
        const unsigned long long b = __ballot(yes);
        if (b == 0) return;
        const uint32_t off = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
        if (yes) q[(head + count + off) & (uint32_t)(kQueueCap - 1)] = (uint8_t)slot;
        count += (uint32_t)__popcll(b);
    }
};

__device__ __forceinline__ void pool_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The burst (round 3).  The round-2 form of this kernel was bound by the NUMBER of vector instructions
// (profiles/r02: SQ_INSTS_VALU x ~4.3 cycles, scripts/alu_probe*.hip, = all of its SIMD time), 80 % of them the
// burst's trial pairs: 144 per Philox block, 58 for the block and 86 for two trials and their bookkeeping.  Now 116:
//   * a pair of trials is straight-line code over four draws x0..x3 picked by the stream's parity once (the
//     second trial is evaluated for every lane; it simply does not count when the first one finished the step) --
//     no exec-mask regions, no moves under them;
//   * the verdict is arithmetic on two per-slot thresholds: with bit = "offset idx may hold a neighbour of t",
//         thrA = in_run ? t0 : (bit ? A : t2)      accept            iff  ru <  thrA
//         thrR = in_run ? t0 : (bit ? R : t2)      look at candidate iff  thrA <= ru < thrR
//     A = R = t1 for an exact mask, (A, R) = (lo, hi) for a filter: one v_bfe_i32 turns the mask bit into a lane
//     mask, one v_bfi per threshold, one select for the run of t;
//   * the 128-bit mask of a slot is 16 contiguous bytes, the word is read at base + ((idx >> 3) & 12); a slot whose
//     run of t does not fit RPOS ("wide") points its base at sixteen 0xFF bytes instead of testing a flag per trial;
//   * the first step is worked out where new walkers are taken on -- 64 at a time, in the wave's reserve (see `refill`) --
//     and a new walker enters the pool with its first entry gather queued (it was a special case inside every trial);
//   * rejections are counted per pair, the runaway cap is looked at once per pair;
//   * the Philox key schedule lives in vector registers (philox4x32_10_keyed).
// 20 words per walker (16 + the 4 mask words): with 122 walkers per wave that is 9.8 KB of LDS, 40.6 KB per workgroup,
// FOUR workgroups per CU (160 KB) -- 22 words fitted three, and both the waves per SIMD and the walkers per wave pay.
enum : int {
    PF_T = 0, PF_PREV, PF_START, PF_DEG, PF_Z,
    PF_RPOS,    // where t sits in prev's row, packed as in the entry: multiplicity << 28 | first offset (0: nowhere)
    PF_FLAGS,   // bits 0..3 kPf*, bits 4..6 PK_* of the queued / in-flight gather, bits 8..28 rejections at this step,
                // bits 29..31 how far the probe chain of the current table lookup has gone
    PF_S, PF_WALKER,
    PF_W1, PF_W2, PF_W3, PF_BLK,  // words 1..3 of the walker's current Philox block, and its number
    PF_X,       // the candidate node a table probe asks about
    PF_AT,      // fat entry index to gather
#if TRW_POOL_QUAD_STORES
    PF_T2,      // the node before t: the fourth column of a 32-byte store
#endif
    PF_COUNT
};

template <bool kCount, class OutT = int64_t>
__global__ void __launch_bounds__(kBlock)
biased_pool_kernel(const uint32_t *__restrict__ row32, const uint32_t *__restrict__ col32,
                   const u32x4_a16 *__restrict__ ext, const uint32_t *__restrict__ table, uint32_t tnb,
                   const int64_t *__restrict__ targets, int64_t n_nodes, int64_t n_walkers, int64_t walker_base, int32_t L,
                   uint32_t seed, BiasedParams bp, OutT *__restrict__ out, int64_t *__restrict__ status)
{
    if (status[kStDescents] != status[kStBoundary]) return;  // unsorted rows: biased_hash_kernel's job
    __shared__ uint32_t pool_mem[kPoolWaves][PF_COUNT][kPoolSlots];
    __shared__ u32x4_a16 mask_mem[kPoolWaves][kPoolSlots + 1];  // [kPoolSlots]: all ones, the mask of a "wide" slot
    __shared__ uint8_t queue_mem[kPoolWaves][3][kQueueCap];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        status[2] = TRW_RNG_WALKER;
        status[3] = 5;
    }
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t(*P)[kPoolSlots] = pool_mem[wv];
    u32x4_a16 *M = mask_mem[wv];
    PoolQueue trial_q = {queue_mem[wv][0], 0u, 0u}, issue_q = {queue_mem[wv][1], 0u, 0u}, free_q = {queue_mem[wv][2], 0u, 0u};
    for (int i = lane; i < kPoolSlots; i += 64) free_q.q[i] = (uint8_t)i;
    free_q.count = kPoolSlots;
    if (lane == 0) M[kPoolSlots] = u32x4_a16{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    pool_fence();
    const bool band_takes_near = bp.t1 > bp.t2;
    const int32_t cols = L + 1;
    // the Philox key schedule and the threshold of a return to t, in vector registers (see philox4x32_10_keyed)
    PhiloxKeysV key_r;
    philox_keys_v(seed, key_r);
    uint32_t t0_v = bp.t0, t2_v = bp.t2;
    asm volatile("" : "+v"(t0_v), "+v"(t2_v));

    uint32_t st = 0;
    uint32_t n_trials = 0, n_fetch = 0, n_tests = 0, n_probes = 0;
    bool hold = false;
    uint32_t hs = 0, hkind = 0;
    u32x4_a4 e0 = {0u, 0u, 0u, 0u}, e1 = {0u, 0u, 0u, 0u};
    bool more_walkers = true;
    // the reserve of new walkers (see `refill`): state 0 empty / 1 claimed / 2 start nodes asked for / 3 rows asked for / 4 ready
    int rs_state = 0;
    uint32_t rs_cnt = 0, rs_next = 0, rs_have = 0, rs_at = 0, rs_tg_lo = 0, rs_tg_hi = 0;
    long long rs_base = 0;
    int64_t rs_walker0 = 0, rs_target = 0;
    u32x2_a4 rs_row = {0u, 0u};
    uint32_t rs_blk[4] = {0u, 0u, 0u, 0u};
    bool rs_bad = false, rs_dead = false;
    uint64_t w_iters = 0, w_pass[3] = {0, 0, 0}, w_lanes[3] = {0, 0, 0}, w_cycles[4] = {0, 0, 0, 0}, w_trial_passes = 0;
    uint64_t t_mark = kCount ? clock64() : 0;
    auto lap = [&](int phase) {
        if (kCount) {
            const uint64_t now = clock64();
            w_cycles[phase] += now - t_mark;
            t_mark = now;
        }
    };
    auto row_of = [&](uint32_t slot) { return out + (int64_t)P[PF_WALKER][slot] * cols; };

    while (true) {
        // ================= adopt: consume the gathers issued in the previous iteration =================
#if TRW_POOL_PRIO
        __builtin_amdgcn_s_setprio(TRW_POOL_PRIO == 1 ? 3 : 0);
#endif
        {
            bool to_trial = false, to_issue = false, to_free = false;
            if (kCount) {
                const unsigned long long hb = __ballot(hold);
                ++w_iters;
                if (hb) {
                    ++w_pass[0];
                    w_lanes[0] += (uint64_t)__popcll(hb);
                }
            }
            if (hold) {
                const uint32_t slot = hs;
                // ONE wait for the gather, here, where the compiler can see it: left to itself it sank the wait into the
                // branches below -- one of them BEHIND the column store, a store's round trip per iteration (adopt 21 ->
                // 34 % of the iteration when the branches were touched in round 4; the edge-list pool had the same)
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
                if (hkind == PK_ENTRY) {  // e0, e1 = the accepted entry: step onto it
                    const uint32_t x = e0.x;
                    int32_t s = (int32_t)P[PF_S][slot];
                    const uint32_t prev = P[PF_PREV][slot];
                    OutT *r = row_of(slot);
#if TRW_POOL_QUAD_STORES
                    // Columns leave in 32-byte ALIGNED groups of four (one HBM write sector, written once): column s with
                    // the three before it -- the nodes the slot keeps anyway, PF_PREV, PF_T, and PF_T2 for this -- when
                    // column s is the last of its group, not at all otherwise.  A row's ends are ragged (rows are
                    // 8-byte aligned, 81 columns): the head of a row and its last group go out column by column.
                    const uint32_t cm2 = P[PF_T][slot], cm3 = P[PF_T2][slot];
                    const uint32_t a = (uint32_t)(((uintptr_t)(r + s) / sizeof(OutT)) & 3u);
#ifndef TRW_DBG_NO_STORES
                    if (a == 3u && s >= 3) {
                        if (sizeof(OutT) == 8) {
                            *reinterpret_cast<i64x2_a16 *>(r + s - 3) = i64x2_a16{(int64_t)cm3, (int64_t)cm2};
                            *reinterpret_cast<i64x2_a16 *>(r + s - 1) = i64x2_a16{(int64_t)prev, (int64_t)x};
                        } else {
                            *reinterpret_cast<u32x4_a16 *>(r + s - 3) = u32x4_a16{cm3, cm2, prev, x};
                        }
                    } else if (a == 3u || s == L) {
                        r[s] = (OutT)x;
                        if (a >= 1u && s >= 1) r[s - 1] = (OutT)prev;
                        if (a >= 2u && s >= 2) r[s - 2] = (OutT)cm2;
                        if (a >= 3u && s >= 3) r[s - 3] = (OutT)cm3;
                    }
#endif
                    P[PF_T2][slot] = cm2;
#elif TRW_POOL_PAIR_STORES
                    // Columns leave in 16-byte ALIGNED pairs: column s together with column s-1 (the node being left, in
                    // the slot anyway) when its address is the upper half of a pair, not at all when it is the lower half
                    // (the next step stores it) -- half the store requests of a kernel that runs at the memory system's
                    // request rate.  (Round 2 paired by step parity: rows are 8-byte aligned, every other walker's pairs
                    // straddled and were split again.)
                    // (uint32 rows -- the multi-GPU gather's wire format, straight from the walk: pairs of 8 bytes)
                    if (((uintptr_t)(r + s) / sizeof(OutT)) & 1u) {
                        if (sizeof(OutT) == 8)
                            *reinterpret_cast<i64x2_a16 *>(r + s - 1) = i64x2_a16{(int64_t)prev, (int64_t)x};
                        else
                            *reinterpret_cast<u32x2_a8 *>(r + s - 1) = u32x2_a8{prev, x};
                    } else if (s == L) {
                        r[s] = (OutT)x;  // (the last column has no partner to wait for)
                    }
#else
                    r[s] = (OutT)x;
#endif
                    ++s;
                    const uint32_t deg = e0.z & ~kMaskIsFilter;
                    if (s > L) {
                        to_free = true;
                    } else if (deg == 0) {  // dead end: reported, the walker stays
                        st |= TRW_ST_DEGREE0;
#if TRW_POOL_QUAD_STORES
                        if (a != 3u) {  // (the columns of x's group that were waiting for it)
                            if (a >= 1u && s >= 2) r[s - 2] = (OutT)prev;
                            if (a >= 2u && s >= 3) r[s - 3] = (OutT)cm2;
                        }
#endif
                        for (int32_t c = s - 1; c <= L; ++c) r[c] = (OutT)x;
                        to_free = true;
                    } else {
                        FastMod fm;
                        fm.init_exact(deg);
                        P[PF_T][slot] = prev;
                        P[PF_PREV][slot] = x;
                        P[PF_START][slot] = e0.y;
                        P[PF_DEG][slot] = deg;
                        P[PF_Z][slot] = fm.z;
                        M[slot] = u32x4_a16{e1.x, e1.y, e1.z, e1.w};
                        P[PF_S][slot] = (uint32_t)s;
                        P[PF_RPOS][slot] = e0.w;
                        const uint32_t keep = P[PF_FLAGS][slot] & kPfOdd;  // (rejections and probe count start at 0)
                        const uint32_t ex = (e0.z & kMaskIsFilter) ? 0u : kPfExact;
                        if (e0.w == kRposLookup) {
                            // t's run in x's row has too many parallel edges / starts too far in for the packed form:
                            // found here, by bisection (rare).  It goes where two mask words were; the mask then says
                            // "look" for every offset.
                            const uint64_t found = row_find32_cold(col32, e0.y, deg, prev);
                            const uint32_t lo = (uint32_t)found, cnt = (uint32_t)(found >> 32);
                            u32x4_a16 mm = M[slot];
                            mm.z = lo;
                            mm.w = cnt;
                            M[slot] = mm;
                            P[PF_FLAGS][slot] = keep | kPfWide;
                        } else {
                            P[PF_FLAGS][slot] = keep | ex;
                        }
                        to_trial = true;
                    }
                } else if (hkind == PK_TEST) {  // e0.x = a candidate whose mask bit is only a filter: ask the table
                    if (kCount) ++n_tests;
                    P[PF_X][slot] = e0.x;
                    P[PF_FLAGS][slot] = (P[PF_FLAGS][slot] & ~(kPfKind | kPfProbe)) | (PK_MEMBER << 4);
                    to_issue = true;
                } else {  // PK_MEMBER: e0, e1 = a block (four keys) of the edge set -- is (t, x) an edge?
                    if (kCount) ++n_probes;
                    const uint32_t ku = P[PF_T][slot], kv = P[PF_X][slot];
                    uint32_t flags = P[PF_FLAGS][slot];
                    bool match = half_has(e0, ku, kv) || half_has(e1, ku, kv);
                    bool empty = !match && (half_ends(e0) || half_ends(e1));
                    if (!empty && !match && (flags & kPfProbe) == kPfProbe) {
                        // the chain is longer than the probe field counts: this lane walks the rest of it here
                        const uint64_t chain = edge_chain_cold(table, tnb, ku, kv, 8u);
                        match = (uint32_t)chain != 0u;
                        empty = !match;
                        if (kCount) n_probes += (uint32_t)(chain >> 32);
                    }
                    if (!empty && !match) {
                        P[PF_FLAGS][slot] = flags + (1u << kPfProbeShift);
                        to_issue = true;
                    } else {  // the trial drew inside the band: a neighbour is accepted iff t1 is the upper threshold
                        bool accept = match == band_takes_near;
                        flags &= ~(kPfKind | kPfProbe);
                        if (!accept) {
                            flags += 1u << kPfRejShift;
                            if (((flags & kPfRejMask) >> kPfRejShift) >= kTrialCap) {
                                st |= (uint32_t)kStTrialCap;
                                accept = true;
                            }
                        }
                        if (accept) {  // gather the entry again, this time to step onto it
                            P[PF_FLAGS][slot] = flags | (PK_ENTRY << 4);
                            if (kCount) ++n_fetch;
                            to_issue = true;
                        } else {
                            P[PF_FLAGS][slot] = flags;
                            to_trial = true;
                        }
                    }
                }
            }
            trial_q.push(to_trial, hs);
            issue_q.push(to_issue, hs);
            free_q.push(to_free, hs);
            hold = false;
            pool_fence();
        }
        lap(0);

        // ================= refill: new walkers into free slots =================
        // New walkers come out of a RESERVE of up to 64 that the wave keeps in registers, one per lane: claimed from the
        // device counter 64 at a time, their start nodes, start rows, first Philox block and first step worked out a
        // stage per iteration, every load long on its way before it is used.  (Claiming them when slots fell free made
        // the wave sit through an atomic and a load, ~20 000 cycles a pass, so a pass was only worth it for 32 free
        // slots at once and 16 slots of a pool stood empty on average: bursts of 58.7 lanes instead of 62.)
        if (rs_state == 1) {  // the claim has returned: which walkers, if any
            const long long base = __shfl(rs_base, 0);
            int64_t left = n_walkers - (int64_t)base;
            left = left < 0 ? 0 : (left > 64 ? 64 : left);
            if (left < 64) more_walkers = false;  // the counter has run out: no further claim
            rs_cnt = (uint32_t)left;
            rs_walker0 = (int64_t)base;
            if ((uint32_t)lane < rs_cnt) rs_target = targets[base + lane];
            rs_state = rs_cnt ? 2 : 0;
        } else if (rs_state == 2) {  // the start nodes have arrived: their rows
            rs_bad = (uint32_t)lane < rs_cnt && (uint64_t)rs_target >= (uint64_t)n_nodes;
            rs_row = u32x2_a4{0u, 0u};
            if ((uint32_t)lane < rs_cnt && !rs_bad) rs_row = *reinterpret_cast<const u32x2_a4 *>(row32 + (uint32_t)rs_target);
            // block 0 of every reserved walker's stream, 64 lanes at once: word 0 is the first step's draw
            const uint64_t gw0 = (uint64_t)walker_base + (uint64_t)rs_walker0 + (uint64_t)lane;
            philox4x32_10_keyed((uint32_t)gw0, (uint32_t)(gw0 >> 32), 0u, 0u, key_r, rs_blk);
            rs_state = 3;
        } else if (rs_state == 3) {  // the rows have arrived: the first step (a plain uniform neighbour, rw_cpu.cpp:159)
            const uint32_t deg = rs_row.y >= rs_row.x ? rs_row.y - rs_row.x : 0u;
            FastMod fm;
            fm.init_exact(deg ? deg : 1u);
            rs_at = rs_row.x + fm.mod31(rs_blk[0] >> 1);
            rs_dead = deg == 0u;
            // The hand-out below must not look at a register a LOAD wrote: the compiler cannot know that load finished
            // many iterations ago and would wait for every memory operation in flight (this iteration's column stores
            // included, ~20 000 cycles) in front of the first use.  Real moves, here, where the loads have been waited for.
            asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3"
                         : "=&v"(rs_tg_lo), "=&v"(rs_tg_hi)
                         : "v"((uint32_t)rs_target), "v"((uint32_t)((uint64_t)rs_target >> 32)));
            rs_next = 0u;
            rs_have = rs_cnt;
            rs_state = 4;
        }
        if (rs_state == 4 && free_q.count > 0u && (free_q.count >= kPoolRefillMin || trial_q.count + issue_q.count < 64u)) {
            uint32_t n, slot;
            uint32_t take = free_q.count < rs_have ? free_q.count : rs_have;  // (rs_have <= 64)
            const uint32_t keep_count = free_q.count;
            free_q.pop(lane, n, slot);
            if (take < n) {  // hand the unused ones back (they are the last ones popped)
                free_q.head = (free_q.head + (uint32_t)kQueueCap - (n - take)) & (uint32_t)(kQueueCap - 1);
                free_q.count = keep_count - take;
            }
            // lane i < take serves reserve entry rs_next + i
            const int src = (int)((rs_next + (uint32_t)lane) & 63u);
            const int64_t tg = (int64_t)(((uint64_t)(uint32_t)__shfl((int)rs_tg_hi, src) << 32) |
                                         (uint64_t)(uint32_t)__shfl((int)rs_tg_lo, src));
            const uint32_t b1 = (uint32_t)__shfl((int)rs_blk[1], src), b2 = (uint32_t)__shfl((int)rs_blk[2], src),
                           b3 = (uint32_t)__shfl((int)rs_blk[3], src), at0 = (uint32_t)__shfl((int)rs_at, src);
            const uint32_t why = (uint32_t)__shfl((int)((rs_bad ? 1u : 0u) | (rs_dead ? 2u : 0u)), src);
            bool to_issue = false, back = false;
            if ((uint32_t)lane < take) {
                const int64_t walker = rs_walker0 + (int64_t)src;
                OutT *rowp = out + walker * cols;
                if (why) {  // reported; a bad id fills its row, a dead end at the start node stays there
                    st |= (why & 1u) ? TRW_ST_NODE_RANGE : TRW_ST_DEGREE0;
                    for (int32_t c = 0; c <= L; ++c) rowp[c] = (OutT)tg;
                    back = true;
                } else {
                    P[PF_WALKER][slot] = (uint32_t)walker;
                    rowp[0] = (OutT)tg;  // column 0
                    P[PF_PREV][slot] = (uint32_t)tg;  // step 1 leaves the start node
                    P[PF_T][slot] = (uint32_t)tg;
                    P[PF_S][slot] = 1u;
                    P[PF_AT][slot] = at0;
                    P[PF_W1][slot] = b1;
                    P[PF_W2][slot] = b2;
                    P[PF_W3][slot] = b3;
                    P[PF_BLK][slot] = 0u;
                    P[PF_FLAGS][slot] = PK_ENTRY << 4;
                    if (kCount) ++n_fetch;
                    to_issue = true;
                }
            }
            rs_next += take;
            rs_have -= take;
            if (rs_have == 0u) rs_state = 0;
            issue_q.push(to_issue, slot);
            free_q.push(back, slot);
            pool_fence();
        }
        if (rs_state == 0 && more_walkers) {  // the reserve is used up: claim the next 64 (looked at next time round)
            if (lane == 0) rs_base = (long long)atomicAdd((unsigned long long *)&status[kStNextWalker], 64ull);
            rs_state = 1;
        }
        lap(1);
        // every walker of this wave is done: none in the pool, none in the reserve, none to claim
        if (!more_walkers && rs_state == 0 && free_q.count == (uint32_t)kPoolSlots) break;

        // ================= issue: gathers for the slots that want one =================
        if (issue_q.count >= kPoolIssueMin || (issue_q.count > 0u && trial_q.count < kPoolTrialMin)) {
            uint32_t n, slot;
            issue_q.pop(lane, n, slot);
            if (kCount) {
                ++w_pass[1];
                w_lanes[1] += n;
            }
            if ((uint32_t)lane < n) {
                const uint32_t flags = P[PF_FLAGS][slot];
                const uint32_t kind = (flags & kPfKind) >> 4;
                if (kind == PK_TEST) {
                    // only the candidate's node is wanted: from the compact shadow (4 bytes of a 0.4 GB array that the
                    // L2s and the Infinity Cache hold a good part of) instead of its 32-byte entry (3.5 GB: a miss)
                    e0.x = col32[P[PF_AT][slot]];
                } else {
                    const uint32_t *addr;
                    if (kind == PK_ENTRY) {
                        addr = reinterpret_cast<const uint32_t *>(ext + 2 * (size_t)P[PF_AT][slot]);
                    } else {
                        uint32_t blk = edge_block(P[PF_T][slot], P[PF_X][slot], tnb) + (flags >> kPfProbeShift);  // (PK_MEMBER)
                        blk = blk >= tnb ? blk - tnb : blk;
                        addr = table + 8 * (size_t)blk;  // a block of the edge set: four keys, an aligned half line
                    }
                    e0 = *reinterpret_cast<const u32x4_a4 *>(addr);
                    e1 = *reinterpret_cast<const u32x4_a4 *>(addr + 4);
                }
                hold = true;
                hs = slot;
                hkind = kind;
            }
        }

        lap(2);
#if TRW_POOL_PRIO
        __builtin_amdgcn_s_setprio(TRW_POOL_PRIO == 1 ? 0 : 3);
#endif
        // ================= burst: pairs of trials for a wave of slots that are between gathers =================
        if (trial_q.count >= kPoolTrialMin || (trial_q.count > 0u && issue_q.count < kPoolIssueMin)) {
            uint32_t n, slot;
            trial_q.pop(lane, n, slot);
            if (kCount) {
                ++w_pass[2];
                w_lanes[2] += n;
            }
            const bool busy = (uint32_t)lane < n;
            const uint32_t t_start = P[PF_START][slot];
            const uint32_t deg = busy ? P[PF_DEG][slot] : 1u, z = P[PF_Z][slot];
            const uint32_t flags = P[PF_FLAGS][slot], rpos = P[PF_RPOS][slot];
            const bool exact = (flags & kPfExact) != 0u, wide = (flags & kPfWide) != 0u;
            uint32_t pos_lo = rpos & 0x0FFFFFFFu, pos_cnt = rpos >> 28;
            // the mask of the slot; a wide slot reads sixteen 0xFF bytes and keeps its run of t where mask words 2 and 3
            // would be
            const uint32_t *mask_at = reinterpret_cast<const uint32_t *>(M + (wide ? (uint32_t)kPoolSlots : slot));
            if (wide) {
                const u32x4_a16 mm = M[slot];
                pos_lo = mm.z;
                pos_cnt = mm.w;
            }
            // thresholds of a candidate whose mask bit is set: an exact bit says "neighbour of t" (t1 either way), a
            // filter bit says "maybe": accepted below lo whatever it is, looked at inside [lo, hi)
            const uint32_t thr_a = exact ? bp.t1 : bp.lo, thr_r = exact ? bp.t1 : bp.hi;
            bool odd = (flags & kPfOdd) != 0u;
            uint32_t rejected = (flags & kPfRejMask) >> kPfRejShift;
            uint32_t w1 = P[PF_W1][slot], w2 = P[PF_W2][slot], w3 = P[PF_W3][slot], blk_at = P[PF_BLK][slot];
            const uint64_t gw = (uint64_t)walker_base + (uint64_t)P[PF_WALKER][slot];
            bool take = false, test = false;
            uint32_t at = 0u;

            // one trial: candidate draw word rc_w, acceptance draw word ru_w (the draws are the words >> 1)
            auto trial = [&](uint32_t rc_w, uint32_t ru_w, uint32_t &idx, bool &acc, bool &tst) {
                const uint32_t rc = rc_w >> 1, ru = ru_w >> 1;
                const uint32_t r = rc - __umulhi(rc, z) * deg, r1 = r - deg;
                idx = r1 < r ? r1 : r;  // FastMod::mod31
                const uint32_t word = mask_at[(idx >> 5) & 3u];
                int32_t bitm;  // all ones iff mask bit idx & 127 is set
                asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(bitm) : "v"(word), "v"(idx));
                const uint32_t m = (uint32_t)bitm;
                const uint32_t band_a = select_bits(m, thr_a, t2_v), band_r = select_bits(m, thr_r, t2_v);
                const bool in_run = idx - pos_lo < pos_cnt;
                const uint32_t ta = in_run ? t0_v : band_a, tr = in_run ? t0_v : band_r;
                acc = ru < ta;
                tst = !acc && ru < tr;
            };

            // (trials are not counted one by one: each moves the stream on by half a block, so a visit's trials are the
            // distance between where the stream stood and where it stands -- see the write-back)
            const uint32_t stood = 2u * blk_at + (odd ? 1u : 0u);
            bool running = busy;
#if TRW_POOL_UNROLL
#pragma unroll
#else
#pragma unroll 1
#endif
            for (int rep = 0; rep < kPoolPairs; ++rep) {
#if !TRW_POOL_UNROLL
                if (__ballot(running) == 0) break;
#endif
                if (kCount) ++w_trial_passes;
                if (running) {
                    uint32_t nb[4];
                    philox4x32_10_keyed((uint32_t)gw, (uint32_t)(gw >> 32), blk_at + 1u, 0u, key_r, nb);
                    // the pair's four draws: words 1, 2, 3 of the current block and word 0 of the next one, or -- the
                    // stream stands one trial further on (odd) -- words 3 | 0, 1, 2
                    const uint32_t x0 = odd ? w3 : w1, x1 = odd ? nb[0] : w2, x2 = odd ? nb[1] : w3, x3 = odd ? nb[2] : nb[0];
                    uint32_t idx1, idx2;
                    bool acc1, tst1, acc2, tst2;
                    trial(x0, x1, idx1, acc1, tst1);
                    trial(x2, x3, idx2, acc2, tst2);
                    // (lane masks combined as masks: a select between two bools would go through 0 / 1 integers)
                    const bool f1 = acc1 || tst1;
                    bool fin = f1 || acc2 || tst2;
                    take = acc1 || (!f1 && acc2);
                    const uint32_t idx = f1 ? idx1 : idx2;
                    test = fin && !take;
                    at = t_start + idx;
                    // where the stream stands now: a pair used up moves on one block at the same parity; a step
                    // finished by the pair's first trial flips the parity (and, from even, stays on its block)
                    const bool stay = !odd && f1;
                    w1 = nb[1];
                    w2 = nb[2];
                    w3 = stay ? w3 : nb[3];
                    blk_at += stay ? 0u : 1u;
                    odd = odd != f1;
                    running = !fin;
                }
            }
            // write back what moved
            bool to_issue = false, to_trial = false;
            if (busy) {
                // every trial of the visit but the one that settled the step (if one did) was a rejection
                const uint32_t trials = 2u * blk_at + (odd ? 1u : 0u) - stood;
                rejected += trials - ((take || test) ? 1u : 0u);
                if (kCount) n_trials += trials;
                if (!(take || test) && rejected >= kTrialCap) {  // runaway rejection loop: take what the last trial drew
                    st |= (uint32_t)kStTrialCap;
                    take = true;
                }
                P[PF_W1][slot] = w1;
                P[PF_W2][slot] = w2;
                P[PF_W3][slot] = w3;
                P[PF_BLK][slot] = blk_at;
                uint32_t nf = (flags & (kPfExact | kPfWide)) | (odd ? kPfOdd : 0u) | (rejected << kPfRejShift);
                if (take || test) {
                    if (kCount) ++n_fetch;
                    P[PF_AT][slot] = at;
                    nf |= (take ? PK_ENTRY : PK_TEST) << 4;
                    to_issue = true;
                } else {
                    to_trial = true;
                }
                P[PF_FLAGS][slot] = nf;
            }
            issue_q.push(to_issue, slot);
            trial_q.push(to_trial, slot);
            pool_fence();
        }
        lap(3);
    }

    if (st) atomicOr((unsigned long long *)&status[0], (unsigned long long)st);
    if (kCount) {
        const uint64_t a = wave_sum(n_trials), b = wave_sum(n_fetch), c = wave_sum(n_tests), d = wave_sum(n_probes);
        if (lane == 0) {
            const uint64_t w[12] = {w_iters, w_pass[0], w_lanes[0], w_pass[1], w_lanes[1], w_pass[2], w_lanes[2],
                                    w_trial_passes, w_cycles[0], w_cycles[1], w_cycles[2], w_cycles[3]};
            for (int k = 0; k < 12; ++k) atomicAdd((unsigned long long *)&status[12 + k], (unsigned long long)w[k]);
            atomicAdd((unsigned long long *)&status[4], (unsigned long long)a);
            atomicAdd((unsigned long long *)&status[5], (unsigned long long)b);
            atomicAdd((unsigned long long *)&status[6], (unsigned long long)c);
            atomicAdd((unsigned long long *)&status[7], (unsigned long long)d);
        }
    }
}

// ===========================================================================
// Biased walk, reference-exact: the serial glibc chain replayed by a whole wave
// ===========================================================================
// TRW_RNG_GLIBC on a biased walk is ONE stream consumed walker after walker, trial after trial (rw_cpu.cpp:118,
// :166-195): which draws a trial sees depends on how many trials everything before it took, so walkers cannot
// run side by side.  What CAN run side by side is the work inside one step: the stream's VALUES do not depend on
// the walk, so the wave evaluates the next 64 trials of the current step at once -- lane k looks at draws
// pos+2k, pos+2k+1 and decides its trial from the prepared entry's rpos / mask exactly as biased_mask_kernel
// does -- and the first lane whose trial accepts is the step (everything after it was speculation).  Trials
// whose mask bit is only a filter are resolved for all such lanes in parallel (candidate entry + table probe)
// before the first definite accept is taken.  One dependent gather per step is left: the accepted entry.
// The generator itself is three lanes wide: r[k] = r[k-3] + r[k-31] gives r[k], r[k+1], r[k+2] from values
// that exist already; the raw words go into an LDS ring indexed by stream position.
// Sorted rows + prepared table and fat entries only; anything else runs biased_glibc_seq_kernel (one lane).
constexpr int kGlRing = 4096;  // stream positions held in LDS (a power of two)

__global__ void __launch_bounds__(64)
biased_glibc_wave_kernel(const uint32_t *__restrict__ row32, const uint32_t *__restrict__ col32, const u32x4_a16 *__restrict__ ext,
                         const uint32_t *__restrict__ table, uint32_t tnb, const int64_t *__restrict__ targets,
                         int64_t n_nodes, int64_t n_walkers, int32_t L, SeedWindow sw, BiasedParams bp,
                         int64_t *__restrict__ out, int64_t *__restrict__ status)
{
    if (!rows_sorted(status)) return;  // biased_glibc_seq_kernel's job
    __shared__ uint32_t R[kGlRing];    // R[i & (kGlRing-1)] = generator word of rand() call number i
    const int lane = threadIdx.x;
    constexpr uint32_t M = kGlRing - 1;
    if (lane == 0) {
        status[2] = TRW_RNG_GLIBC;
        status[3] = 6;
    }
    if (lane < kGlDeg) R[(uint32_t)(kGlRing - kGlDeg + lane) & M] = sw.w[lane];  // calls -31 .. -1
    pool_fence();
    uint64_t pos = 0, gen = 0;  // next draw to consume / to generate (wave-uniform)
    // The generator, 31 words per turn: r[k] = r[k-3] + r[k-31] makes the new words three running sums (one per residue
    // of k mod 3) over the 31 words before them,
    //     r[n+j] = r[n + j%3 - 3] + sum of r[n + i - 31] over i = j, j-3, j-6, ... >= 0,
    // i.e. a strided prefix sum across 31 lanes (four shuffle-and-add steps: strides 3, 6, 12, 24) instead of eleven
    // dependent trips through LDS three words at a time -- 10 cycles a word instead of 40, and off the critical path:
    // the loop below runs it while the step's entry is on its way from memory.
    auto need = [&](uint64_t upto) {
        while (gen < upto) {
            const uint32_t n = (uint32_t)gen, j = (uint32_t)lane;
            uint32_t x = 0u, base = 0u;
            if (j < (uint32_t)kGlDeg) {
                x = R[(n + j - (uint32_t)kGlDeg) & M];
                base = R[(n + j % 3u - 3u) & M];
            }
#pragma unroll
            for (int d = kGlSep; d < kGlDeg; d *= 2) {
                const uint32_t y = (uint32_t)__shfl_up((int)x, d);
                if (j >= (uint32_t)d) x += y;
            }
            if (j < (uint32_t)kGlDeg) R[(n + j) & M] = x + base;
            gen += kGlDeg;
            pool_fence();
        }
    };
    auto draw = [&](uint64_t at) { return R[(uint32_t)at & M] >> 1; };
    const bool band_takes_near = bp.t1 > bp.t2;
    const uint32_t v_clear = band_takes_near ? 0u : 1u, v_set_exact = band_takes_near ? 1u : 0u;

    uint32_t st = 0;
    for (int64_t i = 0; i < n_walkers && !st; ++i) {
        int64_t *row = out + i * (int64_t)(L + 1);
        const int64_t tg = targets[i];
        if (lane == 0) row[0] = tg;
        if ((uint64_t)tg >= (uint64_t)n_nodes) {  // the chain's draw positions are undefined past a fault
            for (int32_t c = 1 + lane; c <= L; c += 64) row[c] = tg;
            st |= TRW_ST_NODE_RANGE;
            break;
        }
        uint32_t t = (uint32_t)tg, prev = t, p_start = 0, p_deg = 0, pos_lo = 0, pos_cnt = 0;
        uint32_t m0 = 0, m1 = 0, m2 = 0, m3 = 0;
        bool exact = false;
        FastMod fm;
        // the replicated state of the step onto entry {e0, e1} (the same values in every lane)
        auto step_from = [&](const u32x4_a16 &e0, const u32x4_a16 &e1, int32_t s) {
            if (lane == 0) row[s] = (int64_t)e0.x;
            t = prev;
            prev = e0.x;
            p_start = e0.y;
            p_deg = e0.z & ~kMaskIsFilter;
            exact = (e0.z & kMaskIsFilter) == 0u;
            m0 = e1.x;
            m1 = e1.y;
            m2 = e1.z;
            m3 = e1.w;
            pos_lo = e0.w & 0x0FFFFFFFu;
            pos_cnt = e0.w >> 28;
            if (p_deg == 0) return;
            fm.init_exact(p_deg);
            if (e0.w == kRposLookup)  // where t sits in prev's row: a run too long / too far in for the packed form
                pos_lo = row_find32(col32, p_start, p_deg, t, pos_cnt);
        };
        // step 1: uniform neighbour of the start node, one draw (rw_cpu.cpp:159)
        {
            uint32_t start, deg;
            load_row32(row32, t, start, deg);
            if (deg == 0) {  // rw_cpu.cpp:20 divides by zero here
                st |= TRW_ST_DEGREE0;
                break;
            }
            need(pos + 1);
            const uint32_t r0 = draw(pos);
            pos += 1;
            const uint32_t entry = start + r0 % deg;
            const u32x4_a16 e0 = ext[2 * (size_t)entry], e1 = ext[2 * (size_t)entry + 1];
            need(pos + 320);  // (while the entry is on its way)
            step_from(e0, e1, 1);
        }
        for (int32_t s = 2; s <= L && !st; ++s) {
            if (p_deg == 0) {
                st |= TRW_ST_DEGREE0;
                break;
            }
            uint32_t rejected = 0;
            u32x4_a16 c0, c1;
            int f = 0;
            while (true) {  // 64 trials at a time
                need(pos + 128);  // (generated ahead of time, below: nothing to do here but for the first round)
                const uint32_t rc = draw(pos + 2 * (uint64_t)lane), ru = draw(pos + 2 * (uint64_t)lane + 1);
                const uint32_t idx = fm.mod31(rc);
                // Every lane fetches the entry of ITS candidate at once: the winner's entry is then in registers when
                // the verdicts are in (the gather overlaps their evaluation instead of following it), a candidate the
                // masks cannot decide has its node for the table, and the draws of the next round are generated while
                // all of that is on its way.  64 gathers for one step: a single wave has the memory system to itself.
                c0 = ext[2 * (size_t)(p_start + idx)];
                c1 = ext[2 * (size_t)(p_start + idx) + 1];
                need(pos + 128 + 192);
                const uint32_t k = (idx >> 5) & 3u;
                const uint32_t word = k == 0u ? m0 : (k == 1u ? m1 : (k == 2u ? m2 : m3));
                const bool bit = (word >> (idx & 31u)) & 1u;
                const uint32_t v_band = bit ? (exact ? v_set_exact : 2u) : v_clear;
                const uint32_t v_other = ru < bp.lo ? 1u : (ru >= bp.hi ? 0u : v_band);
                const uint32_t verdict = (idx - pos_lo < pos_cnt) ? (ru < bp.t0 ? 1u : 0u) : v_other;
                unsigned long long accepts = __ballot(verdict == 1u), tests = __ballot(verdict == 2u);
                // (a use of the entry's second half HERE: it is only needed once the winner is known, and the compiler had
                // moved its load down there -- a second, dependent round trip per step, if only to the L1)
                asm volatile("" ::"v"(c1.x), "v"(c1.y), "v"(c1.z), "v"(c1.w));
                while (true) {  // settle the filter-only trials that come before the first definite accept
                    const int fa = accepts ? (int)__ffsll((long long)accepts) - 1 : 64;
                    const unsigned long long before = fa >= 64 ? ~0ull : ((1ull << fa) - 1ull);
                    const unsigned long long pend = tests & before;
                    if (pend == 0) break;
                    bool ok = false;
                    if ((pend >> lane) & 1ull) ok = edge_table_find(table, tnb, t, c0.x) == band_takes_near;
                    accepts |= __ballot(ok);
                    tests &= ~pend;
                }
                if (accepts) {
                    f = (int)__ffsll((long long)accepts) - 1;
                    pos += 2 * (uint64_t)(f + 1);
                    break;
                }
                pos += 128;
                rejected += 64;
                if (rejected >= kTrialCap) {
                    st |= (uint32_t)kStTrialCap;
                    f = 63;
                    break;
                }
            }
            // the winning lane's entry, to every lane
            u32x4_a16 e0, e1;
            e0.x = (uint32_t)__shfl((int)c0.x, f);
            e0.y = (uint32_t)__shfl((int)c0.y, f);
            e0.z = (uint32_t)__shfl((int)c0.z, f);
            e0.w = (uint32_t)__shfl((int)c0.w, f);
            e1.x = (uint32_t)__shfl((int)c1.x, f);
            e1.y = (uint32_t)__shfl((int)c1.y, f);
            e1.z = (uint32_t)__shfl((int)c1.z, f);
            e1.w = (uint32_t)__shfl((int)c1.w, f);
            step_from(e0, e1, s);
        }
    }
    if (st && lane == 0) atomicOr((unsigned long long *)&status[0], (unsigned long long)st);
}

}  // namespace
