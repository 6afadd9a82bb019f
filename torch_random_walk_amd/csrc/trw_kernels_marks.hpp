// trw_kernels_marks.hpp -- the prepare pass of the biased walk, round 5: the 32-byte entries and their neighbourhood masks.
//
// What a 32-byte entry holds, and why, is written up in trw_kernels_narrow.hpp ("Fat entries for the biased walk").
// Who works out the mask of entry u -> v (rows sorted), by the two rows' lengths deg_u, deg_v:
//   RT_INLINE  deg_u <= inline_cap (8): the thread that owns the entry.  u's row sits in its registers; a row of v of up to
//              kDirectMax entries is read and compared on the spot (rpos too: no probe of the edge set at all), a longer
//              one is asked through the edge set, one probe per distinct entry of u's row.
//   RT_TILE_A  deg_v < deg_u (or equal and u < v), deg_v <= heavy_cap: the tile pass, u's row in LDS, v's row streamed
//              past it -- the items of a task are consecutive entries of u's row, so rows of v of ANY length down to one
//              entry ride along (round 4 kept rows of up to 8 for the owning thread: one probe of the edge set per element).
//   RT_TILE_B  deg_u <= deg_v, inline_cap < deg_u <= heavy_cap: the tile pass, v's row in LDS, u's row streamed -- found
//              through v's in-list (count: entries kernel, segments: plan kernel, fill: scatter kernel).
//   RT_NONE    the shorter row is beyond heavy_cap (or rows are not sorted): all ones + filter, every band trial asks the
//              edge set.
// The entries kernel writes words 0..3 of EVERY entry in their final form (the filter bit included: an entry routed to
// the tile pass is certain to be worked out there -- the item array and the task array are sized for the worst case, so
// nothing is ever dropped; header[kStDropped] counts what would have been and stays 0) and the mask words of the entries
// it settles itself; the tile pass writes the mask words of the others.  No word has two writers.
//
// The tile pass, round 5: a BLOOM FILTER in front of the bisection.  Round 4's pass bisected every element of a
// streamed row into the tile -- 1.0e10 searches of ~13 LDS levels on R-MAT 10M/110M, 36 ms, the LDS 83 % and the
// vector pipes 69 % busy -- and 97.8 % of those searches are misses.  Now the tile row also sets one bit per entry in a
// filter of up to 64 Ki bits of LDS (32 bits per entry up to 2 048 entries: 3 % false positives; 8 bits at 8 192:
// 12 %); an element costs a multiply, a shift, ONE ds_read and a bit test, the survivors -- the 2 % hits and the
// false positives -- are compacted into a per-wave queue (ballot + mbcnt, as the walker pools do) and bisected 64 at
// a time with all lanes busy.  And because only one element in fifteen reaches the tile, the tile can be small (4 096
// entries; longer rows keep every m-th entry in it and finish in global memory, survivors only).
//
// Included by trw_hip.hip after trw_kernels_narrow.hpp (edge set, Mask128, row_find32*, TileTask, HeavyItem).
#pragma once

namespace {

// a wave-uniform value as a scalar the compiler knows to be one (what is decided from it stays a scalar branch)
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

constexpr int kStDropped = 19;   // header slot: in-list items / tasks that found no room (0: the arrays hold the worst case)
constexpr int kStItemCount = 20; // header slot: in-list items made (RT_TILE_B entries)
constexpr uint32_t kDirectMax = 8u;  // RT_INLINE: rows of v up to this long are compared directly

enum : uint32_t { RT_INLINE = 0, RT_TILE_A, RT_TILE_B, RT_NONE };
__device__ __forceinline__ uint32_t route2(uint32_t u, uint32_t v, uint32_t deg_u, uint32_t deg_v, uint32_t inline_cap,
                                           uint32_t heavy_cap)
{
    if (deg_u <= inline_cap) return RT_INLINE;
    const bool a_side = deg_v < deg_u || (deg_v == deg_u && u < v);  // ties: by id, so a regular graph splits in half
    if (a_side) return deg_v <= heavy_cap ? RT_TILE_A : RT_NONE;
    return deg_u <= heavy_cap ? RT_TILE_B : RT_NONE;
}

// ---------------------------------------------------------------------------
// The entries: words 0..3 of every entry, the masks of RT_INLINE / RT_NONE entries, the in-list counts
// ---------------------------------------------------------------------------
// A thread owns 8 consecutive entries (one bisection of row32 finds the row of the first, the others follow).
// `marks` = 0: no tile pass follows (TRW_PREP_BIASED alone): rows of u up to inline_cap as below, everything else RT_NONE
// unless the SHORTER row fits inline_cap (then the owning thread probes along it, as round 2 did).
__global__ void __launch_bounds__(kBlock)
entries_kernel(const uint32_t *__restrict__ row32, const uint32_t *__restrict__ col32, int64_t n_nodes, int64_t nnz,
               const uint32_t *__restrict__ table, uint32_t tnb, uint32_t inline_cap, uint32_t heavy_cap, uint32_t marks,
               u32x4_a16 *__restrict__ ext, uint32_t *__restrict__ in_count, int64_t *__restrict__ header)
{
    constexpr int kPer = 8;
    const bool sorted = header[kStDescents] == header[kStBoundary];
    const uint32_t e_first = row32[0], e_last = row32[n_nodes];  // entries outside belong to no row
    const int64_t chunks = (nnz + kPer - 1) / kPer;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x; c < chunks; c += stride) {
        const uint32_t e0 = (uint32_t)(c * kPer);
        uint32_t u = 0, n = (uint32_t)n_nodes;
        while (n > 1) {  // the row that holds entry e0
            const uint32_t half = n >> 1;
            if (row32[u + half] <= e0) {
                u += half;
                n -= half;
            } else {
                n = half;
            }
        }
        uint32_t row_lo = row32[u], row_hi = row32[u + 1];
        uint32_t ur[kDirectMax];  // u's row when it is short (static indices only: registers)
        uint32_t ur_of = 0xFFFFFFFFu;
        for (int k = 0; k < kPer; ++k) {
            const uint32_t e = e0 + k;
            if (e >= (uint32_t)nnz) break;
            while (e >= row_hi && u + 1 < (uint32_t)n_nodes) {
                ++u;
                row_lo = row_hi;
                row_hi = row32[u + 1];
            }
            const uint32_t v = col32[e];
            uint32_t start, deg;
            load_row32(row32, v, start, deg);
            Mask128 m;
            m.fill();
            uint32_t flag = kMaskIsFilter, rpos = kRposAbsent;
            bool write_mask = true;
            if (e >= e_first && e < e_last) {
                const uint32_t deg_u = row_hi - row_lo;
                uint32_t route = RT_NONE;
                if (sorted) {
                    route = route2(u, v, deg_u, deg, inline_cap, marks ? heavy_cap : 0u);
                    // without a tile pass behind it the owning thread also takes the entries whose row of V is the short one
                    if (!marks && route == RT_NONE && deg <= inline_cap) route = RT_INLINE;
                }
                bool rpos_known = false;
                if (route == RT_INLINE) {
                    m.clear();
                    flag = deg > 128u ? kMaskIsFilter : 0u;
                    if (deg_u <= kDirectMax && deg <= kDirectMax) {
                        // both rows short: v's row against u's, in registers -- the mask and rpos without a probe
                        if (ur_of != u) {
#pragma unroll
                            for (uint32_t i = 0; i < kDirectMax; ++i) ur[i] = i < deg_u ? col32[row_lo + i] : 0xFFFFFFFFu;
                            ur_of = u;
                        }
                        uint32_t first = 0xFFFFFFFFu, cnt = 0;
                        for (uint32_t j = 0; j < deg; ++j) {
                            const uint32_t x = col32[start + j];
                            bool in_u = false;
#pragma unroll
                            for (uint32_t i = 0; i < kDirectMax; ++i) in_u = in_u || ur[i] == x;  // (ids < 2^31: the padding never matches)
                            if (in_u) m.w[0] |= 1u << j;  // (j < 8)
                            if (x == u) {
                                first = first == 0xFFFFFFFFu ? j : first;
                                ++cnt;
                            }
                        }
                        rpos = cnt ? ((cnt << 28) | first) : kRposAbsent;  // (cnt <= 8)
                        rpos_known = true;
                    } else {
                        common_mask_table(col32, table, tnb, u, row_lo, row_hi, v, start, deg, m);
                    }
                } else if (route == RT_TILE_B) {
                    // v's in-list grows by this entry; its place in it travels in the rpos word until the tile pass, which
                    // works out the real one (and the mask), overwrites the entry
                    rpos = atomicAdd(in_count + v, 1u);
                    flag = deg > 128u ? kMaskIsFilter : 0u;
                    write_mask = false;
                    rpos_known = true;
                } else if (route == RT_TILE_A) {
                    flag = deg > 128u ? kMaskIsFilter : 0u;
                    write_mask = false;
                    rpos_known = true;  // (the tile pass: u's own id in v's row as it streams past)
                }
                if (!rpos_known && edge_table_find(table, tnb, v, u)) {  // the reverse edge exists: where u sits in v's row
                    uint32_t cnt = 0;
                    const uint32_t lo = sorted ? row_find32_wide(col32, start, deg, u, cnt) : 0u;
                    rpos = (sorted && cnt <= 15u && lo < (1u << 28)) ? ((cnt << 28) | lo) : kRposLookup;
                }
            }
            u32x4_a16 o = {v, start, deg | flag, rpos};
            ext[2 * (size_t)e] = o;
            if (write_mask) {
                u32x4_a16 mm = {m.w[0], m.w[1], m.w[2], m.w[3]};
                ext[2 * (size_t)e + 1] = mm;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// The plan: tasks for the rows that are a tile, a segment of the item array for every in-list
// ---------------------------------------------------------------------------
// One lane per node, a wave per 64 nodes; the wave takes its segments and its task numbers with ONE atomic each (a
// prefix sum over the lanes) and writes the task records of a long row together.
//
// Two classes of tile rows, two task arrays (round 5).  Most tiles are SMALL -- nine in ten of the 3.1e6 tasks of R-MAT
// 10M/110M belong to a row of fewer than 256 entries, with a tenth of the elements -- and a task of such a row is a chain
// of four dependent memory round trips (record, row, items, their rows) in front of a few hundred elements: run by a
// workgroup of eight waves that meet at every tile change it cost a third of the tile pass.  Rows of up to kSmallRow
// entries get tasks of 64 entries / 64 items for a kernel whose workgroup is ONE wave (nothing to meet, 25 independent
// waves per CU to hide the chain); longer rows tasks of 128 / 64 for the eight-wave kernel, whose tile changes are
// rare next to its elements.
constexpr uint32_t kSmallRow = 255u;  // (+ the sentinel: a tile of 256 entries)
constexpr uint32_t kTileTaskItems = 64u;  // entries of the tile row / in-list items per task: one per lane of the wave that takes it
constexpr uint32_t kSmallTaskA = kTileTaskItems, kSmallTaskB = kTileTaskItems, kBigTaskA = kTileTaskItems, kBigTaskB = kTileTaskItems;
constexpr int kStBigCount = 21, kStBigNext = 22;  // header slots of the long rows' task array (the small rows': kStTaskCount / kStTaskNext)

__global__ void __launch_bounds__(kBlock)
marks_plan_kernel(const uint32_t *__restrict__ row32, int64_t n_nodes, const uint32_t *__restrict__ in_count,
                  uint32_t *__restrict__ in_off, TileTask *__restrict__ tasks_small, uint32_t small_room,
                  TileTask *__restrict__ tasks_big, uint32_t big_room, uint32_t item_room, uint32_t inline_cap,
                  int64_t *__restrict__ header)
{
    if (header[kStDescents] != header[kStBoundary]) return;  // rows not sorted: no masks, no tasks
    // The segments and the task numbers come from three device counters: a prefix sum over the wave, then over the
    // workgroup's four waves (LDS), and ONE atomic per workgroup and counter (one per wave was 470 000 atomics on three
    // addresses: 1.7 of the kernel's 2.3 ms).
    __shared__ uint32_t w_tot[3][kBlock / 64];
    __shared__ unsigned long long blk_base[3];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    auto block_take = [&](int which, uint32_t mine_total, int64_t *counter, uint32_t &before_waves, uint32_t &all) {
        if (lane == 0) w_tot[which][wv] = mine_total;
        __syncthreads();
        before_waves = 0;
        all = 0;
#pragma unroll
        for (int k = 0; k < kBlock / 64; ++k) {
            const uint32_t t = w_tot[which][k];
            before_waves += k < wv ? t : 0u;
            all += t;
        }
        if (threadIdx.x == 0 && all) blk_base[which] = atomicAdd((unsigned long long *)counter, (unsigned long long)all);
        __syncthreads();
        return all ? blk_base[which] : 0ull;
    };
    for (int64_t base = (int64_t)blockIdx.x * kBlock; base < n_nodes; base += stride) {  // (workgroup-uniform trip count)
        const int64_t x0 = base + (wv << 6);
        const int64_t x = x0 + lane;
        uint32_t start = 0, deg = 0;
        if (x < n_nodes) load_row32(row32, (uint32_t)x, start, deg);
        const bool tile = deg > inline_cap;  // (a row of up to inline_cap entries is never the tile: RT_INLINE takes its entries)
        const bool small = deg <= kSmallRow;
        const uint32_t c = tile ? in_count[x] : 0u;
        uint32_t c_total, c_waves, c_all;
        const uint32_t c_before = wave_prefix_excl(c, lane, c_total);
        const uint64_t seg_base = block_take(0, c_total, &header[kStItemCount], c_waves, c_all) + c_waves;
        uint32_t seg = kNoSegment;
        if (c) {
            if (seg_base + c_before + c <= item_room) seg = (uint32_t)(seg_base + c_before);
            else atomicAdd((unsigned long long *)&header[kStDropped], (unsigned long long)c);  // (cannot happen: room = nnz)
        }
        if (x < n_nodes) in_off[x] = seg;
        const uint32_t per_a = small ? kSmallTaskA : kBigTaskA, per_b = small ? kSmallTaskB : kBigTaskB;
        const uint32_t n_a = tile ? (deg + per_a - 1u) / per_a : 0u;
        const uint32_t n_b = seg != kNoSegment ? (c + per_b - 1u) / per_b : 0u;
        const uint32_t n_t = n_a + n_b;
        // the workgroup's small tasks and its big ones
        uint32_t s_total, b_total, s_waves, b_waves, s_all, b_all;
        const uint32_t s_before = wave_prefix_excl(small ? n_t : 0u, lane, s_total);
        const uint32_t b_before = wave_prefix_excl(small ? 0u : n_t, lane, b_total);
        const uint64_t s_blk = block_take(1, s_total, &header[kStTaskCount], s_waves, s_all);
        const uint64_t b_blk = block_take(2, b_total, &header[kStBigCount], b_waves, b_all);
        const uint64_t s_at0 = s_blk + s_waves, b_at0 = b_blk + b_waves;
        if (threadIdx.x == 0 && s_blk + s_all > small_room)
            atomicAdd((unsigned long long *)&header[kStDropped], (unsigned long long)(s_blk + s_all - small_room));  // (cannot happen)
        if (threadIdx.x == 0 && b_blk + b_all > big_room)
            atomicAdd((unsigned long long *)&header[kStDropped], (unsigned long long)(b_blk + b_all - big_room));  // (cannot happen)
        const uint64_t at = small ? s_at0 + s_before : b_at0 + b_before;
        TileTask *const tasks = small ? tasks_small : tasks_big;
        const uint32_t room = small ? small_room : big_room;
        auto record = [&](uint32_t node, uint32_t r_start, uint32_t r_seg, uint32_t r_na, uint32_t r_pa, uint32_t r_pb, uint32_t k) {
            TileTask t;
            if (k < r_na) {
                t.node_mode = node;
                t.start = r_start + k * r_pa;
            } else {
                t.node_mode = node | 0x80000000u;
                t.start = r_seg + (k - r_na) * r_pb;
            }
            return t;
        };
        // rows with many records (a long row, or a short one that many rows point at): the whole wave writes them; the
        // others: their own lane
        uint64_t wide = __ballot(n_t > 16u);
        while (wide) {
            const int src = __builtin_ctzll(wide);
            wide &= wide - 1;
            const uint32_t w_n = (uint32_t)__shfl((int)n_t, src), w_na = (uint32_t)__shfl((int)n_a, src);
            const uint32_t w_start = (uint32_t)__shfl((int)start, src), w_seg = (uint32_t)__shfl((int)seg, src);
            const uint64_t w_at = (uint64_t)__shfl((long long)at, src);
            const bool w_small = __shfl((int)small, src) != 0;
            TileTask *const w_tasks = w_small ? tasks_small : tasks_big;
            const uint32_t w_room = w_small ? small_room : big_room;
            const uint32_t w_pa = w_small ? kSmallTaskA : kBigTaskA, w_pb = w_small ? kSmallTaskB : kBigTaskB;
            for (uint32_t k = (uint32_t)lane; k < w_n; k += 64u)
                if (w_at + k < w_room) w_tasks[w_at + k] = record((uint32_t)(x0 + src), w_start, w_seg, w_na, w_pa, w_pb, k);
        }
        if (n_t <= 16u)
            for (uint32_t k = 0; k < n_t; ++k)
                if (at + k < room) tasks[at + k] = record((uint32_t)x, start, seg, n_a, per_a, per_b, k);
    }
}

// The in-lists: every RT_TILE_B entry u -> v goes into v's segment.  A wave takes 64 consecutive nodes; the rows that can
// be the shorter side of such an entry (inline_cap < deg <= heavy_cap) are scanned by the whole wave.  What the entry
// points at -- v, the length of its row -- and its place in v's in-list are read from the entry itself, which the entries
// kernel has just written (a coalesced stream; round 4 gathered row32[v] again and took the place with a second atomic:
// two random requests per entry more).
__global__ void __launch_bounds__(kBlock)
marks_scatter_kernel(const uint32_t *__restrict__ row32, const u32x4_a16 *__restrict__ ext, int64_t n_nodes,
                     uint32_t inline_cap, uint32_t heavy_cap, const uint32_t *__restrict__ in_off,
                     HeavyItem *__restrict__ items, const int64_t *__restrict__ header)
{
    if (header[kStDescents] != header[kStBoundary]) return;  // rows not sorted: the plan made no segments
    const int lane = threadIdx.x & 63;
    const int64_t waves = (int64_t)gridDim.x * (kBlock / 64);
    for (int64_t w = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); w * 64 < n_nodes; w += waves) {
        const int64_t x = w * 64 + lane;
        uint32_t start = 0, deg = 0;
        if (x < n_nodes) load_row32(row32, (uint32_t)x, start, deg);
        uint64_t todo = __ballot(deg > inline_cap && deg <= heavy_cap);
        while (todo) {
            const int src = __builtin_ctzll(todo);
            todo &= todo - 1;
            const uint32_t u = (uint32_t)(w * 64 + src);
            const uint32_t u_start = __shfl(start, src), u_deg = __shfl(deg, src);
            for (uint32_t i = (uint32_t)lane; i < u_deg; i += 64u) {
                const uint32_t e = u_start + i;
                const u32x4_a16 o = ext[2 * (size_t)e];
                const uint32_t v = o.x, v_deg = o.z & ~kMaskIsFilter;
                if (route2(u, v, u_deg, v_deg, inline_cap, heavy_cap) != RT_TILE_B) continue;
                const uint32_t seg = in_off[v];
                if (seg == kNoSegment) continue;
                items[seg + o.w] = HeavyItem{e, u};  // (o.w: the entry's place in v's in-list, handed out by the entries kernel's count)
            }
        }
    }
}

// ---------------------------------------------------------------------------
// The tile pass with a Bloom filter in front of the bisection
// ---------------------------------------------------------------------------
// kCap: entries of LDS per tile (a power of two); kBloomWords: words of the filter (a power of two); kThreads: workgroup
// size.  A row of fewer than kCap entries is in the tile whole; of a longer one the tile holds every m-th entry -- the last
// of each block of m = ceil(n / (kCap-1)) -- and a survivor's last log2(m) levels read the block in global memory.
// The filter always covers the whole row.
#ifdef TRW_TILE_CLOCKS
static __device__ unsigned long long g_tile_dbg[16];  // (debug counters of the clocked tile kernel)
#endif
#ifndef TRW_BLOOM_K
#define TRW_BLOOM_K 2
#endif
constexpr uint32_t kBloomMul = 0x9E3779B1u;
// the bits of element x in its word of the filter, h = x * kBloomMul: two (bits 10..14 and 5..9 of h name them; the word
// is named by h's top bits).  One bit per element let 3 % of the misses through at 32 bits per entry and 12 % at 8; two
// let 0.4 % and 5 % through, and the searches of the survivors were 38 % of the wave cycles.
__device__ __forceinline__ uint32_t bloom_bits(uint32_t h)
{
#if TRW_BLOOM_K == 1
    return 0x80000000u >> ((h >> 10) & 31u);
#else
    return (0x80000000u >> ((h >> 10) & 31u)) | (0x80000000u >> ((h >> 5) & 31u));
#endif
}
#ifndef TRW_TILE_CAP
#define TRW_TILE_CAP 2048
#endif
#ifndef TRW_BIG_GRAB
#define TRW_BIG_GRAB 16
#endif
#ifndef TRW_SMALL_GRAB
#define TRW_SMALL_GRAB 16
#endif
#ifndef TRW_BLOOM_WORDS
#define TRW_BLOOM_WORDS 2000
#endif

// One kernel, two shapes.  The long rows': a workgroup of 8 waves shares the tile and the filter of one row; its waves
// take WHOLE tasks of that row in turn (64 entries or 64 in-list items each: a wave's flat numbering then spans some
// thousands of elements, thirty batches -- while the waves shared every task, sixteen items each, a wave's share of a task
// was a batch or two and the ~400 instructions of setting a task up were half of what the kernel issued).  The short rows':
// a workgroup IS a wave.
template <uint32_t kCap, uint32_t kBloomWords, int kThreads, int kMinBlocks, uint32_t kGrab, int kCountSlot, int kNextSlot,
          bool kStaticGrab>
__global__ void __launch_bounds__(kThreads, kMinBlocks)
tile_bloom_kernel(const uint32_t *__restrict__ row32, const uint32_t *__restrict__ col32, u32x4_a16 *__restrict__ ext,
                  const HeavyItem *__restrict__ items, const TileTask *__restrict__ tasks, uint32_t task_room,
                  const uint32_t *__restrict__ in_count, const uint32_t *__restrict__ in_off, uint32_t inline_cap,
                  uint32_t heavy_cap, int64_t *__restrict__ header)
{
    constexpr uint32_t kWaves = kThreads / 64;
    constexpr uint32_t kItems = kTileTaskItems;  // items of a task: one per lane
    constexpr uint32_t kQueue = 128;  // survivors waiting for a search, per wave (a stack: fewer than 64 wait when 64 elements arrive)
    static_assert(kItems == 64 && kGrab <= 64 && (kCap & (kCap - 1)) == 0 && kBloomWords >= 64, "layout");
    // per wave: its items' masks (4 words each), where their elements start in the wave's flat numbering (+ a sentinel),
    // where their rows start in col32, where the tile's node sits in them, and the stack of survivors {element, its flat number}
    constexpr uint32_t kWaveWords = kItems * 8 + 4 + 2 * kQueue;
    constexpr uint32_t kHead = kWaves * kWaveWords + 4;
    __shared__ uint32_t lds[kHead + kBloomWords + kCap];
    uint32_t *const bloom = lds + kHead;
    uint32_t *const tile = bloom + kBloomWords;
    uint32_t *const wave_lds = lds + (threadIdx.x >> 6) * kWaveWords;
    uint32_t(*const acc)[4] = reinterpret_cast<uint32_t(*)[4]>(wave_lds);
    uint32_t *const pfx = wave_lds + kItems * 4;
    uint32_t *const row_at = pfx + kItems + 1;
    uint32_t *const rp_first = row_at + kItems + 1;  // (tile row = u's) where u first sits in item j's row of v, and how often:
    uint32_t *const rp_cnt = rp_first + kItems;      // rpos of the entry, found while v's row streams past
    uint32_t *const queue = wave_lds + kItems * 8 + 4;  // (8-byte aligned: kWaveWords and the offset are even)
    uint32_t &grabbed = lds[kHead - 4], &next_task = lds[kHead - 3], &grab_len = lds[kHead - 2];
    const uint32_t tile_lds = (uint32_t)reinterpret_cast<uintptr_t>(tile);
    const int lane = threadIdx.x & 63;
    const uint32_t wave = uni(threadIdx.x >> 6);
    const uint64_t made = (uint64_t)header[kCountSlot];
    const uint32_t n_tasks = made < task_room ? (uint32_t)made : task_room;
    // the tile holds cur_node's row [t_start, t_start + n): every m-th entry, nb of them (m = 1: the row is in the tile
    // `whole`); the bisection's first step is top4 bytes; an element x is in the filter's word mulhi(x * kBloomMul, bwords)
    uint32_t cur_node = 0xFFFFFFFFu, n = 0, t_start = 0, m = 1, nb = 0, top4 = 128, bwords = 64, mtop = 1;
    bool whole = true;
#ifdef TRW_TILE_CLOCKS
    unsigned long long dbg[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    bool tile_ready = false;  // (kWaves == 1: the tile of cur_node is loaded when the first of its tasks has something to do)
#ifdef TRW_TILE_CLOCKS
    unsigned long long ck[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ck_t = clock64();
#define TRW_LAP(k) do { const unsigned long long now_ = clock64(); ck[k] += now_ - ck_t; ck_t = now_; } while (0)
#else
#define TRW_LAP(k) do { } while (0)
#endif
    // the row of cur_node into the tile and the filter (every thread of the workgroup; barriers inside)
    auto load_tile = [&]() {
        if (kWaves > 1) __syncthreads();  // (every wave is done with the old tile)
        // A row of fewer than kCap entries is in the tile whole; of a longer one the tile holds every m-th entry -- the last
        // of each block of m = ceil(n / (kCap-1)) -- and a survivor's search ends in that block, in global memory.  The filter
        // covers the whole row: 32 bits per entry, 2 048 bits at least, kBloomWords words at most.
        m = n < kCap ? 1u : (n + kCap - 2u) / (kCap - 1u);
        whole = m == 1u;
        nb = (n + m - 1u) / m;
        uint32_t pow2 = 64u;
        while (pow2 < nb + 1u) pow2 <<= 1;
        bwords = 64u;
        while (bwords < n && bwords < kBloomWords) bwords <<= 1;
        bwords = bwords < kBloomWords ? bwords : kBloomWords;  // (any number of words: an element's word is mulhi(hash, bwords))
        for (uint32_t i = threadIdx.x; i < bwords; i += kThreads) bloom[i] = 0u;
        for (uint32_t i = threadIdx.x; i < pow2; i += kThreads) {
            const uint32_t last = (i + 1u) * m < n ? (i + 1u) * m : n;  // one past block i's last entry
            tile[i] = i < nb ? col32[t_start + last - 1u] : kTileSentinel;
        }
        top4 = pow2 * 2u;  // bytes of the first bisection step: pow2/2 entries
        mtop = 1u;
        while (mtop < m) mtop <<= 1;  // first step inside a block (m > 1): covers offsets 0 .. m-1
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += kThreads) {
            const uint32_t h = (whole ? tile[i] : col32[t_start + i]) * kBloomMul;
            atomicOr(&bloom[__umulhi(h, bwords)], bloom_bits(h));
        }
        __syncthreads();
        tile_ready = true;
    };

    uint64_t turn = blockIdx.x;  // (kStaticGrab)
    while (true) {
        uint32_t t0;
        if (kStaticGrab) {
            // The short rows' tasks are many and alike: workgroup b takes the kGrab tasks of every gridDim.x-th turn.  (From
            // a device counter, four at a time, 600 000 atomics on ONE address took longer than the tasks: 55 -> 77 ms
            // for the whole prepare pass with one at a time, 50 with sixteen.)
            const uint64_t at = turn * kGrab;
            turn += gridDim.x;
            t0 = at < n_tasks ? (uint32_t)at : n_tasks;
        } else {
            __syncthreads();  // (everyone has read `grabbed` of the last round and taken its last task)
            if (threadIdx.x == 0) {
                // (kGrab = 16 tasks a turn.  64 -- fewer meetings of the workgroup's waves -- left the last turns running 8 ms
                // behind the rest; so did turns sized by what is left, 64 down to 8: the longest tasks, the hubs' in-lists of
                // 64 rows of up to 1 024 entries against a filter that lets most of them through, come FIRST in the array,
                // and sixty-four of them in one turn are a workgroup's whole run.)
                const uint64_t g = kGrab;
                grabbed = (uint32_t)atomicAdd((unsigned long long *)&header[kNextSlot], (unsigned long long)g);
                grab_len = (uint32_t)g;
                next_task = 0u;
            }
            __syncthreads();
            t0 = uni(grabbed);
        }
        TRW_LAP(0);
        if (t0 >= n_tasks) break;
        const uint32_t g_len = kStaticGrab ? kGrab : uni(grab_len);
        const uint32_t g_cnt = (t0 + g_len < n_tasks ? t0 + g_len : n_tasks) - t0;
        // the grab's records, one per lane; its tasks come in RUNS of one row's (the plan kernel writes a row's together)
        TileTask rec_l = {0xFFFFFFFFu, 0u};
        if ((uint32_t)lane < g_cnt) rec_l = tasks[t0 + lane];
        const uint32_t node_l = rec_l.node_mode & 0x7FFFFFFFu;
        // The waves of a workgroup take the tasks of a run one at a time from a counter in LDS (in turn -- wave w the tasks
        // w, w + 8, ... -- a wave with two long tasks kept the other seven waiting at the next tile change: the waits at
        // the workgroup's barriers were 28 % of the wave cycles).  `mine_t`: the task this wave has taken and not yet worked
        // on; one that belongs to a later run waits with it for that run's tile.
        auto take = [&]() {
            uint32_t t = 0;
            if (lane == 0) t = atomicAdd(&next_task, 1u);
            return uni(t);
        };
#ifdef TRW_DBG_STATIC_TASKS
        uint32_t mine_t = wave;
#else
        uint32_t mine_t = kWaves > 1 ? take() : 0u;
#endif
        uint32_t i0 = 0;
        while (i0 < g_cnt) {
            const uint32_t node = (uint32_t)__builtin_amdgcn_readlane((int)node_l, (int)i0);
            const uint64_t others = __builtin_amdgcn_ballot_w64((uint32_t)lane > i0 && (uint32_t)lane < g_cnt && node_l != node);
            const uint32_t i1 = others ? (uint32_t)__builtin_ctzll(others) : g_cnt;
            if (node != cur_node) {
                load_row32(row32, node, t_start, n);
                t_start = uni(t_start);
                n = uni(n);
                cur_node = node;
                tile_ready = false;
                if (kWaves > 1) load_tile();
            }
            TRW_LAP(2);
#ifdef TRW_DBG_STATIC_TASKS
            for (uint32_t ti = mine_t; ti < i1; ti = mine_t = ti + kWaves) {
#else
            for (uint32_t ti = mine_t; ti < i1; ti = mine_t = (kWaves > 1 ? take() : ti + 1u)) {
#endif
                struct {
                    uint32_t start, count, mode;
                } task = {(uint32_t)__builtin_amdgcn_readlane((int)rec_l.start, (int)ti), 0u,
                          (uint32_t)__builtin_amdgcn_readlane((int)rec_l.node_mode, (int)ti) >> 31};
                if (task.mode == 0u) {
                    const uint32_t rest_a = t_start + n - task.start;
                    task.count = rest_a < kItems ? rest_a : kItems;
                } else {
                    const uint32_t rest_b = in_off[node] + in_count[node] - task.start;
                    task.count = uni(rest_b < kItems ? rest_b : kItems);
                }
                // the task's items, one per lane
                uint32_t s_start = 0, s_len = 0, e = 0, other = 0;  // other: v (tile row = u's) / u (tile row = v's)
                bool mine = false;
                if ((uint32_t)lane < task.count) {
                    if (task.mode == 0u) {
                        e = task.start + (uint32_t)lane;
                        other = col32[e];
                        load_row32(row32, other, s_start, s_len);
                        mine = route2(node, other, n, s_len, inline_cap, heavy_cap) == RT_TILE_A;
                    } else {
                        const HeavyItem it = items[task.start + (uint32_t)lane];
                        e = it.e;
                        other = it.u;
                        load_row32(row32, other, s_start, s_len);
                        mine = true;
                    }
                }
                const uint64_t todo = __builtin_amdgcn_ballot_w64(mine);
                TRW_LAP(1);
                if (todo == 0) continue;
                if (kWaves == 1 && !tile_ready) load_tile();  // (a short row none of whose tasks has anything to do is never loaded)
                TRW_LAP(2);
                // The elements of the task's items -- the entries of their shorter rows -- in ONE flat numbering, 256 of them
                // (four per lane) at a time.
                const uint32_t len_j = mine ? s_len : 0u;
                uint32_t total;
                const uint32_t first_j = wave_prefix_excl(len_j, lane, total);
#ifdef TRW_TILE_CLOCKS
                dbg[whole ? 6 : 7] += total;
                dbg[8] += (total + 255u) / 256u;
                dbg[9] += 1;
#endif
                // The items that HAVE elements, numbered densely (d_j): the tables of the flat numbering -- where item d starts
                // in it, where its row starts in col32 -- and the masks the hits go to are indexed by that number, so that
                // "the item of element f" is "how many items start at or before f", a count of bits (below).
                const uint64_t ne = __builtin_amdgcn_ballot_w64(len_j != 0u);
                const uint32_t d_j = __builtin_amdgcn_mbcnt_hi((uint32_t)(ne >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ne, 0u));
                // (a zero made HERE: kept in four registers across the whole kernel it was the one value the register allocator
                // spilled to scratch -- and the private segment is set up for every wave of a kernel that has any)
                uint32_t zero;
                asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
                acc[lane][0] = acc[lane][1] = acc[lane][2] = acc[lane][3] = zero;
                rp_first[lane] = 0xFFFFFFFFu;
                rp_cnt[lane] = 0u;
                pfx[lane] = 0xFFFFFFFFu;  // (behind the last item: beyond every element)
                if (lane == 0) pfx[kItems] = 0xFFFFFFFFu;
                if (len_j != 0u) {
                    pfx[d_j] = first_j;
                    row_at[d_j] = s_start;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const bool odd_levels = (__builtin_clz(top4) & 1) != 0;
                // (wave-uniform) the item that holds the first element of the batch being fetched: its lane, its dense number,
                // and the lanes of the items after it
                uint32_t cur_lane = 0, cur_d = 0;
                uint64_t after = ne;
                if (total) {
                    cur_lane = (uint32_t)__builtin_ctzll(ne);
                    after = ne & (ne - 1);
                }
                auto fetch = [&](uint32_t f0, uint32_t (&xn)[4]) {
                    // the item that holds f0: the last one that starts at or before it
                    while (after) {
                        const uint32_t l = (uint32_t)__builtin_ctzll(after);
                        if ((uint32_t)__builtin_amdgcn_readlane((int)first_j, (int)l) > f0) break;
                        cur_lane = l;
                        ++cur_d;
                        after &= after - 1;
                    }
                    const uint32_t c_first = (uint32_t)__builtin_amdgcn_readlane((int)first_j, (int)cur_lane);
                    const uint32_t c_len = (uint32_t)__builtin_amdgcn_readlane((int)len_j, (int)cur_lane);
                    const uint32_t last = f0 + 256u < total ? f0 + 255u : total - 1u;
                    if (last < c_first + c_len) {  // the whole batch is one item's
                        const uint32_t *const row = col32 + (uint32_t)__builtin_amdgcn_readlane((int)s_start, (int)cur_lane) + (f0 - c_first);
                        const uint32_t span = last - f0;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const uint32_t o = 64u * r + (uint32_t)lane;
                            xn[r] = row[o < span ? o : span];
                        }
                        return;
                    }
                    // Several items in the batch.  The items that start inside the batch set a bit at their first position --
                    // scalar work, a few items a batch --, and the item of position q is cur_d + the number of bits at positions
                    // <= q: two v_mbcnt.  (A lane used to bisect the items' prefix table for each of its four elements: 22 vector
                    // instructions an element.)
                    uint64_t m0 = 0, m1 = 0, m2 = 0, m3 = 0;
                    const uint32_t base_d = cur_d;
                    while (after) {
                        const uint32_t l = (uint32_t)__builtin_ctzll(after);
                        const uint32_t st = (uint32_t)__builtin_amdgcn_readlane((int)first_j, (int)l);
                        if (st > last) break;
                        const uint32_t p = st - f0;  // (1 .. 255: position 0 belongs to cur)
                        const uint64_t bit = 1ull << (p & 63u);
                        const uint32_t w = p >> 6;
                        m0 |= w == 0u ? bit : 0ull;
                        m1 |= w == 1u ? bit : 0ull;
                        m2 |= w == 2u ? bit : 0ull;
                        m3 |= w == 3u ? bit : 0ull;
                        cur_lane = l;  // (the next batch starts behind every item seen here)
                        ++cur_d;
                        after &= after - 1;
                    }
                    const uint64_t mm[4] = {m0, m1, m2, m3};
                    uint32_t before = base_d;  // items that start at or before the end of the previous 64 positions
                    uint32_t dd[4], ra[4], pf[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // bits of mm[r] at positions <= lane: its bit 0, and mbcnt (bits BELOW the lane) of the mask moved down by one
                        const uint64_t sh = mm[r] >> 1;
                        dd[r] = __builtin_amdgcn_mbcnt_hi((uint32_t)(sh >> 32),
                                                          __builtin_amdgcn_mbcnt_lo((uint32_t)sh, before + (uint32_t)(mm[r] & 1ull)));
                        before += (uint32_t)__builtin_popcountll(mm[r]);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {  // (the four pairs of reads in one flight)
                        ra[r] = row_at[dd[r]];
                        pf[r] = pfx[dd[r]];
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        uint32_t f = f0 + 64u * r + (uint32_t)lane;
                        f = f < total ? f : total - 1u;
                        xn[r] = col32[ra[r] + (f - pf[r])];
                    }
                };
                // The survivors of the filter wait on the wave's stack; 64 of them (or what is left at the end of the task) are
                // bisected into the tile together.  A level is add, ds_read, compare, select; `qa` = LDS address of the last tile
                // entry known to be < x (tile[-1] to start with).
                // Where x sits in the tile row: `pos` = entries of the row below x, `run` = how many equal it (0: not there).
                auto locate = [&](uint32_t x, bool live, bool want_run, uint32_t &pos, uint32_t &run, auto whole_row) {
                    constexpr bool kWhole = decltype(whole_row)::value;
                    uint32_t qa = tile_lds - 4u;
                    auto level = [&](uint32_t s4) {
                        const uint32_t a = qa + s4;
                        const uint32_t t = *reinterpret_cast<const lds_u32 *>((uintptr_t)a);
                        qa = t < x ? a : qa;
                    };
                    // TWO levels in one round trip to the LDS: the three entries the two levels could look at are read
                    // together (the wave waits for the LDS, not for its adders: 12 dependent reads were 12 latencies)
                    auto level2 = [&](uint32_t s4) {
                        const uint32_t h4 = s4 >> 1;
                        const uint32_t t1 = *reinterpret_cast<const lds_u32 *>((uintptr_t)(qa + h4));
                        const uint32_t t2 = *reinterpret_cast<const lds_u32 *>((uintptr_t)(qa + s4));
                        const uint32_t t3 = *reinterpret_cast<const lds_u32 *>((uintptr_t)(qa + s4 + h4));
                        const bool up = t2 < x;
                        const uint32_t t = up ? t3 : t1;
                        qa += (up ? s4 : 0u) + (t < x ? h4 : 0u);
                    };
                    uint32_t s4 = top4;
                    if (odd_levels) {
                        level(s4);
                        s4 >>= 1;
                    }
                    for (; s4 >= 8u; s4 >>= 2) level2(s4);
                    const uint32_t jb = (qa + 4u - tile_lds) >> 2;  // tile entries < x
                    run = 0;
                    if (kWhole) {  // the tile is the row
                        pos = jb;
                        if (live && *reinterpret_cast<const lds_u32 *>((uintptr_t)(qa + 4u)) == x) {
                            run = 1u;
                            if (want_run)
                                while (tile[pos + run] == x) ++run;  // (the padding ends every run)
                        }
                    } else {  // x can only be in block jb of the row, whose last entry (>= x) the tile holds; the others: global memory
                        // One load a level, up to six levels, and NO load to look at what was found: the search is for the last
                        // entry <= x, which one of its own probes has read.  (Tried: the block's entries loaded together when
                        // m <= 8 -- one round trip, seven loads a lane --, and four pivots + eight entries (row_find32_wide) for
                        // longer blocks: 12 200 and 14 300 cycles a search against 8 900 -- what a search of these rows costs is
                        // the NUMBER of its scattered loads, not their depth.)
                        const uint32_t lo = jb * m;
                        const uint32_t len = (live && jb < nb) ? (n - lo < m ? n - lo : m) : 0u;
                        const uint32_t last = *reinterpret_cast<const lds_u32 *>((uintptr_t)(qa + 4u));  // entry lo + len - 1
                        uint32_t le = 0, val = 0;  // entries of the block's first len - 1 that are <= x; the largest of them (le > 0)
                        for (uint32_t step = mtop >> 1; step; step >>= 1) {
                            const uint32_t p = le + step;
                            if (p < len) {
                                const uint32_t ev = col32[t_start + lo + p - 1u];
                                if (ev <= x) {
                                    le = p;
                                    val = ev;
                                }
                            }
                        }
                        const bool at_last = len != 0u && last == x;
                        // (le != 0: a lane beyond the survivors reads whatever the stack's memory holds for x -- 0xFFFFFFFF, a
                        // sentinel of this kernel's own tables left in LDS by an earlier launch, once passed for "found")
                        const bool at_val = le != 0u && val == x;
                        if (at_val || at_last) {  // entries lo .. lo + le - 1 are <= x, and the last of them (or the block's last) is x
                            uint32_t first = at_val ? lo + le - 1u : lo + len - 1u, end = first + 1u;
                            if (want_run) {  // (wave-uniform) the whole run: parallel edges are rare, one load either way
                                if (at_val && at_last) end = lo + len;
                                while (first > 0u && col32[t_start + first - 1u] == x) --first;
                                while (end < n && col32[t_start + end] == x) ++end;
                            }
                            pos = first;
                            run = end - first;
                        } else {
                            pos = lo + le;
                        }
                    }
                };
                // rpos of the items of a task whose tile row is v's: where u sits in it -- one search a task, 64 lanes wide
                uint32_t rpos_b = kRposAbsent;
                if (task.mode != 0u) {
                    uint32_t pos, run;
                    if (whole) locate(other, mine, true, pos, run, std::true_type{});
                    else locate(other, mine, true, pos, run, std::false_type{});
                    if (run) rpos_b = (run <= 15u && pos < (1u << 28)) ? ((run << 28) | pos) : kRposLookup;
                }
                uint32_t q_n = 0;  // (wave-uniform) survivors on the stack
                auto search = [&](uint32_t cnt, auto whole_row) {  // the top `cnt` (<= 64) of them
                    q_n -= cnt;
                    const uint32_t at = (q_n + (uint32_t)lane) * 2u;
                    const uint32_t x = queue[at], f = queue[at + 1u];  // the element and its flat number
                    const bool live = (uint32_t)lane < cnt;
                    uint32_t pos, run;
                    locate(x, live, task.mode != 0u, pos, run, whole_row);  // (tile row = u's: is it there; = v's: where, how often)
                    const bool hit = run != 0u;
                    if (__builtin_amdgcn_ballot_w64(hit) == 0) return;
                    if (hit) {
                        uint32_t j = 0;  // the item of element f: the last one that starts at or before it (hits are rare: 2 %)
#pragma unroll
                        for (uint32_t step = kItems >> 1; step; step >>= 1) j = pfx[j + step] <= f ? j + step : j;
                        uint32_t *const am = acc[j];
                        if (task.mode == 0u) {  // element idx of v's row is in u's row
                            const uint32_t idx = f - pfx[j];
                            atomicOr(&am[(idx >> 5) & 3u], 1u << (idx & 31u));
                        } else {  // this element of u's row sits at pos .. of v's row (a run: parallel edges)
                            uint32_t p = pos;
                            for (uint32_t k = 0; k < run && k < 128u; ++k, ++p) atomicOr(&am[(p >> 5) & 3u], 1u << (p & 31u));
                        }
                    }
                };
                auto drain = [&](uint32_t cnt) {
                    TRW_LAP(5);
#ifdef TRW_TILE_CLOCKS
                    const unsigned long long d_t0 = clock64();
#endif
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (whole) search(cnt, std::true_type{});
                    else search(cnt, std::false_type{});
#ifdef TRW_TILE_CLOCKS
                    if (lane == 0 && (kThreads == 64) == (TRW_TILE_CLOCKS == 2)) {
                        const int k = whole ? 0 : 1;
                        dbg[k] += 1;
                        dbg[2 + k] += cnt;
                        dbg[4 + k] += clock64() - d_t0;
                    }
#endif
                    TRW_LAP(6);
                };
                // One batch: 256 elements (four per lane) against the filter -- multiply, two shifts, ds_read, shift, sign test
                // --, the survivors onto the stack (ballot + mbcnt), a search when 64 or more wait.  kFull: every lane has four.
                auto batch = [&](uint32_t f0, const uint32_t (&x)[4], auto full) {
                    constexpr bool kFull = decltype(full)::value;
                    const uint32_t rest = total - f0;  // (wave-uniform)
                    uint32_t bw[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const uint32_t h = x[r] * kBloomMul;
                        const uint32_t bits = bloom_bits(h);
                        bw[r] = (bloom[__umulhi(h, bwords)] & bits) ^ bits;  // 0: both bits are set
                    }
                    if (task.mode == 0u) {  // v's row streams past: the tile's own node in it is rpos of the entry (rare: 1.5 %)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const bool me = x[r] == node && (kFull || 64u * r + (uint32_t)lane < rest);
                            if (__builtin_amdgcn_ballot_w64(me) == 0) continue;
                            if (me) {
                                const uint32_t f = f0 + 64u * r + (uint32_t)lane;
                                uint32_t j = 0;
#pragma unroll
                                for (uint32_t step = kItems >> 1; step; step >>= 1) j = pfx[j + step] <= f ? j + step : j;
                                atomicMin(&rp_first[j], f - pfx[j]);
                                atomicAdd(&rp_cnt[j], 1u);
                            }
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (!kFull && 64u * r >= rest) break;  // (wave-uniform)
                        const bool pass = bw[r] == 0u && (kFull || 64u * r + (uint32_t)lane < rest);
                        const unsigned long long b = __builtin_amdgcn_ballot_w64(pass);
                        if (b) {
                            const uint32_t off = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
                            if (pass) {
                                const uint32_t at = (q_n + off) * 2u;
                                queue[at] = x[r];
                                queue[at + 1u] = f0 + 64u * r + (uint32_t)lane;
                            }
                            q_n += (uint32_t)__popcll(b);
                        }
                        if (q_n >= 64u) drain(64u);
                    }
                };
                // two sets of registers take turns: one is tested while the other's loads are on their way.  (Three sets -- two
                // batches in flight -- took 83 registers instead of 71 and a workgroup per CU with them: 21.6 -> 26.5 ms.)
                uint32_t xa[4], xb[4];
                if (total) fetch(0u, xa);
                for (uint32_t f0 = 0; f0 < total;) {
                    if (f0 + 256u < total) {
                        TRW_LAP(5);
                        fetch(f0 + 256u, xb);
                        TRW_LAP(3);
                        batch(f0, xa, std::true_type{});
                    } else {
                        batch(f0, xa, std::false_type{});
                        break;
                    }
                    f0 += 256u;
                    if (f0 + 256u < total) {
                        TRW_LAP(5);
                        fetch(f0 + 256u, xa);
                        TRW_LAP(3);
                        batch(f0, xb, std::true_type{});
                    } else {
                        batch(f0, xb, std::false_type{});
                        break;
                    }
                    f0 += 256u;
                }
                while (q_n >= 64u) drain(64u);  // (a last batch that ended before its check)
                if (q_n) drain(q_n);
                TRW_LAP(5);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (mine) {  // the whole entry: {v, start(v), deg(v) | filter bit, rpos} + the mask
                    u32x4_a16 mm = {0u, 0u, 0u, 0u};  // (an item without elements -- a dead end's empty row -- has an empty mask)
                    uint32_t rpos = rpos_b;
                    if (len_j != 0u) {
                        mm = u32x4_a16{acc[d_j][0], acc[d_j][1], acc[d_j][2], acc[d_j][3]};
                        if (task.mode == 0u) {
                            const uint32_t cnt = rp_cnt[d_j], first = rp_first[d_j];
                            rpos = cnt ? ((cnt <= 15u && first < (1u << 28)) ? ((cnt << 28) | first) : kRposLookup) : kRposAbsent;
                        }
                    }
                    const uint32_t v_start = task.mode == 0u ? s_start : t_start, v_deg = task.mode == 0u ? s_len : n;
                    ext[2 * (size_t)e] = u32x4_a16{task.mode == 0u ? other : node, v_start, v_deg | (v_deg > 128u ? kMaskIsFilter : 0u), rpos};
                    ext[2 * (size_t)e + 1] = mm;
                }
                __builtin_amdgcn_wave_barrier();  // (the next task clears acc)
                TRW_LAP(4);
            }
            i0 = i1;
        }
    }
#ifdef TRW_TILE_CLOCKS  // (1: the clocks of the long rows' kernel, 2: of the short rows')
    if (lane == 0 && (kThreads == 64) == (TRW_TILE_CLOCKS == 2))
        for (int k = 0; k < 7; ++k) atomicAdd((unsigned long long *)&header[24 + k], ck[k]);
    if (lane == 0 && (kThreads == 64) == (TRW_TILE_CLOCKS == 2))
        for (int k = 0; k < 10; ++k) atomicAdd(&g_tile_dbg[k], dbg[k]);
#endif
#undef TRW_LAP
}

}  // namespace
