/*
 * trw.h -- C ABI of the MI355X (gfx950) CSR random-walk library, libtrw_hip.so.
 *
 * This is the drop-in boundary for ONE hot path of Sachinx0e/torch_random_walk:
 * the CSR walker behind torch_rw.rw.walk.  Each entry point names the reference
 * interface it replaces (paths relative to the reference repo root).  Plain
 * pointers and sizes only -- no torch types.  All pointers except `status`
 * outputs are DEVICE pointers valid on the device that is current when the call
 * is made; the caller owns every buffer.  Calls are asynchronous on
 * `hip_stream` (a hipStream_t, NULL = the null stream) unless stated otherwise
 * and allocate nothing after the first call on a device (which uploads a 190 KB
 * constant jump table once); after that a call enqueues kernels only (no
 * memset/memcpy nodes, no synchronisation) and may be captured into a HIP graph.  Environment knobs (debug/tests): TRW_GLIBC_CHAIN=k
 * walkers chained per lane in the uniform kernel; TRW_FORCE_WIDE=1 use the int64
 * kernels; TRW_NO_POS=1 disable the located-t shortcut of the biased kernel;
 * TRW_HASH=0/1, TRW_EXT=0/1 force the edge set / the fat entries off or on;
 * TRW_HASH_SLACK=k TENTHS of a place of the edge set per edge (default 25: load 0.4);
 * TRW_RESIDENT_BLOCKS=k workgroups per CU of the persistent biased kernel (default 5);
 * TRW_COMMON_MAX=k longest (shorter) row whose neighbourhood mask the thread that owns
 * the entry works out itself (default 32; 8 when the tile pass follows);
 * TRW_HEAVY_MAX=k longest (shorter) row the marking passes take (default 256 / 1024);
 * TRW_PLAN_MARKS_AT / TRW_PLAN_FULL_AT=k steps per CSR entry from which ONE call is
 * planned with TRW_PREP_MARKS / _FULL (default 0 / 9: the first level comes with the entries).
 * None of them changes a result.
 */
#ifndef TRW_H
#define TRW_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRW_VERSION 100 /* 0.1.0 */

/* ---- RNG contract ------------------------------------------------------- */
/* The reference's CPU path draws from ONE process-global glibc rand() stream,
 * seeded once per call (csrc/cpu/rw_cpu.cpp:60,118) and consumed walker-major.
 *
 * TRW_RNG_GLIBC   reference-exact.  Uniform walks (p == 1 && q == 1) consume
 *                 exactly one draw per step, so walker i, step s uses stream
 *                 index (walker_base+i)*L + s-1; every walker jumps straight
 *                 to its window (exact jump-ahead of the lagged-Fibonacci
 *                 recurrence) and the launch is fully parallel.  Biased walks
 *                 consume a data-dependent number of draws, so the stream is
 *                 an inherently serial chain: this mode then runs ONE wave
 *                 that replays the chain in order (bit-exact, slow; meant for
 *                 the small inputs on which the reference itself is
 *                 deterministic).  Requires walker_base == 0 for biased walks.
 * TRW_RNG_WALKER  per-walker counter stream: Philox4x32-10, key (seed, 0),
 *                 counter (walker_lo, walker_hi, draw/4, 0), value =
 *                 word[draw%4] >> 1 (31 bits, like rand()).  Shard- and
 *                 launch-geometry-invariant; fully parallel for biased walks.
 * TRW_RNG_AUTO    uniform -> GLIBC.  biased -> GLIBC when walker_base == 0 and
 *                 n_walkers <= 32768/(walk_length+1) (the reference's
 *                 parallel_for grain, rw_cpu.cpp:137 -- the regime in which
 *                 csrc/cpu output is defined independently of its thread
 *                 count), WALKER otherwise.
 */
#define TRW_RNG_AUTO   0
#define TRW_RNG_GLIBC  1
#define TRW_RNG_WALKER 2

/* ---- error codes (return values; 0 = success) --------------------------- */
#define TRW_OK            0
#define TRW_E_ARG        (-1)  /* null pointer, negative size, walk_length < 1, ... */
#define TRW_E_WORKSPACE  (-2)  /* workspace missing or too small                     */
#define TRW_E_HIP        (-3)  /* a HIP runtime call or kernel launch failed          */
#define TRW_E_RNG_MODE   (-4)  /* rng_mode not usable for this call                   */
#define TRW_E_RANGE      (-5)  /* stream position beyond 2^48 draws                   */
#define TRW_E_DATA       (-6)  /* host leg only: the graph or the start nodes are
                                  faulty (trw_last_error() names the TRW_ST_* bits;
                                  the device legs report those through the status
                                  block instead, they cannot stop a launch)           */
/* device-side status bits, returned by trw_walk_status() */
#define TRW_ST_DEGREE0    1    /* a walker stood on a node with no out-edges (the
                                  reference divides by zero there, rw_cpu.cpp:20)  */
#define TRW_ST_NODE_RANGE 2    /* a start node or a col_idx entry outside [0,n)    */
#define TRW_ST_BAD_CSR    4    /* row_ptr not monotone within [0, nnz] (or, graphs
                                  on the uint32 layout: one row of >= 2^31 entries) */
#define TRW_ST_TRIAL_CAP  8    /* 2^20 consecutive rejections at one step; the
                                  candidate was force-accepted (the reference
                                  would still be spinning, rw_cpu.cpp:166)         */
#define TRW_E_UNSUPPORTED (-7) /* narrow rows asked of a call shape whose kernel family writes int64 only */

int trw_version(void);

/* Text of the last error on the calling thread ("" if none). */
const char *trw_last_error(void);

/* Bytes of device scratch trw_walk_csr_hip needs for this call shape
 * (>= 256): status block, per-lane RNG windows (uniform walks), the uint32
 * shadow of the CSR (4*(n_nodes+1+nnz) bytes), and by call shape the fat
 * entries (16*nnz), or the edge set (20*nnz), the 32-byte entries (32*nnz) and, for a call that
 * makes 5 steps per entry or more, the marking passes' scratch (10.3*nnz + 8*n_nodes).  This function is
 * the authority; the figures are for orientation (about 67 bytes per entry + 12 per node at most). */
size_t trw_walk_csr_workspace_bytes(int64_t n_nodes, int64_t nnz, int64_t n_walkers,
                                    int32_t walk_length, double p, double q, int32_t rng_mode);

/*
 * The CSR walk.  Replaces the reference's
 *     torch::Tensor walk_gpu(row_ptr, column_idx, target_nodes, p, q, walk_length, seed)
 * (csrc/cuda/rw_cuda.h:4-10, called from csrc/rw_init.cpp:20-21) and produces,
 * on the GPU, what walk_cpu (csrc/cpu/rw_cpu.cpp:203-226) produces.
 *
 *   row_ptr  [n_nodes+1] int64   CSR row offsets          (reference arg 1)
 *   col_idx  [nnz]       int64   CSR column indices       (reference arg 2)
 *   targets  [n_walkers] int64   start nodes, contiguous  (reference arg 3)
 *   walker_base                  global index of targets[0]; lets a shard of a
 *                                larger batch reproduce exactly the rows the
 *                                unsharded call would (no reference counterpart:
 *                                the reference is single-device)
 *   p, q, walk_length, seed      as the reference; p == 1.0 && q == 1.0 selects
 *                                the uniform walk by exact comparison
 *                                (rw_cpu.cpp:220)
 *   out      [n_walkers][walk_length+1] int64 row-major; column 0 = start node
 *   workspace / workspace_bytes  device scratch, see above; 16-byte aligned
 *
 * Returns TRW_OK or a negative code (trw_last_error() has the text).  Data
 * problems found on the device (degree-0 node, node id out of range) do not
 * fault: the walker stays where it is, a status bit is set, and
 * trw_walk_status() reports it.
 */
int trw_walk_csr_hip(const int64_t *row_ptr, int64_t n_nodes,
                     const int64_t *col_idx, int64_t nnz,
                     const int64_t *targets, int64_t n_walkers,
                     int64_t walker_base,
                     double p, double q, int32_t walk_length, int32_t seed,
                     int32_t rng_mode,
                     int64_t *out,
                     void *workspace, size_t workspace_bytes,
                     void *hip_stream);

/*
 * The prepared graph.  Everything the walk derives from (row_ptr, col_idx) alone -- validation, the uint32
 * shadow, fat entries, the edge set, the per-edge neighbourhood masks -- is built into ONE caller-owned
 * device buffer and reused by any number of walks (the reference has no such object: it re-reads the int64
 * CSR on every call; SURVEY 8(b) "Ownership" allows cached derived data).  trw_walk_csr_hip above is the
 * one-shot form: it prepares inside its own workspace on every call.
 *
 *   what  OR of TRW_PREP_* bits (0 = validated uint32 shadow only):
 *     TRW_PREP_UNIFORM  + 16-byte fat entries {v, start(v), deg(v)}: the uniform walk is one gather per step
 *     TRW_PREP_BIASED   + the edge set (8-byte keys {u, v} in 32-byte blocks, 2.5 places per edge; TRW_HASH_SLACK
 *                       = tenths of a place per edge, i.e. a load of 0.4 by default) and 32-byte fat entries {v,
 *                       start(v), deg(v), where-the-reverse-edge-sits, 128-bit "is the node at this offset a
 *                       neighbour of the node I came from" mask}: a node2vec rejection trial is decided in
 *                       registers and the graph is read once per step.  Masks are worked out for entries one of
 *                       whose rows has at most 32 entries; the rest ask the edge set.
 *     TRW_PREP_MARKS    (implies BIASED) masks for every entry one of whose rows has at most 256 entries (the
 *                       tile pass: the longer row in LDS behind a Bloom filter): CHEAPER to build than BIASED alone
 *                       (the owning thread then keeps only rows of up to 8 entries) and 40 % faster to walk, at the
 *                       price of 10 bytes per entry of scratch while it is built -- trw_prepare_plan asks for it
 *                       whenever it asks for BIASED
 *     TRW_PREP_MARKS_FULL (implies MARKS) ... at most 1024 entries: about 1.25x the prepare time, another 20 %
 *                       (planned from 9 steps per entry on)
 *   or one of          TRW_PREP_ORDER_ONLY: no shadow, only the row-order probe the int64 biased kernels want;
 *                      TRW_PREP_NOTHING: header only (uniform walk on the per-walker stream)
 * trw_prepare_plan returns the `what` one call of the given shape is worth (what trw_walk_csr_hip builds).
 * The buffer must be 16-byte aligned and stay untouched while walks use it; it is valid for exactly the
 * (row_ptr, col_idx) contents it was built from.  trw_walk_csr_prepared_hip takes the same arguments as
 * trw_walk_csr_hip plus the buffer; its workspace is small (status block + 124 B per lane of a uniform walk).
 */
#define TRW_PREP_UNIFORM     1
#define TRW_PREP_BIASED      2
#define TRW_PREP_MARKS       4
#define TRW_PREP_MARKS_FULL  8
#define TRW_PREP_ORDER_ONLY (-1)
#define TRW_PREP_NOTHING    (-2)
int32_t trw_prepare_plan(int64_t n_nodes, int64_t nnz, int64_t n_walkers, int32_t walk_length,
                         double p, double q, int32_t rng_mode);
size_t trw_prepare_csr_bytes(int64_t n_nodes, int64_t nnz, int32_t what);
int trw_prepare_csr_hip(const int64_t *row_ptr, int64_t n_nodes, const int64_t *col_idx, int64_t nnz,
                        int32_t what, void *prepared, size_t prepared_bytes, void *hip_stream);
/* The marking passes (TRW_PREP_MARKS*) need scratch -- the in-lists' items, the tile pass's task records, two words
 * per node -- that no walk ever reads: trw_prepare_csr_bytes = trw_prepare_csr_persistent_bytes (what the walks read:
 * shadow, edge set, entries; 56 bytes per entry + 4 per node) + trw_prepare_csr_scratch_bytes (about 10.3 per entry + 8
 * per node; 0 without marks), and trw_prepare_csr_hip keeps both in the one buffer.  A caller that wants the scratch
 * back hands it in separately: `prepared` then holds trw_prepare_csr_persistent_bytes, `scratch` (16-byte aligned) is
 * free again once the stream has run the call's kernels.  scratch == NULL: as trw_prepare_csr_hip.  The arrays are
 * sized for the worst case -- nothing a graph can look like makes a pass run out of room; header slot 19 of the prepared
 * buffer (int64) counts what would have been dropped and stays 0. */
size_t trw_prepare_csr_persistent_bytes(int64_t n_nodes, int64_t nnz, int32_t what);
size_t trw_prepare_csr_scratch_bytes(int64_t n_nodes, int64_t nnz, int32_t what);
int trw_prepare_csr_scratch_hip(const int64_t *row_ptr, int64_t n_nodes, const int64_t *col_idx, int64_t nnz,
                                int32_t what, void *prepared, size_t prepared_bytes, void *scratch,
                                size_t scratch_bytes, void *hip_stream);
/* Where a part of the prepared buffer sits (diagnostics and tests: the layout is a function of the shapes alone).
 * *bytes == 0: this `what` has no such part. */
#define TRW_REGION_HEADER 0
#define TRW_REGION_ROW32  1   /* uint32 row offsets, n_nodes + 1 */
#define TRW_REGION_COL32  2   /* uint32 neighbours, nnz */
#define TRW_REGION_EXT16  3   /* 16-byte entries of the uniform walk */
#define TRW_REGION_TABLE  4   /* edge set: blocks of four 8-byte keys {u, v} */
#define TRW_REGION_EXT32  5   /* 32-byte entries of the node2vec walk: {v, start(v), deg(v)|filter, rpos} + 128-bit mask */
#define TRW_REGION_TASKS  6   /* the tile pass's task records {node | mode, start} (scratch of the prepare pass; the offset is the
                                 one inside a buffer that holds the scratch too) */
int trw_prepare_csr_region(int64_t n_nodes, int64_t nnz, int32_t what, int32_t which, size_t *offset, size_t *bytes);
size_t trw_walk_csr_prepared_workspace_bytes(int64_t n_walkers, int32_t walk_length, double p, double q,
                                             int32_t rng_mode);
int trw_walk_csr_prepared_hip(const void *prepared, size_t prepared_bytes, int32_t what,
                              const int64_t *row_ptr, int64_t n_nodes,
                              const int64_t *col_idx, int64_t nnz,
                              const int64_t *targets, int64_t n_walkers, int64_t walker_base,
                              double p, double q, int32_t walk_length, int32_t seed, int32_t rng_mode,
                              int64_t *out, void *workspace, size_t workspace_bytes, void *hip_stream);

/*
 * The same walk with NARROW rows: out32 [n_walkers][walk_length+1] uint32 row-major -- the wire format of the multi-GPU
 * gather (node ids are < n_nodes <= 2^32), written by the walk kernel itself instead of by a pass over int64
 * rows (trw_rows_narrow_hip).  Same walks, value for value.  Only the production kernels write them: the reference-exact
 * uniform walk over the uint32 shadow, and the node2vec walk on the per-walker stream over a graph prepared with
 * TRW_PREP_BIASED (sorted or unsorted rows); any other call shape -- the serial replay, the uniform walk on the
 * per-walker stream, graphs on the int64 kernels, 2^32 walkers or more -- returns TRW_E_UNSUPPORTED and writes no row: the
 * caller walks int64 rows and narrows them.
 */
int trw_walk_csr_prepared_u32_hip(const void *prepared, size_t prepared_bytes, int32_t what,
                                  const int64_t *row_ptr, int64_t n_nodes,
                                  const int64_t *col_idx, int64_t nnz,
                                  const int64_t *targets, int64_t n_walkers, int64_t walker_base,
                                  double p, double q, int32_t walk_length, int32_t seed, int32_t rng_mode,
                                  uint32_t *out32, void *workspace, size_t workspace_bytes, void *hip_stream);

/*
 * The host leg of the same walk.  Replaces the reference's
 *     torch::Tensor walk_cpu(row_ptr, column_idx, target_nodes, p, q, walk_length, seed)
 * (csrc/cpu/rw_cpu.h:5-11, called from csrc/rw_init.cpp:22-23 for CPU tensors).  All pointers are HOST
 * pointers; the call returns when the rows are written.  Arguments as trw_walk_csr_hip; no workspace.
 *   rng_mode   TRW_RNG_AUTO / TRW_RNG_GLIBC: the reference's one glibc rand() stream, i.e. exactly
 *              what csrc/cpu produces on one thread.  Uniform walks are split over `n_threads` host
 *              threads by jump-ahead (same rows for any thread count); a biased walk's stream is a
 *              serial chain and runs on the calling thread (walker_base must be 0).
 *              TRW_RNG_WALKER: the per-walker Philox stream of the HIP kernels, on `n_threads` threads
 *              (row for row what trw_walk_csr_hip writes in that mode).
 *   n_threads  host threads to use (< 1 is taken as 1)
 * Data faults (degree-0 node, node id out of range, broken row_ptr, runaway rejection loop) stop the
 * call: TRW_E_DATA, trw_last_error() says which.
 */
int trw_walk_csr_cpu(const int64_t *row_ptr, int64_t n_nodes,
                     const int64_t *col_idx, int64_t nnz,
                     const int64_t *targets, int64_t n_walkers,
                     int64_t walker_base,
                     double p, double q, int32_t walk_length, int32_t seed,
                     int32_t rng_mode,
                     int64_t *out, int32_t n_threads);

/*
 * The triples (knowledge-graph) walk.  Replaces the reference's
 *     torch::Tensor triples::walk_triples_gpu(triples_indexed, relation_tail_index, target_nodes,
 *                                             walk_length, padding_idx, restart, seed)
 * (csrc/cuda/rw_cuda_triples.h:4-13, called from csrc/rw_init.cpp:57-65) and produces what
 * triples::walk_triples_cpu (csrc/cpu/rw_cpu_triples.cpp:105-127) produces.
 *
 *   triples   [n_triples][3]  int64 (head, relation, tail) sorted by head     (reference arg 1)
 *   rt_index  [n_entities][2] int64 inclusive (first, last) triple per head,
 *                             (-1,-1) for an entity without out-edges          (reference arg 2)
 *   targets   [n_walkers]     int64 start entities                             (reference arg 3)
 *   padding_idx               emitted once a walk can go no further            (reference arg 5)
 *   out       [n_walkers][2*walk_length+1] int64: start, then (relation, tail) per hop
 * The reference's `restart` argument is accepted by its Python surface and never read
 * (rw_cpu_triples.cpp:48-55); it has no counterpart here.  rng_mode as for the CSR walk:
 * TRW_RNG_GLIBC replays the reference's serial stream with one lane (walker_base == 0),
 * TRW_RNG_WALKER uses the per-walker stream (draw k = the k-th draw the walker consumes; hops
 * from single-edge entities and padding consume none, csrc/cpu/cpu_utils.cpp:4-5), TRW_RNG_AUTO
 * picks GLIBC for walker_base == 0 && n_walkers <= 32768 && n_walkers*walk_length <= 2^18.
 */
size_t trw_walk_triples_workspace_bytes(int64_t n_triples, int64_t n_entities, int64_t n_walkers,
                                        int32_t walk_length, int64_t padding_idx, int32_t rng_mode);
int trw_walk_triples_hip(const int64_t *triples, int64_t n_triples,
                         const int64_t *rt_index, int64_t n_entities,
                         const int64_t *targets, int64_t n_walkers, int64_t walker_base,
                         int32_t walk_length, int64_t padding_idx, int32_t seed, int32_t rng_mode,
                         int64_t *out, void *workspace, size_t workspace_bytes, void *hip_stream);

/*
 * The edge-list walk.  Replaces the reference's
 *     torch::Tensor walk_edge_list_gpu(edge_list_indexed, node_edges_idx, target_nodes, p, q,
 *                                      walk_length, seed, padding_idx, restart)
 * (csrc/cuda/rw_cuda_edge_list.h:4-12, called from csrc/rw_init.cpp:39-40) and produces what
 * walk_edge_list_cpu (csrc/cpu/rw_cpu_edge_list.cpp:240-265) produces.
 *
 *   edges            [n_edges][2] int64 (head, tail) sorted by head            (reference arg 1)
 *   node_edge_index  [n_nodes][2] int64 inclusive (first, last) edge per head,
 *                                 (-1,-1) for a node without out-edges          (reference arg 2)
 *   targets          [n_walkers]  int64 start nodes                             (reference arg 3)
 *   padding_idx, restart          a dead end steps to padding_idx; from padding_idx the walk goes to the
 *                                 start node (restart != 0) or stays on padding_idx
 *   out              [n_walkers][walk_length+1] int64
 * rng_mode as for the CSR walk (hops from single-edge nodes and from padding consume no draw, so
 * even the uniform walk's stream positions are data dependent): TRW_RNG_GLIBC replays the
 * reference's serial stream with one lane (walker_base == 0), TRW_RNG_WALKER is the per-walker
 * stream, TRW_RNG_AUTO picks GLIBC for walker_base == 0, n_walkers within the reference's
 * parallel_for grain (32768 uniform, 32768/(walk_length+1) biased) and n_walkers*walk_length <= 2^18.
 */
size_t trw_walk_edge_list_workspace_bytes(int64_t n_edges, int64_t n_nodes, int64_t n_walkers,
                                          int32_t walk_length, int64_t padding_idx, int32_t rng_mode);
int trw_walk_edge_list_hip(const int64_t *edges, int64_t n_edges,
                           const int64_t *node_edge_index, int64_t n_nodes,
                           const int64_t *targets, int64_t n_walkers, int64_t walker_base,
                           double p, double q, int32_t walk_length, int32_t seed,
                           int64_t padding_idx, int32_t restart, int32_t rng_mode,
                           int64_t *out, void *workspace, size_t workspace_bytes, void *hip_stream);

/*
 * The prepared edge list (cf. trw_prepare_csr_hip): uint32 shadow of tails and index; TRW_ELPREP_TABLE adds the edge
 * table {head, tail, positions of the tail in the head's range modulo 32, "not only the head's last edge"};
 * TRW_ELPREP_ENTRIES (implies TABLE) adds one 64-byte entry per edge with the next node's range and two 128-bit
 * masks ("the tail at this offset is a neighbour of the node I came from" / "... IS the node I came from") --
 * the biased walk then decides its trials in registers (elist_pool_kernel).  The blob depends on `padding_idx`.
 * trw_walk_edge_list_hip above prepares inside its workspace on every call; a caller that walks an edge list more
 * than once keeps the blob.  trw_walk_edge_list_prepared_hip needs a 256-byte workspace (the status block).
 */
#define TRW_ELPREP_TABLE   1
#define TRW_ELPREP_ENTRIES 2
int32_t trw_prepare_edge_list_plan(int64_t n_edges, int64_t n_nodes, int64_t n_walkers, int32_t walk_length,
                                   double p, double q, int64_t padding_idx, int32_t rng_mode);
size_t trw_prepare_edge_list_bytes(int64_t n_edges, int64_t n_nodes, int64_t padding_idx, int32_t what);
int trw_prepare_edge_list_hip(const int64_t *edges, int64_t n_edges, const int64_t *node_edge_index, int64_t n_nodes,
                              int64_t padding_idx, int32_t what, void *prepared, size_t prepared_bytes,
                              void *hip_stream);
int trw_walk_edge_list_prepared_hip(const void *prepared, size_t prepared_bytes, int32_t what,
                                    const int64_t *edges, int64_t n_edges,
                                    const int64_t *node_edge_index, int64_t n_nodes,
                                    const int64_t *targets, int64_t n_walkers, int64_t walker_base,
                                    double p, double q, int32_t walk_length, int32_t seed,
                                    int64_t padding_idx, int32_t restart, int32_t rng_mode,
                                    int64_t *out, void *workspace, size_t workspace_bytes, void *hip_stream);

/*
 * Window builders.  Replace the reference's to_windows_gpu / to_windows_cbow_gpu
 * (csrc/cuda/windows_cuda.h:5-17, called from csrc/rw_init.cpp:83-84 and :96-97) and produce what
 * to_windows_cpu / to_windows_cbow_cpu (csrc/cpu/windows_cpu.cpp:5-77, :80-158) produce.
 *
 *   walks [num_walks][walk_cols] int64, contiguous; W = num_walks * (walk_cols - window_size + 1) windows,
 *   walk-major; mid = window_size / 2.
 * to_windows:      target_nodes [W] (the middle node), pos_windows [W][window_size-1] (the rest of the
 *                  window), neg_windows [W][window_size-1] = rand() % num_nodes.  Exactly window_size-1 draws
 *                  per window, so the negatives are the reference's glibc stream itself: reference exact at
 *                  any size and for any shard (`walk_base` = global index of walks[0]).
 * to_windows_cbow: pos_nodes [W], windows [W][window_size-1], neg_nodes [W]: one negative per window, redrawn
 *                  while it equals the positive node (at most 101 times).  rng_mode: TRW_RNG_GLIBC replays the
 *                  reference's serial stream with one lane (window_base == 0), TRW_RNG_WALKER gives window
 *                  window_base + w its own stream, TRW_RNG_AUTO picks GLIBC up to 2^18 windows.
 */
size_t trw_to_windows_workspace_bytes(int64_t num_walks, int64_t walk_cols, int32_t window_size);
int trw_to_windows_hip(const int64_t *walks, int64_t num_walks, int64_t walk_cols, int32_t window_size,
                       int64_t num_nodes, int32_t seed, int64_t walk_base,
                       int64_t *target_nodes, int64_t *pos_windows, int64_t *neg_windows,
                       void *workspace, size_t workspace_bytes, void *hip_stream);
int trw_to_windows_cbow_hip(const int64_t *walks, int64_t num_walks, int64_t walk_cols, int32_t window_size,
                            int64_t num_nodes, int32_t seed, int64_t window_base, int32_t rng_mode,
                            int64_t *pos_nodes, int64_t *neg_nodes, int64_t *windows,
                            void *workspace, size_t workspace_bytes, void *hip_stream);

/*
 * Triples window builders.  Replace to_windows_triples_gpu / to_windows_triples_cbow_gpu
 * (csrc/cuda/windows_cuda.h:19-31, called from csrc/rw_init.cpp:111-112 and :126-127) and produce what
 * to_windows_triples_cpu / to_windows_triples_cbow_cpu (csrc/cpu/windows_cpu.cpp:160-314, :316-474) produce.
 *
 *   walks [num_walks][walk_cols] = start, (relation, tail)*; W = num_walks * ((walk_cols-1)/2) windows (one per hop)
 *   triples [num_triples][3]: the pool negatives are drawn from; padding_idx fills window rows past the walk's ends
 * to_windows_triples:      target_triples [W][3], pos_windows [W][2*window_size][3], neg_windows [W][2*window_size][3]
 *                          (2*window_size draws per window: reference exact at any size, `walk_base` for shards)
 * to_windows_triples_cbow: pos_triples [W][3], neg_triples [W][3] (redrawn while equal to the positive triple),
 *                          pos_windows [W][2*window_size][3]; rng_mode as trw_to_windows_cbow_hip.
 * The reference's `num_nodes` argument is unused there and has no counterpart here.
 */
size_t trw_to_windows_triples_workspace_bytes(int64_t num_walks, int64_t walk_cols, int32_t window_size);
int trw_to_windows_triples_hip(const int64_t *walks, int64_t num_walks, int64_t walk_cols, int32_t window_size,
                               int64_t padding_idx, const int64_t *triples, int64_t num_triples, int32_t seed,
                               int64_t walk_base, int64_t *target_triples, int64_t *pos_windows,
                               int64_t *neg_windows, void *workspace, size_t workspace_bytes, void *hip_stream);
int trw_to_windows_triples_cbow_hip(const int64_t *walks, int64_t num_walks, int64_t walk_cols, int32_t window_size,
                                    int64_t padding_idx, const int64_t *triples, int64_t num_triples, int32_t seed,
                                    int64_t window_base, int32_t rng_mode, int64_t *pos_triples,
                                    int64_t *neg_triples, int64_t *pos_windows, void *workspace,
                                    size_t workspace_bytes, void *hip_stream);

/*
 * The uniform walk and to_windows as ONE call (torch_rw/rw.py:3-4 followed by :29-30; csrc/cpu/rw_cpu.cpp:58-107 then
 * csrc/cpu/windows_cpu.cpp:5-77): target_nodes / pos_windows / neg_windows come out equal, bit for bit, to
 *     trw_to_windows_hip(trw_walk_csr_prepared_hip(..., p = q = 1, seed, TRW_RNG_GLIBC), window_size, num_nodes,
 *                        window_seed, walk_base = walker_base)
 * -- when trw_walk_windows_fused() says 1 for the shape, from a single walk kernel that writes the windows out of its
 * LDS tile (the [n_walkers, walk_length+1] rows never make the round trip through HBM; `walks` may then be NULL, or
 * a buffer that receives them as well); otherwise as the two passes, and `walks` is required.  `prepared` / `what`: a
 * prepared graph as for trw_walk_csr_prepared_hip.  Workspace: trw_walk_windows_workspace_bytes(); its first 256 bytes
 * are the walk's status block (trw_walk_status).
 */
int trw_walk_windows_fused(int64_t n_nodes, int64_t nnz, int64_t n_walkers, int32_t walk_length, int32_t window_size,
                           int32_t what);
size_t trw_walk_windows_workspace_bytes(int64_t n_walkers, int32_t walk_length, int32_t window_size);
int trw_walk_windows_csr_prepared_hip(const void *prepared, size_t prepared_bytes, int32_t what, const int64_t *row_ptr,
                                      int64_t n_nodes, const int64_t *col_idx, int64_t nnz, const int64_t *targets,
                                      int64_t n_walkers, int64_t walker_base, int32_t walk_length, int32_t seed,
                                      int32_t window_size, int64_t num_nodes, int32_t window_seed, int64_t *walks,
                                      int64_t *target_nodes, int64_t *pos_windows, int64_t *neg_windows, void *workspace,
                                      size_t workspace_bytes, void *hip_stream);

/*
 * Host legs of the rows next to the CSR walk: what the reference's dispatcher does with CPU tensors for its other
 * entry points (csrc/rw_init.cpp:27-131 -> walk_edge_list_cpu, triples::walk_triples_cpu, to_windows*_cpu).  HOST
 * pointers, synchronous, one thread, the reference's one glibc rand() stream (that is csrc/cpu on one thread);
 * arguments and result layouts as the *_hip forms above (no walker_base / rng_mode / workspace / stream).
 * `cbow` selects to_windows_cbow / to_windows_triples_cbow; `centre`/`rest`/`neg` are (target_nodes, pos_windows,
 * neg_windows) or, with cbow, (pos_nodes, windows, neg_nodes).  Data faults: TRW_E_DATA.
 */
int trw_walk_edge_list_cpu(const int64_t *edges, int64_t n_edges, const int64_t *node_edge_index, int64_t n_nodes,
                           const int64_t *targets, int64_t n_walkers, double p, double q, int32_t walk_length,
                           int32_t seed, int64_t padding_idx, int32_t restart, int64_t *out);
int trw_walk_triples_cpu(const int64_t *triples, int64_t n_triples, const int64_t *rt_index, int64_t n_entities,
                         const int64_t *targets, int64_t n_walkers, int32_t walk_length, int64_t padding_idx,
                         int32_t seed, int64_t *out);
int trw_to_windows_cpu(const int64_t *walks, int64_t num_walks, int64_t walk_cols, int32_t window_size,
                       int64_t num_nodes, int32_t seed, int32_t cbow, int64_t *centre, int64_t *rest, int64_t *neg);
int trw_to_windows_triples_cpu(const int64_t *walks, int64_t num_walks, int64_t walk_cols, int32_t window_size,
                               int64_t padding_idx, const int64_t *triples, int64_t num_triples, int32_t seed,
                               int32_t cbow, int64_t *target_triples, int64_t *pos_windows, int64_t *neg);

/*
 * Wire format of the multi-GPU gather (no reference counterpart: the reference is single-device,
 * csrc/cuda/rw_cuda.cu:198).  Node ids are < n_nodes, so while n_nodes <= 2^32 a block of rows travels as uint32 --
 * half the bytes per xGMI link -- and is widened where it lands.  `rows` / `wire`: `count` elements each, any
 * alignment (16-byte aligned blocks take the vector path); asynchronous on `hip_stream`.  `overflow` (optional,
 * device int64, zeroed by the caller): bit 0 is set when some value did not fit 32 bits.  The *_cpu forms are the
 * host legs (HOST pointers, synchronous).
 */
int trw_rows_narrow_hip(const int64_t *rows, int64_t count, uint32_t *wire, int64_t *overflow, void *hip_stream);
int trw_rows_widen_hip(const uint32_t *wire, int64_t count, int64_t *rows, void *hip_stream);
int trw_rows_narrow_cpu(const int64_t *rows, int64_t count, uint32_t *wire, int64_t *overflow);
int trw_rows_widen_cpu(const uint32_t *wire, int64_t count, int64_t *rows);

/* Synchronises `hip_stream` and copies the status block of the last walk that
 * used `workspace` to the host: status[0] = OR of TRW_ST_* bits,
 * status[1] = 1 if some CSR row was found unsorted (biased walks only; the
 * membership test then scans linearly like rw_cpu.cpp:32-56 instead of
 * bisecting), status[2] = the RNG mode that ran (TRW_RNG_GLIBC/WALKER),
 * status[3] = 0 int64 kernels, 1 uint32-shadow kernels, 2 shadow + edge hash table,
 * 3 shadow + fat entries (uniform walk, one gather per step).  Returns TRW_OK or TRW_E_HIP. */
int trw_walk_status(const void *workspace, void *hip_stream, int64_t status[4]);

/* Deferred form of the same check, for callers that must not synchronise (the reference's GPU entry
 * point launches and returns, csrc/cuda/rw_cuda.cu:223-245).
 * trw_status_post: enqueues on `hip_stream` an asynchronous copy of the status block of the call that just
 *   used `workspace` into a pinned host slot, followed by an event; never waits.  Does nothing while the
 *   stream is being captured into a HIP graph.
 * trw_status_collect: ORs into *bits the TRW_ST_* bits of every posted read-back that has completed since
 *   the last collect (wait != 0: first waits for all of them) -- each is reported once, process-wide --
 *   and returns how many read-backs are still in flight (>= 0) or a negative error code. */
int trw_status_post(const void *workspace, void *hip_stream);
int trw_status_collect(int wait, int64_t *bits);

/* Helpers for a host layer that keeps a prepared graph across calls and streams (the torch extension's
 * cache): is `hip_stream` being captured into a HIP graph (1/0); an event recorded on a stream (NULL on
 * failure), made a dependency of another stream, destroyed. */
int trw_stream_is_capturing(void *hip_stream);
void *trw_event_record(void *hip_stream);
int trw_event_wait(void *event, void *hip_stream);
void trw_event_destroy(void *event);

/* Integer form of the reference's accept test (rw_cpu.cpp:169-194):
 * (double)r / (double)RAND_MAX < prob  <=>  r < trw_accept_threshold(prob).
 * Host helper, exported so tests can pin it against the fp64 expression. */
uint32_t trw_accept_threshold(double prob);

/* Host helper: fills window[31] with the glibc TYPE_3 generator words in front
 * of draw index `pos` after srand(seed), using the library's jump tables
 * (the same tables the device uses).  Exported for tests. */
void trw_glibc_window_at(int32_t seed, uint64_t pos, uint32_t window[31]);

/* Instrumentation for bench.py (not thread-safe, one call in flight).
 * trw_profile_enable(flags): bit 0 -- every trw_walk_csr_hip records HIP events
 * on its stream around its three phases; trw_profile_last() synchronises the
 * last call's events and returns, in milliseconds, the device time of
 *   prepare  (uint32 shadow CSR + validation; glibc jump-ahead windows),
 *   walk     (the walk kernel itself -- the dominant kernel),
 *   finish   (always ~0: every walk kernel writes the caller's rows itself;
 *            kept so that the three-phase report stays stable).
 * bit 1 -- the biased walker kernel also counts its work into the workspace
 * (a separately compiled variant, not the one that is timed);
 * trw_walk_counters() synchronises and returns
 *   [0] rejection trials, [1] candidate entries loaded from col_idx,
 *   [2] membership tests performed, [3] col_idx entries read by bisections
 *   (membership tests and the once-per-step location of t in prev's row). */
int trw_profile_enable(int flags);
int trw_profile_last(double *prepare_ms, double *walk_ms, double *finish_ms);
/* Device time of the last trw_prepare_csr_hip since profiling was enabled (reported once; TRW_E_ARG if there
 * was none -- e.g. the call found its graph in the host layer's cache). */
int trw_profile_prepare_last(double *prepare_ms);
int trw_walk_counters(const void *workspace, void *hip_stream, uint64_t counters[4]);

/* How often a window-builder call was handed a side lane (a second stream for its negatives, beside the main
 * kernel) since the library was loaded: counts[0] = all calls that got one, counts[1] = those whose stream was being
 * captured into a HIP graph at the time (the lane joins the capture through its fork event).  For tests. */
void trw_side_lane_counts(int64_t counts[2]);

/* Self-test of the device arithmetic the walk kernels are built on, over EVERY divisor a row length can be
 * (1 .. 2^31-1): the exact reciprocal of the walker pools' one-correction `draw % degree` (the stand-in for the
 * reference's `rand() % (row_end - row_start)`, csrc/cpu/rw_cpu.cpp:20), the two-correction form of the other
 * kernels, both against the hardware's own %.  `scratch` = 16 bytes of device memory, 16-byte aligned.  Synchronises;
 * *failures = number of divisors / draws that disagreed (0 on a healthy build). */
int trw_selftest_hip(void *scratch, void *hip_stream, uint64_t *failures);

#ifdef __cplusplus
}
#endif
#endif /* TRW_H */
