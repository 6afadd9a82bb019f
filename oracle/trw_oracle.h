/*
 * trw_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's CSR random-walk hot path
 * (Sachinx0e/torch_random_walk, csrc/cpu/rw_cpu.cpp).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (torch_random_walk_amd/) never links or calls it.
 *
 * Parity pin: the restatement reproduces the reference's own golden vectors
 * (tests/test_rw.py:49-53 uniform, :116-120 biased; committed as data under
 * tests/golden/) and its glibc-rand() restatement is checked draw-for-draw
 * against the real libc srand()/rand() of this image (glibc 2.35), which is
 * the third-party dependency the reference calls at rw_cpu.cpp:20,60,118,169.
 * oracle/_ref (a build of the reference's own sources) is NOT available:
 * rw_cpu.cpp:5 includes ../cuda/utils.cuh, which includes <curand.h>, a CUDA
 * header this ROCm image does not ship.
 */
#ifndef TRW_ORACLE_H
#define TRW_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* RNG modes of the oracle walker. */
enum {
    /* One process-global glibc TYPE_3 stream seeded once by srand(seed) and
     * consumed walker-major, exactly as rw_cpu.cpp does when parallel_for
     * runs inline (1 thread, or N <= grain).  Restated generator. */
    TRW_ORACLE_RNG_GLIBC_SEQ = 0,
    /* Per-walker counter stream: Philox4x32-10, key (seed, 0), counter
     * (walker_lo, walker_hi, draw/4, 0), draw value = word[draw%4] >> 1.
     * The contract the parallel HIP biased kernel is held to. */
    TRW_ORACLE_RNG_WALKER = 1,
    /* Same walk as GLIBC_SEQ but drawing from the real libc srand()/rand().
     * Pins the restatement; also the timed "csrc/cpu" baseline. */
    TRW_ORACLE_RNG_LIBC = 2,
    /* Uniform walks only: draw (walker*L + step-1) of the glibc stream is
     * fetched by O(log) jump-ahead per walker.  Must equal GLIBC_SEQ. */
    TRW_ORACLE_RNG_GLIBC_INDEXED = 3
};

/* Error codes (negative). */
enum {
    TRW_ORACLE_OK = 0,
    TRW_ORACLE_E_ARG = -1,       /* bad argument                                  */
    TRW_ORACLE_E_DEGREE0 = -2,   /* a walker stood on a node with no out-edges
                                    (the reference dies with SIGFPE there,
                                    rw_cpu.cpp:20)                                */
    TRW_ORACLE_E_RANGE = -3      /* node id outside [0, n_nodes)                  */
};

/* Work counters for the roofline's ALGORITHMIC byte count. */
typedef struct {
    uint64_t draws_total;        /* RNG draws consumed                            */
    uint64_t trials_total;       /* rejection trials (biased, steps >= 2)         */
    uint64_t membership_tests;   /* is_neighbor calls (candidate != t)            */
    uint64_t scan_entries_total; /* col_idx entries the reference's linear
                                    early-exit scan reads (rw_cpu.cpp:47-52)      */
} trw_oracle_counters;

/* glibc srand(seed); then n successive rand() values starting at draw index
 * `start` (0-based).  Restated TYPE_3 generator, no libc call. */
void trw_oracle_glibc_stream(int32_t seed, uint64_t start, uint64_t n, int32_t *out);

/* The same through the real libc (start is always 0).  NOT re-entrant. */
void trw_oracle_libc_stream(int32_t seed, uint64_t n, int32_t *out);

/* Jump-ahead: the 31-word generator window in front of draw index `pos`
 * (window[j] = r[pos + 313 + j]), computed by polynomial jump, not stepping. */
void trw_oracle_glibc_window_at(int32_t seed, uint64_t pos, uint32_t window[31]);

/* One Philox4x32-10 block. */
void trw_oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* Draw `draw` of walker `walker` under TRW_ORACLE_RNG_WALKER (31-bit). */
int32_t trw_oracle_walker_draw(int32_t seed, uint64_t walker, uint64_t draw);

/* The normalised acceptance probabilities of rw_cpu.cpp:130-134. */
void trw_oracle_probs(double p, double q, double prob[3]);

/* Smallest r in [0, 2^31] with (double)r / (double)RAND_MAX >= prob, so that
 * the reference's test `(double)rand()/RAND_MAX < prob` (rw_cpu.cpp:169-176)
 * is exactly `r < threshold`.  Used to check the HIP host's integer form. */
uint32_t trw_oracle_accept_threshold(double prob);

/*
 * walk_cpu restated (rw_cpu.cpp:203-226 -> uniform_walk :58-107 or
 * biased_walk :109-201, sample_neighbor :7-30, is_neighbor :32-56).
 *
 *   out              [n_walkers][walk_length+1] row-major int64
 *   walker_base      global index of targets[0] (stream offset for
 *                    GLIBC_INDEXED / key for WALKER); must be 0 for the
 *                    sequential modes of a biased walk
 *   counters         nullable
 *   draws_per_walker nullable, [n_walkers]
 */
int trw_oracle_walk_csr(const int64_t *row_ptr, int64_t n_nodes,
                        const int64_t *col_idx, int64_t nnz,
                        const int64_t *targets, int64_t n_walkers,
                        int64_t walker_base,
                        double p, double q, int32_t walk_length, int32_t seed,
                        int32_t rng_mode,
                        int64_t *out,
                        trw_oracle_counters *counters,
                        int64_t *draws_per_walker);

/* TIMING only: the reference's multi-threaded execution policy (at::parallel_for chunks, all threads on the
 * one locked libc rand()); walks depend on the thread interleaving.  *seconds = wall time of the walk,
 * *threads_used = threads at::parallel_for would have started.  NOT re-entrant. */
int trw_oracle_walk_csr_mt(const int64_t *row_ptr, int64_t n_nodes, const int64_t *col_idx, int64_t nnz,
                           const int64_t *targets, int64_t n_walkers, double p, double q, int32_t walk_length,
                           int32_t seed, int32_t n_threads, int64_t *out, double *seconds, int32_t *threads_used);

/*
 * triples::walk_triples_cpu restated (csrc/cpu/rw_cpu_triples.cpp:105-127 ->
 * uniform_walk_triples :48-103, triples::sample_neighbor :11-46, sample_int
 * csrc/cpu/cpu_utils.cpp:3-10).
 *
 *   triples   [n_triples][3] int64 (head, relation, tail), sorted by head
 *   rt_index  [n_entities][2] int64 inclusive (first, last) triple of each
 *             head, (-1,-1) for entities without out-edges
 *   out       [n_walkers][2*walk_length+1]: start, then (relation, tail) per hop
 * A hop from the padding id, or from an entity without out-edges, yields
 * (padding, padding) and consumes no draw; a single-edge entity consumes no
 * draw either (sample_int returns early when start == end).
 * rng_mode: GLIBC_SEQ / LIBC (serial stream, walker_base must be 0) or WALKER
 * (draw k = the k-th draw this walker actually consumes).
 */
int trw_oracle_walk_triples(const int64_t *triples, int64_t n_triples,
                            const int64_t *rt_index, int64_t n_entities,
                            const int64_t *targets, int64_t n_walkers,
                            int64_t walker_base, int32_t walk_length,
                            int64_t padding_idx, int32_t seed, int32_t rng_mode,
                            int64_t *out, uint64_t *draws_total);

/*
 * walk_edge_list_cpu restated (csrc/cpu/rw_cpu_edge_list.cpp:240-265 ->
 * uniform_walk_edge_list :64-126 or biased_walk_edge_list :128-238,
 * sample_neighbor :8-35, is_neighbor :37-62, sample_int cpu_utils.cpp:3-10).
 *
 *   edges    [n_edges][2] int64 (head, tail), sorted by head
 *   index    [n_nodes][2] int64 inclusive (first, last) edge per head, (-1,-1) if none
 *   out      [n_walkers][walk_length+1]
 * Kept from the reference: a dead-end node steps to padding_idx; a step from
 * padding_idx goes to the jump node (the start node if `restart`, else padding)
 * without a draw; is_neighbor scans [first, last) -- it never looks at a head's
 * LAST edge (:51); in the biased loop the `new == t` test is a separate `if`, so a
 * rejected return to t is tested again as neighbour / non-neighbour (:204-231).
 * Where the reference indexes node_edge_index with the padding id (t == padding,
 * an out-of-bounds read when padding_idx >= n_nodes) this restatement answers
 * "not a neighbour".
 */
int trw_oracle_walk_edge_list(const int64_t *edges, int64_t n_edges,
                              const int64_t *index, int64_t n_nodes,
                              const int64_t *targets, int64_t n_walkers,
                              int64_t walker_base, double p, double q,
                              int32_t walk_length, int32_t seed,
                              int64_t padding_idx, int32_t restart, int32_t rng_mode,
                              int64_t *out, uint64_t *draws_total);

/*
 * to_windows_cpu / to_windows_cbow_cpu restated (csrc/cpu/windows_cpu.cpp:5-77 and :80-158),
 * single-thread semantics (one pass over the walks in order).
 *
 *   walks [num_walks][walk_cols]; windows per walk = walk_cols - window_size + 1; mid = window_size/2
 * to_windows:      target_nodes [W], pos_windows [W][window_size-1] (the window without its middle),
 *                  neg_windows [W][window_size-1] = rand() % num_nodes, window_size-1 draws per window.
 * to_windows_cbow: pos_nodes [W], windows [W][window_size-1], neg_nodes [W] = sample_int(0,num_nodes-1)
 *                  redrawn while it equals the positive node, at most 101 times (:129-135).
 * rng_mode: GLIBC_SEQ / LIBC, or (cbow only) WALKER = one stream per window, keyed window_base + w.
 */
int trw_oracle_to_windows(const int64_t *walks, int64_t num_walks, int64_t walk_cols, int32_t window_size,
                          int64_t num_nodes, int32_t seed, int32_t rng_mode,
                          int64_t *target_nodes, int64_t *pos_windows, int64_t *neg_windows);
int trw_oracle_to_windows_cbow(const int64_t *walks, int64_t num_walks, int64_t walk_cols, int32_t window_size,
                               int64_t num_nodes, int32_t seed, int32_t rng_mode, int64_t window_base,
                               int64_t *pos_nodes, int64_t *neg_nodes, int64_t *windows);

/*
 * to_windows_triples_cpu / to_windows_triples_cbow_cpu restated (csrc/cpu/windows_cpu.cpp:160-314, :316-474).
 *
 *   walks [num_walks][walk_cols] = start, (relation, tail)*; one window per hop: W = num_walks*((walk_cols-1)/2)
 *   pos_windows [W][2*window_size][3]: window_size triples to the left then window_size to the right of the
 *   target triple, padding_idx where the walk ends.  Kept from the reference: the left half writes
 *   window_size+1 rows (the extra one is overwritten by the first right row) and stores walk[rel] in the HEAD
 *   slot of left rows (:221-225).
 * to_windows_triples:      target_triples [W][3], neg_windows [W][2*window_size][3] = uniformly drawn rows of
 *                          `triples` (sample_int(0, num_triples-1), one draw each unless num_triples == 1)
 * to_windows_triples_cbow: pos_triples [W][3], neg_triples [W][3]: one drawn row, redrawn while it equals
 *                          the positive triple (at most 101 times)
 * rng_mode: GLIBC_SEQ / LIBC, or (cbow only) WALKER keyed window_base + w.
 */
int trw_oracle_to_windows_triples(const int64_t *walks, int64_t num_walks, int64_t walk_cols, int32_t window_size,
                                  int64_t padding_idx, const int64_t *triples, int64_t num_triples, int32_t seed,
                                  int32_t rng_mode, int64_t *target_triples, int64_t *pos_windows,
                                  int64_t *neg_windows);
int trw_oracle_to_windows_triples_cbow(const int64_t *walks, int64_t num_walks, int64_t walk_cols,
                                       int32_t window_size, int64_t padding_idx, const int64_t *triples,
                                       int64_t num_triples, int32_t seed, int32_t rng_mode, int64_t window_base,
                                       int64_t *pos_triples, int64_t *neg_triples, int64_t *pos_windows);

#ifdef __cplusplus
}
#endif
#endif
