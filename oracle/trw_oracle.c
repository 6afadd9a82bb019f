/*
 * trw_oracle.c -- TEST INFRASTRUCTURE ONLY (see trw_oracle.h).
 *
 * Plain-C restatement of the reference's CSR random-walk hot path.  Every
 * function cites the reference lines it follows (paths relative to the
 * reference repo root).  Nothing here is compiled into, linked by or called
 * from the product library.
 */
#include "trw_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* glibc srand()/rand(), TYPE_3 additive feedback generator                  */
/* (GNU libc stdlib/random_r.c; called by the reference at                   */
/*  csrc/cpu/rw_cpu.cpp:20,60,118,169).  r[k] = r[k-3] + r[k-31] mod 2^32,   */
/*  output r[k] >> 1, 310 values discarded after seeding.                    */
/* ------------------------------------------------------------------------ */

#define GL_DEG 31
#define GL_SEP 3

typedef struct {
    uint32_t w[GL_DEG]; /* w[(pos+j) % 31] = r[pos + 313 + j], j = 0..30      */
    uint32_t slot;      /* pos % 31                                           */
} glibc_state;

static void glibc_seed_window(int32_t seed, uint32_t win[GL_DEG])
{
    /* random_r.c __srandom_r: seed 0 becomes 1; Schrage's 16807 LCG fills
     * r[1..30] with C's truncating signed division on the int32 word. */
    uint32_t r[344];
    uint32_t s = (uint32_t)seed;
    if (s == 0) s = 1;
    r[0] = s;
    int32_t word = (int32_t)s;
    for (int i = 1; i < GL_DEG; ++i) {
        long hi = word / 127773;
        long lo = word % 127773;
        word = (int32_t)(16807 * lo - 2836 * hi);
        if (word < 0) word += 2147483647;
        r[i] = (uint32_t)word;
    }
    for (int i = 31; i < 34; ++i) r[i] = r[i - 31];
    for (int i = 34; i < 344; ++i) r[i] = r[i - 31] + r[i - 3];
    for (int j = 0; j < GL_DEG; ++j) win[j] = r[313 + j];
}

static void glibc_init(glibc_state *st, int32_t seed)
{
    glibc_seed_window(seed, st->w);
    st->slot = 0;
}

static inline int32_t glibc_next(glibc_state *st)
{
    uint32_t j = st->slot;
    uint32_t k = j + (GL_DEG - GL_SEP);
    if (k >= GL_DEG) k -= GL_DEG;
    uint32_t v = st->w[j] + st->w[k]; /* r[k] = r[k-31] + r[k-3] */
    st->w[j] = v;
    st->slot = (j + 1 == GL_DEG) ? 0 : j + 1;
    return (int32_t)(v >> 1);
}

/* Jump-ahead.  The recurrence is linear over Z/2^32 with characteristic
 * polynomial x^31 = x^28 + 1, so r[q+J+m] = sum_i g_i r[q+i+m] where
 * g = x^J mod (x^31 - x^28 - 1). */
static void poly_mulmod(const uint32_t a[GL_DEG], const uint32_t b[GL_DEG], uint32_t out[GL_DEG])
{
    uint32_t c[2 * GL_DEG - 1];
    memset(c, 0, sizeof c);
    for (int i = 0; i < GL_DEG; ++i) {
        if (!a[i]) continue;
        for (int j = 0; j < GL_DEG; ++j) c[i + j] += a[i] * b[j];
    }
    for (int d = 2 * GL_DEG - 2; d >= GL_DEG; --d) {
        c[d - GL_SEP] += c[d];
        c[d - GL_DEG] += c[d];
    }
    memcpy(out, c, GL_DEG * sizeof(uint32_t));
}

static void poly_xpow(uint64_t J, uint32_t g[GL_DEG])
{
    uint32_t base[GL_DEG], acc[GL_DEG], tmp[GL_DEG];
    memset(base, 0, sizeof base);
    memset(acc, 0, sizeof acc);
    base[1] = 1; /* x   */
    acc[0] = 1;  /* 1   */
    while (J) {
        if (J & 1) { poly_mulmod(acc, base, tmp); memcpy(acc, tmp, sizeof acc); }
        J >>= 1;
        if (J) { poly_mulmod(base, base, tmp); memcpy(base, tmp, sizeof base); }
    }
    memcpy(g, acc, sizeof acc);
}

static void window_apply(const uint32_t g[GL_DEG], const uint32_t win[GL_DEG], uint32_t out[GL_DEG])
{
    uint32_t ext[2 * GL_DEG - 1];
    memcpy(ext, win, GL_DEG * sizeof(uint32_t));
    for (int t = 0; t < GL_DEG - 1; ++t) ext[GL_DEG + t] = ext[t] + ext[t + GL_DEG - GL_SEP];
    for (int m = 0; m < GL_DEG; ++m) {
        uint32_t acc = 0;
        for (int i = 0; i < GL_DEG; ++i) acc += g[i] * ext[i + m];
        out[m] = acc;
    }
}

void trw_oracle_glibc_window_at(int32_t seed, uint64_t pos, uint32_t window[31])
{
    uint32_t w0[GL_DEG], g[GL_DEG];
    glibc_seed_window(seed, w0);
    poly_xpow(pos, g);
    window_apply(g, w0, window);
}

static void glibc_init_at(glibc_state *st, int32_t seed, uint64_t pos)
{
    if (pos == 0) { glibc_init(st, seed); return; }
    trw_oracle_glibc_window_at(seed, pos, st->w);
    st->slot = 0;
}

void trw_oracle_glibc_stream(int32_t seed, uint64_t start, uint64_t n, int32_t *out)
{
    glibc_state st;
    glibc_init_at(&st, seed, start);
    for (uint64_t i = 0; i < n; ++i) out[i] = glibc_next(&st);
}

void trw_oracle_libc_stream(int32_t seed, uint64_t n, int32_t *out)
{
    srand((unsigned)seed);
    for (uint64_t i = 0; i < n; ++i) out[i] = rand();
}

/* ------------------------------------------------------------------------ */
/* Philox4x32-10 (Salmon et al., SC'11) -- the per-walker counter stream.    */
/* ------------------------------------------------------------------------ */

void trw_oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

int32_t trw_oracle_walker_draw(int32_t seed, uint64_t walker, uint64_t draw)
{
    uint32_t ctr[4] = { (uint32_t)walker, (uint32_t)(walker >> 32), (uint32_t)(draw >> 2), 0u };
    uint32_t key[2] = { (uint32_t)seed, 0u };
    uint32_t o[4];
    trw_oracle_philox4x32_10(ctr, key, o);
    return (int32_t)(o[draw & 3] >> 1);
}

/* ------------------------------------------------------------------------ */
/* RNG front-end shared by the walk loops                                     */
/* ------------------------------------------------------------------------ */

typedef struct {
    int mode;
    glibc_state gl;
    /* walker stream */
    int32_t seed;
    uint64_t walker;
    uint64_t draw;
    uint32_t blk[4];
    uint64_t blk_index; /* which philox block is cached, ~0 = none */
    uint64_t ndraws;
} rng_t;

static inline int32_t rng_next(rng_t *g)
{
    g->ndraws++;
    switch (g->mode) {
    case TRW_ORACLE_RNG_LIBC:
        return rand();
    case TRW_ORACLE_RNG_WALKER: {
        uint64_t d = g->draw++;
        uint64_t b = d >> 2;
        if (b != g->blk_index) {
            uint32_t ctr[4] = { (uint32_t)g->walker, (uint32_t)(g->walker >> 32), (uint32_t)b, 0u };
            uint32_t key[2] = { (uint32_t)g->seed, 0u };
            trw_oracle_philox4x32_10(ctr, key, g->blk);
            g->blk_index = b;
        }
        return (int32_t)(g->blk[d & 3] >> 1);
    }
    default:
        return glibc_next(&g->gl);
    }
}

static inline void rng_begin_walker(rng_t *g, uint64_t walker, int32_t walk_length)
{
    if (g->mode == TRW_ORACLE_RNG_WALKER) {
        g->walker = walker;
        g->draw = 0;
        g->blk_index = ~(uint64_t)0;
    } else if (g->mode == TRW_ORACLE_RNG_GLIBC_INDEXED) {
        glibc_init_at(&g->gl, g->seed, walker * (uint64_t)walk_length);
    }
}

/* ------------------------------------------------------------------------ */
/* The walk                                                                   */
/* ------------------------------------------------------------------------ */

void trw_oracle_probs(double p, double q, double prob[3])
{
    /* rw_cpu.cpp:130-134 */
    double max_prob_init = fmax(1.0 / p, 1);
    double max_prob = fmax(max_prob_init, 1.0 / q);
    prob[0] = 1.0 / p / max_prob;
    prob[1] = 1.0 / max_prob;
    prob[2] = 1.0 / q / max_prob;
}

uint32_t trw_oracle_accept_threshold(double prob)
{
    /* (double)r / (double)RAND_MAX is monotone non-decreasing in r, so the
     * accept set { r : r/RAND_MAX < prob } is a prefix [0, T). */
    const double rmax = (double)RAND_MAX;
    if (!(prob > 0.0)) return 0; /* also NaN */
    uint64_t lo = 0, hi = (uint64_t)1 << 31; /* answer in [lo, hi] */
    while (lo < hi) {
        uint64_t mid = (lo + hi) >> 1;
        if ((double)mid / rmax < prob) lo = mid + 1; else hi = mid;
    }
    return (uint32_t)lo;
}

typedef struct {
    const int64_t *row_ptr;
    const int64_t *col_idx;
    int64_t n_nodes, nnz;
    int err;
} graph_t;

/* sample_neighbor, rw_cpu.cpp:7-30.  rand() is promoted to int64 before the
 * modulo; the bounds check falls back to the current node. */
static inline int64_t sample_neighbor(graph_t *g, int64_t node, rng_t *rng)
{
    if (node < 0 || node >= g->n_nodes) { g->err = TRW_ORACLE_E_RANGE; return node; }
    int64_t column_start = g->row_ptr[node];
    int64_t column_end = g->row_ptr[node + 1];
    int64_t deg = column_end - column_start;
    if (deg == 0) { /* the reference divides by zero here (SIGFPE) */
        g->err = TRW_ORACLE_E_DEGREE0;
        (void)rng_next(rng);
        return node;
    }
    int64_t nbr_idx = column_start + ((int64_t)rng_next(rng) % deg);
    if (nbr_idx >= 0 && nbr_idx < g->nnz) return g->col_idx[nbr_idx];
    return node;
}

/* is_neighbor(new_node, previous_node), rw_cpu.cpp:32-56: linear early-exit
 * scan of previous_node's row for new_node. */
static inline int is_neighbor(graph_t *g, int64_t new_node, int64_t previous_node, uint64_t *scanned)
{
    if (previous_node < 0 || previous_node >= g->n_nodes) { g->err = TRW_ORACLE_E_RANGE; return 0; }
    int64_t column_start = g->row_ptr[previous_node];
    int64_t column_end = g->row_ptr[previous_node + 1];
    for (int64_t i = column_start; i < column_end; ++i) {
        if (scanned) ++*scanned;
        if (g->col_idx[i] == new_node) return 1;
    }
    return 0;
}

/* One walker: uniform_walk's hot loop (rw_cpu.cpp:95-104) or biased_walk's (rw_cpu.cpp:159-198). */
static void walk_one(graph_t *g, rng_t *rng, int uniform, const double prob[3], int64_t target_node,
                     int64_t walk_size, int64_t *walk, trw_oracle_counters *cnt, uint64_t *scanned)
{
    walk[0] = target_node; /* rw_cpu.cpp:92 / :156 */
    if (uniform) {
        int64_t previous_node = target_node;
        for (int64_t s = 1; s < walk_size; ++s) {
            int64_t next_node = sample_neighbor(g, previous_node, rng);
            walk[s] = next_node;
            previous_node = next_node;
        }
        return;
    }
    walk[1] = sample_neighbor(g, target_node, rng);
    int64_t previous_node = walk[1];
    for (int64_t s = 2; s < walk_size; ++s) {
        int64_t selected_node = -1;
        int64_t t_node = walk[s - 2];
        while (1) {
            /* both draws are always consumed, candidate first */
            int64_t new_node = sample_neighbor(g, previous_node, rng);
            double random_prob = ((double)rng_next(rng) / (double)RAND_MAX);
            cnt->trials_total++;
            if (g->err) { selected_node = new_node; break; }
            if (new_node == t_node) {
                if (random_prob < prob[0]) { selected_node = new_node; break; }
            } else {
                cnt->membership_tests++;
                if (is_neighbor(g, new_node, t_node, scanned)) {
                    if (random_prob < prob[1]) { selected_node = new_node; break; }
                } else if (random_prob < prob[2]) { selected_node = new_node; break; }
            }
        }
        walk[s] = selected_node;
        previous_node = selected_node;
    }
}

int trw_oracle_walk_csr(const int64_t *row_ptr, int64_t n_nodes,
                        const int64_t *col_idx, int64_t nnz,
                        const int64_t *targets, int64_t n_walkers,
                        int64_t walker_base,
                        double p, double q, int32_t walk_length, int32_t seed,
                        int32_t rng_mode,
                        int64_t *out,
                        trw_oracle_counters *counters,
                        int64_t *draws_per_walker)
{
    if (!row_ptr || (!col_idx && nnz) || (!targets && n_walkers) || (!out && n_walkers)) return TRW_ORACLE_E_ARG;
    if (n_nodes < 0 || nnz < 0 || n_walkers < 0 || walker_base < 0 || walk_length < 1) return TRW_ORACLE_E_ARG;
    if (rng_mode < 0 || rng_mode > 3) return TRW_ORACLE_E_ARG;

    /* rw_cpu.cpp:220: exact-equality dispatch. */
    const int uniform = (p == 1.0 && q == 1.0);
    const int sequential = (rng_mode == TRW_ORACLE_RNG_GLIBC_SEQ || rng_mode == TRW_ORACLE_RNG_LIBC);
    if (rng_mode == TRW_ORACLE_RNG_GLIBC_INDEXED && !uniform) return TRW_ORACLE_E_ARG;
    if (sequential && !uniform && walker_base != 0) return TRW_ORACLE_E_ARG;
    if (rng_mode == TRW_ORACLE_RNG_LIBC && walker_base != 0) return TRW_ORACLE_E_ARG;

    graph_t g = { row_ptr, col_idx, n_nodes, nnz, 0 };
    rng_t rng;
    memset(&rng, 0, sizeof rng);
    rng.mode = rng_mode;
    rng.seed = seed;
    /* rw_cpu.cpp:60 / :118: the stream is seeded once per call. */
    if (rng_mode == TRW_ORACLE_RNG_LIBC) srand((unsigned)seed);
    else if (rng_mode == TRW_ORACLE_RNG_GLIBC_SEQ)
        glibc_init_at(&rng.gl, seed, (uint64_t)walker_base * (uint64_t)walk_length);

    double prob[3];
    trw_oracle_probs(p, q, prob);

    trw_oracle_counters cnt;
    memset(&cnt, 0, sizeof cnt);
    uint64_t *scanned = counters ? &cnt.scan_entries_total : NULL;

    const int64_t walk_size = (int64_t)walk_length + 1; /* rw_cpu.cpp:216 */

    for (int64_t i = 0; i < n_walkers; ++i) {
        uint64_t draws_before = rng.ndraws;
        rng_begin_walker(&rng, (uint64_t)walker_base + (uint64_t)i, walk_length);
        walk_one(&g, &rng, uniform, prob, targets[i], walk_size, out + i * walk_size, &cnt, scanned);
        if (draws_per_walker) draws_per_walker[i] = (int64_t)(rng.ndraws - draws_before);
        if (g.err) break;
    }
    cnt.draws_total = rng.ndraws;
    if (counters) *counters = cnt;
    return g.err;
}

/* ------------------------------------------------------------------------ */
/* The reference's multi-threaded run, for TIMING only                         */
/* ------------------------------------------------------------------------ */
/* walk_cpu runs its loop under at::parallel_for(0, N, grain) with every thread drawing from the one
 * process-global, lock-protected libc rand() (rw_cpu.cpp:62-63 grain 32768, :137 grain 32768/(L+1)):
 * the output depends on the interleaving, so this entry point exists to time that execution policy on
 * the host next to the single-thread run (bench.py's cpu_baseline), not to produce comparable walks. */
#include <pthread.h>
#include <time.h>

typedef struct {
    graph_t g;
    const int64_t *targets;
    int64_t begin, end, walk_size;
    int uniform;
    const double *prob;
    int64_t *out;
} mt_chunk;

static void *mt_worker(void *arg)
{
    mt_chunk *c = (mt_chunk *)arg;
    rng_t rng;
    memset(&rng, 0, sizeof rng);
    rng.mode = TRW_ORACLE_RNG_LIBC;
    trw_oracle_counters cnt;
    memset(&cnt, 0, sizeof cnt);
    for (int64_t i = c->begin; i < c->end && !c->g.err; ++i)
        walk_one(&c->g, &rng, c->uniform, c->prob, c->targets[i], c->walk_size, c->out + i * c->walk_size, &cnt, NULL);
    return NULL;
}

int trw_oracle_walk_csr_mt(const int64_t *row_ptr, int64_t n_nodes, const int64_t *col_idx, int64_t nnz,
                           const int64_t *targets, int64_t n_walkers, double p, double q, int32_t walk_length,
                           int32_t seed, int32_t n_threads, int64_t *out, double *seconds, int32_t *threads_used)
{
    if (!row_ptr || (!col_idx && nnz) || (!targets && n_walkers) || (!out && n_walkers)) return TRW_ORACLE_E_ARG;
    if (n_nodes < 0 || nnz < 0 || n_walkers < 0 || walk_length < 1 || n_threads < 1 || n_threads > 1024)
        return TRW_ORACLE_E_ARG;
    const int uniform = (p == 1.0 && q == 1.0);
    double prob[3];
    trw_oracle_probs(p, q, prob);
    const int64_t walk_size = (int64_t)walk_length + 1;
    int64_t grain = uniform ? 32768 : 32768 / walk_size;
    if (grain < 1) grain = 1;
    /* at::parallel_for: serial when the range fits one grain, else at most ceil(N / grain) threads with
     * equal contiguous chunks */
    int64_t nt = 1;
    if (n_walkers > grain && n_threads > 1) {
        nt = (n_walkers + grain - 1) / grain;
        if (nt > n_threads) nt = n_threads;
    }
    const int64_t chunk = (n_walkers + nt - 1) / nt;
    mt_chunk chunks[1024];
    pthread_t tid[1024];
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    srand((unsigned)seed);
    for (int64_t k = 0; k < nt; ++k) {
        mt_chunk *c = &chunks[k];
        c->g = (graph_t){ row_ptr, col_idx, n_nodes, nnz, 0 };
        c->targets = targets;
        c->begin = k * chunk;
        c->end = (k + 1) * chunk < n_walkers ? (k + 1) * chunk : n_walkers;
        c->walk_size = walk_size;
        c->uniform = uniform;
        c->prob = prob;
        c->out = out;
        if (nt == 1) mt_worker(c);
        else if (pthread_create(&tid[k], NULL, mt_worker, c) != 0) return TRW_ORACLE_E_ARG;
    }
    int err = 0;
    for (int64_t k = 0; k < nt; ++k) {
        if (nt > 1) pthread_join(tid[k], NULL);
        if (chunks[k].g.err) err = chunks[k].g.err;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (seconds) *seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    if (threads_used) *threads_used = (int32_t)nt;
    return err;
}

/* ------------------------------------------------------------------------ */
/* Triples (knowledge-graph) walk                                             */
/* ------------------------------------------------------------------------ */

int trw_oracle_walk_triples(const int64_t *triples, int64_t n_triples,
                            const int64_t *rt_index, int64_t n_entities,
                            const int64_t *targets, int64_t n_walkers,
                            int64_t walker_base, int32_t walk_length,
                            int64_t padding_idx, int32_t seed, int32_t rng_mode,
                            int64_t *out, uint64_t *draws_total)
{
    if ((!triples && n_triples) || (!rt_index && n_entities) || (!targets && n_walkers) || (!out && n_walkers))
        return TRW_ORACLE_E_ARG;
    if (n_triples < 0 || n_entities < 0 || n_walkers < 0 || walker_base < 0 || walk_length < 1) return TRW_ORACLE_E_ARG;
    if (rng_mode != TRW_ORACLE_RNG_GLIBC_SEQ && rng_mode != TRW_ORACLE_RNG_LIBC && rng_mode != TRW_ORACLE_RNG_WALKER)
        return TRW_ORACLE_E_ARG;
    if (rng_mode != TRW_ORACLE_RNG_WALKER && walker_base != 0) return TRW_ORACLE_E_ARG;

    rng_t rng;
    memset(&rng, 0, sizeof rng);
    rng.mode = rng_mode;
    rng.seed = seed;
    if (rng_mode == TRW_ORACLE_RNG_LIBC) srand((unsigned)seed); /* rw_cpu_triples.cpp:57 */
    else if (rng_mode == TRW_ORACLE_RNG_GLIBC_SEQ) glibc_init(&rng.gl, seed);

    int err = 0;
    const int64_t walk_size = 2 * (int64_t)walk_length + 1; /* rw_cpu_triples.cpp:119 */
    for (int64_t i = 0; i < n_walkers && !err; ++i) {
        int64_t *walk = out + i * walk_size;
        rng_begin_walker(&rng, (uint64_t)walker_base + (uint64_t)i, walk_length);
        const int64_t target_node = targets[i];
        walk[0] = target_node;
        int64_t previous_node = target_node;
        for (int64_t s = 1; s < walk_size; s += 2) { /* rw_cpu_triples.cpp:91 */
            int64_t relation = padding_idx, tail = padding_idx;
            if (previous_node != padding_idx) { /* :20 */
                if (previous_node < 0 || previous_node >= n_entities) { err = TRW_ORACLE_E_RANGE; break; }
                const int64_t start_index = rt_index[2 * previous_node];
                const int64_t end_index = rt_index[2 * previous_node + 1];
                if (!(start_index == -1 || end_index == -1)) { /* :26 */
                    int64_t e = start_index; /* sample_int, cpu_utils.cpp:3-10 */
                    if (start_index != end_index) {
                        const int64_t span = (end_index + 1) - start_index;
                        if (span <= 0) { err = TRW_ORACLE_E_ARG; break; }
                        e = start_index + ((int64_t)rng_next(&rng) % span);
                    }
                    if (e < 0 || e >= n_triples) { err = TRW_ORACLE_E_RANGE; break; }
                    relation = triples[3 * e + 1];
                    tail = triples[3 * e + 2];
                }
            }
            walk[s] = relation;
            walk[s + 1] = tail;
            previous_node = tail;
        }
    }
    if (draws_total) *draws_total = rng.ndraws;
    return err;
}

/* ------------------------------------------------------------------------ */
/* Edge-list walk                                                             */
/* ------------------------------------------------------------------------ */

typedef struct {
    const int64_t *edges, *index;
    int64_t n_edges, n_nodes, padding;
    int err;
} elist_t;

/* sample_neighbor, rw_cpu_edge_list.cpp:8-35 */
static int64_t el_sample_neighbor(elist_t *g, int64_t node, int64_t jump_node, rng_t *rng)
{
    if (node == g->padding) return jump_node; /* :31-34 */
    if (node < 0 || node >= g->n_nodes) { g->err = TRW_ORACLE_E_RANGE; return g->padding; }
    const int64_t start_index = g->index[2 * node], end_index = g->index[2 * node + 1];
    if (start_index == -1 || end_index == -1) return g->padding; /* :21-22 */
    int64_t e = start_index; /* sample_int */
    if (start_index != end_index) {
        const int64_t span = (end_index + 1) - start_index;
        if (span <= 0) { g->err = TRW_ORACLE_E_ARG; return g->padding; }
        e = start_index + ((int64_t)rng_next(rng) % span);
    }
    if (e < 0 || e >= g->n_edges) { g->err = TRW_ORACLE_E_RANGE; return g->padding; }
    return g->edges[2 * e + 1];
}

/* is_neighbor, rw_cpu_edge_list.cpp:37-62: scans [start, end), i.e. never the last edge */
static int el_is_neighbor(elist_t *g, int64_t new_node, int64_t previous_node)
{
    if (previous_node < 0 || previous_node >= g->n_nodes) return 0; /* the reference reads out of bounds here */
    const int64_t start_index = g->index[2 * previous_node], end_index = g->index[2 * previous_node + 1];
    if (start_index == -1 || end_index == -1) return 0;
    for (int64_t i = start_index; i < end_index; ++i) {
        if (i < 0 || i >= g->n_edges) { g->err = TRW_ORACLE_E_RANGE; return 0; }
        if (g->edges[2 * i + 1] == new_node) return 1;
    }
    return 0;
}

int trw_oracle_walk_edge_list(const int64_t *edges, int64_t n_edges,
                              const int64_t *index, int64_t n_nodes,
                              const int64_t *targets, int64_t n_walkers,
                              int64_t walker_base, double p, double q,
                              int32_t walk_length, int32_t seed,
                              int64_t padding_idx, int32_t restart, int32_t rng_mode,
                              int64_t *out, uint64_t *draws_total)
{
    if ((!edges && n_edges) || (!index && n_nodes) || (!targets && n_walkers) || (!out && n_walkers))
        return TRW_ORACLE_E_ARG;
    if (n_edges < 0 || n_nodes < 0 || n_walkers < 0 || walker_base < 0 || walk_length < 1) return TRW_ORACLE_E_ARG;
    if (rng_mode != TRW_ORACLE_RNG_GLIBC_SEQ && rng_mode != TRW_ORACLE_RNG_LIBC && rng_mode != TRW_ORACLE_RNG_WALKER)
        return TRW_ORACLE_E_ARG;
    if (rng_mode != TRW_ORACLE_RNG_WALKER && walker_base != 0) return TRW_ORACLE_E_ARG;

    elist_t g = { edges, index, n_edges, n_nodes, padding_idx, 0 };
    rng_t rng;
    memset(&rng, 0, sizeof rng);
    rng.mode = rng_mode;
    rng.seed = seed;
    if (rng_mode == TRW_ORACLE_RNG_LIBC) srand((unsigned)seed); /* :73 / :140 */
    else if (rng_mode == TRW_ORACLE_RNG_GLIBC_SEQ) glibc_init(&rng.gl, seed);

    const int uniform = (p == 1.0 && q == 1.0); /* :260 */
    double prob[3];
    trw_oracle_probs(p, q, prob); /* :158-162 */
    const int64_t walk_size = (int64_t)walk_length + 1;

    for (int64_t i = 0; i < n_walkers && !g.err; ++i) {
        int64_t *walk = out + i * walk_size;
        rng_begin_walker(&rng, (uint64_t)walker_base + (uint64_t)i, walk_length);
        const int64_t target_node = targets[i];
        const int64_t jump_node = restart ? target_node : padding_idx; /* :103-108 */
        walk[0] = target_node;
        if (uniform) {
            int64_t previous_node = target_node;
            for (int64_t s = 1; s < walk_size; ++s) {
                const int64_t next_node = el_sample_neighbor(&g, previous_node, jump_node, &rng);
                walk[s] = next_node;
                previous_node = next_node;
            }
        } else {
            walk[1] = el_sample_neighbor(&g, target_node, jump_node, &rng);
            int64_t previous_node = walk[1];
            for (int64_t s = 2; s < walk_size && !g.err; ++s) {
                int64_t selected_node = -1;
                const int64_t t_node = walk[s - 2];
                while (1) {
                    const int64_t new_node = el_sample_neighbor(&g, previous_node, jump_node, &rng);
                    const double random_prob = ((double)rng_next(&rng) / (double)RAND_MAX);
                    if (g.err) { selected_node = new_node; break; }
                    if (new_node == t_node) { /* :204-209, not chained with what follows */
                        if (random_prob < prob[0]) { selected_node = new_node; break; }
                    }
                    if (new_node == padding_idx) { /* :212-217 */
                        if (random_prob < prob[0]) { selected_node = jump_node; break; }
                    } else if (el_is_neighbor(&g, new_node, t_node)) { /* :220-225 */
                        if (random_prob < prob[1]) { selected_node = new_node; break; }
                    } else if (random_prob < prob[2]) { /* :228-231 */
                        selected_node = new_node;
                        break;
                    }
                }
                walk[s] = selected_node;
                previous_node = selected_node;
            }
        }
    }
    if (draws_total) *draws_total = rng.ndraws;
    return g.err;
}

/* ------------------------------------------------------------------------ */
/* Window builders                                                            */
/* ------------------------------------------------------------------------ */

static int windows_args_ok(int64_t num_walks, int64_t walk_cols, int32_t window_size, int64_t num_nodes)
{
    return num_walks >= 0 && walk_cols >= 1 && window_size >= 1 && window_size <= walk_cols && num_nodes >= 1;
}

int trw_oracle_to_windows(const int64_t *walks, int64_t num_walks, int64_t walk_cols, int32_t window_size,
                          int64_t num_nodes, int32_t seed, int32_t rng_mode,
                          int64_t *target_nodes, int64_t *pos_windows, int64_t *neg_windows)
{
    if (!windows_args_ok(num_walks, walk_cols, window_size, num_nodes)) return TRW_ORACLE_E_ARG;
    if (rng_mode != TRW_ORACLE_RNG_GLIBC_SEQ && rng_mode != TRW_ORACLE_RNG_LIBC) return TRW_ORACLE_E_ARG;
    rng_t rng;
    memset(&rng, 0, sizeof rng);
    rng.mode = rng_mode;
    if (rng_mode == TRW_ORACLE_RNG_LIBC) srand((unsigned)seed); /* windows_cpu.cpp:12 */
    else glibc_init(&rng.gl, seed);
    const int64_t step_end = (walk_cols - window_size) + 1;
    const int64_t mid_pos = window_size / 2;
    const int64_t width = window_size - 1;
    for (int64_t walk_idx = 0; walk_idx < num_walks; ++walk_idx) {
        const int64_t *walk = walks + walk_idx * walk_cols;
        for (int64_t step_idx = 0; step_idx < step_end; ++step_idx) {
            const int64_t pos = walk_idx * step_end + step_idx;
            target_nodes[pos] = walk[step_idx + mid_pos];
            int64_t k = 0;
            for (int64_t i = 0; i < window_size; ++i)
                if (i != mid_pos) pos_windows[pos * width + k++] = walk[step_idx + i];
            for (int64_t i = 0; i < width; ++i) /* :66-69 */
                neg_windows[pos * width + i] = (int64_t)rng_next(&rng) % num_nodes;
        }
    }
    return 0;
}

int trw_oracle_to_windows_cbow(const int64_t *walks, int64_t num_walks, int64_t walk_cols, int32_t window_size,
                               int64_t num_nodes, int32_t seed, int32_t rng_mode, int64_t window_base,
                               int64_t *pos_nodes, int64_t *neg_nodes, int64_t *windows)
{
    if (!windows_args_ok(num_walks, walk_cols, window_size, num_nodes) || window_base < 0) return TRW_ORACLE_E_ARG;
    if (rng_mode != TRW_ORACLE_RNG_GLIBC_SEQ && rng_mode != TRW_ORACLE_RNG_LIBC && rng_mode != TRW_ORACLE_RNG_WALKER)
        return TRW_ORACLE_E_ARG;
    if (rng_mode != TRW_ORACLE_RNG_WALKER && window_base != 0) return TRW_ORACLE_E_ARG;
    rng_t rng;
    memset(&rng, 0, sizeof rng);
    rng.mode = rng_mode;
    rng.seed = seed;
    if (rng_mode == TRW_ORACLE_RNG_LIBC) srand((unsigned)seed); /* windows_cpu.cpp:87 */
    else if (rng_mode == TRW_ORACLE_RNG_GLIBC_SEQ) glibc_init(&rng.gl, seed);
    const int64_t step_end = (walk_cols - window_size) + 1;
    const int64_t mid_pos = window_size / 2;
    const int64_t width = window_size - 1;
    for (int64_t walk_idx = 0; walk_idx < num_walks; ++walk_idx) {
        const int64_t *walk = walks + walk_idx * walk_cols;
        for (int64_t step_idx = 0; step_idx < step_end; ++step_idx) {
            const int64_t pos = walk_idx * step_end + step_idx;
            rng_begin_walker(&rng, (uint64_t)window_base + (uint64_t)pos, 0);
            const int64_t pos_node = walk[step_idx + mid_pos];
            pos_nodes[pos] = pos_node;
            /* sample_int(0, num_nodes-1): no draw when the range is a single value (cpu_utils.cpp:4-5) */
            int64_t neg_node = num_nodes == 1 ? 0 : (int64_t)rng_next(&rng) % num_nodes;
            int max_checks = 0;
            while (neg_node == pos_node && max_checks <= 100) { /* :131-134 */
                neg_node = num_nodes == 1 ? 0 : (int64_t)rng_next(&rng) % num_nodes;
                max_checks = max_checks + 1;
            }
            neg_nodes[pos] = neg_node;
            int64_t k = 0;
            for (int64_t i = 0; i < window_size; ++i)
                if (i != mid_pos) windows[pos * width + k++] = walk[step_idx + i];
        }
    }
    return 0;
}

/* pos_windows of one target triple, windows_cpu.cpp:212-276 (identical at :392-455) */
static void triples_pos_window(const int64_t *walk, int64_t walk_length, int64_t target_rel_index,
                               int32_t window_size, int64_t padding_idx, int64_t *win /* [2*ws][3] */)
{
    for (int64_t hop = 0; hop <= window_size; hop++) { /* sic: window_size+1 rows */
        const int64_t rel_idx = target_rel_index - ((hop + 1) * 2);
        const int64_t head_idx = rel_idx - 1, tail_idx = rel_idx + 1;
        if (hop >= 2 * (int64_t)window_size) break; /* window_size == 0 guard; never taken for ws >= 1 */
        win[hop * 3 + 0] = head_idx >= 0 ? walk[rel_idx] : padding_idx; /* sic: walk[rel_idx] */
        win[hop * 3 + 1] = rel_idx >= 0 ? walk[rel_idx] : padding_idx;
        win[hop * 3 + 2] = tail_idx >= 0 ? walk[tail_idx] : padding_idx;
    }
    for (int64_t hop = 0; hop < window_size; hop++) {
        const int64_t pos = hop + window_size;
        const int64_t rel_idx = target_rel_index + ((hop + 1) * 2);
        const int64_t head_idx = rel_idx - 1, tail_idx = rel_idx + 1;
        win[pos * 3 + 0] = head_idx < walk_length ? walk[head_idx] : padding_idx;
        win[pos * 3 + 1] = rel_idx < walk_length ? walk[rel_idx] : padding_idx;
        win[pos * 3 + 2] = tail_idx < walk_length ? walk[tail_idx] : padding_idx;
    }
}

static int triples_windows_args_ok(int64_t num_walks, int64_t walk_cols, int32_t window_size, int64_t num_triples)
{
    return num_walks >= 0 && walk_cols >= 1 && window_size >= 1 && num_triples >= 1;
}

int trw_oracle_to_windows_triples(const int64_t *walks, int64_t num_walks, int64_t walk_cols, int32_t window_size,
                                  int64_t padding_idx, const int64_t *triples, int64_t num_triples, int32_t seed,
                                  int32_t rng_mode, int64_t *target_triples, int64_t *pos_windows,
                                  int64_t *neg_windows)
{
    if (!triples_windows_args_ok(num_walks, walk_cols, window_size, num_triples)) return TRW_ORACLE_E_ARG;
    if (rng_mode != TRW_ORACLE_RNG_GLIBC_SEQ && rng_mode != TRW_ORACLE_RNG_LIBC) return TRW_ORACLE_E_ARG;
    rng_t rng;
    memset(&rng, 0, sizeof rng);
    rng.mode = rng_mode;
    if (rng_mode == TRW_ORACLE_RNG_LIBC) srand((unsigned)seed);
    else glibc_init(&rng.gl, seed);
    const int64_t per_walk = (walk_cols - 1) / 2;
    const int64_t slots = 2 * (int64_t)window_size;
    for (int64_t walk_idx = 0; walk_idx < num_walks; ++walk_idx) {
        const int64_t *walk = walks + walk_idx * walk_cols;
        int64_t target_index = 0;
        for (int64_t tri = 1; tri < walk_cols - 1; tri += 2) {
            const int64_t pos = per_walk * walk_idx + target_index;
            target_triples[pos * 3 + 0] = walk[tri - 1];
            target_triples[pos * 3 + 1] = walk[tri];
            target_triples[pos * 3 + 2] = walk[tri + 1];
            triples_pos_window(walk, walk_cols, tri, window_size, padding_idx, pos_windows + pos * slots * 3);
            for (int64_t hop = 0; hop < slots; ++hop) { /* :279-290 */
                const int64_t idx = num_triples == 1 ? 0 : (int64_t)rng_next(&rng) % num_triples;
                for (int k = 0; k < 3; ++k) neg_windows[(pos * slots + hop) * 3 + k] = triples[idx * 3 + k];
            }
            target_index++;
        }
    }
    return 0;
}

int trw_oracle_to_windows_triples_cbow(const int64_t *walks, int64_t num_walks, int64_t walk_cols,
                                       int32_t window_size, int64_t padding_idx, const int64_t *triples,
                                       int64_t num_triples, int32_t seed, int32_t rng_mode, int64_t window_base,
                                       int64_t *pos_triples, int64_t *neg_triples, int64_t *pos_windows)
{
    if (!triples_windows_args_ok(num_walks, walk_cols, window_size, num_triples) || window_base < 0)
        return TRW_ORACLE_E_ARG;
    if (rng_mode != TRW_ORACLE_RNG_GLIBC_SEQ && rng_mode != TRW_ORACLE_RNG_LIBC && rng_mode != TRW_ORACLE_RNG_WALKER)
        return TRW_ORACLE_E_ARG;
    if (rng_mode != TRW_ORACLE_RNG_WALKER && window_base != 0) return TRW_ORACLE_E_ARG;
    rng_t rng;
    memset(&rng, 0, sizeof rng);
    rng.mode = rng_mode;
    rng.seed = seed;
    if (rng_mode == TRW_ORACLE_RNG_LIBC) srand((unsigned)seed);
    else if (rng_mode == TRW_ORACLE_RNG_GLIBC_SEQ) glibc_init(&rng.gl, seed);
    const int64_t per_walk = (walk_cols - 1) / 2;
    const int64_t slots = 2 * (int64_t)window_size;
    for (int64_t walk_idx = 0; walk_idx < num_walks; ++walk_idx) {
        const int64_t *walk = walks + walk_idx * walk_cols;
        int64_t target_index = 0;
        for (int64_t tri = 1; tri < walk_cols - 1; tri += 2) {
            const int64_t pos = per_walk * walk_idx + target_index;
            rng_begin_walker(&rng, (uint64_t)window_base + (uint64_t)pos, 0);
            const int64_t ph = walk[tri - 1], pr = walk[tri], pt = walk[tri + 1];
            pos_triples[pos * 3 + 0] = ph;
            pos_triples[pos * 3 + 1] = pr;
            pos_triples[pos * 3 + 2] = pt;
            int64_t idx = num_triples == 1 ? 0 : (int64_t)rng_next(&rng) % num_triples; /* :374 */
            int max_checks = 0;
            while (triples[idx * 3] == ph && triples[idx * 3 + 1] == pr && triples[idx * 3 + 2] == pt &&
                   max_checks <= 100) { /* :381-389 */
                idx = num_triples == 1 ? 0 : (int64_t)rng_next(&rng) % num_triples;
                max_checks = max_checks + 1;
            }
            for (int k = 0; k < 3; ++k) neg_triples[pos * 3 + k] = triples[idx * 3 + k];
            triples_pos_window(walk, walk_cols, tri, window_size, padding_idx, pos_windows + pos * slots * 3);
            target_index++;
        }
    }
    return 0;
}
