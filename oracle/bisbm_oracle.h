/*
 * bisbm_oracle.h -- CPU restatement of the node-label Metropolis-Hastings sweep of
 * junipertcy/bipartiteSBM-MCMC.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker.  Nothing under bipartitesbm-mcmc_amd/ links, includes
 * or calls it; the product path is the HIP library behind include/bisbm.h.
 *
 * Parity status: the reference ships no tests and cannot be built in this image (Boost is
 * absent and stand-in headers are not allowed), so this restatement is pinned by
 *   (1) the reference outputs recorded in SURVEY.md App. C.3 and section 4
 *       (tests/golden/survey_known_answers.json), and
 *   (2) the three Boost-free reference translation units compiled as they lie
 *       (oracle/_ref: spence.cc, graph_utilities.cc, output_functions.cc), and
 *   (3) the real libstdc++ 11 for the <random>/<algorithm> arithmetic (oracle/stdcheck.cc).
 * See DESIGN.md "Oracle and pinning".
 *
 * Each function cites the reference file:line it follows (paths relative to
 * /root/reference/src).
 */
#ifndef BISBM_ORACLE_H
#define BISBM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_model orc_model;

enum { ORC_RNG_COMPAT = 0, ORC_RNG_PHILOX = 1 };
enum {
    ORC_SCHED_EXPONENTIAL = 0,
    ORC_SCHED_LINEAR = 1,
    ORC_SCHED_LOGARITHMIC = 2,
    ORC_SCHED_CONSTANT = 3,
    ORC_SCHED_ABRUPT_COOL = 4
};

/* ---- text I/O (graph_utilities.cc:5-49, output_functions.hh:20-29) ---- */
/* Returns the number of edges (lines) or -1 when the file cannot be opened.  The two arrays are malloced. */
long orc_load_edge_list(const char *path, uint64_t **a, uint64_t **b);
long orc_load_memberships(const char *path, uint32_t **labels);
/* edge_to_adj: undirected, duplicates kept, file order.  rowptr has n+1 entries, col 2*n_edges.
 * Returns 0, or -1 when an id is >= n. */
int orc_edge_to_csr(const uint64_t *a, const uint64_t *b, size_t n_edges, size_t n,
                    uint64_t *rowptr, uint32_t *col);
/* output_vec: "<x> <y> ... <z> \n" (trailing blank).  Returns bytes written (excluding NUL). */
size_t orc_format_vec(const uint32_t *v, size_t n, char *out, size_t cap);

/* ---- numerics (support/cache.{hh,cc}, support/int_part.{hh,cc}, support/spence.cc, util.hh) ---- */
void orc_init_tables(size_t lgamma_size, size_t q_kcap); /* grows the process-wide tables */
double orc_lgamma_fast(size_t x);
double orc_safelog_fast(size_t x);
double orc_log_q(int n, int k);
double orc_log_q_philox(int n, int k); /* Philox-mode definition: get_v to convergence for u >= 2.5 */
double orc_log_q_approx_philox(size_t n, size_t k);
double orc_log_q_approx(size_t n, size_t k);
double orc_q_cache_at(size_t n, size_t k); /* raw table cell, k <= kcap */
double orc_spence(double x);
double orc_lbinom_fast(size_t N, size_t k);
const double *orc_lgamma_table(size_t *size);
const double *orc_q_table(size_t *rows, size_t *stride);

/* ---- schedules (metropolis_hasting.cc:10-37) ---- */
double orc_schedule(int schedule, uint64_t t, float kw0, float kw1);

/* ---- libstdc++-11 compatible RNG pieces exposed for oracle/stdcheck ---- */
typedef struct orc_mt19937 {
    uint32_t mt[624];
    int idx;
} orc_mt19937;
void orc_mt_seed(orc_mt19937 *g, uint64_t seed);
uint32_t orc_mt_next(orc_mt19937 *g);
double orc_mt_canonical(orc_mt19937 *g);
uint32_t orc_mt_lemire(orc_mt19937 *g, uint32_t range);
void orc_mt_shuffle_u32(orc_mt19937 *g, uint32_t *v, size_t n);
void orc_mt_shuffle_u8(orc_mt19937 *g, uint8_t *v, size_t n); /* std::shuffle of a vector<bool> */
size_t orc_mt_discrete(orc_mt19937 *g, const int *w, size_t n);

/* ---- Philox4x32-10 (production RNG; Salmon et al. SC'11) ---- */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* sweep visit order in Philox mode: position i of sweep `sweep` of chain `chain` */
uint32_t orc_philox_visit(uint64_t seed, uint32_t chain, uint64_t sweep, uint32_t na, uint32_t nb, uint32_t i);

/* ---- model (blockmodel.{hh,cc}) ---- */
orc_model *orc_create(size_t n, size_t na, size_t nb, const uint64_t *rowptr, const uint32_t *col,
                      size_t ka, size_t kb, double epsilon, const uint32_t *labels);
void orc_destroy(orc_model *m);
void orc_seed_compat(orc_model *m, uint64_t engine_seed, uint64_t gen_seed);
void orc_seed_philox(orc_model *m, uint64_t seed, uint32_t chain_id);
void orc_set_memberships(orc_model *m, const uint32_t *labels); /* then call orc_init_bisbm */
void orc_init_bisbm(orc_model *m);    /* blockmodel.cc:682-688 */
void orc_shuffle_bisbm(orc_model *m); /* blockmodel.cc:672-680 (engine) / Philox definition */
double orc_anneal(orc_model *m, int schedule, float kw0, float kw1, uint64_t duration,
                  uint64_t steps_await); /* metropolis_hasting.cc:64-101 */
double orc_entropy(orc_model *m);        /* blockmodel.cc:753-787 */
double orc_get_entropy(const orc_model *m); /* running sum of accepted dS, blockmodel.cc:91 */
double orc_compute_dS_vertex(orc_model *m, size_t v, size_t r, size_t s); /* blockmodel.cc:290-333 */
/* one proposal + transition ratio without applying it (for known-answer probes) */
double orc_transition_ratio(orc_model *m, size_t v, size_t s, double *accu_r);
/* diagnostic: how often two consecutive steps of a chunk are independent (see the .c file) */
void orc_pair_probe(orc_model *m, uint64_t sweeps, double temperature, uint64_t out[6]);
void orc_depth_probe(orc_model *m, uint64_t sweeps, double temperature, int depth, uint64_t out[12]);
/* the Philox-mode proposal (single_vertex_change, blockmodel.cc:613-637) for given uniforms */
size_t orc_propose_philox(orc_model *m, size_t v, double u_idx, double u_R, double u_tgt);

size_t orc_n(const orc_model *m);
size_t orc_k(const orc_model *m);
size_t orc_num_edges(const orc_model *m);
size_t orc_max_degree(const orc_model *m);
void orc_get_memberships(const orc_model *m, uint32_t *out);
void orc_get_m(const orc_model *m, int32_t *out); /* K*K row-major */
void orc_get_m_r(const orc_model *m, int32_t *out);
void orc_get_n_r(const orc_model *m, int32_t *out);
void orc_get_eta(const orc_model *m, uint32_t *out); /* K*(maxdeg+1) */
void orc_get_vlist(const orc_model *m, uint32_t *out);
uint64_t orc_last_accepted(const orc_model *m);
uint64_t orc_last_sweeps(const orc_model *m);
uint64_t orc_total_sweeps(const orc_model *m);

/* agglomerative merges between anneals (blockmodel.cc:109-288,335-372,567-611,639-669); see the .c file */
size_t orc_ka(const orc_model *m);
size_t orc_kb(const orc_model *m);
double orc_merge_dS(const orc_model *m, size_t r, size_t s);
/* agg_split(engine, type, nm), blockmodel.cc:505-565 with the intended rank-within-block indexing (see the .c file) */
int orc_agg_split(orc_model *m, int type, int nm);
double orc_last_split_dS(const orc_model *m);
int orc_agg_merge(orc_model *m, int diff_a, int diff_b, int nm);
int orc_agg_merge_total(orc_model *m, int diff, int nm);
size_t orc_geospace(long start_a, long end_a, long start_b, long end_b, double ratio, int *out_a, int *out_b,
                    size_t cap);

#ifdef __cplusplus
}
#endif
#endif
