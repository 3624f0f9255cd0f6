/*
 * bisbm.h -- C ABI of the MI355X-native Metropolis-Hastings sweep engine for the degree-corrected
 * bipartite SBM (libbisbm_hip.so).
 *
 * The reference (junipertcy/bipartiteSBM-MCMC) has no FFI; its CLI (src/mcmc_main.cc) drives the
 * hot path through two C++ classes.  Every entry point below stands behind one of those member
 * functions, cited as <file>:<line> relative to /root/reference/src.  Plain pointers and sizes only:
 * no C++ types, no torch types, no exceptions across the boundary.  Host buffers are caller-owned;
 * device memory is library-owned.  One host thread per handle.
 *
 * All functions return BISBM_OK (0) or a bisbm_status error code; bisbm_last_error() gives the text.
 * There is no CPU fallback: without a HIP device bisbm_create fails with BISBM_ERR_NO_DEVICE.
 */
#ifndef BISBM_H
#define BISBM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: bisbm_get_ka_kb_chain; KA + KB above 256 (wide mode); handles whose chains differ in shape.
 * 3: bisbm_check_shape; several devices behind one handle (bisbm_create_multi); bisbm_last_pass_steps.  Additions only. */
#define BISBM_ABI_VERSION 3

typedef struct bisbm_engine *bisbm_handle;

typedef enum bisbm_status {
    BISBM_OK = 0,
    BISBM_ERR_INVALID_ARG = 1,   /* null pointer, size mismatch, label out of range ... */
    BISBM_ERR_NOT_BIPARTITE = 2, /* an edge joins two nodes of one type, or an id >= n */
    BISBM_ERR_UNSUPPORTED = 3,   /* more blocks than wide mode serves (bisbm_check_shape), more than 2^32-1 adjacency entries ... */
    BISBM_ERR_NO_DEVICE = 4,     /* no HIP device / bad ordinal: the engine has no CPU path */
    BISBM_ERR_HIP = 5,           /* a HIP runtime call failed */
    BISBM_ERR_STATE = 6          /* call order (e.g. anneal before init/shuffle) */
} bisbm_status;

/* Random-number definition of the chain.
 * BISBM_RNG_PHILOX         production: Philox4x32-10 counters keyed by (seed, global chain id),
 *                          Feistel visit order per sweep, integer inverse-CDF proposal draw,
 *                          butterfly FP64 sums (DESIGN.md "Philox-mode definition").
 * BISBM_RNG_MT19937_COMPAT the reference's own draw sequence: std::mt19937 `engine`
 *                          (mcmc_main.cc:242) + the hidden `gen` (blockmodel.hh:17-18),
 *                          libstdc++-11 shuffle / generate_canonical / discrete_distribution,
 *                          serial FP64 sums in source order.  Chain c uses engine seed `seed + c`
 *                          and gen seed `gen_seed + c` (c = global chain id).
 */
typedef enum bisbm_rng { BISBM_RNG_PHILOX = 0, BISBM_RNG_MT19937_COMPAT = 1 } bisbm_rng;

/* metropolis_hasting.cc:10-37 */
typedef enum bisbm_schedule {
    BISBM_SCHED_EXPONENTIAL = 0,
    BISBM_SCHED_LINEAR = 1,
    BISBM_SCHED_LOGARITHMIC = 2,
    BISBM_SCHED_CONSTANT = 3,
    BISBM_SCHED_ABRUPT_COOL = 4
} bisbm_schedule;

#define BISBM_ALL_CHAINS (-1)

/* Replaces blockmodel_t::blockmodel_t (blockmodel.hh:22-23, blockmodel.cc:15-75; call sites
 * mcmc_main.cc:352,420,453).  The graph is CSR of the reference's adj_list_t: row v holds the
 * neighbours of v in edge-file order, duplicates kept (graph_utilities.cc:36-49).  Nodes
 * [0,na) are type a, [na,na+nb) type b (mcmc_main.cc:121-130).  Blocks [0,ka) are type a,
 * [ka,ka+kb) type b.  `n_chains` independent chains are created on HIP device `device`; chain i
 * of this handle has global id first_chain_id + i (the id keys its random stream, so results do
 * not depend on how chains are sharded over GPUs).  Builds the lgamma / log_q tables
 * (support/cache.cc:64-91, support/int_part.cc:34-51) on the host and uploads them.
 * No more blocks than nodes of a type, and a shape bisbm_check_shape accepts.  With ka + kb > 256 (the reference's --merge driver starts from one
 * block per node, mcmc_main.cc:350-353) the handle runs in WIDE MODE: two-byte labels, the block matrix read and updated in
 * HBM, the generic kernel (slow per step; meant for the greedy sweeps between merge stages); bisbm_agg_merge switches it to
 * byte labels and the ordinary kernels as soon as it leaves ka + kb <= 256; a split (negative diff) that takes a handle past
 * 256 blocks switches it to wide mode, and splits are served while wide. */
int bisbm_create(bisbm_handle *out, uint64_t n, uint64_t na, uint64_t nb, const uint64_t *rowptr,
                 const uint32_t *col, uint32_t ka, uint32_t kb, double epsilon, uint32_t n_chains,
                 uint32_t first_chain_id, int device, int rng_mode, uint64_t seed,
                 uint64_t gen_seed);

/* The same for chains spread over several devices of one node (no reference counterpart: the reference is one chain in one
 * process; BASELINE north_star / SURVEY 8e: "chains shard trivially across the 8 GPUs of one node").  The n_chains chains are
 * split into contiguous ranges, one per entry of devices[] (the first n_chains % n_devices ranges one chain longer); graph
 * and tables are replicated per device; chain i keeps the global id first_chain_id + i, so every result equals what ONE
 * device with all the chains gives.  Every other call works on the returned handle as on a single-device one (all-chain calls
 * run the devices side by side, one host thread and one stream per device; there is no exchange during sweeps) except:
 * bisbm_set_stream (refused) and bisbm_marginals_accumulate with a caller-owned buffer (refused: each device accumulates into
 * its own; bisbm_marginals_map pools them on the devices, bisbm_marginals_get on the host).  A device may be listed more than
 * once (rehearsal on a one-GPU box). */
int bisbm_create_multi(bisbm_handle *out, uint64_t n, uint64_t na, uint64_t nb, const uint64_t *rowptr,
                       const uint32_t *col, uint32_t ka, uint32_t kb, double epsilon, uint32_t n_chains,
                       uint32_t first_chain_id, const int *devices, int n_devices, int rng_mode,
                       uint64_t seed, uint64_t gen_seed);

/* Devices behind a handle: their number, their ordinals and the first chain of each (arrays of *n_devices entries; any
 * pointer may be NULL).  1 / {device} / {0} for a bisbm_create handle. */
int bisbm_device_count(bisbm_handle h, int *n_devices, int *devices, uint32_t *first_chain);

int bisbm_destroy(bisbm_handle h);

/* Whether a handle of ka + kb blocks can be served, without creating one (no reference counterpart: blockmodel_t has no
 * block limit; the CLI's --merge driver asks before it starts from one block per node, mcmc_main.cc:350-353).
 * Up to 256 blocks: always.  Above (wide mode): labels are two bytes (ka + kb <= 65535) and a chain's m_r, n_r and k_v histogram
 * stay in LDS, which ends at about 14 000 blocks (even split: 14 700 in Philox mode, 13 700 in mt19937-compat mode; 10 bytes
 * per block plus 4 for the larger type, 160 KiB per CU).  BISBM_ERR_UNSUPPORTED with the numbers in bisbm_last_error(NULL). */
int bisbm_check_shape(uint32_t ka, uint32_t kb, int rng_mode);

/* Initial partition: the `memberships` constructor argument (blockmodel.cc:23), n labels in
 * [0, ka+kb).  chain = BISBM_ALL_CHAINS copies the vector to every chain. */
int bisbm_set_memberships(bisbm_handle h, int64_t chain, const uint32_t *labels);

/* blockmodel_t::init_bisbm (blockmodel.cc:682-688): rebuild n_r, m, m_r, eta from the labels. */
int bisbm_init(bisbm_handle h);

/* blockmodel_t::shuffle_bisbm (blockmodel.cc:672-680): shuffle the type-a labels, then the
 * type-b labels (block sizes preserved), then rebuild the block state. */
int bisbm_shuffle(bisbm_handle h);

/* metropolis_hasting::anneal (metropolis_hasting.hh:48-53, metropolis_hasting.cc:64-101) for
 * every chain: duration_steps / n sweeps of n node updates (step :42-62, transition_ratio
 * :103-192, single_vertex_change blockmodel.cc:613-637, apply_mcmc_moves blockmodel.cc:461-503),
 * early stop per chain when the count of T<1 steps since the last new minimum reaches steps_await.
 * kwargs are the two float cooling parameters (mcmc_main.cc:49).  acc_rate_out[n_chains] receives
 * anneal's return value per chain (may be NULL). */
int bisbm_anneal(bisbm_handle h, int schedule, const float kwargs[2], uint64_t duration_steps,
                 uint64_t steps_await, double *acc_rate_out);

/* blockmodel_t::get_memberships (blockmodel.cc:87) for one chain: n labels. */
int bisbm_get_memberships(bisbm_handle h, uint32_t chain, uint32_t *labels_out);

/* get_m / get_m_r / get_n_r / get_eta_rk_ (blockmodel.cc:93-99) for one chain.
 * m is the reference's full symmetric K*K matrix (row-major), eta is K*(max_degree+1).
 * Any output pointer may be NULL. */
int bisbm_get_block_state(bisbm_handle h, uint32_t chain, int32_t *m, int32_t *m_r, int32_t *n_r,
                          uint32_t *eta);

/* blockmodel_t::get_entropy (blockmodel.cc:91): running sum of accepted dS, per chain.  Launches that keep anneal()'s
 * early-stop bookkeeping (below T = 1 with steps_await in reach), the generic kernel and mt19937-compat mode add it up step by
 * step; a production launch that cannot stop early advances it per bisbm_anneal call by the change of the block-state part of
 * the description length instead -- the same quantity to <= 1e-12 of the description length (BISBM_KEEP_SUM=1: step by step
 * everywhere). */
int bisbm_get_cum_dS(bisbm_handle h, double *out);

/* blockmodel_t::entropy (blockmodel.cc:753-787): full description length, per chain.  The
 * reference's log(na*nb) table growth (cache.hh:47-57) that aborts on large graphs
 * is replaced by a direct log(). */
int bisbm_entropy(bisbm_handle h, double *out);

/* Bookkeeping of the last bisbm_anneal per chain: accepted steps and sweeps executed
 * (metropolis_hasting.cc:72,97,100).  Either pointer may be NULL. */
int bisbm_get_last_counts(bisbm_handle h, uint64_t *accepted, uint64_t *sweeps);

/* Marginal accumulation the README describes for "marginalize" (README.md:49-53; the code at
 * mcmc_main.cc:61-65 parses -b/-f and never uses them): add one sample of every chain's labels
 * to counts[n][kmax], kmax = max(ka,kb), column = block index within the node's type.
 * device_counts is a DEVICE pointer to n*kmax uint32 owned by the caller (e.g. a torch tensor
 * that RCCL then reduces across ranks); NULL uses an internal buffer read by bisbm_marginals_get.
 * Stream contract: the histogram kernel runs on the handle's own non-blocking stream and adds with
 * plain read-modify-writes, so everything the caller has in flight on device_counts (its zero fill,
 * its own kernels) must have COMPLETED before the call (synchronise the stream that wrote it); the
 * call returns after its kernel has finished, so the caller may read the buffer right away. */
int bisbm_marginals_accumulate(bisbm_handle h, uint32_t *device_counts);
int bisbm_marginals_reset(bisbm_handle h);
int bisbm_marginals_get(bisbm_handle h, uint32_t *counts_out /* n*kmax, host */);

/* The marginal estimate README.md:49-53 asks for: the most frequent block of every node over all samples of all chains (ties ->
 * the lowest block), n labels in the reference's numbering, from the internal histogram.  Over several devices this is the
 * exchange of SURVEY 8(e), on the devices: ncclReduceScatter of the per-device histograms by node range (RCCL over xGMI) ->
 * argmax on the owner of the range -> ncclAllGather of the labels; peer copies + an add kernel where RCCL cannot serve (a
 * device listed twice, librccl.so missing, BISBM_POOL=p2p). */
int bisbm_marginals_map(bisbm_handle h, uint32_t *labels_out /* n, host */);

/* blockmodel_t::agg_merge(engine, diff_a, diff_b, nm) (blockmodel.hh, blockmodel.cc:109-206; call sites
 * mcmc_main.cc:385,429,434,446): merge diff_a type-a and diff_b type-b blocks in every chain -- nm proposals per
 * block (single_block_change :639-669), lowest merge dS first (compute_dS :335-372), blocks renumbered in the
 * order of their first node (apply_block_moves :567-611), block state rebuilt.  Afterwards bisbm_get_ka_kb returns
 * the new counts.  A negative diff first splits one block per unit (agg_split :505-565, compute_dS(split) :374-424,
 * apply_split_moves :428-459; type a first, :110-117): every block of the type with more than one node is cut nm
 * times into random halves, the cut with the lowest dS wins, its marked nodes become block KA (type a; the type-b
 * labels move up by one) or block K (type b).  The reference's split dS indexes its split vector with a counter that
 * runs over all nodes (:402), an out-of-range read; the engine implements the intended meaning -- position = rank of
 * the node within its block -- which is what agg_split's own pass (:554-561) uses.
 * The two-type overload changes every chain's counts by the same amounts.  The selection is
 * K-scale host work, as in the reference; ranks, cut evaluation, relabelling and the rebuild run on the device.
 * BISBM_ERR_STATE when no block can be split (asked before anything about the handle changes: a refused split leaves the
 * labels, their width and the block state as they were); a split past 256 blocks switches the handle to two-byte labels.
 * Several devices / several shapes behind one handle: a request that some chain cannot meet is refused before any device or
 * group changes.  What is NOT atomic is a failure in the middle of the work (a device out of memory): the devices run side
 * by side, so the others have merged by then -- the call returns the first failing device's code, bisbm_last_error names
 * every device that failed, and the handle keeps serving per-chain calls (bisbm_get_ka_kb_chain tells which chains
 * changed shape); a caller that needs all-or-nothing keeps the labels (bisbm_get_memberships) and puts them back. */
int bisbm_agg_merge(bisbm_handle h, int diff_a, int diff_b, int nm);

/* blockmodel_t::agg_merge(engine, diff, nm) (blockmodel.cc:208-271; call site mcmc_main.cc:365): diff merges over
 * both types together, proposals redrawn while the last one taken had dS = +inf.  Which types lose blocks is up to
 * each chain's own proposals, so the chains of a handle may end with different (Ka,Kb).  The handle then keeps them
 * grouped by shape internally (kernels are launched for one shape: one launch per group from then on); every call
 * keeps working per chain -- anneal, memberships, block state (array sizes follow bisbm_get_ka_kb_chain), sum dS,
 * entropy, further merges of either overload -- except that the calls which need one common shape (bisbm_get_ka_kb, the
 * marginal histogram) return BISBM_ERR_STATE while the chains' shapes differ (later merges may bring them together again).  diff < 0 is BISBM_ERR_INVALID_ARG (this overload has no split branch). */
int bisbm_agg_merge_total(bisbm_handle h, int diff, int nm);

/* Shape queries (get_KA/get_KB blockmodel.cc:103-105, get_num_edges :81).  bisbm_get_ka_kb: the block counts all chains
 * share (BISBM_ERR_STATE while a one-argument agg_merge has left them with different ones); bisbm_get_ka_kb_chain: one chain's. */
int bisbm_get_ka_kb(bisbm_handle h, uint32_t *ka, uint32_t *kb);
int bisbm_get_ka_kb_chain(bisbm_handle h, uint32_t chain, uint32_t *ka, uint32_t *kb);
int bisbm_get_sizes(bisbm_handle h, uint64_t *n, uint64_t *num_edges, uint32_t *max_degree,
                    uint32_t *n_chains);

/* Run kernels on a caller-provided hipStream_t (NULL = the handle's own stream). */
int bisbm_set_stream(bisbm_handle h, void *hip_stream);

/* Device time of the sweep kernel of the last bisbm_anneal, measured with HIP events on the
 * launch stream, and the number of node updates it executed (all chains). */
int bisbm_last_sweep_timing(bisbm_handle h, double *kernel_ms, uint64_t *node_updates);

/* How many steps the passes of the last sweep launch evaluated at once (1, 2, 4 or 8; the largest over the
 * handle's devices / shape groups).  With at most 32 blocks of a type the depth is chosen per launch from the
 * measured speed of earlier launches; the chain does not depend on it.  Diagnostic: no
 * counterpart in the reference. */
int bisbm_last_pass_steps(bisbm_handle h, uint32_t *steps_per_pass);

/* Device numerics probe (tests): evaluates log_q(n[i], k[i]) on the device (int_part.hh:27-37).
 * fast = 0: the literal evaluation (mt19937-compat mode, entropy()); fast = 1: the Philox-mode
 * evaluation, which uses a closed form of get_v/spence for k/sqrt(n) > 21. */
int bisbm_debug_log_q(bisbm_handle h, const int32_t *n, const int32_t *k, size_t count, int fast,
                      double *out);

const char *bisbm_last_error(bisbm_handle h); /* h may be NULL: error of the last failed create */
int bisbm_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* BISBM_H */
