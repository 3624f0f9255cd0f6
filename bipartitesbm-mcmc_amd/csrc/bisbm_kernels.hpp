// bisbm_kernels.hpp -- kernel parameter blocks and launcher prototypes shared by
// bisbm_kernels.hip, bisbm_sweep_fast.hip (device) and the host side of the C ABI (bisbm_engine.hpp lists its units).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bisbm_device.hpp"

namespace bisbm {

// Per-chain scalar state kept in HBM between kernels.
struct ChainScalars {
    double cum_dS;          // blockmodel_t::entropy_ (blockmodel.hh:100): running sum of accepted dS
    double accu_r;          // metropolis_hasting::accu_r_ (metropolis_hasting.hh:28), survives anneal calls
    double last_rate;       // return value of the last anneal
    uint64_t sweeps_total;  // Philox counter: sweeps executed over the chain's lifetime
    uint64_t last_accepted;
    uint64_t last_sweeps;
    uint32_t shuffle_epoch;  // Philox counter: shuffle_bisbm calls so far
    uint32_t engine_idx;     // std::mt19937 positions (compat mode)
    uint32_t gen_idx;
    uint32_t merge_epoch;  // Philox counter: proposal rounds of agg_merge so far
    // where the last production sweep launch ran this chain: HW_ID of the stepping and of the feeder wave, XCC_ID
    // (diagnostic, see BISBM_PLACEMENT_LOG in bisbm_anneal.hip)
    uint32_t hw_id[2];
    uint32_t xcc_id;
    uint32_t split_epoch;  // Philox counter: agg_split calls so far
    // anneal()'s early-stop bookkeeping (metropolis_hasting.cc:75,85-98) where one call runs as several launches of the
    // production kernel: the minimum of sum dS so far, the count of T < 1 steps before the step that reached it, the
    // count of T < 1 steps so far, and whether the chain has returned already (SweepParams::resume)
    double stop_emin;
    uint64_t stop_mark;
    uint64_t stop_below1;
    uint32_t stopped;
    uint32_t pad_;
};

struct SweepParams {
    // graph (shared by all chains)
    const uint32_t* rowptr;
    const uint32_t* col;
    uint32_t n, na, nb, ka, kb, maxdeg;
    double epsilon;
    // chains
    uint32_t n_chains, first_chain_id;
    const uint32_t* chain_gids;  // global id of every chain of the launch (keys its Philox streams); NULL: first_chain_id + index
    uint8_t* labels;
    size_t label_stride;
    uint32_t* vlist;  // compat: [chain][n]
    int32_t* m;       // [chain][ka*kb]
    int32_t* m_r;     // [chain][K]
    int32_t* n_r;     // [chain][K]
    uint32_t* eta;    // [chain][K*(maxdeg+1)]
    ChainScalars* scalars;
    uint32_t* mt_engine;  // compat: [chain][624]
    uint32_t* mt_gen;
    // tables
    const double* lgamma_tab;
    uint64_t lgamma_size;
    const double* q_tab;
    uint32_t q_stride;
    const double* log_tab;
    const double* T_tab;  // host-evaluated temperatures for the pow/log schedules: entry i is step T_base + i of the call
    uint64_t T_len;
    uint64_t T_base;
    int T_zero_after;
    // schedule / run
    int schedule;
    float kw0, kw1;
    uint64_t duration, steps_await;
    // production kernel, a call that runs as several launches: steps of the call executed by the launches before (whole
    // sweeps), the call's full duration, and resume = 1 from the second launch on (the early-stop bookkeeping continues from
    // the chain's scalars; chains that have returned are skipped).  One launch per call: 0, duration, 0.
    uint64_t t_base, call_duration;
    uint32_t resume;
    uint64_t seed;
    int eta_in_lds;
    // production kernel without all of eta in LDS: the window it keeps there instead -- eta_w consecutive degrees per block of
    // the phase's own type, from eta_lo_a (type-a phase) / eta_lo_b on; nodes of other degrees take the general step
    uint32_t eta_w, eta_lo_a, eta_lo_b;
    int vlist_in_lds;
    // production kernel: one counter per SIMD of the chip (kSimdClaims entries, zeroed before the launch) through which
    // the workgroups keep their stepping waves on different SIMDs; NULL: wave `fixed_stepping_wave` (0 or 1) steps
    uint32_t* simd_claims;
    uint32_t fixed_stepping_wave;
    // production kernel: 0 = one step per pass; 1 = two consecutive steps per pass where both block counts are <= 32;
    // 2 = also four per pass where both are <= 16; 3 = also eight per pass where both are <= 8
    uint32_t pair_steps;
    // depth of this launch's passes where the block counts allow a choice (1 / 2 / 3 = two / four / eight steps per pass): the
    // host sets it from the measured speed of the launches before (bisbm_anneal)
    uint32_t pass_depth;
    // wide mode (KA + KB > 256; generic kernel only): `labels` holds two-byte labels (label_stride counts labels, not
    // bytes), and the a x b quadrant of m is read and updated in HBM
    uint32_t wide;
    // production kernel: keep the running sum of accepted dS (and the early-stop bookkeeping's code path) also in a launch
    // that cannot stop early -- BISBM_KEEP_SUM=1: the tests that check the sum of the kernel's own dS values against the change
    // of the description length, tools/soak.py
    uint32_t keep_sum;
};
constexpr uint32_t kSimdClaims = 1u << 14;  // index: XCC_ID[3:0] | HW_ID se, sh, cu [15:8] | simd [5:4]

struct BuildParams {
    const uint32_t* rowptr;
    const uint32_t* col;
    uint32_t n, na, ka, kb, maxdeg, n_chains;
    const uint8_t* labels;
    size_t label_stride;
    int32_t* m;
    int32_t* m_r;
    int32_t* n_r;
    uint32_t* eta;
    uint32_t wide;  // two-byte labels, m counted in HBM (see SweepParams::wide)
};

struct ShuffleParams {
    uint32_t n, na, nb, n_chains, first_chain_id;
    const uint32_t* chain_gids;  // see SweepParams
    uint64_t seed;
    uint8_t* labels;
    const uint8_t* labels_old;  // Philox: snapshot the gather reads from
    size_t label_stride;
    ChainScalars* scalars;
    uint32_t* mt_engine;
    uint32_t wide;
};

struct EntropyParams {
    uint32_t ka, kb, maxdeg, n_chains;
    const int32_t* m;
    const int32_t* m_r;
    const int32_t* n_r;
    const uint32_t* eta;
    const double* lgamma_tab;
    uint64_t lgamma_size;
    const double* q_tab;
    uint32_t q_stride;
    const double* log_tab;
    double* out;
};

struct MarginalParams {
    uint32_t n, na, ka, kmax, n_chains;
    const uint8_t* labels;
    size_t label_stride;
    uint32_t* counts;
    uint32_t wide;  // two-byte labels (see SweepParams::wide)
};

// agg_split (blockmodel.cc:505-565): evaluation of `n_trials` random half-cuts of every block of one type, all chains
struct SplitParams {
    const uint32_t* rowptr;
    const uint32_t* col;
    uint32_t n, na, ka, kb, n_chains, first_chain_id;
    const uint32_t* chain_gids;  // see SweepParams
    uint32_t type;              // 0: a type-a block is split, 1: a type-b block
    uint32_t trial0, n_trials;  // trials evaluated by this launch: trial0 .. trial0 + n_trials - 1
    uint32_t nm;                // trials per block of the whole call (stride of `bits`)
    uint64_t seed;
    uint8_t* labels;
    size_t label_stride;
    const int32_t* n_r;  // [chain][K]
    const ChainScalars* scalars;
    uint32_t* rank;        // [chain][n_type]: rank of every node of the type within its block (ascending id)
    const uint32_t* bits;  // compat: [chain][nm][bit_words] cut bits at position (offset of the block + rank); NULL: Philox
    uint32_t bit_words;
    int32_t* out_k;    // [chain][n_trials][k_type][k_oth]: edges from the marked nodes of block r to opposite block t
    int32_t* out_deg;  // [chain][n_trials][k_type]: degree sum of the marked nodes
    const uint32_t* chosen;  // apply: [chain][2] = {block (own-type index), trial}
    // wide handles (two-byte labels, more than 256 blocks): the rank counters ([chain][K], zeroed) and, in compat mode, the
    // first position of every block in the cut bits ([chain][k_type]) live in HBM; out_k / out_deg are zeroed by the caller
    uint32_t wide;
    uint32_t* rank_base;
    const uint32_t* block_off;
};
constexpr uint32_t PHX_SPLIT = 6;

hipError_t launch_split_rank(const SplitParams& p, hipStream_t stream);
hipError_t launch_split_eval(const SplitParams& p, hipStream_t stream);
hipError_t launch_split_apply(const SplitParams& p, hipStream_t stream);
hipError_t launch_labels_to_wide(const uint8_t* labels, uint8_t* wide_labels, size_t label_stride, uint32_t n, uint32_t n_chains,
                                 hipStream_t stream);

// global id of chain `chain` of a launch
template <class P>
__device__ __forceinline__ uint32_t chain_gid_of(const P& p, uint32_t chain) {
    return p.chain_gids ? p.chain_gids[chain] : p.first_chain_id + chain;
}

// metropolis_hasting.cc:10-37, arithmetic types as the C++ promotes them
__device__ __forceinline__ double temperature_of(const SweepParams& p, uint64_t t) {
    switch (p.schedule) {
        case SCHED_CONSTANT:
            return (double)p.kw0;
        case SCHED_ABRUPT:
            return ((float)t < p.kw0) ? 1. : 0.;
        case SCHED_LINEAR:
            return (double)(p.kw0 - p.kw1 * (float)t);
        case SCHED_EXPONENTIAL:
            if (t < p.T_len) return p.T_tab[t];  // host table: glibc pow, incl. the subnormal tail
            if (p.T_zero_after) return 0.;
            return (double)p.kw0 * pow((double)p.kw1, (double)t);
        default: {  // SCHED_LOGARITHMIC
            if (t < p.T_len) return p.T_tab[t];
            const float x = (float)t + p.kw1;
            const unsigned long long i = (unsigned long long)x;
            return (double)p.kw0 / (i == 0 ? 0. : log((double)i));
        }
    }
}

// The production kernel's temperatures: the pow / log schedules come from the host table only (glibc, the reference's own
// values) -- bisbm_anneal hands every launch the slice of the call it covers, so no pow() / log() is compiled into that
// kernel (they cost it 30 - 110 spilled vector registers in the cooling-schedule variants).
__device__ __forceinline__ double temperature_tabled(const SweepParams& p, uint64_t t) {
    switch (p.schedule) {
        case SCHED_CONSTANT:
            return (double)p.kw0;
        case SCHED_ABRUPT:
            return ((float)t < p.kw0) ? 1. : 0.;
        case SCHED_LINEAR:
            return (double)(p.kw0 - p.kw1 * (float)t);
        default: {  // SCHED_EXPONENTIAL, SCHED_LOGARITHMIC
            const uint64_t i = t - p.T_base;
            return i < p.T_len ? p.T_tab[i] : 0.;  // (past the table: the exponential schedule after it has underflowed to 0)
        }
    }
}

// Production kernel: does a launch keep anneal()'s early-stop bookkeeping (metropolis_hasting.cc:85-98)?  It can only ever fire
// below T = 1 and when steps_await can be reached within the call.  A launch that does not keep it does not keep the running sum
// of accepted dS either (nobody looks at it during the launch): bisbm_anneal then advances the chain's sum by the change of the
// block-state part of the description length over the call (entropy_kernel before and after; the two agree to ~1e-13 relative,
// tests/test_gpu_scale.py), which takes the sum out of every pass.
__host__ __device__ inline bool sweep_fast_tracks_minimum(int schedule, float kw0, uint64_t steps_await, uint64_t call_duration) {
    return (schedule != SCHED_CONSTANT || (double)kw0 < 1.) && steps_await <= call_duration;
}
// scalars[c].cum_dS += after[c] - before[c]
hipError_t launch_sum_from_entropy(ChainScalars* scalars, const double* before, const double* after, uint32_t n_chains, hipStream_t stream);

hipError_t launch_sweep(const SweepParams& p, int rng_mode, size_t lds_bytes, hipStream_t stream);
hipError_t launch_sweep_fast(const SweepParams& p, size_t lds_bytes, hipStream_t stream);
size_t sweep_fast_lds_bytes(uint32_t ka, uint32_t kb, uint32_t maxdeg, bool eta_in_lds, uint32_t eta_window);
hipError_t launch_state_build(const BuildParams& p, hipStream_t stream);
// `labels` is the byte base of the label array; wide: two-byte labels (label_stride counts labels in both cases)
hipError_t launch_labels_broadcast(const uint32_t* src, uint8_t* labels, bool wide, size_t label_stride, uint32_t n,
                                   uint32_t first_chain, uint32_t n_chains, hipStream_t stream);
hipError_t launch_labels_widen(const uint8_t* labels, bool wide, uint32_t* dst, uint32_t n, hipStream_t stream);
hipError_t launch_labels_narrow(const uint8_t* wide_labels, uint8_t* labels, size_t label_stride, uint32_t n, uint32_t n_chains,
                                hipStream_t stream);
// relabelling after block merges: map1 / fmap are [n_chains][map_len] tables of labels (bytes, or two bytes when wide), first is
// [n_chains][map_len] (preset to ~0); map_len = the block count before the call rounded up to a multiple of 256
hipError_t launch_merge_first(const uint8_t* labels, bool wide, size_t label_stride, uint32_t n, uint32_t n_chains, uint32_t map_len,
                              const void* map1, uint32_t* first, hipStream_t stream);
hipError_t launch_merge_relabel(uint8_t* labels, bool wide, size_t label_stride, uint32_t n, uint32_t n_chains, uint32_t map_len,
                                const void* fmap, hipStream_t stream);
hipError_t launch_shuffle(const ShuffleParams& p, int rng_mode, hipStream_t stream);
hipError_t launch_entropy(const EntropyParams& p, hipStream_t stream);
hipError_t launch_marginals(const MarginalParams& p, hipStream_t stream);
// MAP labels of `rows` nodes from the histogram rows at `counts` (node `first` on); a += b over `count` counters
hipError_t launch_marginal_map(const uint32_t* counts, uint32_t rows, uint32_t kmax, uint32_t first, uint32_t n, uint32_t na,
                               uint32_t ka, uint16_t* labels_out, hipStream_t stream);
hipError_t launch_counts_add(uint32_t* a, const uint32_t* b, size_t count, hipStream_t stream);
hipError_t launch_log_q_probe(const Tables& tab, const int32_t* n, const int32_t* k, size_t count, double* out,
                              int fast, hipStream_t stream);

}  // namespace bisbm
