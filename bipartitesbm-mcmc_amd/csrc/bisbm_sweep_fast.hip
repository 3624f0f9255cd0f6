// bisbm_sweep_fast.hip -- the production sweep kernel: Philox mode, both block counts <= 64.
//
// metropolis_hasting::anneal / step / transition_ratio (metropolis_hasting.cc:42-192),
// single_vertex_change (blockmodel.cc:613-637) and apply_mcmc_moves (blockmodel.cc:461-503) for one
// chain per wavefront, persistent over all sweeps of the call.  Every number it produces is the one
// the generic kernel (bisbm_kernels.hip, sweep_kernel<RNG_PHILOX>) produces; the tests run both.
//
// A Philox-mode sweep visits all type-a nodes, then all type-b nodes, each class in a keyed, id-local
// permutation (TiledOrder; the two colour classes of the bipartite graph).  Inside a phase the visited nodes are
// never neighbours of each other, so everything a step reads about its neighbourhood is frozen:
//   * CSR and labels are consumed per 64 nodes, lane q <-> node q: every lane walks ITS adjacency row
//     straight from HBM, 16 neighbours at a time (four 16-byte id loads, 16 label gathers, all in
//     flight together), and counts the labels into its own row of k_v byte counters in LDS.
//     Per step that leaves one LDS read for k_v; no label pipeline, no replay of moves;
//   * m_r of the opposite type cannot change in the phase: 1/(m_r[t] + eps K) is a per-lane constant
//     of the phase (no division per step); the code of a phase is specialised on the type (no
//     per-step selects);
//   * the pivot neighbour of the proposal depends only on the step's uniform and the row, so its label
//     is picked up during the same walk.
// Two waves per chain: one runs the steps, the other ("feeder") prepares the NEXT chunk meanwhile --
// visit order, row extents, own labels, the label walk into a second set of k_v counters, the pivot
// neighbour's label -- and hands it over through LDS at one workgroup barrier per chunk.  The memory latency of
// a chunk's preparation (several dependent HBM round trips) is thereby off the step path entirely.  Which wave
// steps is settled at kernel start so that no SIMD of the chip hosts the stepping waves of two chains.
// The feeder also PREDICTS each step's inverse-CDF target from the block matrix as it stands a chunk ahead (round 4):
// the two-steps passes issue everything that depends on the target with their first reads and check the prediction
// against their own scan before anything is written (prepare, step_pair, step_pair64).
// State on chip: the a x b quadrant of m (odd row stride: rows and columns conflict-free) and eta in
// LDS; m_r / n_r in registers (lane i <-> block i of each type).  dS and the Hastings sums are DPP
// butterflies; the four log_q values are one SIMT evaluation; the four uniforms of a step come from one
// Philox evaluation per 64 steps per lane; apply_mcmc_moves writes values it already holds.
//
// A lone wave per SIMD pays one issue slot (~4.3 cycles) for EVERY instruction and 25-40 cycles for every
// vector->scalar->vector crossing (tools/probe/issue_latency.hip), so the step exists twice:
//   * the hot step: 1 <= deg <= 255, target drawn from column m[.][t], closed-form log_q tier.
//     Straight-line code, one test for "rare" at the top, r == s leaves right after the proposal, one tier
//     test, one crossing for the accept decision; lane patterns are bit arithmetic on per-lane constant
//     masks, the constants of the closed forms sit in vector registers (the scalar file is full);
//   * the general step: the literal definition (uniform random targets, empty and over-long rows, a
//     single own block, T = 0, every log_q tier).  The hot step hands over before it writes anything.
// No generic pointers (flat loads wait on vmcnt AND lgkmcnt), no workgroup barriers on the step path.
// Diagnostic hooks (BISBM_STAMPS, BISBM_ABLATE) are compiled out of the product build.
#include "bisbm_kernels.hpp"

#include <algorithm>
#include <cstdio>
#include <type_traits>

namespace bisbm {

#ifdef BISBM_STAMPS
__device__ unsigned long long g_fast_stamps[16];
#define FSTAMP(i)                                                                        \
    do {                                                                                 \
        unsigned long long now_;                                                         \
        __builtin_amdgcn_sched_barrier(0);                                               \
        __asm__ volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                               \
        st_acc[i] += now_ - st_prev;                                                     \
        st_prev = now_;                                                                  \
    } while (0)
#else
#define FSTAMP(i) \
    do {          \
    } while (0)
#endif
// BISBM_STAMPS=2: only the two per-chunk stamps (time in the chunk's steps / time waiting for the feeder), so
// that the step loop itself is the production code
#if defined(BISBM_STAMPS) && BISBM_STAMPS == 2
#define FSTAMP_STEP(i) \
    do {               \
    } while (0)
#else
#define FSTAMP_STEP(i) FSTAMP(i)
#endif

__device__ __forceinline__ void wfence() {
    __builtin_amdgcn_wave_barrier();
    __asm__ volatile("" ::: "memory");
}

// 8-byte table entry at a 32-bit element index: keeps the address in saddr + voffset form
__device__ __forceinline__ double tab_at(const double* base, uint32_t idx) {
#if defined(BISBM_ABLATE) && (BISBM_ABLATE & 8)
    return (double)idx * 1e-3 + (double)((size_t)base & 0xff);  // diagnostic build: table gathers removed (wrong results)
#endif
    return *(const double*)((const char*)base + (uint32_t)(idx << 3));  // tables are < 2^28 entries (host check)
}

struct __attribute__((packed, aligned(4))) Id4 {  // four consecutive column ids, 4-byte aligned
    uint32_t x, y, z, w;
};

// min(x, 1) on the scalar unit: x != 0 as a 0 / 1 word.  Written as the instruction itself: the compiler turns the
// C expression into a compare whose boolean it then materialises through the vector unit (v_cndmask +
// v_readfirstlane, ~30 cycles of latency each on a lone wave).
__device__ __forceinline__ uint32_t sflag(uint32_t x) {
    uint32_t r;
    __asm__("s_min_u32 %0, %1, 1" : "=s"(r) : "s"(x) : "scc");
    return r;
}
__device__ __forceinline__ uint32_t smin(uint32_t x, uint32_t y) {
    uint32_t r;
    __asm__("s_min_u32 %0, %1, %2" : "=s"(r) : "s"(x), "s"(y) : "scc");
    return r;
}

// BISBM_PREDICT_TARGET=0 compiles the predicted target out of the two-steps passes (A/B builds)
#ifndef BISBM_PREDICT_TARGET
#define BISBM_PREDICT_TARGET 1
#endif
constexpr uint32_t kHistStride = 68;  // bytes per k_v row: 64 counters + pad (17 dwords: odd, conflict-free)
constexpr uint32_t kHandWords = 9;    // v, row begin, degree, own label, pivot label, proposal word, packed hot-step inputs, accept uniform (2)
constexpr uint32_t kRowCap = 255;     // longest row the feeder walks (a k_v counter is a byte); longer rows: per-step side path

// a double constant held in a vector register pair for the whole kernel (64-bit literals are not encodable
// and the scalar file is full: without this the hot loop rebuilds them with s_mov pairs at every use)
#define BISBM_PIN(name, value) \
    double name = (value);     \
    __asm__ volatile("" : "+v"(name))

// EL: all of eta in LDS; otherwise a WINDOW of it: the rows of the phase's own type, p.eta_w consecutive degrees from
// p.eta_lo_a / p.eta_lo_b on (chosen by the host to hold most nodes), swapped at the phase change -- steps of nodes whose degree
// lies outside take the general path, which reads and writes those entries in HBM.  CT: constant schedule.  K32: both block counts <= 32 (five-level scans and sums).  K16 (with K32): both
// block counts <= 16: four steps per pass in the four 16-lane rows of the wave (step_quad).  K8 (with K16): both <= 8: eight
// steps per pass in groups of eight lanes (step_oct).  Q32 (with K32, without K16): four steps per pass with up to 32 blocks
// of a type, two blocks per lane (step_quad32) -- the kernel of a launch whose pass depth the host set to four.
template <bool EL, bool CT, bool K32, bool K16, bool K8, bool Q32>
__global__ __launch_bounds__(2 * kWave, 2) void sweep_fast_kernel(SweepParams p) {
    extern __shared__ __align__(16) uint32_t lds32[];
    const uint32_t chain = blockIdx.x;
    if (chain >= p.n_chains) return;
    const uint32_t lane = threadIdx.x & 63u;
    if (p.resume != 0u && p.scalars[chain].stopped != 0u) {  // a later launch of a call this chain has returned from (:96-98)
        if (threadIdx.x == 0) {
            p.scalars[chain].last_accepted = 0;
            p.scalars[chain].last_sweeps = 0;
        }
        return;
    }
    const uint32_t ka = p.ka, kb = p.kb, K = ka + kb, na = p.na, nb = p.nb;
    const uint32_t D = p.maxdeg + 1, S = kb | 1u;
    // the two-steps passes (step_pair, step_pair64) start from a predicted inverse-CDF target (see prepare and step_pair)
    constexpr bool kPredictTarget = !K16 && !Q32 && (BISBM_PREDICT_TARGET != 0);
    const uint32_t row_cap = p.maxdeg < kRowCap ? p.maxdeg : kRowCap;  // neighbours walked per row by the feeder
    // LDS layout, dword offsets
    const uint32_t o_mq = 0, o_eta = ka * S;
    const uint32_t eta_w = EL ? D : p.eta_w;                                  // degrees per row of eta in LDS
    const uint32_t o_hist8 = o_eta + (EL ? K * D : (ka > kb ? ka : kb) * eta_w);  // two buffers of k_v rows
    const uint32_t o_hand = o_hist8 + 2 * kWave * (kHistStride / 4);         // two hand-off buffers, kHandWords x 64 dwords
    const uint32_t o_slow = o_hand + 2 * kHandWords * kWave;
    const uint32_t o_flag = o_slow + kWave;
    int32_t* const mq = (int32_t*)(lds32 + o_mq);
    uint32_t* const eta_l = lds32 + o_eta;
    uint32_t* const hist8_base = lds32 + o_hist8;           // k_v byte counters: 2 buffers x 64 rows
    uint32_t* const hand_base = lds32 + o_hand;             // per-chunk lane data handed from the feeder wave
    int32_t* const slow_hist = (int32_t*)(lds32 + o_slow);  // 64 counters for rows longer than 64
    uint32_t* const stop_flag = lds32 + o_flag;
    // steps with T < 1 so far (anneal()'s early stop, see emin_l0): read and written once per chunk, so an LDS word
    // pair rather than a loop-carried scalar
    unsigned long long* const below1_total = (unsigned long long*)(lds32 + ((o_flag + 4u + 1u) & ~1u));
    // Which of the two waves steps and which feeds.  The dispatcher puts the two waves of a workgroup on different
    // SIMDs and, with four workgroups per CU, two waves on every SIMD -- but not always one wave 0 and one wave 1: now
    // and then a SIMD gets the wave 0 of two workgroups.  Two stepping waves on one SIMD run ~20 % slower each, and a
    // launch lasts as long as its slowest chain (one such SIMD on the chip: 0.73 s -> 0.89 s per sweep, measured).  So
    // the workgroup claims a SIMD for its stepping wave in a chip-wide table: wave 0's if that SIMD has no stepping
    // wave yet, else wave 1's.  The roles are symmetric: results do not depend on the choice.
    const uint32_t wave_in_wg = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t hw_id, xcc_id;
    __asm__ volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    __asm__ volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    uint32_t* const role = lds32 + o_flag + 1;  // [0], [1]: the waves' SIMD slots; [2]: the wave that steps
    // New labels of the chunk's nodes, one byte per step (0xff: the node stayed): a step that moves its node writes here, and
    // the chunk's label stores to HBM go out together when its steps are done.  A node is visited once per sweep and the
    // labels of its own type are read by nobody during the phase, so the delay changes nothing -- but a store issued inside
    // the pass is still in flight at the top of the next one, where the compiler's wait for "all vector memory operations"
    // (it cannot prove that no load into a reused register is pending) then waits for it.
    uint8_t* const new_lab = (uint8_t*)(lds32 + o_flag + 8);
    // m_r of the phase's OPPOSITE type (frozen during the phase), published by the stepping wave when the phase begins: the
    // feeder wave works out the proposals' random part with it (prepare)
    int32_t* const mr_pub = (int32_t*)(lds32 + o_flag + 24);
    // 1 / T of the chunk's 64 steps (cooling schedules), written by the stepping wave when the chunk begins and read per pass
    double* const invT_buf = (double*)(lds32 + o_flag + 88);
    if (lane == 0) role[wave_in_wg] = ((xcc_id & 0xfu) << 10) | ((hw_id >> 6) & 0x3fcu) | ((hw_id >> 4) & 3u);
    __syncthreads();
    if (wave_in_wg == 0 && lane == 0) {
        uint32_t pick = p.simd_claims != nullptr ? 0u : (p.fixed_stepping_wave & 1u);
        if (p.simd_claims != nullptr && atomicAdd(&p.simd_claims[role[0]], 1u) != 0u) {
            if (atomicAdd(&p.simd_claims[role[1]], 1u) == 0u) {
                pick = 1;
                atomicSub(&p.simd_claims[role[0]], 1u);
            } else {
                atomicSub(&p.simd_claims[role[1]], 1u);  // both taken: stay with wave 0
            }
        }
        role[2] = pick;
    }
    __syncthreads();
    const bool is_main = (uint32_t)__builtin_amdgcn_readfirstlane((int)role[2]) == wave_in_wg;

    uint8_t* const labels = p.labels + (size_t)chain * p.label_stride;
    int32_t* const m_g = p.m + (size_t)chain * ka * kb;
    int32_t* const mr_g = p.m_r + (size_t)chain * K;
    int32_t* const nr_g = p.n_r + (size_t)chain * K;
    uint32_t* const eta_g = p.eta + (size_t)chain * K * D;
    ChainScalars* const sc = p.scalars + chain;
    const Tables tab{p.lgamma_tab, p.lgamma_size, p.q_tab, p.q_stride, p.log_tab};
    if (lane == 0) {  // placement record: which SIMD of which CU the two waves landed on
        sc->hw_id[is_main ? 0 : 1] = hw_id;
        if (is_main) sc->xcc_id = xcc_id;
    }

    // chain state -> LDS / registers (wave 0 owns it)
    if (is_main) {
        for (uint32_t i = lane; i < ka * kb; i += kWave) mq[(i / kb) * S + (i % kb)] = m_g[i];
        if (EL)
            for (uint32_t i = lane; i < K * D; i += kWave) eta_l[i] = eta_g[i];
        if (lane == 0) {
            *stop_flag = 0;
            *below1_total = p.resume != 0u ? sc->stop_below1 : 0ull;
        }
    }
    // The stepping wave shares its SIMD with the feeder wave of another chain (four chains per CU): it is the
    // critical path, so it issues first whenever both have an instruction ready.
    if (is_main) __builtin_amdgcn_s_setprio(3);
    // lane <-> block: m_r / n_r of block i sit in lane i; the K <= 32 variants keep a second copy in lane 32 + i, so that
    // the upper half of the wave can run a step of its own (step_pair below); the K <= 16 variants keep four copies (step_quad)
    const uint32_t lb = K8 ? (lane & 7u) : K16 ? (lane & 15u) : K32 ? (lane & 31u) : lane;
    int mrA = lb < ka ? mr_g[lb] : 0, nrA = lb < ka ? nr_g[lb] : 0;
    int mrB = lb < kb ? mr_g[ka + lb] : 0, nrB = lb < kb ? nr_g[ka + lb] : 0;
    __syncthreads();
    uint64_t sweeps_total = sc->sweeps_total;

    const double eps = p.epsilon;
    const double Kd = (double)K;
    const double epsK = eps * Kd;
    const uint32_t chain_gid = chain_gid_of(p, chain);
    const uint32_t n = p.n;
    const uint64_t all_sweeps = p.duration / n;
    // A temperature so small that 1 / T overflows (the subnormal tail of an exponential schedule on its way to 0) acts like
    // T = 0 in the reference's own arithmetic: -1 / T * dS is -inf, +inf or (dS = 0, r == s included) NaN, so a step is accepted
    // exactly when dS < 0 (metropolis_hasting.cc:54-59).  Such a T is replaced by 0 here, once, where the temperatures are made.
    // (A constant schedule cannot get there: its T is a float.)
    const double T_const = (double)p.kw0;  // CT: constant schedule (metropolis_hasting.cc:25-28)
    // the early-stop bookkeeping can only ever fire below T = 1, and only if steps_await can be reached within the call
    // (the counter starts at 0 and gains at most 1 per step) -- a scalar word, not a lane mask: one s_cmp to test
    // two steps per pass (step_pair): K <= 32 (a constant schedule at T = 0 takes the general step anyway);
    // p.pair_steps == 0 switches it off (A/B runs, tests); 2: four steps per pass where both block counts are <= 16
    const bool pair_mode = (uint32_t)__builtin_amdgcn_readfirstlane((K32 && (!CT || T_const > 0.) && p.pair_steps != 0) ? 1 : 0) != 0u;
    // two steps per pass with more than 32 blocks of a type (step_pair64: two leaves per lane)
    const bool pair64_mode = (uint32_t)__builtin_amdgcn_readfirstlane((!K32 && (!CT || T_const > 0.) && p.pair_steps != 0) ? 1 : 0) != 0u;
    const bool quad_mode = (uint32_t)__builtin_amdgcn_readfirstlane((K16 && !K8 && (!CT || T_const > 0.) && p.pair_steps > 1u) ? 1 : 0) != 0u;
    const bool quad32_mode = (uint32_t)__builtin_amdgcn_readfirstlane((Q32 && (!CT || T_const > 0.) && p.pair_steps > 1u) ? 1 : 0) != 0u;
    const bool oct_mode = (uint32_t)__builtin_amdgcn_readfirstlane((K8 && (!CT || T_const > 0.) && p.pair_steps > 2u) ? 1 : 0) != 0u;
    const uint32_t track_min =
        (uint32_t)__builtin_amdgcn_readfirstlane((sweep_fast_tracks_minimum(p.schedule, p.kw0, p.steps_await, p.call_duration) || p.keep_sum != 0u) ? 1 : 0);
    // constants of the hot step (log_q closed form, accept filter)
    BISBM_PIN(c_l2e, 0x1.71547652b82fep+0);        // log2(e)
    double c_tol = 1e-5;                           // accept filter margin
    if (!(Q32 && !CT) && !(!K32 && !EL && !CT)) __asm__ volatile("" : "+v"(c_tol));  // (pinned, except in the variants where registers are scarcest)
    LogQConsts lqc = log_q_consts();  // log_q closed form
    if (!(Q32 && !EL))  // (held in vector registers for the whole kernel, except where registers are scarcest: built at the use there)
        __asm__ volatile("" : "+v"(lqc.nc0l2e), "+v"(lqc.c1c0), "+v"(lqc.c1), "+v"(lqc.c2c0), "+v"(lqc.lfc));
    double c_576 = kDirectU2;                      // 18^2: tier test k^2 > 324 n (the name is of the round when it was 24^2)
    if ((K32 && !Q32) || CT) __asm__ volatile("" : "+v"(c_576));  // (pinned like the others, except where registers are scarcest)
    double c_169 = 169.0;                          // 13^2: tier test k^2 >= 169 n
    if (K32 && !Q32) __asm__ volatile("" : "+v"(c_169));   // (pinned like the others, except in the two-blocks-per-lane variants: registers)
    uint64_t sweeps_done = 0;
    // Sum of accepted dS (blockmodel_t::entropy_) and accepted count: lane 0's copy is the value.  They are bumped
    // inside the lane-0 region of an accepted step (a vector add under the execution mask, no LDS round trip).
    double cum_l0 = sc->cum_dS;
    unsigned long long acc_l0 = 0;
    // anneal()'s early stop (metropolis_hasting.cc:75-76,85-98): `u`, the number of steps with T < 1 since entropy_ last
    // reached a new minimum, is only looked at when a sweep ends.  Kept as u = (steps with T < 1 so far) - (steps with
    // T < 1 before the step of the last minimum): the first count advances once per chunk, the second (and the minimum
    // itself) lives in lane 0's registers and is touched by accepted moves only -- a rejected step does nothing.
    double emin_l0 = p.resume != 0u ? sc->stop_emin : INFINITY;
    unsigned long long mark_l0 = p.resume != 0u ? sc->stop_mark : 0ull;
    double rate = 0.;
    bool stopped = false;
#ifdef BISBM_STAMPS
    unsigned long long st_prev, st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    __asm__ volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
#endif

    for (uint64_t sweep = 0; sweep < all_sweeps; ++sweep) {
        const uint64_t sweep_step0 = p.t_base + (uint64_t)n * sweep;  // metropolis_hasting.cc:82

        // One phase = every node of one type.  TB: the phase's nodes are type b.
        auto run_phase = [&](auto tb_tag) {
            constexpr bool TB = decltype(tb_tag)::value;
            const uint32_t n_own = TB ? nb : na, node_base = TB ? na : 0u;
            const uint32_t k_own = TB ? kb : ka, k_oth = TB ? ka : kb;
            const uint32_t own_base = TB ? ka : 0u, oth_base = TB ? 0u : ka;
            int& mr_own = TB ? mrB : mrA;
            int& nr_own = TB ? nrB : nrA;
            const int mr_oth = TB ? mrA : mrB;             // frozen during the phase
            // 1 / (m_r[t] + eps K), lane t <-> opposite block t.  With epsilon = 0 the denominator is zero for idle lanes and
            // for blocks without edges; their k is zero, and 0 * finite * 0 keeps their leaves at zero (0 * inf would not)
            const double den_oth = mr_oth + epsK;
            const double inv_blk = den_oth > 0. ? 1.0 / den_oth : 0.;  // lane <-> block (one copy per half / row / group below 33 blocks)
            // m[own block i][opposite block j] in the a x b quadrant
            auto mq_at = [&](uint32_t i_own, uint32_t j_oth) -> uint32_t {
                return TB ? j_oth * S + i_own : i_own * S + j_oth;
            };
            // eta[own block i][degree]: index into eta_l (hot paths: the degree is inside the window), and the accessors of
            // the general path, which also serve the degrees outside it (from HBM)
            const uint32_t eta_lo = EL ? 0u : (TB ? p.eta_lo_b : p.eta_lo_a);
            auto eta_at = [&](uint32_t i_own, uint32_t deg) -> uint32_t {
                return EL ? (own_base + i_own) * D + deg : i_own * eta_w + (deg - eta_lo);
            };
            auto eta_any_rd = [&](uint32_t i_own, uint32_t deg) -> uint32_t {
                if (EL || deg - eta_lo < eta_w) return eta_l[eta_at(i_own, deg)];
                return eta_g[(own_base + i_own) * D + deg];
            };
            auto eta_any_wr = [&](uint32_t i_own, uint32_t deg, uint32_t val) {
                if (EL || deg - eta_lo < eta_w)
                    eta_l[eta_at(i_own, deg)] = val;
                else
                    eta_g[(own_base + i_own) * D + deg] = val;
            };
            if (!EL && is_main) {  // the window of this phase's type: HBM -> LDS (written back when the phase ends)
                for (uint32_t i = lane; i < k_own * eta_w; i += kWave) {
                    const uint32_t row = i / eta_w, d = eta_lo + i % eta_w;
                    eta_l[i] = d < D ? eta_g[(own_base + row) * D + d] : 0u;
                }
                wfence();
            }
            // step_pair64 (more than 32 blocks of a type, two steps per pass): lane l of a half holds the opposite-type blocks
            // l and l + 32 (two leaves per lane, added first: level 32 of the summation tree) and, for the inverse CDF, the
            // own-type blocks l and l + 32.  The K > 32 variant keeps 1 / (m_r[t] + eps K) for exactly these two blocks; the
            // one-step paths pick theirs (lane <-> block) out of the pair.
            const uint32_t lh = lane & 31u;
            const double inv_lo = __hiloint2double(__builtin_amdgcn_ds_bpermute((int)(lh << 2), __double2hiint(inv_blk)),
                                                   __builtin_amdgcn_ds_bpermute((int)(lh << 2), __double2loint(inv_blk)));
            const double inv_hi = __hiloint2double(__builtin_amdgcn_ds_bpermute((int)((lh + 32u) << 2), __double2hiint(inv_blk)),
                                                   __builtin_amdgcn_ds_bpermute((int)((lh + 32u) << 2), __double2loint(inv_blk)));
            auto inv_oth_of_lane = [&]() -> double { return K32 ? inv_blk : (lane < 32u ? inv_lo : inv_hi); };
            // the scalar terms of dS sit in leaves 0..7 (lgamma) / 0..3 (log_q) of a step's sum: lane pattern by the lane's
            // position in its half / row / group (K > 32: in its half, for step_pair64)
            // step_quad32 (17..32 blocks of a type, four steps per pass): lane l of a 16-lane row holds the opposite-type blocks
            // l and l + 16 (two leaves per lane) and, for the inverse CDF, the own-type blocks l and l + 16
            const uint32_t l16 = lane & 15u;
            const double invq_lo = __hiloint2double(__builtin_amdgcn_ds_bpermute((int)(l16 << 2), __double2hiint(inv_blk)),
                                                    __builtin_amdgcn_ds_bpermute((int)(l16 << 2), __double2loint(inv_blk)));
            const double invq_hi = __hiloint2double(__builtin_amdgcn_ds_bpermute((int)((l16 + 16u) << 2), __double2hiint(inv_blk)),
                                                    __builtin_amdgcn_ds_bpermute((int)((l16 + 16u) << 2), __double2loint(inv_blk)));
            const uint32_t ls = Q32 ? l16 : (K32 ? lb : lh);
            const double sign_tail = ls >= 8 ? 0. : ((ls < 2 || ls >= 6) ? -1. : 1.);
            const double sign_q = ls >= 4 ? 0. : (ls < 2 ? -1. : 1.);
            // the one-step evaluations (step_general, step) sum lanes 0..31 or 0..63 as ONE step: only its first row carries
            // the scalar terms (with several copies per wave -- K <= 16, or the two halves of the K > 32 variant -- the other
            // copies would add them again).  K <= 16: registers of their own; K > 32: selected at the use (registers are
            // scarce there); K <= 32: the same registers as sign_tail / sign_q.
            const double sign_tail1_r = (K16 || Q32) ? (lane < 8u ? sign_tail : 0.) : sign_tail;
            const double sign_q1_r = (K16 || Q32) ? (lane < 8u ? sign_q : 0.) : sign_q;
            auto sign_tail1_of = [&]() -> double { return K32 ? sign_tail1_r : (lane < 32u ? sign_tail : 0.); };
            auto sign_q1_of = [&]() -> double { return K32 ? sign_q1_r : (lane < 32u ? sign_q : 0.); };
            const int eoff_l = (lane & 7u) < 6 ? 1 : ((lane & 1u) ? 2 : 0);       // eta_r+1, eta_s+1, eta_r, eta_s+2
            const int dq_l = (lane & 2u) ? ((lane & 1u) ? 1 : -1) : 0;            // n_r - 1, n_s + 1 in lanes 2,3 (mod 4)
            const int dsgn_l = dq_l;                                              // -deg, +deg in the same lanes
            int odd_mask_l = (lane & 1u) ? -1 : 0;
            int eta_mask_l = (lane & 4u) ? -1 : 0;                                // lanes 4..7 (mod 8): the eta terms
            __asm__ volatile("" : "+v"(odd_mask_l), "+v"(eta_mask_l));            // (opaque: stay bit operations on vector registers)
            const int toff_l = (lane & 4u) ? eoff_l : 1;                          // table index = argument + this
            TiledOrder order;
            order.init(phx_draw(p.seed, chain_gid, PHX_SWEEP_KEY, 2 * sweeps_total + (TB ? 1 : 0)), n_own);
            const uint32_t n_chunks = (n_own + kWave - 1) / kWave;
            const unsigned long long lanes_koth = __builtin_amdgcn_ballot_w64(lb < k_oth);  // both halves in the K <= 32 variants
            const unsigned long long lanes_koth64_lo = __builtin_amdgcn_ballot_w64((lane & 31u) < k_oth);       // step_pair64: the lane's
            const unsigned long long lanes_koth64_hi = __builtin_amdgcn_ballot_w64((lane & 31u) + 32u < k_oth);  // two opposite-type blocks
            const unsigned long long lanes_koth32_lo = __builtin_amdgcn_ballot_w64((lane & 15u) < k_oth);        // step_quad32: the lane's
            const unsigned long long lanes_koth32_hi = __builtin_amdgcn_ballot_w64((lane & 15u) + 16u < k_oth);  // two opposite-type blocks
            const uint32_t node_other0 = TB ? 0u : na;  // some node of the opposite type: what idle slots of the walk load

            // The proposal's random part (blockmodel.cc:619-628), worked out for all 64 steps of a chunk at once (lane = step) by
            // the feeder wave: the opposite type's labels and m_r are frozen during the phase, so the R test (:622-624) and the
            // inverse-CDF target x do not depend on the moves made inside the chunk.  Bit 31 set: uniform random block (low
            // bits); clear: x, to be looked up in the current column m[.][t] at step time.
            auto draw_target = [&](int32_t mrt, double u_R, double u_tgt) -> uint32_t {
                if (u_R * (mrt + epsK) < epsK) {  // :622-624
                    uint32_t sR = (uint32_t)(u_tgt * Kd);
                    if (sR >= K) sR = K - 1;
                    return 0x80000000u | sR;
                }
                uint32_t x = (uint32_t)(u_tgt * (double)mrt);  // m_r < 2^31
                if (x >= (uint32_t)mrt) x = (uint32_t)mrt - 1u;
                return x;
            };

            // ---- feeder wave: everything of chunk c that does not depend on the chain's block state ----
            auto prepare = [&](uint32_t c) {
                const uint32_t vi0 = c * kWave;
                const uint32_t cnt = (n_own - vi0) < (uint32_t)kWave ? (n_own - vi0) : (uint32_t)kWave;
                uint32_t* const hist8w = hist8_base + (c & 1u) * kWave * (kHistStride / 4);
                uint32_t* const hand = hand_base + (c & 1u) * kHandWords * kWave;
                uint32_t v_l = 0, beg_l = 0, deg_l = 0, r_l = 0, which_l = 0;
                if (lane < cnt) {
                    v_l = node_base + order(vi0 + lane);
                    beg_l = p.rowptr[v_l];
                    deg_l = p.rowptr[v_l + 1] - beg_l;
                    r_l = labels[v_l];  // own label: stable until the node's own step
                    const U4 A = phx_draw(p.seed, chain_gid, PHX_STEP_A, sweeps_total * (uint64_t)n + node_base + vi0 + lane);
                    which_l = (uint32_t)(u53(A.x, A.y) * (double)deg_l);  // pivot neighbour, blockmodel.cc:619
                    if (which_l >= deg_l) which_l = deg_l ? deg_l - 1 : 0;
                }
                // k_v rows: zero, then every lane walks its own adjacency row straight from HBM, 16 neighbours at a
                // time: four 16-byte loads of ids (a lane's ids are contiguous; col[] carries four spare entries so the
                // last row may be read past), 16 label gathers, 16 counter bumps.  No per-neighbour branches: idle
                // slots load node 0 and add zero.
                for (uint32_t w = 0; w < kHistStride / 4; ++w) hist8w[lane * (kHistStride / 4) + w] = 0;
                wfence();
                const bool mine = lane < cnt && deg_l <= kRowCap;  // longer rows: per-step side path
#if defined(BISBM_ABLATE) && (BISBM_ABLATE & 64)
                const uint32_t my_deg = 0u;  // diagnostic build: the feeder walks nothing, every k_v is zero (wrong results)
#else
                const uint32_t my_deg = mine ? deg_l : 0u;
#endif
                // the pivot neighbour's label (:619-620): two dependent loads, in flight during the walk
                const int piv_l = (int)labels[my_deg ? p.col[beg_l + which_l] : node_other0];
                {
                    constexpr int U = 16;
                    const uint32_t row_byte0 = lane * kHistStride;
                    for (uint32_t j0 = 0; j0 < row_cap; j0 += U) {
                        if (__ballot(j0 < my_deg) == 0) break;
                        uint32_t idv[U];
#pragma unroll
                        for (int u = 0; u < U; u += 4) {  // (only lanes whose row reaches this far: no sectors fetched for nothing)
                            Id4 v4 = {0u, 0u, 0u, 0u};
                            if (j0 + u < my_deg) v4 = *(const Id4*)(p.col + beg_l + j0 + u);
                            idv[u + 0] = v4.x, idv[u + 1] = v4.y, idv[u + 2] = v4.z, idv[u + 3] = v4.w;
                        }
                        int lab[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
#if defined(BISBM_ABLATE) && (BISBM_ABLATE & 256)
                            lab[u] = (int)(oth_base + (idv[u] & 31u) % k_oth);  // diagnostic (with bit 8 only): no label gathers
#else
                            // (plain cached loads: with the id-local visit order the label sectors are re-used by the
                            // following nodes out of the L2 -- non-temporal loads, +3.5 % under the old scattered
                            // order, cost 5 % here)
                            lab[u] = labels[j0 + u < my_deg ? idv[u] : node_other0];
#endif
                        }
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const uint32_t byte = row_byte0 + (uint32_t)(lab[u] - (int)oth_base);
                            const uint32_t one = j0 + u < my_deg ? 1u : 0u;
#if defined(BISBM_ABLATE) && (BISBM_ABLATE & 128)
                            if (byte == 0xffffffffu) hist8w[0] = one;  // diagnostic (with bit 8 only): no counter bumps
#else
                            atomicAdd(&hist8w[byte >> 2], one << ((byte & 3u) * 8u));
#endif
                        }
                    }
                }
                hand[0 * kWave + lane] = v_l;
                hand[1 * kWave + lane] = beg_l;
                hand[2 * kWave + lane] = deg_l;
                hand[3 * kWave + lane] = r_l;
                hand[4 * kWave + lane] = (uint32_t)piv_l;
                // Per-lane inputs of the hot step.  Bit 31 of the proposal word sends the step down the general path: uniform
                // random target, a single own block, empty or over-long rows, T == 0, a degree outside the eta window.
                // (the other three uniforms of step vi0 + lane are drawn here, after the walk -- counter-based: the draw for the
                // pivot above is simply repeated -- so that they do not occupy registers during it)
                double ud_R = 0., ud_tgt = 0., ud_acc = 0.;
                if (lane < cnt) {
                    const uint64_t gs = sweeps_total * (uint64_t)n + node_base + vi0 + lane;
                    const U4 A = phx_draw(p.seed, chain_gid, PHX_STEP_A, gs);
                    const U4 B = phx_draw(p.seed, chain_gid, PHX_STEP_B, gs);
                    ud_R = u53(A.z, A.w);
                    ud_tgt = u53(B.x, B.y);
                    ud_acc = u53(B.z, B.w);
                }
                const uint32_t tloc_l = ((uint32_t)piv_l - oth_base) & 63u;
                const uint32_t rloc_l = r_l - own_base;
                uint32_t prop_l = draw_target(mr_pub[tloc_l], ud_R, ud_tgt);
                if (k_own == 1u || deg_l == 0u || deg_l > kRowCap || (CT && T_const == 0.)) prop_l |= 0x80000000u;
                if (!EL && deg_l - eta_lo >= eta_w) prop_l |= 0x80000000u;  // eta[.][deg] is not in the LDS window
                hand[5 * kWave + lane] = prop_l;
                // The inverse-CDF target of the step as the block state stands NOW, one chunk ahead of the step (kPredictTarget:
                // the kernels of the two-steps passes): the first own-type block whose running sum of column m[.][t] exceeds x
                // (:627-628).  A PREDICTION -- the stepping wave is moving nodes while this reads the matrix, and up to 128 steps
                // will have run before the step itself -- that lets the step issue everything that depends on its target together
                // with its first reads; the step still does the scan on the state of its own moment and goes ahead only when the
                // two agree (a boundary of the running sums moves by a few units per chunk against bins of ~m_r / K: they do in
                // all but ~1e-3 of the steps at BASELINE configs[2]).  Cost: one LDS read, four vector instructions per own-type
                // block and step, on the wave that has the time.
                uint32_t spred_l = 0u;
                if constexpr (kPredictTarget) {
                    const uint32_t x = prop_l & 0x7fffffffu;
                    uint32_t cum = 0u;
                    for (uint32_t i = 0; i < k_own; ++i) {
                        cum += (uint32_t)mq[mq_at(i, tloc_l)];
                        spred_l += cum <= x ? 1u : 0u;
                    }
                    spred_l = spred_l < k_own ? spred_l : k_own - 1u;
                }
                // degree (<= 255 in the hot step), own block, pivot block and predicted target in one word (one cross-lane move per pass)
                hand[6 * kWave + lane] = (deg_l & 255u) | ((rloc_l & 63u) << 8) | (tloc_l << 16) | (spred_l << 24);
                *(double*)&hand[7 * kWave + 2 * lane] = ud_acc;  // (words 7, 8: the 64 accept uniforms as doubles)
                wfence();
            };

            // ---- main wave: the 64 steps of chunk c (metropolis_hasting.cc:42-62 each) ----
            auto run_steps = [&](uint32_t c) {
                const uint32_t vi0 = c * kWave;
                const uint32_t cnt = (n_own - vi0) < (uint32_t)kWave ? (n_own - vi0) : (uint32_t)kWave;
                const uint8_t* const hist8_cur = (const uint8_t*)(hist8_base + (c & 1u) * kWave * (kHistStride / 4));
                const uint32_t* const hand = hand_base + (c & 1u) * kHandWords * kWave;
                // Per-lane data of the chunk that lives through its steps, all of it handed over by the feeder wave: the node, the
                // proposal word, the packed inputs of the hot step, the accept uniform.  Everything else the rare general step
                // needs -- row begin, full degree, own and pivot label, the other three uniforms -- it reads back from the hand-off
                // buffer / draws again itself (counter-based).
                // (Only the packed word, which the passes fetch with v_readlane, is kept in a register through the chunk; the
                // proposal word, the accept uniform and 1 / T are read from LDS at a per-half / per-row address by every pass.)
                new_lab[lane] = 0xffu;
                const uint32_t pack_l = hand[6 * kWave + lane];
                unsigned long long gen_mask = __builtin_amdgcn_ballot_w64((int32_t)hand[5 * kWave + lane] < 0);  // steps that need the general path
                auto prop_of = [&](uint32_t step) -> uint32_t { return hand[5 * kWave + step]; };
                auto u_acc_of = [&](uint32_t step) -> double { return *(const double*)&hand[7 * kWave + 2 * step]; };
                // the temperatures of the 64 steps (:84), lane = step: one table read or one pow / log per lane per chunk
                double T_l = T_const;
                if (!CT) T_l = temperature_tabled(p, sweep_step0 + node_base + vi0 + lane);
                const double invT_l = 1.0 / T_l;  // (T == 0: not used, the step is decided by the sign of dS)
                if (!CT && invT_l == INFINITY) T_l = 0.;  // (see T_const)
                if (!CT) {
                    invT_buf[lane] = invT_l;
                    wfence();
                }
                auto invT_of = [&](uint32_t step) -> double { return invT_buf[step]; };
                // (the general step evaluates its temperature again -- a table read at a wave-uniform index -- instead of keeping
                // the chunk's 64 temperatures alive through the step loop for it)
                auto T_of_step = [&](uint32_t q) -> double {
                    if (CT) return T_const;
                    const double T = temperature_tabled(p, sweep_step0 + node_base + vi0 + q);
                    return 1.0 / T == INFINITY ? 0. : T;
                };
                // anneal()'s bookkeeping, metropolis_hasting.cc:85-94, see emin_l0 above
                const unsigned long long below1_mask =
                    track_min != 0u ? __builtin_amdgcn_ballot_w64(lane < cnt && T_l < 1.) : 0ull;
                const unsigned long long below1_before = track_min != 0u ? *below1_total : 0ull;
                // (:85-91 look at EVERY accepted step, also one whose proposal was r == s: that matters exactly once per call --
                // the first accepted step finds entropy_min = +inf and sets it to the current sum whether or not it moved its
                // node; afterwards an accepted r == s step finds the sum unchanged, hence not below the minimum)
                auto new_minimum = [&](uint32_t q) {  // after accepted step q (cum_l0 is updated)
                    const unsigned long long before =
                        below1_before + (unsigned long long)__builtin_popcountll(below1_mask & ((1ull << q) - 1ull));
                    const bool better = cum_l0 < emin_l0;  // (:87-90)
                    emin_l0 = better ? cum_l0 : emin_l0;
                    mark_l0 = better ? before : mark_l0;
                };
                auto book = [&](bool ok, double T) {
                    if (ok && lane == 0) acc_l0 += 1;
                };

                // ---- any step (all the rare cases included): the definition the hot path below specialises ----
                auto step_general = [&](uint32_t q, double T) {
                    auto handed = [&](uint32_t word) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)hand[word * kWave + q]); };
                    const uint32_t deg = handed(2), r = handed(3);
                    const uint32_t r_loc = r - own_base;
                    // the step's four uniforms (:619-628, :57), drawn again here: wave-uniform values
                    const uint64_t gs_q = sweeps_total * (uint64_t)n + node_base + vi0 + q;
                    const U4 gA = phx_draw(p.seed, chain_gid, PHX_STEP_A, gs_q), gB = phx_draw(p.seed, chain_gid, PHX_STEP_B, gs_q);
                    const double u_idx = u53(gA.x, gA.y), u_R = u53(gA.z, gA.w), u_tgt = u53(gB.x, gB.y), u_acc = u53(gB.z, gB.w);
                    // k_v counter of lane's block, row r of m, eta[r][deg] (unconditional reads, idle lanes masked
                    // afterwards: the k_v bytes past k_oth are zero, and the m reads past the quadrant stay inside
                    // the kernel's LDS allocation, see sweep_fast_lds_bytes)
                    // (k_v read as a dword + shift: a different instruction from the hot step's byte read, so the
                    // compiler does not merge the two above the branch between the paths and wait for it there)
                    const uint32_t kbyte = q * kHistStride + lane;
                    int k = (int)((((const uint32_t*)hist8_cur)[kbyte >> 2] >> ((kbyte & 3u) * 8u)) & 0xffu);
                    const uint32_t a_rt = mq_at(r_loc, lane);
                    const int32_t m_rt_raw = mq[a_rt];
                    const int32_t m_rt = lane < k_oth ? m_rt_raw : 0;
                    const int eta_r = (int)eta_any_rd(r_loc, deg);
                    uint32_t t_piv = handed(4);
                    if (deg > kRowCap) {  // rows the feeder does not walk: straight from HBM
                        const uint32_t beg = handed(1);
                        uint32_t which = (uint32_t)(u_idx * (double)deg);  // pivot neighbour, blockmodel.cc:619
                        if (which >= deg) which = deg - 1;
                        slow_hist[lane] = 0;
                        wfence();
                        for (uint32_t j = lane; j < deg; j += kWave)
                            atomicAdd(&slow_hist[(int)labels[p.col[beg + j]] - (int)oth_base], 1);
                        wfence();
                        k = lane < k_oth ? slow_hist[lane] : 0;
                        t_piv = labels[p.col[beg + which]];
                    }
                    // column t of m over v's own type; lanes >= k_own read past it and are ignored by the ballot
                    const int w_piv = mq[mq_at(lane, t_piv - oth_base)];

                    // ---- proposal: single_vertex_change, blockmodel.cc:613-637 ----
                    uint32_t s;
                    if (k_own == 1) {
                        s = r;
                    } else if (deg == 0) {
                        s = (uint32_t)(u_idx * Kd);
                        if (s >= K) s = K - 1;
                    } else {
                        const uint32_t prop = draw_target(readlane(mr_oth, t_piv - oth_base), u_R, u_tgt);
                        if ((int32_t)prop < 0) {
                            s = prop & 0x7fffffffu;
                        } else {  // integer inverse CDF over column m[.][t] restricted to v's own type (:627-628)
                            const int scan = wave_inclusive_scan(w_piv);
                            const unsigned long long hit = __ballot(lane < k_own && (uint32_t)scan > prop);
                            s = hit ? own_base + (uint32_t)__ffsll((long long)hit) - 1 : own_base + k_own - 1;
                        }
                    }

                    // ---- transition_ratio, metropolis_hasting.cc:103-192 (production arithmetic) ----
                    // r == s (always accepted at T > 0, :109-112) and cross-type targets (dS = +inf, :121-123):
                    // same straight-line code with s replaced by r, outcome overridden.
                    const bool same = (r == s);
                    const bool cross = !same && ((r < ka) != (s < ka));
                    const uint32_t s_eff = (same || cross) ? r : s;
                    const uint32_t s_loc = s_eff - own_base;
                    const int ideg = (int)deg;
                    const uint32_t a_st = mq_at(s_loc, lane);
                    const int32_t m_st_raw = mq[a_st];
                    const int32_t m_st = lane < k_oth ? m_st_raw : 0;
                    const int eta_s = (int)eta_any_rd(s_loc, deg);
                    const int m0r = readlane(mr_own, r_loc);
                    const int m0s = readlane(mr_own, s_loc);
                    const int n_r_r = readlane(nr_own, r_loc), n_r_s = readlane(nr_own, s_loc);
                    // lanes 0..7 (pattern repeated): the scalar lgamma terms (:164-177); lanes 0..3: log_q arguments
                    const bool odd = lane & 1u;
                    const int mm = odd ? m0s : m0r;
                    const int dd = (lane & 2u) ? (odd ? ideg : -ideg) : 0;  // -> m1r, m1s in lanes 2,3 (mod 4)
                    const int ee = odd ? eta_s : eta_r;
                    const int eoff = (lane & 7u) < 6 ? 1 : (odd ? 2 : 0);  // eta_r+1, eta_s+1, eta_r, eta_s+2
                    const uint32_t tail_idx = (lane & 4u) ? (uint32_t)(ee + eoff) : (uint32_t)(mm + dd + 1);
                    const int qn = mm + dd;
                    const int qk = (odd ? n_r_s : n_r_r) + ((lane & 2u) ? (odd ? 1 : -1) : 0);
                    const double tail_lg = tab_at(tab.lg, tail_idx);
                    const double logn = tab_at(tab.logtab, (uint32_t)qn);  // log(n) of the log_q arguments
                    const uint32_t kk = (uint32_t)k;
                    const double L1 = tab_at(tab.lg, (uint32_t)(m_rt + 1));
                    const double L2 = tab_at(tab.lg, (uint32_t)(m_st + 1));
                    const double L3 = tab_at(tab.lg, (uint32_t)(m_rt + 1) - kk);
                    const double L4 = tab_at(tab.lg, (uint32_t)(m_st + 1) + kk);
                    // Hastings sums.  k == 0 lanes give exact zeros (0 * x = +0, identical table entries cancel).
                    const double a0 = k * (m_st + eps) * inv_oth_of_lane();
                    const double a1 = k * (m_rt - k + eps) * inv_oth_of_lane();
                    double accu0, accu1;
                    if (k_oth <= 32u) {
                        butterfly_pair32(a0, a1, accu0, accu1);
                    } else {
                        butterfly_pair64(a0, a1, accu0, accu1);
                    }
                    if (deg == 0) accu0 = accu1 = 1.;
                    const double lq = log_q<true>(tab, qn, qk, logn);
                    double d = (L1 + L2) - (L3 + L4);
                    // fold the scalar terms into leaves 0..7 / 0..3 with their signs
                    // (sign_tail: -lg(m0r+1) -lg(m0s+1) +lg(m1r+1) +lg(m1s+1) +lg(eta_r+1) +lg(eta_s+1) -lg(eta_r)
                    // -lg(eta_s+2), zero from lane 8 on; sign_q: -,-,+,+, zero from lane 4 on.  x * (+-1) is exact and
                    // idle lanes add a zero: the table values are finite)
                    d = d + tail_lg * sign_tail1_of();
                    d = d + lq * sign_q1_of();
                    double dS = k_oth <= 32u ? butterfly_sum_low32(d) : butterfly_sum(d);
                    // accept (:47-61): T == 0: dS < 0;  else u < exp(-dS/T) accu1/accu0
                    bool accept;
                    if (T == 0.)
                        accept = dS < 0;
                    else
                        accept = less_than_scaled_exp(u_acc * accu0, accu1, -dS * (1.0 / T));
                    if (same) {
                        accept = (T != 0.);
                        dS = 0.;
                    }
                    if (cross) accept = false;
                    // ---- apply_mcmc_moves, blockmodel.cc:461-503 ----
                    // (every lane holds the same outcome; taking it through a ballot tells the compiler so, and the
                    // step counters stay scalar)
                    const bool ok = __builtin_amdgcn_ballot_w64(accept && (n_r_r - 1 != 0)) != 0;  // :467-471 veto
                    if (ok && !same) {
                        wfence();
                        if (lane == 0) {
                            eta_any_wr(r_loc, deg, (uint32_t)(eta_r - 1));
                            eta_any_wr(s_loc, deg, (uint32_t)(eta_s + 1));
                            new_lab[q] = (uint8_t)s;
                        }
                        mr_own += (lb == s_loc ? ideg : 0) - (lb == r_loc ? ideg : 0);
                        nr_own += (lb == s_loc ? 1 : 0) - (lb == r_loc ? 1 : 0);
                        if (lane < k_oth) {  // k == 0: rewrites the same values
                            mq[a_rt] = m_rt - k;
                            mq[a_st] = m_st + k;
                        }
                        if (lane == 0) cum_l0 += dS;  // :500
                        wfence();
                        if (track_min != 0u) new_minimum(q);
                    } else if (ok && track_min != 0u) {
                        new_minimum(q);  // an accepted r == s step (see new_minimum)
                    }
                    book(ok, T);
                    // Nothing of this (rare) step may still be in flight when the hot loop goes on: the register allocator reuses
                    // the destination registers of its gathers for per-pass values of the hot step, and the compiler's wait-count
                    // bookkeeping, which merges every path into the loop head, would otherwise make EVERY hot pass wait for all
                    // outstanding vector memory operations -- i.e. for the label store of the pass before -- at its top.
                    __builtin_amdgcn_s_waitcnt(0);
                };

                // ---- the 64 steps.  Hot path: 1 <= deg <= 255, target drawn from column m[.][t], T > 0 ----
                const uint32_t last_own = k_own - 1;
                const double invT_const = 1.0 / T_const;
                // log_q of the hot steps: four (n, k) pairs per step, one per lane (mod 4).  Above the table, tier u = k / sqrt(n) > 18
                // (k^2 > 324 n: blocks of more than ~6 500 nodes at mean degree 20) is log_q_closed, 13 <= u <= 18 log_q_closed2,
                // 8 <= u < 13 log_q_mid (bisbm_device.hpp): the functions and the exact tier tests of log_q_approx<true>, on pinned
                // constants.  The tier is the LANE's -- a pass holds the arguments of up to eight steps, and a value that depended on
                // which tiers the other lanes are in would depend on the depth of the pass -- but only the tiers some lane needs are
                // evaluated (wave-uniform branches): all lanes far out, the common case on large blocks, is one straight line.
                // Anything else (arguments inside and outside the table in one pass, u < 8) goes through log_q<true> itself.
                // mid: whether the 8 <= u < 13 tier is inlined here (not where registers are scarcest).
                auto hot_log_q = [&](auto mid_tag, int qn, int qk, double logn) -> double {
                    constexpr bool MID = decltype(mid_tag)::value;
                    const int qk2 = qk < qn ? qk : qn;
                    const double nd = (double)qn, kd = (double)qk2;
                    const double k2 = kd * kd;
                    // (the tier tests as lane masks, combined on the scalar side: every lane of the wave is active in a pass)
                    const unsigned long long m_big = __builtin_amdgcn_ballot_w64(qn > kQNmax);
                    const unsigned long long m_direct = m_big & __builtin_amdgcn_ballot_w64(k2 > c_576 * nd);
                    if (__builtin_expect(m_direct == ~0ull, 1)) {
                        double sq, rr;
                        sqrt_rsqrt(nd, sq, rr);
                        return log_q_closed(kd, sq, rr, logn, lqc);
                    }
                    if (m_big == 0ull) return log_q_table(tab, qn, qk2);  // small graphs (int_part.hh:27-37)
                    const unsigned long long m_ge13 = m_big & __builtin_amdgcn_ballot_w64(k2 >= c_169 * nd);
                    const unsigned long long m_ge8 = m_big & __builtin_amdgcn_ballot_w64(k2 >= ldexp(nd, 6));
                    if ((MID ? m_ge8 : m_ge13) != ~0ull) return log_q<true>(tab, qn, qk, logn);
                    double sq, rr;
                    sqrt_rsqrt(nd, sq, rr);
                    // u >= 13: one evaluation of the closed form for both tiers -- with the 1e-7 exponential in the lanes of u > 18
                    // (log_q_closed's own bits) and, in the lanes of 13 <= u <= 24, the more accurate one plus the second-order
                    // term (log_q_closed2's bits); lanes of u < 13 are overwritten below
                    const double u = kd * rr;
                    const unsigned long long m_t2 = m_ge13 & ~m_direct;
                    double x = 0., d2 = 0.;
                    if (m_direct != 0ull) x = exp2_filter(u * lqc.nc0l2e);
                    if (m_t2 != 0ull) {
                        const double x0 = exp_neg7(0x1.48552f88091a8p+0 * u);
                        const bool t2 = __builtin_amdgcn_inverse_ballot_w64(m_t2);
                        x = t2 ? x0 : x;
                        d2 = t2 ? log_q_delta2(u, x0, sq, lqc) : 0.;
                    }
                    double lq = log_q_closed_x(kd, u, x, sq, logn, lqc) + d2;  // (+ 0.0 in the lanes of u > 18: the same bits)
                    if (MID && m_ge13 != ~0ull) {
                        const double lq_mid = log_q_mid(kd, sq, rr, logn, lqc);
                        lq = __builtin_amdgcn_inverse_ballot_w64(m_ge13) ? lq : lq_mid;
                    }
                    return lq;
                };
                // One step.  Early returns only (each is a jump to the loop latch); the rare cases leave through
                // step_general at the top, before anything is computed.  A rejected step changes nothing, not even a
                // register, unless the early-stop bookkeeping is on (T < 1).
                // (tm: whether the early-stop bookkeeping is on, as a type -- the loop exists once per value, so the
                // steps of a T >= 1 run carry no test of it)
                auto step = [&](auto tm, uint32_t q) {
                    constexpr bool TM = decltype(tm)::value;
                    const double T = CT ? T_const : readlane(T_l, q);  // :84
                    FSTAMP_STEP(0);
                    const uint32_t prop = (uint32_t)__builtin_amdgcn_readfirstlane((int)prop_of(q));
                    if (__builtin_expect((int32_t)prop < 0, 0)) {
                        step_general(q, T);
                        return;
                    }
                    const uint32_t pack = readlane(pack_l, q);
                    const uint32_t deg = pack & 255u, r_loc = (pack >> 8) & 63u, t_loc = (pack >> 16) & 63u;
                    // early LDS reads: k_v counter of lane's block, row r of m, column t of m over v's own type
                    const int k = (int)hist8_cur[q * kHistStride + lane];
                    const uint32_t a_rt = mq_at(r_loc, lane);
                    const int32_t m_rt_raw = mq[a_rt];
                    const int w_piv = mq[mq_at(lane, t_loc)];
                    const int n_r_r = readlane(nr_own, r_loc);
                    // the two lgamma gathers that only need row r go out now: their latency runs under the proposal
                    // Lanes with k == 0 (idle lanes included) contribute exact zeros to all three sums whatever m is, so
                    // they all read table entry 1: a gather costs per distinct address, not per lane
                    // (tools/probe/gather_lanes.hip), and a node touches only a few of the blocks.
                    const int32_t kmask = (0 - k) >> 31;
                    const int32_t m_rt = m_rt_raw & kmask;
                    const uint32_t kk = (uint32_t)k;
                    const double L1 = tab_at(tab.lg, (uint32_t)(m_rt + 1));
                    const double L3 = tab_at(tab.lg, (uint32_t)(m_rt + 1) - kk);
                    __asm__ volatile("" ::: "memory");  // keeps the loads above from being sunk below the r == s test
                    FSTAMP_STEP(1);
                    // integer inverse CDF (:627-628): first own block whose running total exceeds x.  Lanes past
                    // k_own hold garbage, but block k_own - 1 always qualifies (its total is m_r[t] > x).
                    const int scan = K32 ? wave_inclusive_scan32(w_piv) : wave_inclusive_scan(w_piv);
                    const unsigned long long hit = __builtin_amdgcn_ballot_w64((uint32_t)scan > prop);
                    uint32_t first_hit;  // (s_ff1 gives ~0u for an empty mask: the clamp below is the safety net)
                    __asm__("s_ff1_i32_b64 %0, %1" : "=s"(first_hit) : "s"(hit));
                    const uint32_t s_loc = min(first_hit, last_own);
                    FSTAMP_STEP(2);
                    if (s_loc == r_loc) {  // r == s: dS = 0, accepted as is unless T == 0; nothing changes (:109-112, :49-50)
                        const bool ok = n_r_r != 1 && (CT || T != 0.);
                        if (ok && lane == 0) acc_l0 += 1;
                        if constexpr (TM)
                            if (ok) new_minimum(q);
                        return;
                    }
                    const uint32_t r = own_base + r_loc, s = own_base + s_loc;
                    const int ideg = (int)deg;
                    const uint32_t a_st = mq_at(s_loc, lane);
                    const int32_t m_st_raw = mq[a_st];
                    const int eta_r = (int)eta_l[eta_at(r_loc, deg)];
                    const int eta_s = (int)eta_l[eta_at(s_loc, deg)];
                    const int32_t m_st = m_st_raw & kmask;
                    const int m0r = readlane(mr_own, r_loc);
                    const int m0s = readlane(mr_own, s_loc);
                    const int n_r_s = readlane(nr_own, s_loc);
                    // lanes 0..7 (pattern repeated): the scalar lgamma terms (:164-177); lanes 0..3: log_q arguments.
                    // Lane-pattern selects as bit arithmetic on per-lane constant masks (a ^ ((a ^ b) & mask)).
                    const int mm = m0r ^ ((m0r ^ m0s) & odd_mask_l);  // odd lanes: s, even lanes: r
                    const int ee = eta_r ^ ((eta_r ^ eta_s) & odd_mask_l);
                    const int qn = mm + __mul24(ideg, dsgn_l);        // m0r, m0s, m0r - deg, m0s + deg
                    const uint32_t tail_idx = (uint32_t)((qn ^ ((qn ^ ee) & eta_mask_l)) + toff_l);
                    const int qk = (n_r_r ^ ((n_r_r ^ n_r_s) & odd_mask_l)) + dq_l;
                    const double tail_lg = tab_at(tab.lg, tail_idx);
                    const double logn = tab_at(tab.logtab, (uint32_t)qn);  // log(n) of the log_q arguments
                    const double L2 = tab_at(tab.lg, (uint32_t)(m_st + 1));
                    const double L4 = tab_at(tab.lg, (uint32_t)(m_st + 1) + kk);
                    FSTAMP_STEP(3);
                    // Hastings sums: on-chip data only, they run while the table gathers are in flight.
                    // k == 0 lanes give exact zeros (0 * x = +0, identical table entries cancel).
                    const double a0 = k * (m_st + eps) * inv_oth_of_lane();
                    const double a1 = k * (m_rt - k + eps) * inv_oth_of_lane();
                    double accu0, accu1;
                    if (K32) {
                        butterfly_pair32(a0, a1, accu0, accu1);
                    } else {
                        butterfly_pair64(a0, a1, accu0, accu1);
                    }
                    FSTAMP_STEP(4);
                    // log_q of the four (n, k) pairs, one per lane (mod 4).  Above the table, tier u = k / sqrt(n) > 18
                    // (k^2 > 576 n: blocks of more than ~12 000 nodes at mean degree 20) is the closed form of
                    // bisbm_device.hpp on pinned constants and tier 8 <= u <= 24 the converged evaluation, the same
                    // functions and the same exact tier tests as log_q_approx<true>; anything else goes through
                    // log_q<true> itself.
                    const double lq = hot_log_q(std::true_type{}, qn, qk, logn);
                    FSTAMP_STEP(5);
                    double d = (L1 + L2) - (L3 + L4);
                    d = d + tail_lg * sign_tail1_of();  // scalar terms folded into leaves 0..7 / 0..3, see step_general
                    d = d + lq * sign_q1_of();
                    const double dS = K32 ? butterfly_sum_low32(d) : butterfly_sum(d);
                    FSTAMP_STEP(6);
                    if (!CT && T == 0.) {  // the greedy tail of a cooling schedule (:49-50): dS < 0 decides
                        if (!(dS < 0.) || n_r_r == 1) return;
                    } else {
                        // accept (:47-61): u accu0 < accu1 exp(-dS/T), decided on a 1e-7-accurate exponential unless
                        // the two sides are within 1e-5 of each other (then the exact one decides)
                        const double z = -dS * (CT ? invT_const : invT_of(q));
                        const double est = accu1 * exp2_filter(z * c_l2e);
                        const double lhs = u_acc_of(q) * accu0;
                        const unsigned long long b_lt = __builtin_amdgcn_ballot_w64(lhs < est);
                        const unsigned long long b_far = __builtin_amdgcn_ballot_w64(fabs(lhs - est) > c_tol * est);
                        FSTAMP_STEP(7);
                        if ((b_lt | ~b_far) == 0) return;  // clearly rejected
                        unsigned long long b_acc = b_lt;
                        if (__builtin_expect(b_far == 0, 0)) b_acc = __builtin_amdgcn_ballot_w64(lhs < accu1 * exp(z));
                        if (b_acc == 0 || n_r_r == 1) return;  // (:467-471: veto after the draw)
                    }
                    // ---- apply_mcmc_moves, blockmodel.cc:461-503 ----
                    wfence();
                    if (lane == 0) {
                        eta_l[eta_at(r_loc, deg)] = (uint32_t)(eta_r - 1);
                        eta_l[eta_at(s_loc, deg)] = (uint32_t)(eta_s + 1);
                        new_lab[q] = (uint8_t)s;
                        if constexpr (TM) cum_l0 += dS;  // :500 (without the bookkeeping: kSumFromEntropy)
                        acc_l0 += 1;
                    }
                    const int dl = (int)min(lb ^ r_loc, 1u) - (int)min(lb ^ s_loc, 1u);  // +1 on lane s_loc, -1 on r_loc
                    mr_own += __mul24(ideg, dl);
                    nr_own += dl;
                    if (lane < k_oth) {  // k == 0: rewrites the same values
                        mq[a_rt] = m_rt_raw - k;
                        mq[a_st] = m_st_raw + k;
                    }
                    wfence();
                    FSTAMP_STEP(8);
                    if constexpr (TM) new_minimum(q);
                };
                // ---- two steps per pass (K <= 32, constant T > 0, no early-stop bookkeeping) ----
                // The hot step uses lanes 0..31 (one lane per block).  Here lanes 32..63 evaluate step q + 1 in the same
                // instructions, against the same state, i.e. the state BEFORE step q.  That is step q + 1's true outcome
                // unless step q moves its node (r -> s, r != s) AND touches something step q + 1 read:
                //   rows r, s of m, m_r / n_r / eta of r, s       <=>  {r', s'} meets {r, s};
                //   column t' of m (it feeds the inverse CDF)      <=>  k_q[t'] != 0 -- and then only rows r and s of the
                //     column change, by -k and +k: the running sums move only for blocks in [min(r,s), max(r,s)), so a
                //     target s' outside (min, max) is still the first block whose sum exceeds x.
                // In those cases step q + 1 is evaluated again as the first step of the next pass; otherwise both
                // steps are committed (their writes touch different rows).  The chain is the serial chain, bit for
                // bit: the CPU checker steps one node at a time and the parity tests compare against it (measured on the
                // bench workload: the second step stands in ~80 % of the passes, 1.8 steps per pass).
                int half_mask_l = lane >= 32u ? -1 : 0;  // (all ones in the upper half: a per-lane select of two scalar words as bit operations)
                __asm__ volatile("" : "+v"(half_mask_l));
                uint32_t acc_chunk = 0;  // accepted steps of the chunk's pair passes (a scalar word; added to acc_l0 per chunk)
                // q: first step of the pass; pairable: 1 = lanes 32..63 evaluate step q + 1, 0 = nothing to pair with
                // (last step of the chunk, or a step that needs the general path next): both halves evaluate step q
                // (cooling schedules: the temperature is the lane's own step's; steps at T = 0 are decided by the sign of dS)
                const unsigned long long zeroT_mask = CT ? 0ull : __builtin_amdgcn_ballot_w64(T_l == 0.);
                auto step_pair = [&](auto tm, uint32_t q, uint32_t pairable) -> uint32_t {
                    constexpr bool TM = decltype(tm)::value;
                    const uint32_t qB = q + pairable;
                    const uint32_t qs = q + ((uint32_t)half_mask_l & pairable);  // (flags are 0 / 1 words and selections arithmetic: a bool
                                                                         // select of uniform values goes through the vector unit and back)
                    const int sel = (int)(qs << 2);
                    FSTAMP_STEP(0);
                    const uint32_t prop = prop_of(qs);
                    const double u_acc = u_acc_of(qs);
                    const uint32_t packA = readlane(pack_l, q), packB = readlane(pack_l, qB);
                    const uint32_t r_locA = (packA >> 8) & 63u, r_locB = (packB >> 8) & 63u;
                    const uint32_t degA = packA & 255u, degB = packB & 255u, t_locA = (packA >> 16) & 63u, t_locB = (packB >> 16) & 63u;
                    // the lane's own step: lower half step q, upper half step qB (from the two scalars: no LDS round trip
                    // in front of the first reads)
                    // (one select of the whole packed word per lane, then the fields out of it: 6 instructions where four selects
                    // of the scalar fields took 12)
                    const uint32_t pack_v = packA ^ ((packA ^ packB) & (uint32_t)half_mask_l);
                    const uint32_t deg = pack_v & 255u, r_loc = (pack_v >> 8) & 63u, t_loc = (pack_v >> 16) & 63u;
                    const int k = (int)hist8_cur[qs * kHistStride + lb];
                    const uint32_t a_rt = mq_at(r_loc, lb);
                    const int32_t m_rt_raw = mq[a_rt];
                    const int w_piv = mq[mq_at(lb, t_loc)];
                    const uint32_t liveA = sflag((uint32_t)readlane(nr_own, r_locA) ^ 1u);  // n_r != 1 (:467-471: a block is never emptied)
                    const uint32_t liveB = smin((uint32_t)readlane(nr_own, r_locB) ^ 1u, pairable);
                    const int32_t kmask = (0 - k) >> 31;
                    const int32_t m_rt = m_rt_raw & kmask;
                    const uint32_t kk = (uint32_t)k;
                    const int ideg = (int)deg;
                    // Everything that depends on the target s, as a function of the target's block index per half.  kPredictTarget:
                    // evaluated on the feeder's PREDICTION (bits 24.. of the packed word), i.e. the reads of row s, eta and m_r / n_r
                    // go out with the first reads and all six table gathers right behind them, while the scan that finds the
                    // target on the state of this moment runs in their shadow; the pass goes ahead only where the two agree (below).
                    // Otherwise (and in every other kind of pass): evaluated on the scan's result, behind it.
                    uint32_t s_loc, idx_l, a_st, e_idx, tail_idx;
                    int32_t m_st_raw, m_st;
                    int ee, qn, qk;
                    double logn, tail_lg, L1, L2, L3, L4;
                    auto target_lds = [&](uint32_t sA, uint32_t sB, bool predicted) {  // row s of m, eta, m_r / n_r of r and s: LDS and cross-lane reads
                        if (predicted)
                            s_loc = pack_v >> 24;
                        else
                            s_loc = sA + ((sB - sA) & (uint32_t)half_mask_l);
                        idx_l = r_loc ^ ((r_loc ^ s_loc) & (uint32_t)odd_mask_l);  // odd lanes: s, even lanes: r
                        a_st = mq_at(s_loc, lb);
                        m_st_raw = mq[a_st];
                        e_idx = eta_at(idx_l, deg);
                        ee = (int)eta_l[e_idx];
                        const int mm = __builtin_amdgcn_ds_bpermute((int)(idx_l << 2), mr_own);
                        const int nn = __builtin_amdgcn_ds_bpermute((int)(idx_l << 2), nr_own);
                        m_st = m_st_raw & kmask;
                        qn = mm + __mul24(ideg, dsgn_l);        // m0r, m0s, m0r - deg, m0s + deg
                        tail_idx = (uint32_t)((qn ^ ((qn ^ ee) & eta_mask_l)) + toff_l);
                        qk = nn + dq_l;
                    };
                    auto target_gathers = [&]() {  // the four table gathers that depend on the target
                        logn = tab_at(tab.logtab, (uint32_t)qn);
                        tail_lg = tab_at(tab.lg, tail_idx);
                        L2 = tab_at(tab.lg, (uint32_t)(m_st + 1));
                        L4 = tab_at(tab.lg, (uint32_t)(m_st + 1) + kk);
                    };
                    auto row_r_gathers = [&]() {  // the two that only need row r
                        L1 = tab_at(tab.lg, (uint32_t)(m_rt + 1));
                        L3 = tab_at(tab.lg, (uint32_t)(m_rt + 1) - kk);
                    };
                    const uint32_t s_prdA = packA >> 24, s_prdB = packB >> 24;
                    if constexpr (kPredictTarget) {
                        target_lds(s_prdA, s_prdB, true);
                        target_gathers();
                        row_r_gathers();
                        __asm__ volatile("" ::: "memory");
                    }
                    FSTAMP_STEP(1);
                    // inverse CDF per half (:627-628): the scan does not cross lane 31 -> 32
                    const int scan = wave_inclusive_scan32(w_piv);
                    const unsigned long long hit = __builtin_amdgcn_ballot_w64((uint32_t)scan > prop);
                    if constexpr (!kPredictTarget) {
                        // the two lgamma gathers that only need row r go out behind the vote (round 4: they are used on the tail and have
                        // slack; in front of the scan their address arithmetic delayed the chain scan -> s -> the s-dependent gathers;
                        // +0.5 %, profiles/r04_ab_pass_scheduling.txt)
                        __asm__ volatile("" ::: "memory");
                        row_r_gathers();
                        __asm__ volatile("" ::: "memory");
                    }
                    uint32_t fhA, fhB;
                    __asm__("s_ff1_i32_b32 %0, %1" : "=s"(fhA) : "s"((uint32_t)hit));
                    __asm__("s_ff1_i32_b32 %0, %1" : "=s"(fhB) : "s"((uint32_t)(hit >> 32)));
                    const uint32_t s_locA = min(fhA, last_own), s_locB = min(fhB, last_own);
                    const uint32_t selfA = 1u - sflag(s_locA ^ r_locA), selfB = 1u - sflag(s_locB ^ r_locB);
                    // T = 0 (the greedy tail of a cooling schedule): r == s is not accepted (dS = 0 is not < 0, :49-50)
                    const uint32_t warmA = CT ? 1u : ((uint32_t)(zeroT_mask >> q) & 1u) ^ 1u;
                    const uint32_t warmB = CT ? 1u : ((uint32_t)(zeroT_mask >> qB) & 1u) ^ 1u;
                    FSTAMP_STEP(2);
                    if ((selfA & selfB) != 0u) {  // both r == s: nothing changes (:109-112)
                        acc_chunk += (liveA & warmA) + (liveB & warmB);
                        if constexpr (TM) {
                            if ((liveA & warmA) != 0u) new_minimum(q);
                            if ((liveB & warmB) != 0u) new_minimum(qB);
                        }
                        return 1u + pairable;
                    }
                    if constexpr (kPredictTarget) {
                        // a target that is not the predicted one: the reads that depend on it, again (nothing has been written).  (The
                        // test on the scalar side as written: as a C expression of two inequalities it became two compares, two
                        // s_cselect_b64, two s_and_b64 and a branch on vcc -- 1 % of the pass)
                        if (__builtin_expect(sflag((s_locA ^ s_prdA) | (s_locB ^ s_prdB)) != 0u, 0)) {
                            target_lds(s_locA, s_locB, false);
                            target_gathers();
                        }
                    } else {
                        target_lds(s_locA, s_locB, false);
                        target_gathers();
                    }
                    FSTAMP_STEP(3);
                    // (the verdict logic's inputs are worked out here, while the table gathers are in flight -- after the gathers have
                    // been issued, not in front of them -- and pinned so that the compiler does not sink them behind the verdicts)
                    // Would step q, if it moves its node, touch what step q + 1 read?  (block sets as bit masks)
                    const uint32_t setA = (1u << r_locA) | (1u << s_locA), setB = (1u << r_locB) | (1u << s_locB);
                    const uint32_t lo = min(r_locA, s_locA), hi = max(r_locA, s_locA);
                    const uint32_t between = ((1u << hi) - 1u) & ~((2u << lo) - 1u);  // blocks strictly between r and s
                    const uint32_t kAtB = readlane(kk, t_locB);                        // (lanes 0..31 hold k_q[.])
                    const uint32_t clash = sflag(setA & setB) | (((between >> s_locB) & 1u) & sflag(kAtB));
                    // what the verdicts will be combined with, in one word (the scalar file is full): bit 0 step q can
                    // move, bit 1 step q is an accepted r == s, bits 2, 3 the same for step q + 1, bit 4 the clash
                    const uint32_t flags = (liveA & (selfA ^ 1u)) | ((liveA & selfA & warmA) << 1) | ((liveB & (selfB ^ 1u)) << 2) |
                                           ((liveB & selfB & warmB) << 3) | (clash << 4);

                    uint32_t flags_pin = flags;
                    __asm__ volatile("" : "+s"(flags_pin));
                    (void)flags_pin;
                    // (+1 on lane s, -1 on lane r, as bit `lane` of the two one-bit block masks: v_bfe_u32 twice and a subtraction --
                    // against min(lb ^ r, 1) - min(lb ^ s, 1), which the compiler lowers through v_cmp / v_cndmask / v_subbrev, +1.3 % on the
                    // bench line, same box, tools/ab.sh)
                    int dlA = (int)__builtin_amdgcn_ubfe(1u << s_locA, lb, 1u) - (int)__builtin_amdgcn_ubfe(1u << r_locA, lb, 1u);
                    int dlB = (int)__builtin_amdgcn_ubfe(1u << s_locB, lb, 1u) - (int)__builtin_amdgcn_ubfe(1u << r_locB, lb, 1u);
                    int dmA = __mul24((int)degA, dlA), dmB = __mul24((int)degB, dlB);
                    __asm__ volatile("" : "+v"(dlA), "+v"(dlB), "+v"(dmA), "+v"(dmB));
                    // (what apply_mcmc_moves will write, worked out while the table gathers are in flight and pinned there)
                    int wr_rt = m_rt_raw - k, wr_st = m_st_raw + k, wr_eta = ee + ((int)(lb & 1u) * 2 - 1);
                    __asm__ volatile("" : "+v"(wr_rt), "+v"(wr_st), "+v"(wr_eta));
                    const double a0 = k * (m_st + eps) * inv_blk;
                    const double a1 = k * (m_rt - k + eps) * inv_blk;
                    double accu0, accu1;  // lanes 16..31: step q, lanes 48..63: step q + 1
                    butterfly_accu_rows32(a0, a1, accu0, accu1);
                    FSTAMP_STEP(4);
#ifdef BISBM_PROBE_NOPS  // diagnostic: idle issue slots in the shadow of the table gathers (how much of their latency is still exposed?)
#pragma unroll
                    for (int i_ = 0; i_ < BISBM_PROBE_NOPS; ++i_) __asm__ volatile("s_nop 0");
#endif
                    const double lq = hot_log_q(std::true_type{}, qn, qk, logn);
                    FSTAMP_STEP(5);
                    double d = (L1 + L2) - (L3 + L4);
                    d = d + tail_lg * sign_tail;
                    d = d + lq * sign_q;
                    const double dS = butterfly_rows32(d);
#ifdef BISBM_PROBE_NOPS_TAIL  // diagnostic: the same number of idle slots on the tail, where nothing is in flight
#pragma unroll
                    for (int i_ = 0; i_ < BISBM_PROBE_NOPS_TAIL; ++i_) __asm__ volatile("s_nop 0");
#endif
                    FSTAMP_STEP(6);
                    // accept (:47-61) in the lanes that hold the sums; bit 31 is step q's verdict, bit 63 step q + 1's
                    double invT = invT_const;
                    if (!CT) invT = invT_of(qs);
                    const double z = -dS * invT;
                    const double est = accu1 * exp2_filter(z * c_l2e);
                    const double lhs = u_acc * accu0;
                    unsigned long long b_acc = __builtin_amdgcn_ballot_w64(lhs < est);
                    unsigned long long b_far = __builtin_amdgcn_ballot_w64(fabs(lhs - est) > c_tol * est);
                    if (!CT) {  // steps at T = 0: dS < 0 decides (:49-50); nothing to be close to
                        const unsigned long long cold = ((unsigned long long)(0u - (warmB ^ 1u)) << 32) | (0u - (warmA ^ 1u));
                        const unsigned long long b_neg = __builtin_amdgcn_ballot_w64(dS < 0.);
                        b_acc = (b_acc & ~cold) | (b_neg & cold);
                        b_far |= cold;
                    }
                    if (__builtin_expect((((uint32_t)b_far & (uint32_t)(b_far >> 32)) >> 31) == 0u, 0)) {  // a verdict too close to call
                        const unsigned long long exact = __builtin_amdgcn_ballot_w64(lhs < accu1 * exp(z));
                        b_acc = (b_acc & b_far) | (exact & ~b_far);
                    }
                    FSTAMP_STEP(7);
                    const uint32_t yesA = (uint32_t)(b_acc >> 31) & 1u, yesB = (uint32_t)(b_acc >> 63) & 1u;
                    const uint32_t chA = flags & yesA;                                  // step q moves its node
                    const uint32_t okA = chA | ((flags >> 1) & 1u);                     // ... counts as accepted
                    const uint32_t stands = pairable & ((chA & (flags >> 4)) ^ 1u);     // step q + 1's evaluation stands
                    const uint32_t chB = stands & (flags >> 2) & yesB;
                    const uint32_t okB = chB | (stands & (flags >> 3) & 1u);
                    acc_chunk += okA + okB;
                    if ((TM ? (okA | okB) : (chA | chB)) != 0u) {  // (with the early-stop bookkeeping on: every accepted step is looked at)
                        // ---- apply_mcmc_moves, blockmodel.cc:461-503, for the step(s) that move: their rows differ ----
                        const uint32_t mA = 0u - chA, mB = 0u - chB;  // all ones / zero
                        const unsigned long long movers = ((unsigned long long)mB << 32) | mA;
                        wfence();
                        if (__builtin_amdgcn_inverse_ballot_w64(movers & lanes_koth)) {  // k == 0: rewrites the same values
                            mq[a_rt] = wr_rt;
                            mq[a_st] = wr_st;
                        }
                        if (__builtin_amdgcn_inverse_ballot_w64(movers & 0x0000003000000030ull))  // lanes 4, 5: eta_r - 1, eta_s + 1
                            eta_l[e_idx] = (uint32_t)wr_eta;
                        if (__builtin_amdgcn_inverse_ballot_w64(movers & 0x0000000100000001ull)) new_lab[qs] = (uint8_t)(own_base + s_loc);
                        mr_own += (dmA & (int)mA) + (dmB & (int)mB);
                        nr_own += (dlA & (int)mA) + (dlB & (int)mB);
                        if constexpr (TM) {  // :500, in step order; without the early-stop bookkeeping nobody looks at the running sum
                                             // during the launch: the host takes it from the description length (kSumFromEntropy)
                            const int dS_A_lo = __builtin_amdgcn_readlane(__double2loint(dS), 31), dS_A_hi = __builtin_amdgcn_readlane(__double2hiint(dS), 31);
                            const int dS_B_lo = __builtin_amdgcn_readlane(__double2loint(dS), 63), dS_B_hi = __builtin_amdgcn_readlane(__double2hiint(dS), 63);
                            // (a step that does not move adds +0.0: x + 0.0 is x, the running sum is never -0.0)
                            cum_l0 += __hiloint2double(dS_A_hi & (int)mA, dS_A_lo & (int)mA);
                            if (okA) new_minimum(q);  // (:86-90, after step q's move and before step q + 1's)
                            cum_l0 += __hiloint2double(dS_B_hi & (int)mB, dS_B_lo & (int)mB);
                            if (okB) new_minimum(qB);
                        }
                        wfence();
                    }
                    FSTAMP_STEP(8);
                    return 1u + stands;
                };

                // ---- two steps per pass with more than 32 blocks of a type (the K > 32 variant) ----
                // step_pair with TWO leaves per lane: lane l of a half (lower half: step q, upper half: step q + 1) holds the
                // opposite-type blocks l and l + 32 -- their k, their entries of rows r and s, their table gathers, their leaves of
                // the three sums, added first (level 32 of the summation tree, bisbm_device.hpp) -- and, for the inverse CDF,
                // entries l and l + 32 of column t.  m_r / n_r sit one block per lane across the whole wave and are read by lane
                // index.  The stand rule, the verdict logic and the writes are step_pair's (block sets as 64-bit masks).
                                auto step_pair64 = [&](auto tm, uint32_t q, uint32_t pairable) -> uint32_t {
                    constexpr bool TM = decltype(tm)::value;
                    const uint32_t qB = q + pairable;
                    const uint32_t qs = q + ((uint32_t)half_mask_l & pairable);
                    const int sel = (int)(qs << 2);
                    const uint32_t prop = prop_of(qs);
                    const double u_acc = u_acc_of(qs);
                    const uint32_t packA = readlane(pack_l, q), packB = readlane(pack_l, qB);
                    const uint32_t r_locA = (packA >> 8) & 63u, r_locB = (packB >> 8) & 63u;
                    const uint32_t degA = packA & 255u, degB = packB & 255u, t_locA = (packA >> 16) & 63u, t_locB = (packB >> 16) & 63u;
                    const uint32_t pack_v = packA ^ ((packA ^ packB) & (uint32_t)half_mask_l);  // (see step_pair)
                    const uint32_t deg = pack_v & 255u, r_loc = (pack_v >> 8) & 63u, t_loc = (pack_v >> 16) & 63u;
                    const int k0 = (int)hist8_cur[qs * kHistStride + lh], k1 = (int)hist8_cur[qs * kHistStride + lh + 32u];
                    const uint32_t a_rt0 = mq_at(r_loc, lh), a_rt1 = mq_at(r_loc, lh + 32u);
                    const int32_t m_rt_raw0 = mq[a_rt0], m_rt_raw1 = mq[a_rt1];
                    const int w0 = mq[mq_at(lh, t_loc)], w1 = mq[mq_at(lh + 32u, t_loc)];
                    const uint32_t liveA = sflag((uint32_t)readlane(nr_own, r_locA) ^ 1u);
                    const uint32_t liveB = smin((uint32_t)readlane(nr_own, r_locB) ^ 1u, pairable);
                    const int32_t kmask0 = (0 - k0) >> 31, kmask1 = (0 - k1) >> 31;
                    const int32_t m_rt0 = m_rt_raw0 & kmask0, m_rt1 = m_rt_raw1 & kmask1;
                    const uint32_t kk0 = (uint32_t)k0, kk1 = (uint32_t)k1;
                    const int ideg = (int)deg;
                    // what depends on the target s (see step_pair): on the feeder's prediction, in front of the scan (kPredictTarget), or on
                    // the scan's result, behind it
                    uint32_t s_loc, idx_l, a_st0, a_st1, e_idx, tail_idx;
                    int32_t m_st_raw0, m_st_raw1, m_st0, m_st1;
                    int ee, qn, qk;
                    double tail_lg, logn, L1_0, L1_1, L2_0, L2_1, L3_0, L3_1, L4_0, L4_1;
                    auto target_lds = [&](uint32_t sA, uint32_t sB, bool predicted) {
                        if (predicted)
                            s_loc = pack_v >> 24;
                        else
                            s_loc = sA + ((sB - sA) & (uint32_t)half_mask_l);
                        idx_l = r_loc ^ ((r_loc ^ s_loc) & (uint32_t)odd_mask_l);  // odd lanes: s, even lanes: r
                        a_st0 = mq_at(s_loc, lh), a_st1 = mq_at(s_loc, lh + 32u);
                        m_st_raw0 = mq[a_st0], m_st_raw1 = mq[a_st1];
                        e_idx = eta_at(idx_l, deg);
                        ee = (int)eta_l[e_idx];
                        const int mm = __builtin_amdgcn_ds_bpermute((int)(idx_l << 2), mr_own);
                        const int nn = __builtin_amdgcn_ds_bpermute((int)(idx_l << 2), nr_own);
                        m_st0 = m_st_raw0 & kmask0, m_st1 = m_st_raw1 & kmask1;
                        qn = mm + __mul24(ideg, dsgn_l);        // m0r, m0s, m0r - deg, m0s + deg
                        tail_idx = (uint32_t)((qn ^ ((qn ^ ee) & eta_mask_l)) + toff_l);
                        qk = nn + dq_l;
                    };
                    auto target_gathers = [&]() {
                        tail_lg = tab_at(tab.lg, tail_idx);
                        logn = tab_at(tab.logtab, (uint32_t)qn);
                        L2_0 = tab_at(tab.lg, (uint32_t)(m_st0 + 1)), L4_0 = tab_at(tab.lg, (uint32_t)(m_st0 + 1) + kk0);
                        L2_1 = tab_at(tab.lg, (uint32_t)(m_st1 + 1)), L4_1 = tab_at(tab.lg, (uint32_t)(m_st1 + 1) + kk1);
                    };
                    auto row_r_gathers = [&]() {
                        L1_0 = tab_at(tab.lg, (uint32_t)(m_rt0 + 1)), L3_0 = tab_at(tab.lg, (uint32_t)(m_rt0 + 1) - kk0);
                        L1_1 = tab_at(tab.lg, (uint32_t)(m_rt1 + 1)), L3_1 = tab_at(tab.lg, (uint32_t)(m_rt1 + 1) - kk1);
                    };
                    const uint32_t s_prdA = packA >> 24, s_prdB = packB >> 24;
                    if constexpr (kPredictTarget) {
                        target_lds(s_prdA, s_prdB, true);
                        target_gathers();
                        row_r_gathers();
                        __asm__ volatile("" ::: "memory");
                    }
                    // inverse CDF per half over 64 own blocks (:627-628): the scan of blocks 0..31, its total, the scan of blocks
                    // 32..63 on top.  (Lanes past k_own hold garbage; block k_own - 1 always qualifies, and with k_own <= 32 the first
                    // hit lies in the lower scan, whose prefix sums up to it are clean.)
                    const int scan0 = wave_inclusive_scan32(w0);
                    const int tot = __builtin_amdgcn_ds_bpermute((int)(((lane & 32u) | 31u) << 2), scan0);
                    const int scan1 = wave_inclusive_scan32(w1) + tot;
                    const unsigned long long hit0 = __builtin_amdgcn_ballot_w64((uint32_t)scan0 > prop);
                    const unsigned long long hit1 = __builtin_amdgcn_ballot_w64((uint32_t)scan1 > prop);
                    if constexpr (!kPredictTarget) {
                        // (the four row-r gathers go out behind the votes, as in step_pair: +0.45 % on the config-5 shape, tools/ab_config5.sh)
                        __asm__ volatile("" ::: "memory");
                        row_r_gathers();
                        __asm__ volatile("" ::: "memory");
                    }
                    const unsigned long long hitsA = (hit0 & 0xffffffffull) | (hit1 << 32), hitsB = (hit0 >> 32) | (hit1 & 0xffffffff00000000ull);
                    uint32_t fhA, fhB;
                    __asm__("s_ff1_i32_b64 %0, %1" : "=s"(fhA) : "s"(hitsA));
                    __asm__("s_ff1_i32_b64 %0, %1" : "=s"(fhB) : "s"(hitsB));
                    const uint32_t s_locA = min(fhA, last_own), s_locB = min(fhB, last_own);
                    const uint32_t selfA = 1u - sflag(s_locA ^ r_locA), selfB = 1u - sflag(s_locB ^ r_locB);
                    const uint32_t warmA = CT ? 1u : ((uint32_t)(zeroT_mask >> q) & 1u) ^ 1u;
                    const uint32_t warmB = CT ? 1u : ((uint32_t)(zeroT_mask >> qB) & 1u) ^ 1u;
                    if ((selfA & selfB) != 0u) {  // both r == s: nothing changes (:109-112)
                        acc_chunk += (liveA & warmA) + (liveB & warmB);
                        if constexpr (TM) {
                            if ((liveA & warmA) != 0u) new_minimum(q);
                            if ((liveB & warmB) != 0u) new_minimum(qB);
                        }
                        return 1u + pairable;
                    }
                    uint32_t pair_ok = pairable;  // step q + 1 is there and has been evaluated on its own target
                    if constexpr (kPredictTarget) {
                        // step q's target is not the predicted one: nothing has been written -- the caller sends step q down the general
                        // path.  Step q + 1's is not: its evaluation does not stand, whatever step q does (it opens the next pass, and
                        // goes the same way).  (step_pair issues the reads again in place instead; here the second copy of the reads
                        // costs registers the K > 32 variants do not have)
                        if (__builtin_expect(s_locA != s_prdA, 0)) return 0u;
                        pair_ok = pairable & (sflag(s_locB ^ s_prdB) ^ 1u);
                    } else {
                        target_lds(s_locA, s_locB, false);
                        target_gathers();
                    }
                    // (worked out while the gathers are in flight and pinned there, see step_pair: the verdict logic's inputs ...)
                    // would step q, if it moves its node, touch what step q + 1 read?  (step_pair's rule on 64-bit block sets)
                    const unsigned long long setA = (1ull << r_locA) | (1ull << s_locA), setB = (1ull << r_locB) | (1ull << s_locB);
                    const uint32_t lo = min(r_locA, s_locA), hi = max(r_locA, s_locA);
                    const unsigned long long between = ((1ull << hi) - 1ull) & ~((2ull << lo) - 1ull);
                    // (lanes 0..31 hold step q's k: block t_locB is the lower or the upper leaf of lane t_locB & 31; selections
                    // stay arithmetic on scalar words, see sflag)
                    const uint32_t kA0 = readlane(kk0, t_locB & 31u), kA1 = readlane(kk1, t_locB & 31u);
                    const uint32_t kAtB = kA0 + (t_locB >> 5) * (kA1 - kA0);
                    const unsigned long long common = setA & setB;
                    const uint32_t clash = sflag((uint32_t)common | (uint32_t)(common >> 32)) | ((uint32_t)((between >> s_locB) & 1ull) & sflag(kAtB));
                    const uint32_t flags = (liveA & (selfA ^ 1u)) | ((liveA & selfA & warmA) << 1) | ((liveB & (selfB ^ 1u)) << 2) |
                                           ((liveB & selfB & warmB) << 3) | (clash << 4);
                    uint32_t flags_pin = flags;
                    __asm__ volatile("" : "+s"(flags_pin));
                    (void)flags_pin;
                    // (... and what apply_mcmc_moves will write, to be masked by the movers)
                    int dlA = (int)min(lane ^ r_locA, 1u) - (int)min(lane ^ s_locA, 1u);  // +1 on lane s, -1 on lane r (lane <-> block)
                    int dlB = (int)min(lane ^ r_locB, 1u) - (int)min(lane ^ s_locB, 1u);
                    int dmA = __mul24((int)degA, dlA), dmB = __mul24((int)degB, dlB);
                    __asm__ volatile("" : "+v"(dlA), "+v"(dlB), "+v"(dmA), "+v"(dmB));
                    // the lane's two leaves of each sum, added first (level 32)
                    const double a0 = k0 * (m_st0 + eps) * inv_lo + k1 * (m_st1 + eps) * inv_hi;
                    const double a1 = k0 * (m_rt0 - k0 + eps) * inv_lo + k1 * (m_rt1 - k1 + eps) * inv_hi;
                    double accu0, accu1;  // lanes 16..31: step q, lanes 48..63: step q + 1
                    butterfly_accu_rows32(a0, a1, accu0, accu1);
                    const double lq = hot_log_q(std::false_type{}, qn, qk, logn);
                    double d = (L1_0 + L2_0) - (L3_0 + L4_0);
                    d = d + tail_lg * sign_tail;  // the scalar terms sit in leaves 0..7 / 0..3: the lane's lower leaf
                    d = d + lq * sign_q;
                    d = d + ((L1_1 + L2_1) - (L3_1 + L4_1));
                    const double dS = butterfly_rows32(d);
                    double invT = invT_const;
                    if (!CT) invT = invT_of(qs);
                    const double z = -dS * invT;
                    const double est = accu1 * exp2_filter(z * c_l2e);
                    const double lhs = u_acc * accu0;
                    unsigned long long b_acc = __builtin_amdgcn_ballot_w64(lhs < est);
                    unsigned long long b_far = __builtin_amdgcn_ballot_w64(fabs(lhs - est) > c_tol * est);
                    if (!CT) {  // steps at T = 0: dS < 0 decides (:49-50)
                        const unsigned long long cold = ((unsigned long long)(0u - (warmB ^ 1u)) << 32) | (0u - (warmA ^ 1u));
                        const unsigned long long b_neg = __builtin_amdgcn_ballot_w64(dS < 0.);
                        b_acc = (b_acc & ~cold) | (b_neg & cold);
                        b_far |= cold;
                    }
                    if (__builtin_expect((((uint32_t)b_far & (uint32_t)(b_far >> 32)) >> 31) == 0u, 0)) {  // a verdict too close to call
                        const unsigned long long exact = __builtin_amdgcn_ballot_w64(lhs < accu1 * exp(z));
                        b_acc = (b_acc & b_far) | (exact & ~b_far);
                    }
                    const uint32_t yesA = (uint32_t)(b_acc >> 31) & 1u, yesB = (uint32_t)(b_acc >> 63) & 1u;
                    const uint32_t chA = flags & yesA;
                    const uint32_t okA = chA | ((flags >> 1) & 1u);
                    const uint32_t stands = pair_ok & ((chA & (flags >> 4)) ^ 1u);
                    const uint32_t chB = stands & (flags >> 2) & yesB;
                    const uint32_t okB = chB | (stands & (flags >> 3) & 1u);
                    acc_chunk += okA + okB;
                    if ((TM ? (okA | okB) : (chA | chB)) != 0u) {
                        const uint32_t mA = 0u - chA, mB = 0u - chB;
                        const unsigned long long movers = ((unsigned long long)mB << 32) | mA;
                        wfence();
                        if (__builtin_amdgcn_inverse_ballot_w64(movers & lanes_koth64_lo)) {  // k == 0: rewrites the same values
                            mq[a_rt0] = m_rt_raw0 - k0;
                            mq[a_st0] = m_st_raw0 + k0;
                        }
                        if (__builtin_amdgcn_inverse_ballot_w64(movers & lanes_koth64_hi)) {
                            mq[a_rt1] = m_rt_raw1 - k1;
                            mq[a_st1] = m_st_raw1 + k1;
                        }
                        if (__builtin_amdgcn_inverse_ballot_w64(movers & 0x0000003000000030ull))  // lanes 4, 5: eta_r - 1, eta_s + 1
                            eta_l[e_idx] = (uint32_t)(ee + ((int)(lh & 1u) * 2 - 1));
                        if (__builtin_amdgcn_inverse_ballot_w64(movers & 0x0000000100000001ull)) new_lab[qs] = (uint8_t)(own_base + s_loc);
                        mr_own += (dmA & (int)mA) + (dmB & (int)mB);
                        nr_own += (dlA & (int)mA) + (dlB & (int)mB);
                        if constexpr (TM) {  // (see step_pair)
                            const int dS_A_lo = __builtin_amdgcn_readlane(__double2loint(dS), 31), dS_A_hi = __builtin_amdgcn_readlane(__double2hiint(dS), 31);
                            const int dS_B_lo = __builtin_amdgcn_readlane(__double2loint(dS), 63), dS_B_hi = __builtin_amdgcn_readlane(__double2hiint(dS), 63);
                            cum_l0 += __hiloint2double(dS_A_hi & (int)mA, dS_A_lo & (int)mA);
                            if (okA) new_minimum(q);
                            cum_l0 += __hiloint2double(dS_B_hi & (int)mB, dS_B_lo & (int)mB);
                            if (okB) new_minimum(qB);
                        }
                        wfence();
                    }
                    return 1u + stands;
                };

                // ---- four steps per pass (both block counts <= 16) ----
                // The same idea one level further: row g of the wave (lanes 16 g .. 16 g + 15, one lane per block) evaluates
                // step q + g against the state before step q.  Steps are committed in order as long as each one's
                // evaluation stands, i.e. no step committed before it in this pass moved its node AND touched what it
                // read (the rule of step_pair, applied to every earlier mover: their writes touch disjoint rows, so the
                // conditions compose); the first step that does not stand opens the next pass.  The six pairwise tests
                // are evaluated lane-parallel (lane 4 i + j: steps i and j) and arrive as one 16-bit word.
                const uint32_t row = lane >> 4;
                auto step_quad = [&](auto tm, uint32_t q, uint32_t nst) -> uint32_t {  // nst: steps of this pass that exist (1..4)
                    constexpr bool TM = decltype(tm)::value;
                    const uint32_t qs = q + min(row, nst - 1u);  // (rows past nst repeat the last step; their results are ignored)
                    const int sel = (int)(qs << 2);
                    const uint32_t prop = prop_of(qs);
                    const uint32_t pack = (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)pack_l);
                    const double u_acc = u_acc_of(qs);
                    const uint32_t deg = pack & 255u, r_loc = (pack >> 8) & 63u, t_loc = (pack >> 16) & 63u;
                    const int k = (int)hist8_cur[qs * kHistStride + lb];
                    const uint32_t a_rt = mq_at(r_loc, lb);
                    const int32_t m_rt_raw = mq[a_rt];
                    const int w_piv = mq[mq_at(lb, t_loc)];
                    const int nn_r = __builtin_amdgcn_ds_bpermute((int)(r_loc << 2), nr_own);
                    const int32_t kmask = (0 - k) >> 31;
                    const int32_t m_rt = m_rt_raw & kmask;
                    const uint32_t kk = (uint32_t)k;
                    const double L1 = tab_at(tab.lg, (uint32_t)(m_rt + 1));
                    const double L3 = tab_at(tab.lg, (uint32_t)(m_rt + 1) - kk);
                    __asm__ volatile("" ::: "memory");
                    // inverse CDF per row (:627-628): the row_shr scan does not leave its 16 lanes
                    const int scan = row_inclusive_scan16(w_piv);
                    const unsigned long long hit = __builtin_amdgcn_ballot_w64((uint32_t)scan > prop);
                    const uint32_t field = (uint32_t)(hit >> (row << 4)) & 0xffffu;
                    const uint32_t s_loc = min((uint32_t)__builtin_ctz(field | 0x10000u), last_own);
                    // (masks combined on the scalar side: one compare per ballot, no boolean round trips through the vector unit)
                    const unsigned long long m_valid = nst >= 4u ? ~0ull : ((1ull << (16u * nst)) - 1ull);  // lanes of the steps that exist
                    const unsigned long long m_live = __builtin_amdgcn_ballot_w64(nn_r != 1);  // (:467-471: a block is never emptied)
                    const unsigned long long m_self = __builtin_amdgcn_ballot_w64(s_loc == r_loc);
                    // T = 0: r == s is not accepted, and dS < 0 decides the others (:49-50)
                    const unsigned long long m_warm = CT ? ~0ull : ~__builtin_amdgcn_ballot_w64(((zeroT_mask >> qs) & 1ull) != 0ull);
                    constexpr unsigned long long kRowRep = 0x8000800080008000ull;  // one lane per row (its last)
                    const unsigned long long b_can = m_valid & m_live & ~m_self & kRowRep;
                    const unsigned long long b_selfok = m_valid & m_live & m_self & m_warm & kRowRep;
                    if (b_can == 0ull) {  // every step of the pass is an r == s (or a vetoed one): nothing changes (:109-112)
                        acc_chunk += (uint32_t)__builtin_popcountll(b_selfok);
                        if constexpr (TM)
                            if (b_selfok != 0ull) new_minimum(q + ((uint32_t)__builtin_ctzll(b_selfok) >> 4));  // (the first accepted one; the sum does not change)
                        return nst;
                    }
                    const uint32_t idx_l = r_loc ^ ((r_loc ^ s_loc) & (uint32_t)odd_mask_l);  // odd lanes: s, even lanes: r
                    const uint32_t a_st = mq_at(s_loc, lb);
                    const int32_t m_st_raw = mq[a_st];
                    const uint32_t e_idx = eta_at(idx_l, deg);
                    const int ee = (int)eta_l[e_idx];
                    const int mm = __builtin_amdgcn_ds_bpermute((int)(idx_l << 2), mr_own);
                    const int nn = __builtin_amdgcn_ds_bpermute((int)(idx_l << 2), nr_own);
                    const int32_t m_st = m_st_raw & kmask;
                    const int ideg = (int)deg;
                    const int qn = mm + __mul24(ideg, dsgn_l);        // m0r, m0s, m0r - deg, m0s + deg
                    const uint32_t tail_idx = (uint32_t)((qn ^ ((qn ^ ee) & eta_mask_l)) + toff_l);
                    const int qk = nn + dq_l;
                    const double tail_lg = tab_at(tab.lg, tail_idx);
                    const double logn = tab_at(tab.logtab, (uint32_t)qn);
                    const double L2 = tab_at(tab.lg, (uint32_t)(m_st + 1));
                    const double L4 = tab_at(tab.lg, (uint32_t)(m_st + 1) + kk);
                    const double a0 = k * (m_st + eps) * inv_blk;
                    const double a1 = k * (m_rt - k + eps) * inv_blk;
                    const double accu0 = butterfly_rows16(a0);  // every lane of a row: the row's sum
                    const double accu1 = butterfly_rows16(a1);
                    const double lq = hot_log_q(std::true_type{}, qn, qk, logn);
                    // pairwise: would step i, if it moves its node, touch what step j read?  (lane 4 i + j, any row)
                    // (worked out while the table gathers are in flight: after ALL of them have been issued -- the table tier of log_q is one more --, see step_pair)
                    uint32_t clash_bits;
                    {
                        uint32_t r_c = r_loc, s_c = s_loc, t_c = t_loc;
                        __asm__ volatile("" : "+v"(r_c), "+v"(s_c), "+v"(t_c)::"memory");
                        const int li = (int)(((lane >> 2) & 3u) << 6), lj = (int)((lane & 3u) << 6);  // lane 16 i, lane 16 j
                        const uint32_t r_i = (uint32_t)__builtin_amdgcn_ds_bpermute(li, (int)r_c);
                        const uint32_t s_i = (uint32_t)__builtin_amdgcn_ds_bpermute(li, (int)s_c);
                        const uint32_t r_j = (uint32_t)__builtin_amdgcn_ds_bpermute(lj, (int)r_c);
                        const uint32_t s_j = (uint32_t)__builtin_amdgcn_ds_bpermute(lj, (int)s_c);
                        const uint32_t t_j = (uint32_t)__builtin_amdgcn_ds_bpermute(lj, (int)t_c);
                        const uint32_t k_i_tj = (uint32_t)__builtin_amdgcn_ds_bpermute(li + (int)(t_j << 2), (int)kk);  // k of step i at block t_j
                        const uint32_t set_i = (1u << r_i) | (1u << s_i), set_j = (1u << r_j) | (1u << s_j);
                        const uint32_t lo = min(r_i, s_i), hi = max(r_i, s_i);
                        const uint32_t between = ((1u << hi) - 1u) & ~((2u << lo) - 1u);  // blocks strictly between r_i and s_i
                        // (bit arithmetic, no short-circuit: a lane-divergent `||` becomes a branch over the execution mask)
                        const uint32_t in_between = (between >> s_j) & min(k_i_tj, 1u);
                        clash_bits = (uint32_t)__builtin_amdgcn_ballot_w64(((set_i & set_j) | in_between) != 0u) & 0xffffu;  // bit 4 i + j
                    }
                    double d = (L1 + L2) - (L3 + L4);
                    d = d + tail_lg * sign_tail;
                    d = d + lq * sign_q;
                    const double dS = butterfly_rows16(d);
                    // accept (:47-61), per row
                    double invT = invT_const;
                    if (!CT) invT = invT_of(qs);
                    const double z = -dS * invT;
                    const double est = accu1 * exp2_filter(z * c_l2e);
                    const double lhs = u_acc * accu0;
                    unsigned long long b_acc = __builtin_amdgcn_ballot_w64(lhs < est);
                    unsigned long long b_far = __builtin_amdgcn_ballot_w64(fabs(lhs - est) > c_tol * est);
                    if (!CT) {  // steps at T = 0: dS < 0 decides (:49-50); nothing to be close to
                        b_acc = (b_acc & m_warm) | (__builtin_amdgcn_ballot_w64(dS < 0.) & ~m_warm);
                        b_far |= ~m_warm;
                    }
                    if (__builtin_expect((~b_far & b_can) != 0ull, 0)) {  // a verdict too close to call
                        const unsigned long long exact = __builtin_amdgcn_ballot_w64(lhs < accu1 * exp(z));
                        b_acc = (b_acc & b_far) | (exact & ~b_far);
                    }
                    // verdicts, in step order: bits 15 / 31 / 47 / 63 -> bits 0..3
                    auto rows4 = [](unsigned long long b) -> uint32_t {
                        const unsigned long long x = b >> 15;
                        return (uint32_t)(x | (x >> 15) | (x >> 30) | (x >> 45)) & 0xfu;
                    };
                    const uint32_t mv4 = rows4(b_can & b_acc), selfok4 = rows4(b_selfok);
                    // commit_j: steps 0..j all stand.  moved bits of committed steps only.
                    uint32_t moved = mv4 & 1u, commit = 1u;
                    {
                        const uint32_t c01 = (clash_bits >> 1) & 1u, c02 = (clash_bits >> 2) & 1u, c03 = (clash_bits >> 3) & 1u;
                        const uint32_t c12 = (clash_bits >> 6) & 1u, c13 = (clash_bits >> 7) & 1u, c23 = (clash_bits >> 11) & 1u;
                        const uint32_t m0 = moved & 1u;
                        const uint32_t k1 = sflag(nst - 1u) & ((m0 & c01) ^ 1u);  // (nst >= 2)
                        const uint32_t m1 = k1 & (mv4 >> 1) & 1u;
                        const uint32_t k2 = k1 & (nst > 2u ? 1u : 0u) & ((m0 & c02) ^ 1u) & ((m1 & c12) ^ 1u);
                        const uint32_t m2 = k2 & (mv4 >> 2) & 1u;
                        const uint32_t k3 = k2 & (nst > 3u ? 1u : 0u) & ((m0 & c03) ^ 1u) & ((m1 & c13) ^ 1u) & ((m2 & c23) ^ 1u);
                        const uint32_t m3 = k3 & (mv4 >> 3) & 1u;
                        commit = 1u | (k1 << 1) | (k2 << 2) | (k3 << 3);
                        moved = m0 | (m1 << 1) | (m2 << 2) | (m3 << 3);
                    }
                    acc_chunk += (uint32_t)__builtin_popcount(moved | (commit & selfok4));
                    if ((TM ? (moved | (commit & selfok4)) : moved) != 0u) {
                        // ---- apply_mcmc_moves, blockmodel.cc:461-503, for the steps that move: their rows of m differ ----
                        const unsigned long long movers = __builtin_amdgcn_ballot_w64(((moved >> row) & 1u) != 0u);  // all lanes of every mover's row
                        wfence();
                        if (__builtin_amdgcn_inverse_ballot_w64(movers & lanes_koth)) {  // k == 0: rewrites the same values
                            mq[a_rt] = m_rt_raw - k;
                            mq[a_st] = m_st_raw + k;
                        }
                        if (__builtin_amdgcn_inverse_ballot_w64(movers & 0x0030003000300030ull))  // lanes 4, 5 of a row: eta_r - 1, eta_s + 1
                            eta_l[e_idx] = (uint32_t)(ee + ((int)(lb & 1u) * 2 - 1));
                        if (__builtin_amdgcn_inverse_ballot_w64(movers & 0x0001000100010001ull)) new_lab[qs] = (uint8_t)(own_base + s_loc);
                        // the register copies of m_r / n_r, sum dS (:500) and the early-stop bookkeeping, in step order
#pragma unroll
                        for (uint32_t g = 0; g < 4u; ++g) {
                            if ((moved >> g) & 1u) {
                                const uint32_t rg = readlane(r_loc, 16u * g), sg = readlane(s_loc, 16u * g), dg = readlane(deg, 16u * g);
                                const int dl = (int)min(lb ^ rg, 1u) - (int)min(lb ^ sg, 1u);  // +1 on lane s, -1 on lane r
                                mr_own += __mul24((int)dg, dl);
                                nr_own += dl;
                                if constexpr (TM) cum_l0 += readlane(dS, 16u * g + 15u);
                                if constexpr (TM) new_minimum(q + g);
                            } else if (TM && (((commit & selfok4) >> g) & 1u)) {
                                new_minimum(q + g);  // an accepted r == s step (see new_minimum)
                            }
                        }
                        wfence();
                    }
                    return (uint32_t)__builtin_popcount(commit);
                };
                auto quad_loop = [&](auto tm) {
                    uint32_t q = 0;
                    acc_chunk = 0;
                    while (q < cnt) {
                        const uint32_t four = (uint32_t)(gen_mask >> q) & 0xfu;
                        if (__builtin_expect((four & 1u) != 0u, 0)) {
                            step_general(q, T_of_step(q));
                            q += 1u;
                        } else {  // the steps up to the next one that needs the general path, or to the end of the chunk
                            const uint32_t nst = min(min((uint32_t)__builtin_ctz(four | 0x10u), 4u), cnt - q);
                            q += step_quad(tm, q, nst);
                        }
                    }
                    acc_l0 += (unsigned long long)acc_chunk;
                };
                // ---- four steps per pass with 17..32 blocks of a type (the Q32 variant) ----
                // step_quad with TWO leaves per lane, the way step_pair64 extends step_pair: lane l of row g (step q + g) holds the
                // opposite-type blocks l and l + 16 -- their k, their entries of rows r and s, their table gathers, their leaves of the
                // three sums -- and, for the inverse CDF, entries l and l + 16 of column t.  The summation trees are the 64-leaf ones
                // (bisbm_device.hpp): the Hastings sums have level 16 second, so the lane adds its two leaves first; dS has it last,
                // so its two 16-leaf halves are summed separately and added at the end.  m_r / n_r sit one block per lane (two copies,
                // lanes 0..31 and 32..63) and are read by lane index.  Stand rule, verdicts and writes are step_quad's.
                auto step_quad32 = [&](auto tm, uint32_t q, uint32_t nst) -> uint32_t {  // nst: steps of this pass that exist (1..4)
                    constexpr bool TM = decltype(tm)::value;
                    const uint32_t qs = q + min(row, nst - 1u);  // (rows past nst repeat the last step; their results are ignored)
                    const int sel = (int)(qs << 2);
                    const uint32_t prop = prop_of(qs);
                    const uint32_t pack = (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)pack_l);
                    const double u_acc = u_acc_of(qs);
                    const uint32_t deg = pack & 255u, r_loc = (pack >> 8) & 63u, t_loc = (pack >> 16) & 63u;
                    const int k0 = (int)hist8_cur[qs * kHistStride + l16], k1 = (int)hist8_cur[qs * kHistStride + l16 + 16u];
                    const uint32_t a_rt0 = mq_at(r_loc, l16), a_rt1 = mq_at(r_loc, l16 + 16u);
                    const int32_t m_rt_raw0 = mq[a_rt0], m_rt_raw1 = mq[a_rt1];
                    const int w0 = mq[mq_at(l16, t_loc)], w1 = mq[mq_at(l16 + 16u, t_loc)];
                    const int nn_r = __builtin_amdgcn_ds_bpermute((int)(r_loc << 2), nr_own);
                    const int32_t kmask0 = (0 - k0) >> 31, kmask1 = (0 - k1) >> 31;
                    const int32_t m_rt0 = m_rt_raw0 & kmask0, m_rt1 = m_rt_raw1 & kmask1;
                    const uint32_t kk0 = (uint32_t)k0, kk1 = (uint32_t)k1;
                    const double L1_0 = tab_at(tab.lg, (uint32_t)(m_rt0 + 1)), L3_0 = tab_at(tab.lg, (uint32_t)(m_rt0 + 1) - kk0);
                    const double L1_1 = tab_at(tab.lg, (uint32_t)(m_rt1 + 1)), L3_1 = tab_at(tab.lg, (uint32_t)(m_rt1 + 1) - kk1);
                    __asm__ volatile("" ::: "memory");
                    // inverse CDF per row over 32 own blocks (:627-628): the scan of blocks 0..15, its total, the scan of blocks
                    // 16..31 on top (garbage past k_own, see step_pair64)
                    const int scan0 = row_inclusive_scan16(w0);
                    int tot = w0;  // (the row's total by four xor levels: every lane gets it without an LDS round trip)
                    tot += __builtin_amdgcn_update_dpp(0, tot, kDppXor1, 0xF, 0xF, false);
                    tot += __builtin_amdgcn_update_dpp(0, tot, kDppXor2, 0xF, 0xF, false);
                    tot += __builtin_amdgcn_update_dpp(0, tot, kDppHalfMirror, 0xF, 0xF, false);
                    tot += __builtin_amdgcn_update_dpp(0, tot, kDppMirror, 0xF, 0xF, false);
                    const int scan1 = row_inclusive_scan16(w1) + tot;
                    const unsigned long long hit0 = __builtin_amdgcn_ballot_w64((uint32_t)scan0 > prop);
                    const unsigned long long hit1 = __builtin_amdgcn_ballot_w64((uint32_t)scan1 > prop);
                    const uint32_t field = ((uint32_t)(hit0 >> (row << 4)) & 0xffffu) | ((uint32_t)(hit1 >> (row << 4)) << 16);
                    const uint32_t s_loc = min((uint32_t)__builtin_ctz(field | 0x80000000u), last_own);
                    // (masks combined on the scalar side: one compare per ballot, no boolean round trips through the vector unit)
                    const unsigned long long m_valid = nst >= 4u ? ~0ull : ((1ull << (16u * nst)) - 1ull);  // lanes of the steps that exist
                    const unsigned long long m_live = __builtin_amdgcn_ballot_w64(nn_r != 1);  // (:467-471: a block is never emptied)
                    const unsigned long long m_self = __builtin_amdgcn_ballot_w64(s_loc == r_loc);
                    // T = 0: r == s is not accepted, and dS < 0 decides the others (:49-50)
                    const unsigned long long m_warm = CT ? ~0ull : ~__builtin_amdgcn_ballot_w64(((zeroT_mask >> qs) & 1ull) != 0ull);
                    constexpr unsigned long long kRowRep = 0x8000800080008000ull;  // one lane per row (its last)
                    const unsigned long long b_can = m_valid & m_live & ~m_self & kRowRep;
                    const unsigned long long b_selfok = m_valid & m_live & m_self & m_warm & kRowRep;
                    if (b_can == 0ull) {  // every step of the pass is an r == s (or a vetoed one): nothing changes (:109-112)
                        acc_chunk += (uint32_t)__builtin_popcountll(b_selfok);
                        if constexpr (TM)
                            if (b_selfok != 0ull) new_minimum(q + ((uint32_t)__builtin_ctzll(b_selfok) >> 4));
                        return nst;
                    }
                    const uint32_t idx_l = r_loc ^ ((r_loc ^ s_loc) & (uint32_t)odd_mask_l);  // odd lanes: s, even lanes: r
                    const uint32_t a_st0 = mq_at(s_loc, l16), a_st1 = mq_at(s_loc, l16 + 16u);
                    const int32_t m_st_raw0 = mq[a_st0], m_st_raw1 = mq[a_st1];
                    const uint32_t e_idx = eta_at(idx_l, deg);
                    const int ee = (int)eta_l[e_idx];
                    const int mm = __builtin_amdgcn_ds_bpermute((int)(idx_l << 2), mr_own);
                    const int nn = __builtin_amdgcn_ds_bpermute((int)(idx_l << 2), nr_own);
                    const int32_t m_st0 = m_st_raw0 & kmask0, m_st1 = m_st_raw1 & kmask1;
                    const int ideg = (int)deg;
                    const int qn = mm + __mul24(ideg, dsgn_l);        // m0r, m0s, m0r - deg, m0s + deg
                    const uint32_t tail_idx = (uint32_t)((qn ^ ((qn ^ ee) & eta_mask_l)) + toff_l);
                    const int qk = nn + dq_l;
                    const double logn = tab_at(tab.logtab, (uint32_t)qn);
                    const double tail_lg = tab_at(tab.lg, tail_idx);
                    const double L2_0 = tab_at(tab.lg, (uint32_t)(m_st0 + 1)), L4_0 = tab_at(tab.lg, (uint32_t)(m_st0 + 1) + kk0);
                    const double L2_1 = tab_at(tab.lg, (uint32_t)(m_st1 + 1)), L4_1 = tab_at(tab.lg, (uint32_t)(m_st1 + 1) + kk1);
                    // the lane's two leaves of each Hastings sum, added first (level 16 of their tree)
                    const double a0 = k0 * (m_st0 + eps) * invq_lo + k1 * (m_st1 + eps) * invq_hi;
                    const double a1 = k0 * (m_rt0 - k0 + eps) * invq_lo + k1 * (m_rt1 - k1 + eps) * invq_hi;
                    const double accu0 = butterfly_rows16(a0);  // every lane of a row: the row's sum
                    const double accu1 = butterfly_rows16(a1);
                    const double lq = hot_log_q(std::false_type{}, qn, qk, logn);
                    // pairwise: would step i, if it moves its node, touch what step j read?  (lane 4 i + j, any row)
                    // (worked out while the table gathers are in flight: after ALL of them have been issued -- the table tier of log_q is one more --, see step_pair)
                    uint32_t clash_bits;
                    {
                        uint32_t r_c = r_loc, s_c = s_loc, t_c = t_loc;
                        __asm__ volatile("" : "+v"(r_c), "+v"(s_c), "+v"(t_c)::"memory");
                        const int li = (int)(((lane >> 2) & 3u) << 6), lj = (int)((lane & 3u) << 6);  // lane 16 i, lane 16 j
                        const uint32_t r_i = (uint32_t)__builtin_amdgcn_ds_bpermute(li, (int)r_c);
                        const uint32_t s_i = (uint32_t)__builtin_amdgcn_ds_bpermute(li, (int)s_c);
                        const uint32_t r_j = (uint32_t)__builtin_amdgcn_ds_bpermute(lj, (int)r_c);
                        const uint32_t s_j = (uint32_t)__builtin_amdgcn_ds_bpermute(lj, (int)s_c);
                        const uint32_t t_j = (uint32_t)__builtin_amdgcn_ds_bpermute(lj, (int)t_c);
                        const int src = li + (int)((t_j & 15u) << 2);  // k of step i at block t_j: leaf t_j >> 4 of lane t_j & 15 of row i
                        const uint32_t k_lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)kk0);
                        const uint32_t k_hi = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)kk1);
                        const uint32_t k_i_tj = k_lo ^ ((k_lo ^ k_hi) & (0u - (t_j >> 4)));
                        const uint32_t set_i = (1u << r_i) | (1u << s_i), set_j = (1u << r_j) | (1u << s_j);
                        const uint32_t lo = min(r_i, s_i), hi = max(r_i, s_i);
                        const uint32_t between = ((1u << hi) - 1u) & ~((2u << lo) - 1u);  // blocks strictly between r_i and s_i
                        // (bit arithmetic, no short-circuit: a lane-divergent `||` becomes a branch over the execution mask)
                        const uint32_t in_between = (between >> s_j) & min(k_i_tj, 1u);
                        clash_bits = (uint32_t)__builtin_amdgcn_ballot_w64(((set_i & set_j) | in_between) != 0u) & 0xffffu;  // bit 4 i + j
                    }
                    double d0 = (L1_0 + L2_0) - (L3_0 + L4_0);
                    d0 = d0 + tail_lg * sign_tail;  // the scalar terms sit in leaves 0..7 / 0..3: the lane's lower leaf
                    d0 = d0 + lq * sign_q;
                    const double d1 = (L1_1 + L2_1) - (L3_1 + L4_1);
                    const double dS = butterfly_rows16(d1) + butterfly_rows16(d0);  // level 16 of the dS tree comes last
                    // accept (:47-61), per row
                    double invT = invT_const;
                    if (!CT) invT = invT_of(qs);
                    const double z = -dS * invT;
                    const double est = accu1 * exp2_filter(z * c_l2e);
                    const double lhs = u_acc * accu0;
                    unsigned long long b_acc = __builtin_amdgcn_ballot_w64(lhs < est);
                    unsigned long long b_far = __builtin_amdgcn_ballot_w64(fabs(lhs - est) > c_tol * est);
                    if (!CT) {  // steps at T = 0: dS < 0 decides (:49-50); nothing to be close to
                        b_acc = (b_acc & m_warm) | (__builtin_amdgcn_ballot_w64(dS < 0.) & ~m_warm);
                        b_far |= ~m_warm;
                    }
                    if (__builtin_expect((~b_far & b_can) != 0ull, 0)) {  // a verdict too close to call
                        const unsigned long long exact = __builtin_amdgcn_ballot_w64(lhs < accu1 * exp(z));
                        b_acc = (b_acc & b_far) | (exact & ~b_far);
                    }
                    // verdicts, in step order: bits 15 / 31 / 47 / 63 -> bits 0..3
                    auto rows4 = [](unsigned long long b) -> uint32_t {
                        const unsigned long long x = b >> 15;
                        return (uint32_t)(x | (x >> 15) | (x >> 30) | (x >> 45)) & 0xfu;
                    };
                    const uint32_t mv4 = rows4(b_can & b_acc), selfok4 = rows4(b_selfok);
                    uint32_t moved = mv4 & 1u, commit = 1u;
                    {
                        const uint32_t c01 = (clash_bits >> 1) & 1u, c02 = (clash_bits >> 2) & 1u, c03 = (clash_bits >> 3) & 1u;
                        const uint32_t c12 = (clash_bits >> 6) & 1u, c13 = (clash_bits >> 7) & 1u, c23 = (clash_bits >> 11) & 1u;
                        const uint32_t m0 = moved & 1u;
                        const uint32_t k1c = sflag(nst - 1u) & ((m0 & c01) ^ 1u);  // (nst >= 2)
                        const uint32_t m1 = k1c & (mv4 >> 1) & 1u;
                        const uint32_t k2c = k1c & (nst > 2u ? 1u : 0u) & ((m0 & c02) ^ 1u) & ((m1 & c12) ^ 1u);
                        const uint32_t m2 = k2c & (mv4 >> 2) & 1u;
                        const uint32_t k3c = k2c & (nst > 3u ? 1u : 0u) & ((m0 & c03) ^ 1u) & ((m1 & c13) ^ 1u) & ((m2 & c23) ^ 1u);
                        const uint32_t m3 = k3c & (mv4 >> 3) & 1u;
                        commit = 1u | (k1c << 1) | (k2c << 2) | (k3c << 3);
                        moved = m0 | (m1 << 1) | (m2 << 2) | (m3 << 3);
                    }
                    acc_chunk += (uint32_t)__builtin_popcount(moved | (commit & selfok4));
                    if ((TM ? (moved | (commit & selfok4)) : moved) != 0u) {
                        // ---- apply_mcmc_moves, blockmodel.cc:461-503, for the steps that move: their rows of m differ ----
                        const unsigned long long movers = __builtin_amdgcn_ballot_w64(((moved >> row) & 1u) != 0u);  // all lanes of every mover's row
                        wfence();
                        if (__builtin_amdgcn_inverse_ballot_w64(movers & lanes_koth32_lo)) {  // k == 0: rewrites the same values
                            mq[a_rt0] = m_rt_raw0 - k0;
                            mq[a_st0] = m_st_raw0 + k0;
                        }
                        if (__builtin_amdgcn_inverse_ballot_w64(movers & lanes_koth32_hi)) {
                            mq[a_rt1] = m_rt_raw1 - k1;
                            mq[a_st1] = m_st_raw1 + k1;
                        }
                        if (__builtin_amdgcn_inverse_ballot_w64(movers & 0x0030003000300030ull))  // lanes 4, 5 of a row: eta_r - 1, eta_s + 1
                            eta_l[e_idx] = (uint32_t)(ee + ((int)(lane & 1u) * 2 - 1));
                        if (__builtin_amdgcn_inverse_ballot_w64(movers & 0x0001000100010001ull)) new_lab[qs] = (uint8_t)(own_base + s_loc);
                        // the register copies of m_r / n_r, sum dS (:500) and the early-stop bookkeeping, in step order
#pragma unroll
                        for (uint32_t g = 0; g < 4u; ++g) {
                            if ((moved >> g) & 1u) {
                                const uint32_t rg = readlane(r_loc, 16u * g), sg = readlane(s_loc, 16u * g), dg = readlane(deg, 16u * g);
                                const int dl = (int)min(lb ^ rg, 1u) - (int)min(lb ^ sg, 1u);  // +1 on lane s, -1 on lane r
                                mr_own += __mul24((int)dg, dl);
                                nr_own += dl;
                                if constexpr (TM) cum_l0 += readlane(dS, 16u * g + 15u);
                                if constexpr (TM) new_minimum(q + g);
                            } else if (TM && (((commit & selfok4) >> g) & 1u)) {
                                new_minimum(q + g);  // an accepted r == s step (see new_minimum)
                            }
                        }
                        wfence();
                    }
                    return (uint32_t)__builtin_popcount(commit);
                };
                auto quad32_loop = [&](auto tm) {
                    uint32_t q = 0;
                    acc_chunk = 0;
                    while (q < cnt) {
                        const uint32_t four = (uint32_t)(gen_mask >> q) & 0xfu;
                        if (__builtin_expect((four & 1u) != 0u, 0)) {
                            step_general(q, T_of_step(q));
                            q += 1u;
                        } else {  // the steps up to the next one that needs the general path, or to the end of the chunk
                            const uint32_t nst = min(min((uint32_t)__builtin_ctz(four | 0x10u), 4u), cnt - q);
                            q += step_quad32(tm, q, nst);
                        }
                    }
                    acc_l0 += (unsigned long long)acc_chunk;
                };
                // ---- eight steps per pass (both block counts <= 8) ----
                // step_quad once more: group g of eight lanes evaluates step q + g.  The 28 pairwise tests fill the wave (lane
                // 8 j + i: earlier step i, later step j), so byte j of their ballot is the set of earlier steps step j clashes
                // with, and the commit chain tests one byte against the movers so far per step.
                const uint32_t grp = lane >> 3;
                auto step_oct = [&](auto tm, uint32_t q, uint32_t nst) -> uint32_t {  // nst: steps of this pass that exist (1..8)
                    constexpr bool TM = decltype(tm)::value;
                    const uint32_t qs = q + min(grp, nst - 1u);
                    const int sel = (int)(qs << 2);
                    const uint32_t prop = prop_of(qs);
                    const uint32_t pack = (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)pack_l);
                    const double u_acc = u_acc_of(qs);
                    const uint32_t deg = pack & 255u, r_loc = (pack >> 8) & 63u, t_loc = (pack >> 16) & 63u;
                    const int k = (int)hist8_cur[qs * kHistStride + lb];
                    const uint32_t a_rt = mq_at(r_loc, lb);
                    const int32_t m_rt_raw = mq[a_rt];
                    const int w_piv = mq[mq_at(lb, t_loc)];
                    const int nn_r = __builtin_amdgcn_ds_bpermute((int)(r_loc << 2), nr_own);
                    const int32_t kmask = (0 - k) >> 31;
                    const int32_t m_rt = m_rt_raw & kmask;
                    const uint32_t kk = (uint32_t)k;
                    const double L1 = tab_at(tab.lg, (uint32_t)(m_rt + 1));
                    const double L3 = tab_at(tab.lg, (uint32_t)(m_rt + 1) - kk);
                    __asm__ volatile("" ::: "memory");
                    const int scan = group_inclusive_scan8(w_piv, lb);  // inverse CDF per group of eight (:627-628)
                    const unsigned long long hit = __builtin_amdgcn_ballot_w64((uint32_t)scan > prop);
                    const uint32_t field = (uint32_t)(hit >> (grp << 3)) & 0xffu;
                    const uint32_t s_loc = min((uint32_t)__builtin_ctz(field | 0x100u), last_own);
                    // (masks combined on the scalar side: one compare per ballot, no boolean round trips through the vector unit)
                    const unsigned long long m_valid = nst >= 8u ? ~0ull : ((1ull << (8u * nst)) - 1ull);  // lanes of the steps that exist
                    const unsigned long long m_live = __builtin_amdgcn_ballot_w64(nn_r != 1);  // (:467-471: a block is never emptied)
                    const unsigned long long m_self = __builtin_amdgcn_ballot_w64(s_loc == r_loc);
                    // T = 0: r == s is not accepted, and dS < 0 decides the others (:49-50)
                    const unsigned long long m_warm = CT ? ~0ull : ~__builtin_amdgcn_ballot_w64(((zeroT_mask >> qs) & 1ull) != 0ull);
                    constexpr unsigned long long kGrpRep = 0x8080808080808080ull;  // one lane per group (its last)
                    const unsigned long long b_can = m_valid & m_live & ~m_self & kGrpRep;
                    const unsigned long long b_selfok = m_valid & m_live & m_self & m_warm & kGrpRep;
                    if (b_can == 0ull) {
                        acc_chunk += (uint32_t)__builtin_popcountll(b_selfok);
                        if constexpr (TM)
                            if (b_selfok != 0ull) new_minimum(q + ((uint32_t)__builtin_ctzll(b_selfok) >> 3));
                        return nst;
                    }
                    const uint32_t idx_l = r_loc ^ ((r_loc ^ s_loc) & (uint32_t)odd_mask_l);
                    const uint32_t a_st = mq_at(s_loc, lb);
                    const int32_t m_st_raw = mq[a_st];
                    const uint32_t e_idx = eta_at(idx_l, deg);
                    const int ee = (int)eta_l[e_idx];
                    const int mm = __builtin_amdgcn_ds_bpermute((int)(idx_l << 2), mr_own);
                    const int nn = __builtin_amdgcn_ds_bpermute((int)(idx_l << 2), nr_own);
                    const int32_t m_st = m_st_raw & kmask;
                    const int ideg = (int)deg;
                    const int qn = mm + __mul24(ideg, dsgn_l);
                    const uint32_t tail_idx = (uint32_t)((qn ^ ((qn ^ ee) & eta_mask_l)) + toff_l);
                    const int qk = nn + dq_l;
                    const double tail_lg = tab_at(tab.lg, tail_idx);
                    const double logn = tab_at(tab.logtab, (uint32_t)qn);
                    const double L2 = tab_at(tab.lg, (uint32_t)(m_st + 1));
                    const double L4 = tab_at(tab.lg, (uint32_t)(m_st + 1) + kk);
                    const double a0 = k * (m_st + eps) * inv_blk;
                    const double a1 = k * (m_rt - k + eps) * inv_blk;
                    const double accu0 = butterfly_groups8(a0);
                    const double accu1 = butterfly_groups8(a1);
                    const double lq = hot_log_q(std::true_type{}, qn, qk, logn);
                    unsigned long long clash_bits;  // byte j, bit i: step i (earlier), if it moves, touches what step j read
                    {  // (worked out while the table gathers are in flight: after ALL of them have been issued -- the table tier of log_q is one more --, see step_pair)
                        uint32_t r_c = r_loc, s_c = s_loc, t_c = t_loc;
                        __asm__ volatile("" : "+v"(r_c), "+v"(s_c), "+v"(t_c)::"memory");
                        const int li = (int)((lane & 7u) << 5), lj = (int)((lane >> 3) << 5);  // lane 8 i, lane 8 j
                        const uint32_t r_i = (uint32_t)__builtin_amdgcn_ds_bpermute(li, (int)r_c);
                        const uint32_t s_i = (uint32_t)__builtin_amdgcn_ds_bpermute(li, (int)s_c);
                        const uint32_t r_j = (uint32_t)__builtin_amdgcn_ds_bpermute(lj, (int)r_c);
                        const uint32_t s_j = (uint32_t)__builtin_amdgcn_ds_bpermute(lj, (int)s_c);
                        const uint32_t t_j = (uint32_t)__builtin_amdgcn_ds_bpermute(lj, (int)t_c);
                        const uint32_t k_i_tj = (uint32_t)__builtin_amdgcn_ds_bpermute(li + (int)(t_j << 2), (int)kk);
                        const uint32_t set_i = (1u << r_i) | (1u << s_i), set_j = (1u << r_j) | (1u << s_j);
                        const uint32_t lo = min(r_i, s_i), hi = max(r_i, s_i);
                        const uint32_t between = ((1u << hi) - 1u) & ~((2u << lo) - 1u);
                        const uint32_t in_between = (between >> s_j) & min(k_i_tj, 1u);  // (bit arithmetic: no lane-divergent branch)
                        clash_bits = __builtin_amdgcn_ballot_w64(((set_i & set_j) | in_between) != 0u);
                    }
                    double d = (L1 + L2) - (L3 + L4);
                    d = d + tail_lg * sign_tail;
                    d = d + lq * sign_q;
                    const double dS = butterfly_groups8(d);
                    double invT = invT_const;
                    if (!CT) invT = invT_of(qs);
                    const double z = -dS * invT;
                    const double est = accu1 * exp2_filter(z * c_l2e);
                    const double lhs = u_acc * accu0;
                    unsigned long long b_acc = __builtin_amdgcn_ballot_w64(lhs < est);
                    unsigned long long b_far = __builtin_amdgcn_ballot_w64(fabs(lhs - est) > c_tol * est);
                    if (!CT) {  // steps at T = 0: dS < 0 decides (:49-50); nothing to be close to
                        b_acc = (b_acc & m_warm) | (__builtin_amdgcn_ballot_w64(dS < 0.) & ~m_warm);
                        b_far |= ~m_warm;
                    }
                    if (__builtin_expect((~b_far & b_can) != 0ull, 0)) {
                        const unsigned long long exact = __builtin_amdgcn_ballot_w64(lhs < accu1 * exp(z));
                        b_acc = (b_acc & b_far) | (exact & ~b_far);
                    }
                    // bits 7 / 15 / ... / 63 -> bits 0..7
                    auto groups8 = [](unsigned long long b) -> uint32_t { return (uint32_t)(((b >> 7) * 0x0102040810204080ull) >> 56) & 0xffu; };
                    const uint32_t mv8 = groups8(b_can & b_acc), selfok8 = groups8(b_selfok);
                    uint32_t moved = mv8 & 1u, commit = 1u, stands = 1u;
#pragma unroll
                    for (uint32_t j = 1; j < 8u; ++j) {  // step j stands: it exists, every earlier one stood, no earlier mover clashes
                        const uint32_t clash_j = (uint32_t)(clash_bits >> (8u * j)) & moved;
                        stands &= (nst > j ? 1u : 0u) & (clash_j == 0u ? 1u : 0u);
                        commit |= stands << j;
                        moved |= (stands & (mv8 >> j) & 1u) << j;
                    }
                    acc_chunk += (uint32_t)__builtin_popcount(moved | (commit & selfok8));
                    if ((TM ? (moved | (commit & selfok8)) : moved) != 0u) {
                        const unsigned long long movers = __builtin_amdgcn_ballot_w64(((moved >> grp) & 1u) != 0u);  // all eight lanes of every mover's group
                        wfence();
                        if (__builtin_amdgcn_inverse_ballot_w64(movers & lanes_koth)) {
                            mq[a_rt] = m_rt_raw - k;
                            mq[a_st] = m_st_raw + k;
                        }
                        if (__builtin_amdgcn_inverse_ballot_w64(movers & 0x3030303030303030ull))  // lanes 4, 5 of a group: eta_r - 1, eta_s + 1
                            eta_l[e_idx] = (uint32_t)(ee + ((int)(lb & 1u) * 2 - 1));
                        if (__builtin_amdgcn_inverse_ballot_w64(movers & 0x0101010101010101ull)) new_lab[qs] = (uint8_t)(own_base + s_loc);
#pragma unroll
                        for (uint32_t g = 0; g < 8u; ++g) {
                            if ((moved >> g) & 1u) {
                                const uint32_t rg = readlane(r_loc, 8u * g), sg = readlane(s_loc, 8u * g), dg = readlane(deg, 8u * g);
                                const int dl = (int)min(lb ^ rg, 1u) - (int)min(lb ^ sg, 1u);
                                mr_own += __mul24((int)dg, dl);
                                nr_own += dl;
                                if constexpr (TM) cum_l0 += readlane(dS, 8u * g + 7u);
                                if constexpr (TM) new_minimum(q + g);
                            } else if (TM && (((commit & selfok8) >> g) & 1u)) {
                                new_minimum(q + g);
                            }
                        }
                        wfence();
                    }
                    return (uint32_t)__builtin_popcount(commit);
                };
                auto oct_loop = [&](auto tm) {
                    uint32_t q = 0;
                    acc_chunk = 0;
                    while (q < cnt) {
                        const uint32_t eight = (uint32_t)(gen_mask >> q) & 0xffu;
                        if (__builtin_expect((eight & 1u) != 0u, 0)) {
                            step_general(q, T_of_step(q));
                            q += 1u;
                        } else {
                            const uint32_t nst = min(min((uint32_t)__builtin_ctz(eight | 0x100u), 8u), cnt - q);
                            q += step_oct(tm, q, nst);
                        }
                    }
                    acc_l0 += (unsigned long long)acc_chunk;
                };
                // steps that need the general path (bit 31 of prop_l) go one at a time
                auto pair_loop = [&](auto tm) {
                    uint32_t q = 0;
                    acc_chunk = 0;
                    while (q < cnt) {
                        const uint32_t two = (uint32_t)(gen_mask >> q) & 3u;
                        if (__builtin_expect((two & 1u) != 0u, 0)) {
                            step_general(q, T_of_step(q));
                            q += 1u;
                        } else if (K32) {
                            q += step_pair(tm, q, ((two >> 1) ^ 1u) & sflag(cnt - 1u - q));
                        } else {  // (one step per pass, BISBM_SINGLE_STEPS=1: a pass whose two halves evaluate the same step)
                            const uint32_t done = step_pair64(tm, q, ((two >> 1) ^ 1u) & sflag(cnt - 1u - q) & (pair64_mode ? 1u : 0u));
                            if constexpr (kPredictTarget) {  // (0: the predicted target of step q was not its target -- the general path takes it, see step_pair64)
                                if (__builtin_expect(done == 0u, 0)) gen_mask |= 1ull << q;
                            }
                            q += done;
                        }
                    }
                    acc_l0 += (unsigned long long)acc_chunk;
                };
                if constexpr (!K32) {  // more than 32 blocks of a type: step_pair64 is the only hot step this variant holds (registers)
                    if (track_min != 0u)
                        pair_loop(std::true_type{});
                    else
                        pair_loop(std::false_type{});
                } else if (K8 && oct_mode) {
                    if (track_min != 0u)
                        oct_loop(std::true_type{});
                    else
                        oct_loop(std::false_type{});
                } else if (K16 && quad_mode) {
                    if (track_min != 0u)
                        quad_loop(std::true_type{});
                    else
                        quad_loop(std::false_type{});
                } else if (Q32 && quad32_mode) {
                    if (track_min != 0u)
                        quad32_loop(std::true_type{});
                    else
                        quad32_loop(std::false_type{});
                } else if (K32 && !K16 && !Q32 && pair_mode) {  // (the K <= 16 / K <= 8 kernels hold their own kind of pass only: registers)
                    if (track_min != 0u)
                        pair_loop(std::true_type{});
                    else
                        pair_loop(std::false_type{});
                } else if (track_min != 0u) {
                    for (uint32_t q = 0; q < cnt; ++q) step(std::true_type{}, q);
                } else {
                    for (uint32_t q = 0; q < cnt; ++q) step(std::false_type{}, q);
                }
                {  // the chunk's label stores (see new_lab)
                    wfence();
                    const uint32_t moved_to = new_lab[lane];
                    if (moved_to != 0xffu) labels[hand[0 * kWave + lane]] = (uint8_t)moved_to;
                }
                if (track_min != 0u) {
                    wfence();
                    *below1_total = below1_before + (unsigned long long)__builtin_popcountll(below1_mask);
                    wfence();
                }
            };

            // chunk pipeline: the feeder is one chunk ahead; one workgroup barrier per chunk
            if (is_main) mr_pub[lane] = mr_oth;
            __syncthreads();
            if (!is_main) prepare(0);
            __syncthreads();
            for (uint32_t c = 0; c < n_chunks; ++c) {
                if (is_main) {
                    run_steps(c);
                    FSTAMP(9);
                } else if (c + 1 < n_chunks) {
                    prepare(c + 1);
                }
                __syncthreads();
                if (is_main) FSTAMP(10);  // (diagnostic builds: time spent waiting for the feeder)
            }
            if (!EL && is_main) {  // the window goes back to HBM: the other type's rows take its place
                wfence();
                for (uint32_t i = lane; i < k_own * eta_w; i += kWave) {
                    const uint32_t row = i / eta_w, d = eta_lo + i % eta_w;
                    if (d < D) eta_g[(own_base + row) * D + d] = eta_l[i];
                }
                wfence();
            }
        };
        run_phase(std::false_type{});
        run_phase(std::true_type{});

        ++sweeps_total;
        sweeps_done = sweep + 1;
        {  // metropolis_hasting.cc:96-98 (without the bookkeeping u stays 0)
            const unsigned long long mark = ((unsigned long long)readlane((uint32_t)(mark_l0 >> 32), 0u) << 32) | readlane((uint32_t)mark_l0, 0u);
            const unsigned long long u_now = track_min != 0u ? *below1_total - mark : 0ull;
            if (is_main && lane == 0 && u_now >= p.steps_await) *stop_flag = 1;
        }
        __syncthreads();
        if (__builtin_amdgcn_readfirstlane((int)*stop_flag)) {  // (scalar: the sweep loop has no divergent exit)
            rate = (double)acc_l0 / (double)((sweep + 1) * (uint64_t)n);
            stopped = true;
            break;
        }
    }
    if (!stopped) rate = (double)acc_l0 / (double)p.duration;  // :100

    // chain state -> HBM (stepping wave)
    __syncthreads();
    if (!is_main) return;
    for (uint32_t i = lane; i < ka * kb; i += kWave) m_g[i] = mq[(i / kb) * S + (i % kb)];
    if (lane < ka) {
        mr_g[lane] = mrA;
        nr_g[lane] = nrA;
    }
    if (lane < kb) {
        mr_g[ka + lane] = mrB;
        nr_g[ka + lane] = nrB;
    }
    if (EL)
        for (uint32_t i = lane; i < K * D; i += kWave) eta_g[i] = eta_l[i];
#ifdef BISBM_STAMPS
    if (lane == 0)
        for (int i = 0; i < 12; ++i) atomicAdd(&g_fast_stamps[i], st_acc[i]);
#endif
    if (lane == 0) {
        // (a launch without the early-stop bookkeeping has not kept the running sum: bisbm_anneal sets it from the change of the
        // description length, sweep_fast_sum_from_entropy)
        if (track_min != 0u) sc->cum_dS = cum_l0;
        sc->sweeps_total = sweeps_total;
        sc->last_rate = rate;
        sc->last_accepted = acc_l0;
        sc->last_sweeps = sweeps_done;
        sc->stop_emin = emin_l0;  // (lane 0's copies are the values)
        sc->stop_mark = mark_l0;
        sc->stop_below1 = *below1_total;
        sc->stopped = stopped ? 1u : 0u;
        // give the SIMD back: workgroups of a later round (more chains than the chip holds at once) claim afresh
        if (p.simd_claims != nullptr) atomicSub(&p.simd_claims[role[wave_in_wg]], 1u);
    }
}

template <bool EL, bool CT, bool K32, bool K16, bool K8, bool Q32 = false>
static hipError_t launch_fast_variant3(const SweepParams& p, size_t lds_bytes, hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute((const void*)sweep_fast_kernel<EL, CT, K32, K16, K8, Q32>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((sweep_fast_kernel<EL, CT, K32, K16, K8, Q32>), dim3(p.n_chains), dim3(2 * kWave), lds_bytes, stream, p);
    return hipGetLastError();
}

// One kernel per kind of pass (a kernel that held all of them ran out of registers): p.pass_depth, set by the host, says
// which one a launch with few blocks takes.
template <bool EL, bool CT>
hipError_t launch_fast_variant(const SweepParams& p, size_t lds_bytes, hipStream_t stream) {
    const uint32_t depth = p.pair_steps < p.pass_depth ? p.pair_steps : p.pass_depth;  // 0 / 1 / 2 / 3: one, two, four, eight steps per pass
    if (p.ka <= 8u && p.kb <= 8u && depth >= 3u) return launch_fast_variant3<EL, CT, true, true, true>(p, lds_bytes, stream);
    if (p.ka <= 16u && p.kb <= 16u && depth >= 2u) return launch_fast_variant3<EL, CT, true, true, false>(p, lds_bytes, stream);
    if (p.ka <= 32u && p.kb <= 32u && depth >= 2u) return launch_fast_variant3<EL, CT, true, false, false, true>(p, lds_bytes, stream);
    return (p.ka <= 32u && p.kb <= 32u) ? launch_fast_variant3<EL, CT, true, false, false>(p, lds_bytes, stream)
                                        : launch_fast_variant3<EL, CT, false, false, false>(p, lds_bytes, stream);
}

// The 20 variants compile side by side: the build (build.py) compiles this file once per BISBM_FAST_PART = 0 .. 3 -- the five
// kernels of one (EL, CT) each -- and once with BISBM_FAST_PART = 4 for the dispatch below; undefined: everything in one unit
// (diagnostic builds with in-kernel stamps, whose counters are one device symbol).
#ifndef BISBM_FAST_PART
#define BISBM_FAST_PART -1
#endif
#define BISBM_FAST_HERE(part) (BISBM_FAST_PART == -1 || BISBM_FAST_PART == (part))
#if BISBM_FAST_HERE(0)
template hipError_t launch_fast_variant<false, false>(const SweepParams&, size_t, hipStream_t);
#endif
#if BISBM_FAST_HERE(1)
template hipError_t launch_fast_variant<false, true>(const SweepParams&, size_t, hipStream_t);
#endif
#if BISBM_FAST_HERE(2)
template hipError_t launch_fast_variant<true, false>(const SweepParams&, size_t, hipStream_t);
#endif
#if BISBM_FAST_HERE(3)
template hipError_t launch_fast_variant<true, true>(const SweepParams&, size_t, hipStream_t);
#endif

#if BISBM_FAST_HERE(4)
#if BISBM_FAST_PART == 4
extern template hipError_t launch_fast_variant<false, false>(const SweepParams&, size_t, hipStream_t);
extern template hipError_t launch_fast_variant<false, true>(const SweepParams&, size_t, hipStream_t);
extern template hipError_t launch_fast_variant<true, false>(const SweepParams&, size_t, hipStream_t);
extern template hipError_t launch_fast_variant<true, true>(const SweepParams&, size_t, hipStream_t);
#endif

size_t sweep_fast_lds_bytes(uint32_t ka, uint32_t kb, uint32_t maxdeg, bool eta_in_lds, uint32_t eta_window) {
    const uint32_t K = ka + kb, D = maxdeg + 1, S = kb | 1u;
    const size_t dwords = (size_t)ka * S + (eta_in_lds ? (size_t)K * D : (size_t)std::max(ka, kb) * eta_window) +
                          2 * (size_t)kWave * (kHistStride / 4) + 2 * (size_t)kHandWords * kWave + kWave + 4 + 10 + 16 + 64 + 128;
    // the step reads m[.][lane] for all 64 lanes whatever ka, kb are (idle lanes are masked after the read):
    // dword index <= 63 * S + 63 must be inside the allocation
    const size_t reach = 63 * (size_t)S + 64;
    return ((dwords > reach ? dwords : reach) * 4 + 15) & ~(size_t)15;
}

hipError_t launch_sweep_fast(const SweepParams& p, size_t /*generic_lds_bytes*/, hipStream_t stream) {
    const bool ct = p.schedule == SCHED_CONSTANT;
    const size_t lds_bytes = sweep_fast_lds_bytes(p.ka, p.kb, p.maxdeg, p.eta_in_lds != 0, p.eta_w);
    hipError_t e;
    if (p.eta_in_lds)
        e = ct ? launch_fast_variant<true, true>(p, lds_bytes, stream) : launch_fast_variant<true, false>(p, lds_bytes, stream);
    else
        e = ct ? launch_fast_variant<false, true>(p, lds_bytes, stream) : launch_fast_variant<false, false>(p, lds_bytes, stream);
#ifdef BISBM_STAMPS
    if (e == hipSuccess) {
        (void)hipStreamSynchronize(stream);
        unsigned long long h[16] = {0};
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fast_stamps), sizeof(h));
        const double steps = (double)p.n_chains * (double)(p.duration / p.n) * (double)p.n;
        static const char* names[11] = {"step entry (+ early exits)", "early reads", "proposal", "lds+gather issue", "accu",
                                        "log_q", "dS butterfly", "accept", "apply", "chunk tail", "barrier wait"};
        double tot = 0;
        for (int i = 0; i < 11; ++i) tot += (double)h[i];
        for (int i = 0; i < 11; ++i) fprintf(stderr, "[stamps] %-28s %8.2f ticks/step\n", names[i], (double)h[i] / steps);
        fprintf(stderr, "[stamps] %-20s %8.1f cyc/step\n", "total", tot / steps);
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fast_stamps), z, sizeof(z));
    }
#endif
    return e;
}
#endif  // BISBM_FAST_HERE(4)

}  // namespace bisbm
