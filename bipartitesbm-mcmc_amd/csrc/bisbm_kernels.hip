// bisbm_kernels.hip -- HIP kernels of the MH sweep engine (gfx950 / CDNA4, wave64).
//
//   sweep_kernel<RNG>        metropolis_hasting::anneal -> step -> transition_ratio
//                            (metropolis_hasting.cc:42-192) + single_vertex_change
//                            (blockmodel.cc:613-637) + apply_mcmc_moves (blockmodel.cc:461-503):
//                            one wavefront = one chain, persistent over all sweeps of the call.
//   state_build_kernel       init_bisbm (blockmodel.cc:682-746): n_r, m, m_r, eta from labels.
//   shuffle_*_kernel         shuffle_bisbm (blockmodel.cc:672-680).
//   entropy_kernel           the block-state part of entropy() (blockmodel.cc:758-771).
//   marginals_kernel         per-node label histogram over chains.
//   log_q_probe_kernel       numerics probe for tests.
//
// Data layout in HBM (per handle; chain-major so a chain's arrays are contiguous):
//   rowptr u32[n+1], col u32[2E] (edge-file order)          shared by all chains
//   labels u8 [chain][label_stride]                          node -> block (K <= 256)
//   m      i32[chain][ka*kb]   a x b quadrant of the symmetric block matrix
//   m_r,n_r i32[chain][K];  eta u32[chain][K*(maxdeg+1)]
//   lgamma f64[lg_size], log_q f64[10001*(kcap+1)]           host-built tables (glibc values)
// During a sweep kernel the chain's m (row stride padded to an odd number of banks so that both
// row and column walks are conflict-free), m_r, n_r, the k_v histogram and (when it fits) eta
// live in LDS; labels stay in HBM and are gathered through the CSR walk.
//
// Compiled with -ffp-contract=off: FP64 expressions must round exactly like the host code they
// are compared with (no FMA contraction).
#include "bisbm_kernels.hpp"

#include <cstdio>
#include <type_traits>

namespace bisbm {

// Diagnostic build only (-DBISBM_GSTAMPS): s_memtime per stage of the generic kernel's step, summed per launch.
#ifdef BISBM_GSTAMPS
__device__ unsigned long long g_generic_stamps[16];
#define GSTAMP(c, i)                                                                     \
    do {                                                                                 \
        unsigned long long now_;                                                         \
        __builtin_amdgcn_sched_barrier(0);                                               \
        __asm__ volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                               \
        (c).st_acc[i] += now_ - (c).st_prev;                                             \
        (c).st_prev = now_;                                                              \
    } while (0)
#else
#define GSTAMP(c, i) \
    do {             \
    } while (0)
#endif

// ------------------------------------------------------------------------------------------
// per-chain context held in registers / LDS while a sweep kernel runs
// ------------------------------------------------------------------------------------------
struct ChainCtx {
    // LDS
    int32_t* mq;     // ka * S
    int32_t* mr;     // K
    int32_t* nr;     // K
    int32_t* hist;   // max(ka, kb)
    uint32_t* eta_l;  // K * (maxdeg+1) in LDS when it fits (eta_lds), else
    uint32_t* eta_g;  // the chain's array in HBM
    bool eta_lds;
    // global
    uint8_t* labels;
    uint16_t* labels16;  // wide mode (K > 256): two-byte labels, and ...
    int32_t* mg;         // ... the a x b quadrant of m stays in HBM (row stride kb): it no longer fits the LDS
    // shape
    uint32_t ka, kb, K, S, D;  // S = LDS row stride of mq, D = maxdeg + 1
    uint32_t na;
    double epsilon;
    // MH object state
    double cum_dS;
    double accu_r;
#ifdef BISBM_GSTAMPS
    unsigned long long st_acc[10];
    unsigned long long st_prev;
#endif
};

// eta lives in LDS or in HBM, chosen at compile time (template EL): one generic pointer would turn the
// accesses into flat loads, which wait on vmcnt AND lgkmcnt and so drain the prefetch pipeline
template <bool EL>
__device__ __forceinline__ uint32_t eta_load(const ChainCtx& c, uint32_t idx) {
    if constexpr (EL)
        return c.eta_l[idx];
    else
        return c.eta_g[idx];
}
template <bool EL>
__device__ __forceinline__ void eta_store(const ChainCtx& c, uint32_t idx, uint32_t val) {
    if constexpr (EL)
        c.eta_l[idx] = val;
    else
        c.eta_g[idx] = val;
}

// m[own block i][opposite block j] for a node of the given type, from the a x b quadrant
template <bool W>
__device__ __forceinline__ int32_t& Mx(const ChainCtx& c, bool type_b, uint32_t i_own, uint32_t j_oth) {
    if constexpr (W)
        return type_b ? c.mg[j_oth * c.kb + i_own] : c.mg[i_own * c.kb + j_oth];
    else
        return type_b ? c.mq[j_oth * c.S + i_own] : c.mq[i_own * c.S + j_oth];
}

// label of a node: a byte, or two in wide mode
template <bool W>
__device__ __forceinline__ uint32_t lab_at(const ChainCtx& c, uint32_t v) {
    if constexpr (W)
        return c.labels16[v];
    else
        return c.labels[v];
}
template <bool W>
__device__ __forceinline__ void lab_set(const ChainCtx& c, uint32_t v, uint32_t s) {
    if constexpr (W)
        c.labels16[v] = (uint16_t)s;
    else
        c.labels[v] = (uint8_t)s;
}

// ------------------------------------------------------------------------------------------
// one MH step for node v (metropolis_hasting.cc:42-62)
//
// The caller hands over the node's CSR row already in registers: lane j holds neighbour id nb and
// its label lab for j < min(deg, 64) (prefetched one step ahead, see sweep_kernel); neighbours
// beyond 64 are read here.  No workgroup barrier is used on this path: a barrier makes the compiler
// drain every outstanding global load (vmcnt(0)), which would serialise the prefetch pipeline.  The
// wave's LDS operations execute in issue order, so wave_fence() (a code-motion barrier only) is
// enough between a lane's LDS write and another lane's read.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_wave_barrier();
    __asm__ volatile("" ::: "memory");
}

template <int RNG, bool EL, bool W>
__device__ __forceinline__ bool mh_step(const SweepParams& p, const Tables& tab, ChainCtx& c, Mt& engine, Mt& gen,
                                        uint32_t v, uint32_t beg, uint32_t deg, uint32_t r, uint32_t nb_reg,
                                        int lab_reg, double T, uint64_t gstep, uint32_t chain_gid, uint32_t* s_out) {
    const int lane = lane_id();
    const bool type_b = v >= c.na;
    const uint32_t K = c.K;
    const uint32_t k_own = type_b ? c.kb : c.ka, k_oth = type_b ? c.ka : c.kb;
    const uint32_t own_base = type_b ? c.ka : 0, oth_base = type_b ? 0 : c.ka;
    const uint32_t r_loc = r - own_base;
    *s_out = r;
    GSTAMP(c, 1);

    // ---- k_v: neighbour-label histogram of the CSR row (replaces the dense k_[v] row of
    //      blockmodel.cc:691-700), reduced in LDS ----
    for (uint32_t t = lane; t < k_oth; t += kWave) c.hist[t] = 0;
    wave_fence();
    if ((uint32_t)lane < deg) atomicAdd(&c.hist[lab_reg - (int)oth_base], 1);
    for (uint32_t j = kWave + lane; j < deg; j += kWave)  // rows longer than one wave (rare)
        atomicAdd(&c.hist[(int)lab_at<W>(c, p.col[beg + j]) - (int)oth_base], 1);
    wave_fence();
    GSTAMP(c, 2);

    // ---- proposal: single_vertex_change, blockmodel.cc:613-637 ----
    uint32_t s;
    double u_acc = 0.;
    const bool win_ok = (RNG == RNG_COMPAT) && engine.window4_ok();
    double win = 0.;
    uint32_t win_used = 0;
    if (RNG == RNG_COMPAT && win_ok) win = engine.window4();
    auto draw_engine = [&]() -> double {
        if (win_ok) return readlane(win, win_used++);
        return engine.canonical();
    };
    if (RNG == RNG_PHILOX) {
#if defined(BISBM_ABLATE) && (BISBM_ABLATE & 2)
        const uint32_t hsh = mix32((uint32_t)gstep ^ chain_gid);  // diagnostic build: Philox removed
        const U4 A{hsh, hsh * 3u, hsh * 5u, hsh * 7u}, B{hsh * 11u, hsh * 13u, hsh * 17u, hsh * 19u};
#else
        const U4 A = phx_draw(p.seed, chain_gid, PHX_STEP_A, gstep);
        const U4 B = phx_draw(p.seed, chain_gid, PHX_STEP_B, gstep);
#endif
        const double u_idx = u53(A.x, A.y), u_R = u53(A.z, A.w), u_tgt = u53(B.x, B.y);
        u_acc = u53(B.z, B.w);
        if (k_own == 1) {
            s = r;
        } else if (deg == 0) {
            s = (uint32_t)(u_idx * (double)K);
            if (s >= K) s = K - 1;
        } else {
            uint32_t which = (uint32_t)(u_idx * (double)deg);
            if (which >= deg) which = deg - 1;
            const uint32_t t = which < (uint32_t)kWave ? (uint32_t)readlane(lab_reg, which)
                                                        : lab_at<W>(c, p.col[beg + which]);
            const int32_t mrt = c.mr[t];
            if (u_R * (mrt + c.epsilon * (double)K) < c.epsilon * (double)K) {  // :622-624 without the division
                s = (uint32_t)(u_tgt * (double)K);
                if (s >= K) s = K - 1;
            } else {
                // integer inverse CDF over row m[t][.] restricted to v's own type
                long long x = (long long)(u_tgt * (double)mrt);
                if (x >= (long long)mrt) x = (long long)mrt - 1;
                const uint32_t t_loc = t - oth_base;
                long long carry = 0;  // row sums are < 2^31 (m_r is int32), so the in-wave scan is 32-bit
                s = own_base + k_own - 1;
                for (uint32_t c0 = 0; c0 < k_own; c0 += kWave) {
                    const uint32_t i = c0 + lane;
                    const int w = i < k_own ? Mx<W>(c, type_b, i, t_loc) : 0;
                    const int scan = wave_inclusive_scan(w);
                    const long long cum = carry + (long long)scan;
                    const unsigned long long hit = __ballot(i < k_own && cum > x);
                    if (hit) {
                        s = own_base + c0 + (uint32_t)__ffsll((long long)hit) - 1;
                        break;
                    }
                    carry += (long long)readlane(scan, (uint32_t)(kWave - 1));
                }
            }
        }
    } else {
        // the reference's draw order: engine for idx / R / uniform target,
        // `gen` for the discrete draw.  size_t(u * K), size_t(u * deg): the products are below 2^32, so the conversion
        // to uint32_t (one instruction; there is no f64 -> u64 one) truncates to the same integer.  The (at most four) uniforms a step takes from `engine` come out of one
        // lane-parallel read of its tempered words whenever no regeneration falls inside the step.
        if (k_own == 1) {
            s = r;
        } else if (deg == 0) {
            s = (uint32_t)(draw_engine() * (double)K);
        } else {
            const uint32_t which = (uint32_t)(draw_engine() * (double)deg);
            const uint32_t t = which < (uint32_t)kWave ? (uint32_t)readlane(lab_reg, which)
                                                        : lab_at<W>(c, p.col[beg + which]);
            const int32_t mrt = c.mr[t];
            const double R_t = c.epsilon * (double)K / (mrt + c.epsilon * (double)K);
            if (draw_engine() < R_t) {
                s = (uint32_t)(draw_engine() * (double)K);
            } else {
                // std::discrete_distribution over the full row m_[t][0..K) (random.tcc:2656-2714):
                // p = w / sum, serial partial sums, last = 1.0, lower_bound(u)
                const uint32_t t_loc = t - oth_base;
                const double u = gen.canonical();
                const double sum = (double)mrt;  // accumulate() of integers is exact
                double acc = 0.;
                s = K - 1;
                bool found = false;
                for (uint32_t c0 = 0; c0 < K && !found; c0 += kWave) {
                    const uint32_t g = c0 + lane;
                    int32_t w = 0;
                    if (g < K && g >= own_base && g < own_base + k_own) w = Mx<W>(c, type_b, g - own_base, t_loc);
                    const double pr = (double)w / sum;
                    // partial sums in index order; a zero weight adds +0.0 (the sum keeps its bits), so only the
                    // non-zero entries take a serial step and every lane keeps the sum up to its own index
                    double cp = acc;
                    for (unsigned long long nz = __ballot(w != 0); nz; nz &= nz - 1) {
                        const uint32_t jj = (uint32_t)__builtin_ctzll(nz);
                        acc = acc + readlane(pr, jj);
                        if ((uint32_t)lane >= jj) cp = acc;
                    }
                    if (g == K - 1) cp = 1.0;
                    const unsigned long long hit = __ballot(g < K && cp >= u);
                    if (hit) {
                        s = c0 + (uint32_t)__ffsll((long long)hit) - 1;
                        found = true;
                    }
                }
            }
        }
    }
    *s_out = s;
    GSTAMP(c, 3);

    // ---- transition_ratio, metropolis_hasting.cc:103-192 ----
    double dS;
    double phx_accu0 = 1., phx_accu1 = 1.;
    const bool same = (r == s);
    const bool cross = !same && ((r < c.ka) != (s < c.ka));
    if (same) {
        c.accu_r = 1.;  // :109-112
        dS = 0.;
    } else if (cross) {
        dS = INFINITY;  // :121-123, accu_r left stale
    } else {
        const uint32_t s_loc = s - own_base;
        const double Kd = (double)K;
        const double eps = c.epsilon;
        const int ideg = (int)deg;
        const int m0r = c.mr[r], m1r = m0r - ideg;
        const int m0s = c.mr[s], m1s = m0s + ideg;
        const int n_r_r = c.nr[r], n_r_s = c.nr[s];
        const int eta_r = (int)eta_load<EL>(c, r * c.D + deg), eta_s = (int)eta_load<EL>(c, s * c.D + deg);

        // (1) every table gather of the step is issued up front: the eight scalar-tail lgamma values
        //     ride in lanes 0..7 of one wave-wide load ...
        long long tail_idx = 1;
        switch (lane) {
            case 0: tail_idx = (long long)m0r + 1; break;    // :164
            case 1: tail_idx = (long long)m0s + 1; break;    // :165
            case 2: tail_idx = (long long)m1r + 1; break;    // :167
            case 3: tail_idx = (long long)m1s + 1; break;    // :168
            case 4: tail_idx = (long long)eta_r + 1; break;  // :173
            case 5: tail_idx = (long long)eta_s + 1; break;  // :174
            case 6: tail_idx = (long long)eta_r; break;      // :176  (eta_r - 1 + 1)
            case 7: tail_idx = (long long)eta_s + 2; break;  // :177
            default: break;
        }
        const double tail_lg = lgamma_fast(tab, tail_idx);

        // ... and the per-block terms of the first 64 opposite-type blocks in lane t
        struct Terms {
            int k;
            int32_t m_rt, m_st, mr_t;
            double L1, L2, L3, L4;
        };
        auto load_terms = [&](uint32_t j) {
            Terms t{0, 0, 0, 0, 0., 0., 0., 0.};
            if (j < k_oth) t.k = c.hist[j];
            if (t.k != 0) {
                t.m_rt = Mx<W>(c, type_b, r_loc, j);
                t.m_st = Mx<W>(c, type_b, s_loc, j);
                t.mr_t = c.mr[oth_base + j];
                t.L1 = lgamma_fast(tab, (long long)t.m_rt + 1);
                t.L2 = lgamma_fast(tab, (long long)t.m_st + 1);
                t.L3 = lgamma_fast(tab, (long long)t.m_rt - t.k + 1);
                t.L4 = lgamma_fast(tab, (long long)t.m_st + t.k + 1);
            }
            return t;
        };
        Terms t0 = load_terms((uint32_t)lane);

        // (2) while those loads are in flight: the four log_q values (:179-183), one per lane, in one
        //     SIMT evaluation
        int qn = 0, qk = 0;
        if (lane == 0) { qn = m0r; qk = n_r_r; }
        if (lane == 1) { qn = m0s; qk = n_r_s; }
        if (lane == 2) { qn = m1r; qk = n_r_r - 1; }
        if (lane == 3) { qn = m1s; qk = n_r_s + 1; }
#if defined(BISBM_ABLATE) && (BISBM_ABLATE & 1)
        const double lq = (double)(qn + qk) * 1e-9;  // diagnostic build: log_q removed (wrong results)
#else
        const double lq = log_q<RNG == RNG_PHILOX>(tab, qn, qk, (RNG == RNG_PHILOX && qn > 0 && (uint64_t)qn < tab.lg_size) ? tab.logtab[qn] : 0.);
#endif

        GSTAMP(c, 4);
        // (3) sums over opposite-type blocks (:150-163)
        double accu0 = 0., accu1 = 0., entropy0 = 0., entropy1 = 0.;
        for (uint32_t c0 = 0; c0 < k_oth; c0 += kWave) {
            const Terms t = (c0 == 0) ? t0 : load_terms(c0 + lane);
            if (RNG == RNG_PHILOX) {
                // production arithmetic (DESIGN.md "Philox-mode definition"): per-lane leaves over chunks
                if (t.k != 0) {
                    const double inv = 1.0 / (t.mr_t + eps * Kd);
                    accu0 += t.k * (t.m_st + eps) * inv;
                    accu1 += t.k * (t.m_rt - t.k + eps) * inv;
                    entropy1 += (t.L1 + t.L2) - (t.L3 + t.L4);  // the leaf's share of dS = S1 - S0
                }
            } else {
                double A0 = 0., A1 = 0.;
                if (t.k != 0) {
                    A0 = t.k * (t.m_st + eps) / (t.mr_t + eps * Kd) / ideg;
                    A1 = t.k * (t.m_rt - t.k + eps) / (t.mr_t + eps * Kd) / ideg;
                }
                // the reference's serial sums in ascending block index: only blocks with k != 0 contribute (:152), so
                // walk the set bits of that mask (lanes past k_oth hold k = 0)
                for (unsigned long long nz = __ballot(t.k != 0); nz; nz &= nz - 1) {
                    const uint32_t jj = (uint32_t)__builtin_ctzll(nz);
                    accu0 += readlane(A0, jj);
                    accu1 += readlane(A1, jj);
                    entropy0 -= readlane(t.L1, jj);
                    entropy0 -= readlane(t.L2, jj);
                    entropy1 -= readlane(t.L3, jj);
                    entropy1 -= readlane(t.L4, jj);
                }
            }
        }
        if (RNG == RNG_PHILOX) {
            // scalar terms folded into leaves 0..7 / 0..3 with their signs, then three butterflies
            double d = entropy1;
            const bool neg_tail = (lane < 2) || (lane >= 6);
            if (lane < 8) d = d + (neg_tail ? -tail_lg : tail_lg);
            if (lane < 4) d = d + (lane < 2 ? -lq : lq);
            dS = butterfly_sum(d);
            phx_accu0 = (deg == 0) ? 1. : butterfly_sum_accu(accu0);
            phx_accu1 = (deg == 0) ? 1. : butterfly_sum_accu(accu1);
        } else {
        // (4) scalar tail in the reference's statement order
        entropy0 -= -readlane(tail_lg, 0);  // :164-168
        entropy0 -= -readlane(tail_lg, 1);
        entropy1 -= -readlane(tail_lg, 2);
        entropy1 -= -readlane(tail_lg, 3);
        entropy0 += -readlane(tail_lg, 4);  // :173-177
        entropy0 += -readlane(tail_lg, 5);
        entropy1 += -readlane(tail_lg, 6);
        entropy1 += -readlane(tail_lg, 7);
        entropy0 += readlane(lq, 0);  // :179-183
        entropy0 += readlane(lq, 1);
        entropy1 += readlane(lq, 2);
        entropy1 += readlane(lq, 3);
            c.accu_r = (deg == 0) ? 1. : accu1 / accu0;  // :185-189
            dS = entropy1 - entropy0;
        }
    }

    GSTAMP(c, 5);
    // ---- accept, metropolis_hasting.cc:47-61 ----
    bool accept;
    if (RNG == RNG_PHILOX) {
        if (cross)
            accept = false;
        else if (T == 0.)
            accept = dS < 0;
        else
            accept = u_acc * phx_accu0 < phx_accu1 * exp(-dS * (1.0 / T));  // u < exp(-dS/T) accu1/accu0
    } else if (T == 0.) {
        accept = dS < 0;
    } else {
        const double a = -1. / T * dS + log(c.accu_r);
        accept = (a > 0.) ? true : (draw_engine() < exp(a));
    }
    if (RNG == RNG_COMPAT && win_ok) engine.idx += 2 * (int)win_used;
    GSTAMP(c, 6);
    if (!accept) return false;

    // ---- apply_mcmc_moves, blockmodel.cc:461-503 ----
    if (c.nr[r] - 1 == 0) return false;  // :467-471: a move that empties a block is vetoed after the draw
    if (same) return true;               // n_r, eta, m updates cancel; entropy_ += 0
    wave_fence();                        // all lanes have read nr/mr/eta before lane 0 rewrites them
    // the counters are updated with LDS / memory atomics that return nothing: fire and forget, where a read-modify-write
    // in C++ would be six dependent round trips on lane 0
    if (lane == 0) {
        atomicSub(&c.nr[r], 1);
        atomicAdd(&c.nr[s], 1);
        if constexpr (EL) {
            atomicSub(&c.eta_l[r * c.D + deg], 1u);
            atomicAdd(&c.eta_l[s * c.D + deg], 1u);
        } else {
            atomicSub(&c.eta_g[r * c.D + deg], 1u);
            atomicAdd(&c.eta_g[s * c.D + deg], 1u);
        }
        atomicSub(&c.mr[r], (int)deg);
        atomicAdd(&c.mr[s], (int)deg);
        lab_set<W>(c, v, s);
    }
    {
        const uint32_t s_loc = s - own_base;
        for (uint32_t j = lane; j < k_oth; j += kWave) {  // :479-487 (mirror entries are the same cells here)
            const int k = c.hist[j];
            if (k != 0) {
                atomicSub(&Mx<W>(c, type_b, r_loc, j), k);
                atomicAdd(&Mx<W>(c, type_b, s_loc, j), k);
            }
        }
    }
    c.cum_dS += dS;  // :500
    wave_fence();
    GSTAMP(c, 7);
    return true;
}

// ------------------------------------------------------------------------------------------
// sweep kernel: metropolis_hasting::anneal (metropolis_hasting.cc:64-101), one wave per chain.
//
// CSR staging: the visit order of a sweep is known in advance (a keyed permutation, or the shuffled
// visit list), so everything that does not depend on the chain state is fetched ahead of use:
//   * per 64 positions, lane q evaluates node v_q and loads its row extent and own label at once;
//   * the neighbour ids of step q+2 and the neighbour labels of step q+1 are in flight while step q
//     computes (labels fetched early are patched when step q moves one of those neighbours).
// ------------------------------------------------------------------------------------------
template <int RNG, bool EL, bool W>
__global__ __launch_bounds__(kWave) void sweep_kernel(SweepParams p) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const uint32_t chain = blockIdx.x;
    if (chain >= p.n_chains) return;
    const int lane = lane_id();
    const uint32_t K = p.ka + p.kb;
    const uint32_t D = p.maxdeg + 1;
    const uint32_t S = p.kb | 1u;
    const uint32_t kmax = p.ka > p.kb ? p.ka : p.kb;

    // carve LDS
    ChainCtx c;
    unsigned char* cur = lds_raw;
    c.mq = (int32_t*)cur;
    if (!W) cur += sizeof(int32_t) * p.ka * S;
    c.mr = (int32_t*)cur;
    cur += sizeof(int32_t) * K;
    c.nr = (int32_t*)cur;
    cur += sizeof(int32_t) * K;
    c.hist = (int32_t*)cur;
    cur += sizeof(int32_t) * kmax;
    uint32_t* ids_lds = (uint32_t*)cur;  // 64 rows x 64 neighbour ids of the current chunk
    cur += sizeof(uint32_t) * kWave * kWave;
    uint32_t* eta_g = p.eta + (size_t)chain * K * D;
    c.eta_g = eta_g;
    c.eta_l = (uint32_t*)cur;
    c.eta_lds = EL;
    if (EL) cur += sizeof(uint32_t) * K * D;
    Mt engine{nullptr, 624}, gen{nullptr, 624};
    uint32_t* vl = nullptr;
    if (RNG == RNG_COMPAT) {
        engine.mt = (uint32_t*)cur;
        cur += sizeof(uint32_t) * 624;
        gen.mt = (uint32_t*)cur;
        cur += sizeof(uint32_t) * 624;
        engine.tm = (uint32_t*)cur;  // tempered outputs beside both states
        cur += sizeof(uint32_t) * 624;
        gen.tm = (uint32_t*)cur;
        cur += sizeof(uint32_t) * 624;
        if (p.vlist_in_lds) {
            vl = (uint32_t*)cur;
            cur += sizeof(uint32_t) * p.n;
        } else {
            vl = p.vlist + (size_t)chain * p.n;
        }
    }
    c.labels = p.labels + (size_t)chain * p.label_stride;
    c.labels16 = (uint16_t*)p.labels + (size_t)chain * p.label_stride;
    c.ka = p.ka;
    c.kb = p.kb;
    c.K = K;
    c.S = S;
    c.D = D;
    c.na = p.na;
    c.epsilon = p.epsilon;

    // load the chain's block state into LDS
    int32_t* m_g = p.m + (size_t)chain * p.ka * p.kb;
    int32_t* mr_g = p.m_r + (size_t)chain * K;
    int32_t* nr_g = p.n_r + (size_t)chain * K;
    c.mg = m_g;
    if (!W)
        for (uint32_t i = lane; i < p.ka * p.kb; i += kWave) c.mq[(i / p.kb) * S + (i % p.kb)] = m_g[i];
    for (uint32_t i = lane; i < K; i += kWave) {
        c.mr[i] = mr_g[i];
        c.nr[i] = nr_g[i];
    }
    if (EL)
        for (uint32_t i = lane; i < K * D; i += kWave) c.eta_l[i] = eta_g[i];
    ChainScalars* sc = p.scalars + chain;
    if (RNG == RNG_COMPAT) {
        const uint32_t* eg = p.mt_engine + (size_t)chain * 624;
        const uint32_t* gg = p.mt_gen + (size_t)chain * 624;
        for (uint32_t i = lane; i < 624; i += kWave) {
            engine.mt[i] = eg[i];
            gen.mt[i] = gg[i];
        }
        engine.idx = (int)sc->engine_idx;
        gen.idx = (int)sc->gen_idx;
        if (p.vlist_in_lds) {
            const uint32_t* vg = p.vlist + (size_t)chain * p.n;
            for (uint32_t i = lane; i < p.n; i += kWave) vl[i] = vg[i];
        }
    }
    c.cum_dS = sc->cum_dS;
    c.accu_r = sc->accu_r;
#ifdef BISBM_GSTAMPS
    for (int i = 0; i < 10; ++i) c.st_acc[i] = 0;
    __asm__ volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c.st_prev)::"memory");
#endif
    uint64_t sweeps_total = sc->sweeps_total;
    __syncthreads();
    if (RNG == RNG_COMPAT) {
        engine.retemper();
        gen.retemper();
    }

    Tables tab{p.lgamma_tab, p.lgamma_size, p.q_tab, p.q_stride, p.log_tab};
    const uint32_t chain_gid = chain_gid_of(p, chain);
    const uint64_t num_nodes = p.n;
    const uint64_t all_sweeps = p.duration / num_nodes;
    uint64_t accepted_steps = 0, u = 0, sweeps_done = 0;
    double entropy_min = INFINITY;  // :75
    double rate = 0.;
    bool stopped = false;
    for (uint64_t sweep = 0; sweep < all_sweeps; ++sweep) {
        TiledOrder order_a, order_b;  // Philox mode: all type-a nodes, then all type-b nodes, each class permuted
        if (RNG == RNG_COMPAT) {
            GSTAMP(c, 1);
            mt_shuffle(engine, vl, (uint32_t)num_nodes);  // :80
            GSTAMP(c, 0);
        } else {
            order_a.init(phx_draw(p.seed, chain_gid, PHX_SWEEP_KEY, 2 * sweeps_total), p.na);
            order_b.init(phx_draw(p.seed, chain_gid, PHX_SWEEP_KEY, 2 * sweeps_total + 1), p.nb);
        }
        const uint64_t current_step = num_nodes * sweep;  // :82
        for (uint64_t vi0 = 0; vi0 < num_nodes; vi0 += kWave) {
            // ---- chunk header: 64 positions of the visit order at once ----
            // lane q: node, row extent and own label of position vi0 + q.  A node is visited once per
            // sweep, so its own label cannot change before its step.
            const uint32_t cnt = (num_nodes - vi0) < (uint64_t)kWave ? (uint32_t)(num_nodes - vi0) : (uint32_t)kWave;
            uint32_t v_l = 0, beg_l = 0, deg_l = 0, r_l = 0;
            if ((uint32_t)lane < cnt) {
                const uint32_t pos = (uint32_t)(vi0 + lane);
                v_l = (RNG == RNG_COMPAT) ? vl[pos] : (pos < p.na ? order_a(pos) : p.na + order_b(pos - p.na));
                beg_l = p.rowptr[v_l];
                deg_l = p.rowptr[v_l + 1] - beg_l;
                r_l = lab_at<W>(c, v_l);
            }
            // CSR staging: the chunk's 64 adjacency rows (first 64 ids of each) go HBM -> LDS by
            // LDS-DMA, one 256-B row slot per instruction, all of them in flight together
            wave_fence();
            for (uint32_t q = 0; q < cnt; ++q) {
                const uint32_t b0 = readlane(beg_l, q), d0 = readlane(deg_l, q);
                if ((uint32_t)lane < d0)
                    __builtin_amdgcn_global_load_lds(p.col + b0 + lane, ids_lds + q * kWave, 4, 0, 0);
            }
            __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0): the DMA writes have landed before any ds_read
            wave_fence();

            // label pipeline: the labels of step q + kDepth are gathered while step q runs; moves made
            // in between are replayed from a small ring when the stage is consumed
            // The label load is unconditional (idle lanes read node 0): a register that receives loads must
            // not also be written by VALU code, or the compiler drains vmcnt before that write.
            auto gather = [&](uint32_t qq, uint32_t& nb, int& lab) {
                if (qq < cnt) {
                    const uint32_t d = readlane(deg_l, qq);
                    const uint32_t id = ids_lds[qq * kWave + lane];
                    const bool on = (uint32_t)lane < d;
                    nb = on ? id : 0xFFFFFFFFu;
                    lab = (int)lab_at<W>(c, on ? id : 0u);
                }
            };
            // three stage registers in fixed roles (the loop is unrolled by the depth): a register that is
            // the target of an in-flight load must never be copied, or the copy waits for the load
            uint32_t nb1 = 0xFFFFFFFFu, nb2 = 0xFFFFFFFFu, nb3 = 0xFFFFFFFFu;
            int lab1 = 0, lab2 = 0, lab3 = 0;
            gather(0, nb1, lab1);
            gather(1, nb2, lab2);
            gather(2, nb3, lab3);
            GSTAMP(c, 8);
            const uint32_t kNoMove = 0xFFFFFFFEu;
            uint32_t mv_v1 = kNoMove, mv_v2 = kNoMove, mv_v3 = kNoMove;  // moves of steps q-3, q-2, q-1
            int mv_s1 = 0, mv_s2 = 0, mv_s3 = 0;

            auto do_step = [&](uint32_t q, uint32_t& nbS, int& labS) {
                const uint32_t v = readlane(v_l, q), beg = readlane(beg_l, q), deg = readlane(deg_l, q);
                const uint32_t r = readlane(r_l, q);
                // this step's row: replay the moves made since its labels were requested
                const uint32_t nbC = nbS;
                int labC = labS;
                if (nbC == mv_v1) labC = mv_s1;
                if (nbC == mv_v2) labC = mv_s2;
                if (nbC == mv_v3) labC = mv_s3;
                auto prefetch = [&]() { gather(q + 3, nbS, labS); };  // the stage is free again: refill it
                const uint64_t vi = vi0 + q;
                const double T = temperature_of(p, current_step + vi);  // :84
                uint32_t s = r;
                prefetch();
                const bool ok = mh_step<RNG, EL, W>(p, tab, c, engine, gen, v, beg, deg, r, nbC, labC, T,
                                                 sweeps_total * num_nodes + vi, chain_gid, &s);
                mv_v1 = mv_v2;
                mv_s1 = mv_s2;
                mv_v2 = mv_v3;
                mv_s2 = mv_s3;
                mv_v3 = kNoMove;
                if (ok) {  // :85-91
                    ++accepted_steps;
                    if (c.cum_dS < entropy_min) {
                        entropy_min = c.cum_dS;
                        u = 0;
                    }
                    if (s != r) {
                        mv_v3 = v;
                        mv_s3 = (int)s;
                    }
                }
                if (T < 1.) ++u;  // :92-94
            };
            if (RNG == RNG_COMPAT) {
                // compat mode: ONE copy of the step in the loop.  Its step is ~1000 instructions long (serial sums,
                // libstdc++'s distributions), so the labels of step q+1, requested when step q starts, have long
                // landed when the stage registers are rotated; three unrolled copies of the body (90 KB of code) ran
                // a lone wave out of the instruction cache.
#pragma nounroll
                for (uint32_t q = 0; q < cnt; ++q) {
                    do_step(q, nb1, lab1);  // consumes stage 1, refills it with step q + 3
                    const uint32_t nb_t = nb1;
                    const int lab_t = lab1;
                    nb1 = nb2;
                    lab1 = lab2;
                    nb2 = nb3;
                    lab2 = lab3;
                    nb3 = nb_t;
                    lab3 = lab_t;
                }
            } else {
                for (uint32_t q = 0; q < cnt; q += 3) {
                    do_step(q, nb1, lab1);
                    if (q + 1 < cnt) do_step(q + 1, nb2, lab2);
                    if (q + 2 < cnt) do_step(q + 2, nb3, lab3);
                }
            }
        }
        ++sweeps_total;
        sweeps_done = sweep + 1;
        if (u >= p.steps_await) {  // :96-98
            rate = (double)accepted_steps / (double)((sweep + 1) * num_nodes);
            stopped = true;
            break;
        }
    }
    if (!stopped) rate = (double)accepted_steps / (double)p.duration;  // :100

    // store the chain back
    __syncthreads();
    if (!W)
        for (uint32_t i = lane; i < p.ka * p.kb; i += kWave) m_g[i] = c.mq[(i / p.kb) * S + (i % p.kb)];
    for (uint32_t i = lane; i < K; i += kWave) {
        mr_g[i] = c.mr[i];
        nr_g[i] = c.nr[i];
    }
    if (EL)
        for (uint32_t i = lane; i < K * D; i += kWave) eta_g[i] = c.eta_l[i];
    if (RNG == RNG_COMPAT) {
        uint32_t* eg = p.mt_engine + (size_t)chain * 624;
        uint32_t* gg = p.mt_gen + (size_t)chain * 624;
        for (uint32_t i = lane; i < 624; i += kWave) {
            eg[i] = engine.mt[i];
            gg[i] = gen.mt[i];
        }
        if (p.vlist_in_lds) {
            uint32_t* vg = p.vlist + (size_t)chain * p.n;
            for (uint32_t i = lane; i < p.n; i += kWave) vg[i] = vl[i];
        }
    }
#ifdef BISBM_GSTAMPS
    if (lane == 0)
        for (int i = 0; i < 10; ++i) atomicAdd(&g_generic_stamps[i], c.st_acc[i]);
#endif
    if (lane == 0) {
        sc->cum_dS = c.cum_dS;
        sc->accu_r = c.accu_r;
        sc->sweeps_total = sweeps_total;
        sc->last_rate = rate;
        sc->last_accepted = accepted_steps;
        sc->last_sweeps = sweeps_done;
        sc->engine_idx = (uint32_t)engine.idx;
        sc->gen_idx = (uint32_t)gen.idx;
    }
}

template __global__ void sweep_kernel<RNG_PHILOX, true, false>(SweepParams);
template __global__ void sweep_kernel<RNG_PHILOX, false, false>(SweepParams);
template __global__ void sweep_kernel<RNG_COMPAT, true, false>(SweepParams);
template __global__ void sweep_kernel<RNG_COMPAT, false, false>(SweepParams);
template __global__ void sweep_kernel<RNG_PHILOX, false, true>(SweepParams);  // wide mode: eta stays in HBM as well
template __global__ void sweep_kernel<RNG_COMPAT, false, true>(SweepParams);

// ------------------------------------------------------------------------------------------
// state build: init_bisbm (blockmodel.cc:682-688, compute_n_r :740-746, compute_m :702-714,
// compute_m_r :716-727, compute_eta_rk :729-738).  One 256-thread workgroup per chain; the
// a x b quadrant is histogrammed in LDS from the type-a rows of the CSR (the b rows are its
// transpose), eta / n_r with global atomics on the chain's own (zeroed) arrays.
// ------------------------------------------------------------------------------------------
template <bool W>
__global__ __launch_bounds__(256) void state_build_kernel(BuildParams p) {
    using LabelT = std::conditional_t<W, uint16_t, uint8_t>;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const uint32_t chain = blockIdx.x;
    const uint32_t K = p.ka + p.kb, D = p.maxdeg + 1;
    const LabelT* labels = (const LabelT*)p.labels + (size_t)chain * p.label_stride;
    int32_t* m_g = p.m + (size_t)chain * p.ka * p.kb;
    int32_t* mq = W ? m_g : (int32_t*)lds_raw;  // ka*kb: in LDS, or (wide mode) counted straight into the chain's array
    int32_t* mr_g = p.m_r + (size_t)chain * K;
    int32_t* nr_g = p.n_r + (size_t)chain * K;
    uint32_t* eta_g = p.eta + (size_t)chain * K * D;
    for (uint32_t i = threadIdx.x; i < p.ka * p.kb; i += blockDim.x) mq[i] = 0;
    for (uint32_t i = threadIdx.x; i < K; i += blockDim.x) {
        mr_g[i] = 0;
        nr_g[i] = 0;
    }
    for (uint32_t i = threadIdx.x; i < K * D; i += blockDim.x) eta_g[i] = 0;
    __syncthreads();
    for (uint32_t v = threadIdx.x; v < p.n; v += blockDim.x) {
        const uint32_t r = labels[v];
        const uint32_t beg = p.rowptr[v], end = p.rowptr[v + 1];
        atomicAdd(&nr_g[r], 1);
        atomicAdd(&eta_g[r * D + (end - beg)], 1u);
        if (v < p.na) {
            for (uint32_t e = beg; e < end; ++e) {
                const uint32_t t = labels[p.col[e]];
                atomicAdd(&mq[r * p.kb + (t - p.ka)], 1);
            }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < p.ka * p.kb; i += blockDim.x) {
        const int32_t x = mq[i];
        if (!W) m_g[i] = x;
        if (x) {
            atomicAdd(&mr_g[i / p.kb], x);
            atomicAdd(&mr_g[p.ka + i % p.kb], x);
        }
    }
}

// set_memberships: u32 host labels (staged on device) -> u8, one chain or broadcast to all
template <class LabelT>
__global__ void labels_broadcast_kernel(const uint32_t* src, LabelT* labels, size_t label_stride,
                                        uint32_t n, uint32_t first_chain, uint32_t n_chains) {
    const uint32_t chain = first_chain + blockIdx.y;
    if (blockIdx.y >= n_chains) return;
    LabelT* dst = labels + (size_t)chain * label_stride;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        dst[i] = (LabelT)src[i];
}

template <class LabelT>
__global__ void labels_widen_kernel(const LabelT* labels, uint32_t* dst, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        dst[i] = labels[i];
}

// wide -> narrow once merges have brought K down to 256 blocks or fewer: every chain's two-byte labels into a byte array
__global__ void labels_narrow_kernel(const uint16_t* src, uint8_t* dst, size_t label_stride, uint32_t n) {
    const uint16_t* s = src + (size_t)blockIdx.y * label_stride;
    uint8_t* d = dst + (size_t)blockIdx.y * label_stride;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) d[i] = (uint8_t)s[i];
}

// shuffle_bisbm, Philox definition: each type's labels are gathered through a keyed Feistel
// permutation (block sizes preserved).  grid = (tiles, chains).
template <class LabelT>
__global__ void shuffle_philox_kernel(ShuffleParams p) {
    const uint32_t chain = blockIdx.y;
    const LabelT* src = (const LabelT*)p.labels_old + (size_t)chain * p.label_stride;
    LabelT* dst = (LabelT*)p.labels + (size_t)chain * p.label_stride;
    const uint32_t gid = chain_gid_of(p, chain);
    const uint32_t epoch = p.scalars[chain].shuffle_epoch;
    Feistel fa, fb;
    fa.init(phx_draw(p.seed, gid, PHX_INIT_SHUFFLE, ((uint64_t)0 << 32) | epoch), p.na);
    fb.init(phx_draw(p.seed, gid, PHX_INIT_SHUFFLE, ((uint64_t)1 << 32) | epoch), p.nb);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += gridDim.x * blockDim.x)
        dst[i] = i < p.na ? src[fa(i)] : src[p.na + fb(i - p.na)];
}

__global__ void shuffle_epoch_bump_kernel(ChainScalars* sc, uint32_t n_chains) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_chains) sc[i].shuffle_epoch += 1;
}

// shuffle_bisbm, compat: two std::shuffle calls on `engine` (blockmodel.cc:673-674)
template <class LabelT>
__global__ __launch_bounds__(kWave) void shuffle_compat_kernel(ShuffleParams p) {
    __shared__ uint32_t mt_lds[624];
    const uint32_t chain = blockIdx.x;
    const int lane = lane_id();
    uint32_t* eg = p.mt_engine + (size_t)chain * 624;
    for (uint32_t i = lane; i < 624; i += kWave) mt_lds[i] = eg[i];
    __syncthreads();
    Mt engine{mt_lds, (int)p.scalars[chain].engine_idx};
    LabelT* labels = (LabelT*)p.labels + (size_t)chain * p.label_stride;
    mt_shuffle(engine, labels, p.na);
    mt_shuffle(engine, labels + p.na, p.nb);
    __syncthreads();
    for (uint32_t i = lane; i < 624; i += kWave) eg[i] = mt_lds[i];
    if (lane == 0) p.scalars[chain].engine_idx = (uint32_t)engine.idx;
}

// ------------------------------------------------------------------------------------------
// entropy(): the chain-dependent part of blockmodel.cc:753-787 (upper triangle of m, eta, m_r,
// log_q).  One wave per chain, butterfly reduction.  The chain-independent terms (degrees,
// multi-edge multiplicities, binomials) are added by the host.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWave) void entropy_kernel(EntropyParams p) {
    const uint32_t chain = blockIdx.x;
    const int lane = lane_id();
    const uint32_t K = p.ka + p.kb, D = p.maxdeg + 1;
    Tables tab{p.lgamma_tab, p.lgamma_size, p.q_tab, p.q_stride, p.log_tab};
    const int32_t* m_g = p.m + (size_t)chain * p.ka * p.kb;
    const int32_t* mr_g = p.m_r + (size_t)chain * K;
    const int32_t* nr_g = p.n_r + (size_t)chain * K;
    const uint32_t* eta_g = p.eta + (size_t)chain * K * D;
    double ent = 0.;
    // m_[r][s], s > r: the a x b quadrant covers every non-zero upper-triangle cell; the zero cells of
    // the same-type blocks contribute lgamma(1) = 0
    for (uint32_t i = lane; i < p.ka * p.kb; i += kWave) ent -= lgamma_fast(tab, (long long)m_g[i] + 1);
    for (uint32_t i = lane; i < K * D; i += kWave) ent -= lgamma_fast(tab, (long long)eta_g[i] + 1);
    for (uint32_t r = lane; r < K; r += kWave) {
        ent += lgamma_fast(tab, (long long)mr_g[r] + 1);
        ent += log_q<false>(tab, mr_g[r], nr_g[r]);
    }
    ent = butterfly_sum_levels_up(ent);
    if (lane == 0) p.out[chain] = ent;
}

// ------------------------------------------------------------------------------------------
// marginals: counts[v][block index within v's type] += #chains whose label of v is that block.
// Thread = node; labels are read coalesced along the node axis of the chain-major array.
// ------------------------------------------------------------------------------------------
template <class LabelT, bool IN_LDS>
__global__ __launch_bounds__(256) void marginals_kernel(MarginalParams p) {
    extern __shared__ __align__(16) uint32_t hist[];  // IN_LDS: one row of (kmax | 1) counters per thread
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    const LabelT* labels = (const LabelT*)p.labels;
    if constexpr (IN_LDS) {
        const uint32_t stride = p.kmax | 1u;
        uint32_t* row = hist + threadIdx.x * stride;
        for (uint32_t j = 0; j < p.kmax; ++j) row[j] = 0;
        if (v < p.n) {
            const uint32_t base = v < p.na ? 0 : p.ka;
            for (uint32_t c = 0; c < p.n_chains; ++c) row[(uint32_t)labels[(size_t)c * p.label_stride + v] - base] += 1;
            uint32_t* out = p.counts + (size_t)v * p.kmax;
            for (uint32_t j = 0; j < p.kmax; ++j)
                if (row[j]) out[j] += row[j];
        }
    } else if (v < p.n) {
        // many blocks (a row of counters per thread no longer fits the LDS): the thread owns row v of the histogram, so
        // it can count straight into it
        const uint32_t base = v < p.na ? 0 : p.ka;
        uint32_t* out = p.counts + (size_t)v * p.kmax;
        for (uint32_t c = 0; c < p.n_chains; ++c) out[(uint32_t)labels[(size_t)c * p.label_stride + v] - base] += 1;
    }
}

// MAP block of the nodes first .. first + rows - 1 from (a slice of) the pooled histogram: the most frequent block of the
// node's type, ties -> the lowest index (README.md:49-53: "the marginal estimate"), in the reference's numbering (type-b
// blocks offset by ka).  Thread = node; `counts` points at the row of node `first`.
__global__ __launch_bounds__(256) void marginal_map_kernel(const uint32_t* counts, uint32_t rows, uint32_t kmax, uint32_t first,
                                                           uint32_t n, uint32_t na, uint32_t ka, uint16_t* labels_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    const uint32_t v = first + i;
    uint32_t best = 0, arg = 0;
    if (v < n) {
        const uint32_t* row = counts + (size_t)i * kmax;
        for (uint32_t j = 0; j < kmax; ++j) {
            const uint32_t c = row[j];
            if (c > best) best = c, arg = j;
        }
        arg += v < na ? 0u : ka;
    }
    labels_out[i] = (uint16_t)arg;
}

// a[i] += b[i]: the owner of a node range adds another device's slice of the histogram (peer-copy pooling path)
__global__ __launch_bounds__(256) void counts_add_kernel(uint32_t* a, const uint32_t* b, size_t count) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < count) a[i] += b[i];
}

__global__ void log_q_probe_kernel(Tables tab, const int32_t* n, const int32_t* k, size_t count, double* out,
                                   int fast) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < count) {
        // the sweep kernels hand log(n) over from the host table; outside the table the probe uses the device log
        const int nn = n[i];
        const double logn = (nn > 0 && (uint64_t)nn < tab.lg_size) ? tab.logtab[nn] : log((double)(nn > 0 ? nn : 1));
        out[i] = fast ? log_q<true>(tab, nn, k[i], logn) : log_q<false>(tab, nn, k[i]);
    }
}

// ------------------------------------------------------------------------------------------
// launchers (called from the host runtime)
// ------------------------------------------------------------------------------------------
template <int RNG, bool EL, bool W = false>
static hipError_t launch_sweep_variant(const SweepParams& p, size_t lds_bytes, hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute((const void*)sweep_kernel<RNG, EL, W>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((sweep_kernel<RNG, EL, W>), dim3(p.n_chains), dim3(kWave), lds_bytes, stream, p);
    return hipGetLastError();
}

hipError_t launch_sweep(const SweepParams& p, int rng_mode, size_t lds_bytes, hipStream_t stream) {
    hipError_t e;
    if (p.wide)  // K > 256: two-byte labels, m and eta in HBM
        e = rng_mode == RNG_COMPAT ? launch_sweep_variant<RNG_COMPAT, false, true>(p, lds_bytes, stream)
                                   : launch_sweep_variant<RNG_PHILOX, false, true>(p, lds_bytes, stream);
    else if (rng_mode == RNG_COMPAT)
        e = p.eta_in_lds ? launch_sweep_variant<RNG_COMPAT, true>(p, lds_bytes, stream)
                         : launch_sweep_variant<RNG_COMPAT, false>(p, lds_bytes, stream);
    else
        e = p.eta_in_lds ? launch_sweep_variant<RNG_PHILOX, true>(p, lds_bytes, stream)
                         : launch_sweep_variant<RNG_PHILOX, false>(p, lds_bytes, stream);
    if (e != hipSuccess) return e;
#ifdef BISBM_GSTAMPS
    {
        (void)hipStreamSynchronize(stream);
        unsigned long long h[16] = {0};
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_generic_stamps), sizeof(h));
        const double steps = (double)p.n_chains * (double)(p.duration / p.n) * (double)p.n;
        static const char* names[9] = {"visit-list shuffle", "step entry / bookkeeping", "k_v histogram", "proposal",
                                       "terms + log_q", "serial sums + tail", "accept", "apply", "chunk header"};
        for (int i = 0; i < 9; ++i) fprintf(stderr, "[gstamps] %-26s %8.2f ticks/step\n", names[i], (double)h[i] / steps);
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_generic_stamps), z, sizeof(z));
    }
#endif
    return hipGetLastError();
}

hipError_t launch_state_build(const BuildParams& p, hipStream_t stream) {
    if (p.wide) {
        hipLaunchKernelGGL(state_build_kernel<true>, dim3(p.n_chains), dim3(256), 0, stream, p);
        return hipGetLastError();
    }
    const size_t lds = sizeof(int32_t) * p.ka * p.kb;
    hipError_t e = hipFuncSetAttribute((const void*)state_build_kernel<false>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(state_build_kernel<false>, dim3(p.n_chains), dim3(256), lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_labels_broadcast(const uint32_t* src, uint8_t* labels, bool wide, size_t label_stride, uint32_t n,
                                   uint32_t first_chain, uint32_t n_chains, hipStream_t stream) {
    const uint32_t tiles = (n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024;
    if (wide)
        hipLaunchKernelGGL(labels_broadcast_kernel<uint16_t>, dim3(tiles ? tiles : 1, n_chains), dim3(256), 0, stream, src,
                           (uint16_t*)labels, label_stride, n, first_chain, n_chains);
    else
        hipLaunchKernelGGL(labels_broadcast_kernel<uint8_t>, dim3(tiles ? tiles : 1, n_chains), dim3(256), 0, stream, src,
                           labels, label_stride, n, first_chain, n_chains);
    return hipGetLastError();
}

hipError_t launch_labels_narrow(const uint8_t* wide_labels, uint8_t* labels, size_t label_stride, uint32_t n, uint32_t n_chains,
                                hipStream_t stream) {
    const uint32_t tiles = (n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024;
    hipLaunchKernelGGL(labels_narrow_kernel, dim3(tiles ? tiles : 1, n_chains), dim3(256), 0, stream, (const uint16_t*)wide_labels,
                       labels, label_stride, n);
    return hipGetLastError();
}

// ---- relabelling after block merges (apply_block_moves, blockmodel.cc:567-611) ----
// first[chain][l] = lowest node id whose label maps to l under map1 (the order in which the reference compacts)
// Maps are [chain][map_len] tables of LabelT (map_len = the block count before the call, rounded up to 256), `first` is
// [chain][map_len].  The tables are staged in dynamic LDS (map_len entries each).
template <class LabelT>
__global__ void merge_first_kernel(const LabelT* labels, size_t label_stride, uint32_t n, uint32_t map_len, const LabelT* map1,
                                   uint32_t* first) {
    extern __shared__ __align__(16) unsigned char merge_lds[];
    uint32_t* lfirst = (uint32_t*)merge_lds;
    LabelT* lmap = (LabelT*)(lfirst + map_len);
    const uint32_t chain = blockIdx.y;
    for (uint32_t i = threadIdx.x; i < map_len; i += blockDim.x) {
        lfirst[i] = 0xffffffffu;
        lmap[i] = map1[(size_t)chain * map_len + i];
    }
    __syncthreads();
    const LabelT* lab = labels + (size_t)chain * label_stride;
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x)
        atomicMin(&lfirst[lmap[lab[v]]], v);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < map_len; i += blockDim.x)
        if (lfirst[i] != 0xffffffffu) atomicMin(&first[(size_t)chain * map_len + i], lfirst[i]);
}

template <class LabelT>
__global__ void merge_relabel_kernel(LabelT* labels, size_t label_stride, uint32_t n, uint32_t map_len, const LabelT* fmap) {
    extern __shared__ __align__(16) unsigned char merge_lds[];
    LabelT* lmap = (LabelT*)merge_lds;
    const uint32_t chain = blockIdx.y;
    for (uint32_t i = threadIdx.x; i < map_len; i += blockDim.x) lmap[i] = fmap[(size_t)chain * map_len + i];
    __syncthreads();
    LabelT* lab = labels + (size_t)chain * label_stride;
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x) lab[v] = lmap[lab[v]];
}

template <class LabelT>
static hipError_t launch_merge_first_t(const LabelT* labels, size_t label_stride, uint32_t n, uint32_t n_chains, uint32_t map_len,
                                       const LabelT* map1, uint32_t* first, hipStream_t stream) {
    const uint32_t tiles = std::min<uint32_t>((n + 256 * 16 - 1) / (256 * 16), 1024u);
    const size_t lds = (sizeof(uint32_t) + sizeof(LabelT)) * map_len;
    hipError_t e = hipFuncSetAttribute((const void*)merge_first_kernel<LabelT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(merge_first_kernel<LabelT>, dim3(tiles ? tiles : 1, n_chains), dim3(256), lds, stream, labels, label_stride, n,
                       map_len, map1, first);
    return hipGetLastError();
}

hipError_t launch_merge_first(const uint8_t* labels, bool wide, size_t label_stride, uint32_t n, uint32_t n_chains, uint32_t map_len,
                              const void* map1, uint32_t* first, hipStream_t stream) {
    return wide ? launch_merge_first_t<uint16_t>((const uint16_t*)labels, label_stride, n, n_chains, map_len, (const uint16_t*)map1,
                                                 first, stream)
                : launch_merge_first_t<uint8_t>(labels, label_stride, n, n_chains, map_len, (const uint8_t*)map1, first, stream);
}

template <class LabelT>
static hipError_t launch_merge_relabel_t(LabelT* labels, size_t label_stride, uint32_t n, uint32_t n_chains, uint32_t map_len,
                                         const LabelT* fmap, hipStream_t stream) {
    const uint32_t tiles = std::min<uint32_t>((n + 256 * 16 - 1) / (256 * 16), 1024u);
    const size_t lds = sizeof(LabelT) * map_len;
    hipError_t e = hipFuncSetAttribute((const void*)merge_relabel_kernel<LabelT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(merge_relabel_kernel<LabelT>, dim3(tiles ? tiles : 1, n_chains), dim3(256), lds, stream, labels, label_stride, n,
                       map_len, fmap);
    return hipGetLastError();
}

hipError_t launch_merge_relabel(uint8_t* labels, bool wide, size_t label_stride, uint32_t n, uint32_t n_chains, uint32_t map_len,
                                const void* fmap, hipStream_t stream) {
    return wide ? launch_merge_relabel_t<uint16_t>((uint16_t*)labels, label_stride, n, n_chains, map_len, (const uint16_t*)fmap, stream)
                : launch_merge_relabel_t<uint8_t>(labels, label_stride, n, n_chains, map_len, (const uint8_t*)fmap, stream);
}

// ------------------------------------------------------------------------------------------
// agg_split, blockmodel.cc:505-565 (+ compute_dS(split) :374-424, apply_split_moves :428-459), intended semantics:
// the position of a node in a block's split vector is its rank within the block.
//   split_rank_kernel   rank of every node of the type within its block, ascending node id (one wave per chain)
//   split_eval_kernel   one workgroup per (trial, chain): for every block of the type with more than one node, the
//                       edge counts k[t] and the degree sum of the nodes the trial's cut marks (LDS histogram);
//                       the K-scale dS arithmetic and the choice stay on the host, as in the merges
//   split_apply_kernel  the chosen cut: marked nodes get the new label (type a: label KA, after every label >= KA
//                       moved up by one, :434-443; type b: label K)
// A cut marks node of rank i of block b in trial j iff
//   compat: bit (offset of b + i) of the host-shuffled splitter_ of that trial (std::shuffle on the chain's engine),
//   Philox: feistel(key(b, j), n_b)(i) >= floor(n_b / 2), key = Philox(seed, chain, PHX_SPLIT, epoch << 32 | b << 16 | j).
// ------------------------------------------------------------------------------------------
// LabelT: bytes, or two bytes while the handle is wide (more than 256 blocks).  HBM: the per-label counters live in p.rank_base
// ([chain][K], zeroed by the host) instead of 256 LDS words.
template <class LabelT, bool HBM>
__global__ __launch_bounds__(kWave) void split_rank_kernel(SplitParams p) {
    __shared__ uint32_t base_lds[256];
    const uint32_t chain = blockIdx.x, lane = threadIdx.x;
    uint32_t* const base = HBM ? p.rank_base + (size_t)chain * (p.ka + p.kb) : base_lds;
    if (!HBM) {
        for (uint32_t i = lane; i < 256; i += kWave) base_lds[i] = 0;
        __syncthreads();
    }
    const uint32_t v_lo = p.type ? p.na : 0u, n_type = p.type ? p.n - p.na : p.na;
    const LabelT* lab = (const LabelT*)p.labels + (size_t)chain * p.label_stride + v_lo;
    uint32_t* rank = p.rank + (size_t)chain * n_type;
    for (uint32_t t0 = 0; t0 < n_type; t0 += kWave) {
        const uint32_t i = t0 + lane;
        const bool active = i < n_type;
        const uint32_t L = active ? (uint32_t)lab[i] : 0xffffffffu;
        unsigned long long remaining = __ballot(active);
        while (remaining) {  // one round per distinct label among the 64 nodes
            const int leader = __ffsll((long long)remaining) - 1;
            const uint32_t cur = (uint32_t)__shfl((int)L, leader, kWave);
            const unsigned long long same = __ballot(active && L == cur);
            uint32_t before = 0;  // nodes of this label seen so far (the atomic returns it: no second pass over the counter)
            if ((int)lane == leader) before = atomicAdd(&base[cur], (uint32_t)__popcll(same));
            before = (uint32_t)__shfl((int)before, leader, kWave);
            if (active && L == cur) rank[i] = before + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
            remaining &= ~same;
        }
    }
}

// the cut of (block, trial): marks by rank
struct SplitCut {
    Feistel f;
    uint32_t half;  // floor(n_b / 2): ranks mapped below it stay
};

// HBM = false: the counts of one (trial, chain) are gathered in LDS ([k_type][k_oth] + per-block helpers) and written out.
// HBM = true (wide handles: the table does not fit the LDS): counted straight into the (zeroed) output with global atomics;
// a node's cut is made on the spot from its block's key instead of being kept per block.
template <class LabelT, bool HBM>
__global__ __launch_bounds__(256) void split_eval_kernel(SplitParams p) {
    extern __shared__ int32_t sp_lds[];
    const uint32_t trial = p.trial0 + blockIdx.x, chain = blockIdx.y;
    const uint32_t K = p.ka + p.kb;
    const uint32_t b_lo = p.type ? p.ka : 0u, k_type = p.type ? p.kb : p.ka;
    const uint32_t t_lo = p.type ? 0u : p.ka, k_oth = p.type ? p.ka : p.kb;
    const uint32_t v_lo = p.type ? p.na : 0u, n_type = p.type ? p.n - p.na : p.na;
    const uint32_t epoch = p.scalars[chain].split_epoch;
    const LabelT* lab = (const LabelT*)p.labels + (size_t)chain * p.label_stride;
    const uint32_t* rank = p.rank + (size_t)chain * n_type;
    const uint32_t* bits = p.bits ? p.bits + ((size_t)chain * p.nm + trial) * p.bit_words : nullptr;
    int32_t* ok = p.out_k + ((size_t)chain * p.n_trials + blockIdx.x) * k_type * k_oth;
    int32_t* od = p.out_deg + ((size_t)chain * p.n_trials + blockIdx.x) * k_type;
    if constexpr (HBM) {
        const int32_t* n_r = p.n_r + (size_t)chain * K + b_lo;
        const uint32_t* off = p.block_off + (size_t)chain * k_type;  // first position of every block in the cut bits (compat)
        for (uint32_t i = threadIdx.x; i < n_type; i += blockDim.x) {
            const uint32_t v = v_lo + i, r = (uint32_t)lab[v] - b_lo, nb = (uint32_t)n_r[r];
            if (nb <= 1) continue;
            const uint32_t rk = rank[i];
            bool marked;
            if (bits) {
                const uint32_t pos = off[r] + rk;
                marked = (bits[pos >> 5] >> (pos & 31u)) & 1u;
            } else {
                const uint64_t idx = ((uint64_t)epoch << 32) | ((uint64_t)(b_lo + r) << 16) | (uint64_t)trial;
                Feistel f;
                f.init(phx_draw(p.seed, chain_gid_of(p, chain), PHX_SPLIT, idx), nb);
                marked = f(rk) >= nb / 2;
            }
            if (!marked) continue;
            const uint32_t e0 = p.rowptr[v], e1 = p.rowptr[v + 1];
            for (uint32_t e = e0; e < e1; ++e) atomicAdd(&ok[(size_t)r * k_oth + ((uint32_t)lab[p.col[e]] - t_lo)], 1);
            atomicAdd(&od[r], (int32_t)(e1 - e0));
        }
        return;
    }
    int32_t* const k = sp_lds;                                  // [k_type][k_oth]
    int32_t* const deg = k + k_type * k_oth;                    // [k_type]
    uint32_t* const nr = (uint32_t*)(deg + k_type);             // [k_type]
    uint32_t* const off = nr + k_type;                          // [k_type]
    SplitCut* const cut = (SplitCut*)(off + k_type + (k_type & 1u));  // [k_type]
    for (uint32_t i = threadIdx.x; i < k_type * k_oth + k_type; i += blockDim.x) k[i] = 0;
    for (uint32_t b = threadIdx.x; b < k_type; b += blockDim.x) {
        const uint32_t nb = (uint32_t)p.n_r[(size_t)chain * K + b_lo + b];
        nr[b] = nb;
        if (!p.bits && nb > 1) {
            const uint64_t idx = ((uint64_t)epoch << 32) | ((uint64_t)(b_lo + b) << 16) | (uint64_t)trial;
            cut[b].f.init(phx_draw(p.seed, chain_gid_of(p, chain), PHX_SPLIT, idx), nb);
            cut[b].half = nb / 2;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (uint32_t b = 0; b < k_type; ++b) {
            off[b] = acc;
            acc += nr[b];
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_type; i += blockDim.x) {
        const uint32_t v = v_lo + i, r = (uint32_t)lab[v] - b_lo, nb = nr[r];
        if (nb <= 1) continue;
        const uint32_t rk = rank[i];
        bool marked;
        if (bits) {
            const uint32_t pos = off[r] + rk;
            marked = (bits[pos >> 5] >> (pos & 31u)) & 1u;
        } else {
            marked = cut[r].f(rk) >= cut[r].half;
        }
        if (!marked) continue;
        const uint32_t e0 = p.rowptr[v], e1 = p.rowptr[v + 1];
        for (uint32_t e = e0; e < e1; ++e) atomicAdd(&k[r * k_oth + ((uint32_t)lab[p.col[e]] - t_lo)], 1);
        atomicAdd(&deg[r], (int32_t)(e1 - e0));
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < k_type * k_oth; i += blockDim.x) ok[i] = k[i];
    for (uint32_t i = threadIdx.x; i < k_type; i += blockDim.x) od[i] = deg[i];
}

template <class LabelT>
__global__ __launch_bounds__(256) void split_apply_kernel(SplitParams p) {
    __shared__ SplitCut cut;
    __shared__ uint32_t s_off, s_nb;
    const uint32_t chain = blockIdx.y;
    const uint32_t K = p.ka + p.kb;
    const uint32_t b_lo = p.type ? p.ka : 0u, k_type = p.type ? p.kb : p.ka;
    const uint32_t v_lo = p.type ? p.na : 0u, n_type = p.type ? p.n - p.na : p.na;
    const uint32_t block = p.chosen[2 * chain], trial = p.chosen[2 * chain + 1];
    if (threadIdx.x == 0) {
        const int32_t* n_r = p.n_r + (size_t)chain * K + b_lo;
        uint32_t acc = 0;
        for (uint32_t b = 0; b < block && b < k_type; ++b) acc += (uint32_t)n_r[b];
        s_off = acc;
        s_nb = (uint32_t)n_r[block];
        if (!p.bits) {
            const uint64_t idx = ((uint64_t)p.scalars[chain].split_epoch << 32) | ((uint64_t)(b_lo + block) << 16) | (uint64_t)trial;
            cut.f.init(phx_draw(p.seed, chain_gid_of(p, chain), PHX_SPLIT, idx), s_nb);
            cut.half = s_nb / 2;
        }
    }
    __syncthreads();
    LabelT* lab = (LabelT*)p.labels + (size_t)chain * p.label_stride;
    const uint32_t* rank = p.rank + (size_t)chain * n_type;
    const uint32_t* bits = p.bits ? p.bits + ((size_t)chain * p.nm + trial) * p.bit_words : nullptr;
    const uint32_t new_label = p.type ? K : p.ka;
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < p.n; v += gridDim.x * blockDim.x) {
        uint32_t L = lab[v];
        const bool own = p.type ? v >= p.na : v < p.na;
        if (!own) {
            if (!p.type) lab[v] = (LabelT)(L + 1);  // type-a split: the type-b labels move up by one (:434-443)
            continue;
        }
        if (L != b_lo + block) continue;
        const uint32_t rk = rank[v - v_lo];
        bool marked;
        if (bits) {
            const uint32_t pos = s_off + rk;
            marked = (bits[pos >> 5] >> (pos & 31u)) & 1u;
        } else {
            marked = cut.f(rk) >= cut.half;
        }
        if (marked) lab[v] = (LabelT)new_label;
    }
}

hipError_t launch_split_rank(const SplitParams& p, hipStream_t stream) {
    if (p.wide)
        hipLaunchKernelGGL((split_rank_kernel<uint16_t, true>), dim3(p.n_chains), dim3(kWave), 0, stream, p);
    else
        hipLaunchKernelGGL((split_rank_kernel<uint8_t, false>), dim3(p.n_chains), dim3(kWave), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_split_eval(const SplitParams& p, hipStream_t stream) {
    if (p.wide) {  // (out_k / out_deg zeroed by the caller)
        hipLaunchKernelGGL((split_eval_kernel<uint16_t, true>), dim3(p.n_trials, p.n_chains), dim3(256), 0, stream, p);
        return hipGetLastError();
    }
    const uint32_t k_type = p.type ? p.kb : p.ka, k_oth = p.type ? p.ka : p.kb;
    const size_t lds = sizeof(int32_t) * ((size_t)k_type * k_oth + 3 * (size_t)k_type + 2) + sizeof(SplitCut) * k_type;
    hipError_t e = hipFuncSetAttribute((const void*)split_eval_kernel<uint8_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((split_eval_kernel<uint8_t, false>), dim3(p.n_trials, p.n_chains), dim3(256), lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_split_apply(const SplitParams& p, hipStream_t stream) {
    const uint32_t tiles = std::min<uint32_t>((p.n + 256 * 16 - 1) / (256 * 16), 1024u);
    if (p.wide)
        hipLaunchKernelGGL(split_apply_kernel<uint16_t>, dim3(tiles ? tiles : 1, p.n_chains), dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL(split_apply_kernel<uint8_t>, dim3(tiles ? tiles : 1, p.n_chains), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// byte labels -> two-byte labels for every chain (a split takes a handle past 256 blocks)
__global__ void labels_to_wide_kernel(const uint8_t* src, uint16_t* dst, size_t label_stride, uint32_t n) {
    const uint8_t* s = src + (size_t)blockIdx.y * label_stride;
    uint16_t* d = dst + (size_t)blockIdx.y * label_stride;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) d[i] = (uint16_t)s[i];
}

hipError_t launch_labels_to_wide(const uint8_t* labels, uint8_t* wide_labels, size_t label_stride, uint32_t n, uint32_t n_chains,
                                 hipStream_t stream) {
    const uint32_t tiles = (n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024;
    hipLaunchKernelGGL(labels_to_wide_kernel, dim3(tiles ? tiles : 1, n_chains), dim3(256), 0, stream, labels, (uint16_t*)wide_labels,
                       label_stride, n);
    return hipGetLastError();
}

// one chain's labels (`labels` points at the chain's first label) -> u32
hipError_t launch_labels_widen(const uint8_t* labels, bool wide, uint32_t* dst, uint32_t n, hipStream_t stream) {
    const uint32_t tiles = (n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024;
    if (wide)
        hipLaunchKernelGGL(labels_widen_kernel<uint16_t>, dim3(tiles ? tiles : 1), dim3(256), 0, stream, (const uint16_t*)labels, dst, n);
    else
        hipLaunchKernelGGL(labels_widen_kernel<uint8_t>, dim3(tiles ? tiles : 1), dim3(256), 0, stream, labels, dst, n);
    return hipGetLastError();
}

hipError_t launch_shuffle(const ShuffleParams& p, int rng_mode, hipStream_t stream) {
    if (rng_mode == RNG_COMPAT) {
        if (p.wide)
            hipLaunchKernelGGL(shuffle_compat_kernel<uint16_t>, dim3(p.n_chains), dim3(kWave), 0, stream, p);
        else
            hipLaunchKernelGGL(shuffle_compat_kernel<uint8_t>, dim3(p.n_chains), dim3(kWave), 0, stream, p);
    } else {
        const uint32_t tiles = (p.n + 255) / 256 < 1024 ? (p.n + 255) / 256 : 1024;
        if (p.wide)
            hipLaunchKernelGGL(shuffle_philox_kernel<uint16_t>, dim3(tiles ? tiles : 1, p.n_chains), dim3(256), 0, stream, p);
        else
            hipLaunchKernelGGL(shuffle_philox_kernel<uint8_t>, dim3(tiles ? tiles : 1, p.n_chains), dim3(256), 0, stream, p);
        hipLaunchKernelGGL(shuffle_epoch_bump_kernel, dim3((p.n_chains + 255) / 256), dim3(256), 0, stream,
                           p.scalars, p.n_chains);
    }
    return hipGetLastError();
}

__global__ void sum_from_entropy_kernel(ChainScalars* scalars, const double* before, const double* after, uint32_t n_chains) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n_chains) scalars[c].cum_dS += after[c] - before[c];
}

hipError_t launch_sum_from_entropy(ChainScalars* scalars, const double* before, const double* after, uint32_t n_chains, hipStream_t stream) {
    hipLaunchKernelGGL(sum_from_entropy_kernel, dim3((n_chains + 255) / 256), dim3(256), 0, stream, scalars, before, after, n_chains);
    return hipGetLastError();
}

hipError_t launch_entropy(const EntropyParams& p, hipStream_t stream) {
    hipLaunchKernelGGL(entropy_kernel, dim3(p.n_chains), dim3(kWave), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_marginals(const MarginalParams& p, hipStream_t stream) {
    const size_t lds = sizeof(uint32_t) * 256 * (p.kmax | 1u);
    const dim3 grid((p.n + 255) / 256), block(256);
    if (lds > 160 * 1024) {  // more than ~159 blocks of a type: a row of counters per thread does not fit the LDS, count in HBM
        if (p.wide)
            hipLaunchKernelGGL((marginals_kernel<uint16_t, false>), grid, block, 0, stream, p);
        else
            hipLaunchKernelGGL((marginals_kernel<uint8_t, false>), grid, block, 0, stream, p);
        return hipGetLastError();
    }
    if (p.wide) {
        hipError_t e = hipFuncSetAttribute((const void*)marginals_kernel<uint16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((marginals_kernel<uint16_t, true>), grid, block, lds, stream, p);
        return hipGetLastError();
    }
    hipError_t e = hipFuncSetAttribute((const void*)marginals_kernel<uint8_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((marginals_kernel<uint8_t, true>), grid, block, lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_log_q_probe(const Tables& tab, const int32_t* n, const int32_t* k, size_t count, double* out,
                              int fast, hipStream_t stream) {
    hipLaunchKernelGGL(log_q_probe_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, tab, n, k,
                       count, out, fast);
    return hipGetLastError();
}

hipError_t launch_marginal_map(const uint32_t* counts, uint32_t rows, uint32_t kmax, uint32_t first, uint32_t n, uint32_t na,
                               uint32_t ka, uint16_t* labels_out, hipStream_t stream) {
    if (rows == 0) return hipSuccess;
    hipLaunchKernelGGL(marginal_map_kernel, dim3((rows + 255) / 256), dim3(256), 0, stream, counts, rows, kmax, first, n, na, ka, labels_out);
    return hipGetLastError();
}

hipError_t launch_counts_add(uint32_t* a, const uint32_t* b, size_t count, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(counts_add_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, a, b, count);
    return hipGetLastError();
}

}  // namespace bisbm
