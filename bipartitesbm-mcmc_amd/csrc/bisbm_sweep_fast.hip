// bisbm_sweep_fast.hip -- the production sweep kernel: Philox mode, both block counts <= 64.
//
// metropolis_hasting::anneal / step / transition_ratio (metropolis_hasting.cc:42-192),
// single_vertex_change (blockmodel.cc:613-637) and apply_mcmc_moves (blockmodel.cc:461-503) for one
// chain per wavefront, persistent over all sweeps of the call.  Every number it produces is the one
// the generic kernel (bisbm_kernels.hip, sweep_kernel<RNG_PHILOX>) produces; the tests run both.
//
// What makes it the fast one:
//   * state on chip: the a x b quadrant of m (odd row stride: rows and columns conflict-free) and
//     eta in LDS; m_r and n_r mirrored in registers (lane i <-> block i of each type), so their
//     wave-uniform reads are v_readlane, not LDS round trips;
//   * CSR staged through LDS: per 64 positions of the visit order the 64 adjacency rows are pulled
//     HBM -> LDS by LDS-DMA, all in flight together; neighbour labels are gathered three steps ahead
//     into fixed stage registers (loop unrolled by the depth: a register that receives loads is never
//     copied or written by VALU code, which would drain vmcnt), moves made meanwhile are replayed
//     from a three-entry ring when a stage is consumed;
//   * k_v from wave ballots (no LDS atomics); proposal CDF by a DPP scan; dS by DPP butterflies;
//   * the four uniforms of a step come from one Philox evaluation per 64 steps per lane;
//   * all table gathers of a step are issued together, the label prefetch right after them (vmcnt
//     retires in order), and the log_q evaluation (four values in four lanes) runs under their latency;
//   * no generic pointers (flat loads wait on vmcnt and lgkmcnt), no workgroup barriers on the step path;
//   * apply_mcmc_moves writes values it already holds: no read-modify-write round trips.
// Diagnostic hooks (BISBM_STAMPS) are compiled out of the product build.
#include "bisbm_kernels.hpp"

#include <cstdio>

namespace bisbm {

#ifdef BISBM_STAMPS
__device__ unsigned long long g_fast_stamps[16];
#define FSTAMP(i)                                                                                   \
    do {                                                                                            \
        unsigned long long now_;                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        __asm__ volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");            \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        st_acc[i] += now_ - st_prev;                                                                \
        st_prev = now_;                                                                             \
    } while (0)
#else
#define FSTAMP(i) \
    do {          \
    } while (0)
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
    return *(const double*)((const char*)base + (uint32_t)(idx << 3));  // tables are < 2^29 entries (host check)
}

template <bool EL, bool CT>
__global__ __launch_bounds__(kWave) void sweep_fast_kernel(SweepParams p) {
    extern __shared__ __align__(16) uint32_t lds32[];
    const uint32_t chain = blockIdx.x;
    if (chain >= p.n_chains) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t ka = p.ka, kb = p.kb, K = ka + kb, na = p.na;
    const uint32_t D = p.maxdeg + 1, S = kb | 1u;
    // LDS layout, dword offsets
    const uint32_t o_mq = 0, o_mr = ka * S, o_nr = o_mr + K, o_eta = o_nr + K;
    const uint32_t o_hist = o_eta + (EL ? K * D : 0u);
    const uint32_t o_ids = o_hist + kWave;
    int32_t* const mq = (int32_t*)(lds32 + o_mq);
    int32_t* const mr_l = (int32_t*)(lds32 + o_mr);
    int32_t* const nr_l = (int32_t*)(lds32 + o_nr);
    uint32_t* const eta_l = lds32 + o_eta;
    uint32_t* const ids = lds32 + o_ids;
    int32_t* const hist = (int32_t*)(lds32 + o_hist);  // k_v of the node being moved, one counter per opposite-type block

    uint8_t* const labels = p.labels + (size_t)chain * p.label_stride;
    int32_t* const m_g = p.m + (size_t)chain * ka * kb;
    int32_t* const mr_g = p.m_r + (size_t)chain * K;
    int32_t* const nr_g = p.n_r + (size_t)chain * K;
    uint32_t* const eta_g = p.eta + (size_t)chain * K * D;
    ChainScalars* const sc = p.scalars + chain;
    const Tables tab{p.lgamma_tab, p.lgamma_size, p.q_tab, p.q_stride, p.log_tab};

    // chain state -> LDS / registers
    for (uint32_t i = lane; i < ka * kb; i += kWave) mq[(i / kb) * S + (i % kb)] = m_g[i];
    for (uint32_t i = lane; i < K; i += kWave) {
        mr_l[i] = mr_g[i];
        nr_l[i] = nr_g[i];
    }
    if (EL)
        for (uint32_t i = lane; i < K * D; i += kWave) eta_l[i] = eta_g[i];
    __syncthreads();
    int mrA = lane < ka ? mr_l[lane] : 0, nrA = lane < ka ? nr_l[lane] : 0;
    int mrB = lane < kb ? mr_l[ka + lane] : 0, nrB = lane < kb ? nr_l[ka + lane] : 0;
    double cum_dS = sc->cum_dS;
    uint64_t sweeps_total = sc->sweeps_total;

    auto eta_rd = [&](uint32_t idx) -> uint32_t { return EL ? eta_l[idx] : eta_g[idx]; };
    auto eta_wr = [&](uint32_t idx, uint32_t val) {
        if (EL)
            eta_l[idx] = val;
        else
            eta_g[idx] = val;
    };

    const double eps = p.epsilon;
    const double Kd = (double)K;
    const double epsK = eps * Kd;
    const uint32_t chain_gid = p.first_chain_id + chain;
    const uint32_t n = p.n;
    const uint64_t all_sweeps = p.duration / n;
    const double T_const = (double)p.kw0;  // CT: constant schedule (metropolis_hasting.cc:25-28)
    uint64_t accepted_steps = 0, u_cnt = 0, sweeps_done = 0;
    double entropy_min = INFINITY;  // metropolis_hasting.cc:75
    double rate = 0.;
    bool stopped = false;
#ifdef BISBM_STAMPS
    unsigned long long st_prev, st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    __asm__ volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
#endif

    for (uint64_t sweep = 0; sweep < all_sweeps; ++sweep) {
        Feistel order;
        order.init(phx_draw(p.seed, chain_gid, PHX_SWEEP_KEY, sweeps_total), n);
        const uint64_t sweep_step0 = (uint64_t)n * sweep;  // metropolis_hasting.cc:82
        for (uint32_t vi0 = 0; vi0 < n; vi0 += kWave) {
            // ---- chunk header: 64 positions of the visit order at once (lane q <-> position vi0+q) ----
            const uint32_t cnt = (n - vi0) < (uint32_t)kWave ? (n - vi0) : (uint32_t)kWave;
            uint32_t v_l = 0, beg_l = 0, deg_l = 0, r_l = 0;
            double ud_idx = 0., ud_R = 0., ud_tgt = 0., ud_acc = 0.;
            if (lane < cnt) {
                v_l = order(vi0 + lane);
                beg_l = p.rowptr[v_l];
                deg_l = p.rowptr[v_l + 1] - beg_l;
                r_l = labels[v_l];  // a node is visited once per sweep: its own label is stable until its step
                const uint64_t gs = sweeps_total * (uint64_t)n + vi0 + lane;
                const U4 A = phx_draw(p.seed, chain_gid, PHX_STEP_A, gs);
                const U4 B = phx_draw(p.seed, chain_gid, PHX_STEP_B, gs);
                ud_idx = u53(A.x, A.y);
                ud_R = u53(A.z, A.w);
                ud_tgt = u53(B.x, B.y);
                ud_acc = u53(B.z, B.w);
            }
            // CSR staging: 64 rows (first 64 ids each) HBM -> LDS by LDS-DMA, one 256-B slot per instruction
            wfence();
            for (uint32_t q = 0; q < cnt; ++q) {
                const uint32_t b0 = readlane(beg_l, q), d0 = readlane(deg_l, q);
                if (lane < d0) __builtin_amdgcn_global_load_lds(p.col + b0 + lane, ids + q * kWave, 4, 0, 0);
            }
            __builtin_amdgcn_s_waitcnt(0);  // the DMA writes have landed before any ds_read of ids
            wfence();

            // label pipeline (depth 3).  The label load is unconditional (idle lanes read node 0); the ids of
            // the stage to refill are read from LDS at the start of a step so that the load itself never
            // waits on LDS.
            auto stage_ids = [&](uint32_t qq) -> uint32_t {  // neighbour id of lane, 0xFFFFFFFF when idle
                uint32_t nb = 0xFFFFFFFFu;
                if (qq < cnt) {
                    const uint32_t id = ids[qq * kWave + lane];
                    nb = lane < readlane(deg_l, qq) ? id : 0xFFFFFFFFu;
                }
                return nb;
            };
            auto stage_load = [&](uint32_t qq, uint32_t nb, int& lab) {
#if defined(BISBM_ABLATE) && (BISBM_ABLATE & 16)
                if (qq < cnt) lab = labels[nb == 0xFFFFFFFFu ? 0u : ((nb & 1023u) + (nb >= na ? na : 0u))];  // diagnostic: cache-resident labels
#elif defined(BISBM_ABLATE) && (BISBM_ABLATE & 32)
                if (qq < cnt) lab = (int)((nb == 0xFFFFFFFFu ? 0u : nb) >= na ? ka + (nb & 31u) : (nb & 31u));  // diagnostic: no label loads at all
#else
                if (qq < cnt) lab = labels[nb == 0xFFFFFFFFu ? 0u : nb];
#endif
            };
            uint32_t nb1 = stage_ids(0), nb2 = stage_ids(1), nb3 = stage_ids(2);
            int lab1 = 0, lab2 = 0, lab3 = 0;
            stage_load(0, nb1, lab1);
            stage_load(1, nb2, lab2);
            stage_load(2, nb3, lab3);
            const uint32_t kNoMove = 0xFFFFFFFEu;
            uint32_t mv_v1 = kNoMove, mv_v2 = kNoMove, mv_v3 = kNoMove;  // moves of steps q-3, q-2, q-1
            int mv_s1 = 0, mv_s2 = 0, mv_s3 = 0;
            // k_v of a step is accumulated in LDS: zero the counters, add one per neighbour label (labels patched with every move
            // made so far), for rows longer than a wave the remainder straight from HBM.
            auto hist_build = [&](uint32_t qq, uint32_t nbX, int labX) {
                if (qq < cnt) {
                    const uint32_t vq = readlane(v_l, qq), dq = readlane(deg_l, qq);
                    const uint32_t ob = vq >= na ? 0u : ka;  // base of the opposite type's block ids
                    int lab = labX;
                    if (nbX == mv_v1) lab = mv_s1;
                    if (nbX == mv_v2) lab = mv_s2;
                    if (nbX == mv_v3) lab = mv_s3;
                    hist[lane] = 0;
                    wfence();
                    if (lane < dq) atomicAdd(&hist[lab - (int)ob], 1);
                    if (__builtin_expect(dq > (uint32_t)kWave, 0)) {
                        const uint32_t beg = readlane(beg_l, qq);
                        for (uint32_t j = kWave + lane; j < dq; j += kWave)
                            atomicAdd(&hist[(int)labels[p.col[beg + j]] - (int)ob], 1);
                    }
                    wfence();
                }
            };

            auto do_step = [&](const uint32_t q, uint32_t& nbS, int& labS, const uint32_t& nbNext, const int& labNext) {
                FSTAMP(0);
                const uint32_t v = readlane(v_l, q), deg = readlane(deg_l, q), r = readlane(r_l, q);
                const bool type_b = v >= na;
                const uint32_t k_own = type_b ? kb : ka, k_oth = type_b ? ka : kb;
                const uint32_t own_base = type_b ? ka : 0u, oth_base = type_b ? 0u : ka;
                const uint32_t r_loc = r - own_base;
                const int mr_own = type_b ? mrB : mrA, mr_oth = type_b ? mrA : mrB;
                const int nr_own = type_b ? nrB : nrA;
                const double T = CT ? T_const : temperature_of(p, sweep_step0 + vi0 + q);  // :84
                // m[own block i][opposite block j] from the a x b quadrant
                auto mq_at = [&](uint32_t i_own, uint32_t j_oth) -> uint32_t {
                    return type_b ? j_oth * S + i_own : i_own * S + j_oth;
                };
                // this step's row: replay the moves made since its labels were requested
                const uint32_t nbC = nbS;
                int labC = labS;
                if (nbC == mv_v1) labC = mv_s1;
                if (nbC == mv_v2) labC = mv_s2;
                if (nbC == mv_v3) labC = mv_s3;
                // early LDS reads that only need r, and the row of the proposal's pivot block t (it depends on
                // the uniform and the row only, not on k_v): their latency overlaps the histogram's
                const uint32_t a_rt = mq_at(r_loc, lane);
                const int32_t m_rt = lane < k_oth ? mq[a_rt] : 0;
                const int eta_r = (int)eta_rd(r * D + deg);
                const double u_idx = readlane(ud_idx, q);
                uint32_t which = (uint32_t)(u_idx * (double)deg);
                if (which >= deg) which = deg ? deg - 1 : 0;
                uint32_t t_piv = oth_base;  // pivot block: label of the which-th neighbour (blockmodel.cc:619-621)
                if (deg != 0)
                    t_piv = which < (uint32_t)kWave ? (uint32_t)readlane(labC, which)
                                                     : (uint32_t)labels[p.col[readlane(beg_l, q) + which]];
                const int w_piv = lane < k_own ? mq[mq_at(lane, t_piv - oth_base)] : 0;

                // ---- k_v (replaces the dense k_[v] row, blockmodel.cc:691-700): LDS counters ----
                hist_build(q, nbS, labS);
                const int k = lane < k_oth ? hist[lane] : 0;
                const uint32_t nbN = stage_ids(q + 3);  // ids of the stage this step will refill
                FSTAMP(1);

                // ---- proposal: single_vertex_change, blockmodel.cc:613-637 ----
                uint32_t s;
                if (__builtin_expect(k_own == 1, 0)) {
                    s = r;
                } else if (__builtin_expect(deg == 0, 0)) {
                    s = (uint32_t)(u_idx * Kd);
                    if (s >= K) s = K - 1;
                } else {
                    const int32_t mrt = readlane(mr_oth, t_piv - oth_base);
                    const double u_tgt = readlane(ud_tgt, q);
                    if (__builtin_expect(readlane(ud_R, q) * (mrt + epsK) < epsK, 0)) {  // u < eps K / (m_r[t] + eps K), :622-624
                        s = (uint32_t)(u_tgt * Kd);
                        if (s >= K) s = K - 1;
                    } else {  // integer inverse CDF over row m[t][.] restricted to v's own type (:627-628)
                        long long x = (long long)(u_tgt * (double)mrt);
                        if (x >= (long long)mrt) x = (long long)mrt - 1;
                        const int scan = wave_inclusive_scan(w_piv);
                        const unsigned long long hit = __ballot(lane < k_own && (long long)scan > x);
                        s = hit ? own_base + (uint32_t)__ffsll((long long)hit) - 1 : own_base + k_own - 1;
                    }
                }
                FSTAMP(2);

                // ---- transition_ratio, metropolis_hasting.cc:103-192 (production arithmetic) ----
                // r == s (a = 0: always accepted at T > 0, :109-112) and cross-type targets (dS = +inf, :121-123)
                // are rare; they run the same straight-line code with s replaced by r and the outcome
                // overridden, so that the step has ONE prefetch site and no control-flow joins on registers
                // that receive loads.
                const bool same = (r == s);
                const bool cross = !same && ((r < ka) != (s < ka));
                const bool plain = !same && !cross;
                const uint32_t s_eff = plain ? s : r;
                const uint32_t s_loc = s_eff - own_base;
                const int ideg = (int)deg;
                const uint32_t a_st = mq_at(s_loc, lane);
                const int32_t m_st = lane < k_oth ? mq[a_st] : 0;
                const int eta_s = (int)eta_rd(s_eff * D + deg);
                const int m0r = readlane(mr_own, r_loc);
                const int m0s = readlane(mr_own, s_loc);
                const int n_r_r = readlane(nr_own, r_loc), n_r_s = readlane(nr_own, s_loc);
                // lanes 0..7: the scalar lgamma terms (:164-177); lanes 0..3 also carry the log_q arguments.  The
                // pattern repeats every 8 / 4 lanes so that every lane runs the same control flow (no exec
                // juggling, no skipped-region branches); only lanes 0..7 / 0..3 are used.
                const bool odd = lane & 1u;
                const int mm = odd ? m0s : m0r;                         // m0r, m0s
                const int dd = (lane & 2u) ? (odd ? ideg : -ideg) : 0;  // -> m1r, m1s in lanes 2,3 (mod 4)
                const int ee = odd ? eta_s : eta_r;
                const int eoff = (lane & 7u) < 6 ? 1 : (odd ? 2 : 0);  // eta_r+1, eta_s+1, eta_r, eta_s+2
                const uint32_t tail_idx = (lane & 4u) ? (uint32_t)(ee + eoff) : (uint32_t)(mm + dd + 1);
                const int qn = mm + dd;
                const int qk = (odd ? n_r_s : n_r_r) + ((lane & 2u) ? (odd ? 1 : -1) : 0);
                const double tail_lg = tab_at(tab.lg, tail_idx);
                const double logn = tab_at(tab.logtab, (uint32_t)qn);  // log(n) of the four log_q arguments
                const uint32_t kk = (uint32_t)k;
                const double L1 = tab_at(tab.lg, (uint32_t)(m_rt + 1));
                const double L2 = tab_at(tab.lg, (uint32_t)(m_st + 1));
                const double L3 = tab_at(tab.lg, (uint32_t)(m_rt + 1) - kk);
                const double L4 = tab_at(tab.lg, (uint32_t)(m_st + 1) + kk);
                FSTAMP(3);
                nbS = nbN;
                stage_load(q + 3, nbN, labS);  // younger than the gathers above: vmcnt retires in order
                // the Hastings sums need only on-chip data: they run while the table gathers are in flight
                // k == 0 lanes give exact zeros (0 * x = +0, identical table entries cancel): no branch needed
                const double inv = 1.0 / (mr_oth + epsK);
                const double a0 = k * (m_st + eps) * inv;
                const double a1 = k * (m_rt - k + eps) * inv;
                double accu0, accu1;
                if (k_oth <= 32u) {
                    butterfly_pair32(a0, a1, accu0, accu1);
                } else {
                    accu0 = butterfly_sum(a0);
                    accu1 = butterfly_sum(a1);
                }
                if (deg == 0) accu0 = accu1 = 1.;
                FSTAMP(4);
                const double lq = log_q<true>(tab, qn, qk, logn);
                FSTAMP(5);
                double d = (L1 + L2) - (L3 + L4);
                // fold the scalar terms into leaves 0..7 / 0..3 with their signs
                const bool neg_tail = (lane < 2) || (lane >= 6);  // -lg(m0r+1) -lg(m0s+1) ... -lg(eta_r) -lg(eta_s+2)
                d = lane < 8 ? d + (neg_tail ? -tail_lg : tail_lg) : d;
                d = lane < 4 ? d + (lane < 2 ? -lq : lq) : d;
                double dS = butterfly_sum(d);
                FSTAMP(6);
                // accept (:47-61): T == 0: dS < 0;  else u < exp(-dS/T) accu1/accu0
                bool accept;
                if (__builtin_expect(T == 0., 0))
                    accept = dS < 0;
                else
                    accept = readlane(ud_acc, q) * accu0 < accu1 * exp(-dS * (1.0 / T));
                if (same) {
                    accept = (T != 0.);
                    dS = 0.;
                }
                if (cross) accept = false;
                FSTAMP(7);
                // ---- apply_mcmc_moves, blockmodel.cc:461-503 ----
                bool ok = accept && (readlane(nr_own, r_loc) - 1 != 0);  // :467-471 veto after the draw
                uint32_t moved = kNoMove;
                if (ok && !same) {
                    wfence();
                    const int ideg = (int)deg;
                    if (lane == 0) {  // m_r / n_r live in the lane registers; LDS copies are rebuilt at kernel end
                        eta_wr(r * D + deg, (uint32_t)(eta_r - 1));
                        eta_wr(s * D + deg, (uint32_t)(eta_s + 1));
                        labels[v] = (uint8_t)s;
                    }
                    const int dm = (lane == s_loc ? ideg : 0) - (lane == r_loc ? ideg : 0);
                    const int dn = (lane == s_loc ? 1 : 0) - (lane == r_loc ? 1 : 0);
                    if (type_b) {
                        mrB += dm;
                        nrB += dn;
                    } else {
                        mrA += dm;
                        nrA += dn;
                    }
                    if (k != 0) {
                        mq[a_rt] = m_rt - k;
                        mq[a_st] = m_st + k;
                    }
                    cum_dS += dS;  // :500
                    moved = v;
                    wfence();
                }
                FSTAMP(8);
                // ---- bookkeeping of anneal(), metropolis_hasting.cc:85-94 ----
                mv_v1 = mv_v2;
                mv_s1 = mv_s2;
                mv_v2 = mv_v3;
                mv_s2 = mv_s3;
                mv_v3 = moved;
                mv_s3 = (int)s;
                if (ok) ++accepted_steps;
                if (!CT || T_const < 1.) {
                    if (ok && cum_dS < entropy_min) {
                        entropy_min = cum_dS;
                        u_cnt = 0;
                    }
                    if (T < 1.) ++u_cnt;
                }
            };
            for (uint32_t q = 0; q < cnt; q += 3) {
                do_step(q, nb1, lab1, nb2, lab2);
                if (__builtin_expect(q + 1 < cnt, 1)) do_step(q + 1, nb2, lab2, nb3, lab3);
                if (__builtin_expect(q + 2 < cnt, 1)) do_step(q + 2, nb3, lab3, nb1, lab1);
            }
        }
        ++sweeps_total;
        sweeps_done = sweep + 1;
        if (u_cnt >= p.steps_await) {  // metropolis_hasting.cc:96-98
            rate = (double)accepted_steps / (double)((sweep + 1) * (uint64_t)n);
            stopped = true;
            break;
        }
    }
    if (!stopped) rate = (double)accepted_steps / (double)p.duration;  // :100

    // chain state -> HBM
    __syncthreads();
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
        sc->cum_dS = cum_dS;
        sc->sweeps_total = sweeps_total;
        sc->last_rate = rate;
        sc->last_accepted = accepted_steps;
        sc->last_sweeps = sweeps_done;
    }
}

template <bool EL, bool CT>
static hipError_t launch_fast_variant(const SweepParams& p, size_t lds_bytes, hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute((const void*)sweep_fast_kernel<EL, CT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((sweep_fast_kernel<EL, CT>), dim3(p.n_chains), dim3(kWave), lds_bytes, stream, p);
    return hipGetLastError();
}

hipError_t launch_sweep_fast(const SweepParams& p, size_t lds_bytes, hipStream_t stream) {
    const bool ct = p.schedule == SCHED_CONSTANT;
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
        static const char* names[9] = {"loop+consume", "hist", "proposal", "lds+gather issue", "prefetch+accu",
                                       "log_q", "dS butterfly", "accept", "apply"};
        double tot = 0;
        for (int i = 0; i < 9; ++i) tot += (double)h[i];
        for (int i = 0; i < 9; ++i) fprintf(stderr, "[stamps] %-18s %8.1f cyc/step\n", names[i], (double)h[i] / steps);
        fprintf(stderr, "[stamps] %-18s %8.1f cyc/step\n", "total", tot / steps);
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fast_stamps), z, sizeof(z));
    }
#endif
    return e;
}

}  // namespace bisbm
