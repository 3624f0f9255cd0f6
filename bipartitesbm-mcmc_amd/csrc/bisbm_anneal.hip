// bisbm_anneal.hip -- bisbm_anneal (metropolis_hasting::anneal, metropolis_hasting.cc:64-101) on the host side: which kernel
// runs (production / generic), its LDS plan, the temperature tables of the pow / log schedules, and how a call becomes
// launches -- one, or several launches of whole sweeps (table slices of a long cooling call; launches of ~10^5 steps per chain
// whose pass depth follows the chain, bisbm_pass_policy.hpp), with anneal()'s early-stop bookkeeping carried in the chain's scalars.
//
// Reference lines cited as <file>:<line> relative to /root/reference/src.
#include "bisbm_engine.hpp"

using namespace bisbm;

extern "C" {

int bisbm_anneal(bisbm_handle h, int schedule, const float kwargs[2], uint64_t duration_steps,
                 uint64_t steps_await, double* acc_rate_out) {
    if (!h) return BISBM_ERR_INVALID_ARG;
    if (!kwargs) return fail(h, BISBM_ERR_INVALID_ARG, "kwargs is NULL");
    if (schedule < BISBM_SCHED_EXPONENTIAL || schedule > BISBM_SCHED_ABRUPT_COOL)
        return fail(h, BISBM_ERR_INVALID_ARG, "unknown schedule %d", schedule);
    if (!h->devs.empty()) return multi_anneal(h, schedule, kwargs, duration_steps, steps_await, acc_rate_out);
    if (!h->groups.empty()) {
        // one launch per shape, all in flight together: every group has a stream of its own, and a host thread per group
        // makes the (blocking) call; kernel time is reported as the longest group's, updates as the sum
        const size_t G = h->groups.size();
        std::vector<int> rcs(G, BISBM_OK);
        std::vector<std::vector<double>> rates(G);
        std::vector<std::thread> th;
        for (size_t gi = 0; gi < G; ++gi) {
            rates[gi].resize(h->groups[gi]->n_chains);
            th.emplace_back([&, gi] { rcs[gi] = bisbm_anneal(h->groups[gi], schedule, kwargs, duration_steps, steps_await, rates[gi].data()); });
        }
        for (auto& t : th) t.join();
        h->last_kernel_ms = 0;
        h->last_updates = 0;
        for (size_t gi = 0; gi < G; ++gi) {
            bisbm_engine* g = h->groups[gi];
            if (rcs[gi]) {
                h->err = g->err;
                return rcs[gi];
            }
            h->last_kernel_ms = std::max(h->last_kernel_ms, g->last_kernel_ms);
            h->last_updates += g->last_updates;
            h->last_pass_steps = gi == 0 ? g->last_pass_steps : std::max(h->last_pass_steps, g->last_pass_steps);
            if (acc_rate_out)
                for (size_t j = 0; j < rates[gi].size(); ++j) acc_rate_out[g->ridx[j]] = rates[gi][j];
        }
        return BISBM_OK;
    }
    if (!h->state_ready) return fail(h, BISBM_ERR_STATE, "call bisbm_init or bisbm_shuffle before bisbm_anneal");
    HIPCHK(h, hipSetDevice(h->device));

    SweepParams p{};
    p.rowptr = h->d_rowptr;
    p.col = h->d_col;
    p.n = (uint32_t)h->n;
    p.na = (uint32_t)h->na;
    p.nb = (uint32_t)h->nb;
    p.ka = h->ka;
    p.kb = h->kb;
    p.maxdeg = h->maxdeg;
    p.epsilon = h->epsilon;
    p.n_chains = h->n_chains;
    p.first_chain_id = h->first_chain_id;
    p.chain_gids = h->d_gids;
    p.labels = h->d_labels;
    p.label_stride = h->label_stride;
    p.wide = h->wide ? 1u : 0u;
    p.vlist = h->d_vlist;
    p.m = h->d_m;
    p.m_r = h->d_m_r;
    p.n_r = h->d_n_r;
    p.eta = h->d_eta;
    p.scalars = h->d_scalars;
    p.mt_engine = h->d_mt_engine;
    p.mt_gen = h->d_mt_gen;
    p.lgamma_tab = h->d_lgamma;
    p.lgamma_size = h->tab->lg.size();
    p.q_tab = h->d_q;
    p.q_stride = h->q_stride;
    p.log_tab = h->d_logtab;
    p.schedule = schedule;
    p.kw0 = kwargs[0];
    p.kw1 = kwargs[1];
    p.duration = duration_steps;
    p.steps_await = steps_await;
    p.seed = h->seed;

    p.t_base = 0;
    p.call_duration = duration_steps;
    p.resume = 0;
    // the production kernel covers Philox mode with both block counts <= 64; mt19937-compat mode and
    // wider partitions run the generic kernel (BISBM_FORCE_GENERIC=1 forces it, for A/B checks)
    const char* force = getenv("BISBM_FORCE_GENERIC");
    const bool fast = h->rng_mode == BISBM_RNG_PHILOX && h->ka <= 64 && h->kb <= 64 && !h->wide && !(force && force[0] == '1');
    // Temperatures of the pow / log schedules are evaluated with the host libm (the reference's own values) into a table of
    // at most kTabCap steps.  The generic kernel evaluates pow / log itself beyond it; the production kernel holds no
    // pow / log at all: a longer call runs as several launches of whole sweeps, each with the slice of the table it covers
    // (the early-stop bookkeeping carries over in the chain's scalars, SweepParams::resume).
    const bool tabled = schedule == BISBM_SCHED_EXPONENTIAL || schedule == BISBM_SCHED_LOGARITHMIC;
    const uint64_t kTabCap = getenv("BISBM_T_TABLE_CAP") ? std::max<uint64_t>(1, strtoull(getenv("BISBM_T_TABLE_CAP"), nullptr, 10)) : (1ull << 22);  // (the variable: tests)
    p.T_tab = nullptr;
    p.T_len = 0;
    p.T_base = 0;
    p.T_zero_after = 0;
    auto upload_table = [&](const std::vector<double>& T, uint64_t t0, int zero_after) -> int {
        if (T.size() > h->d_T_cap) {
            if (h->d_T) (void)hipFree(h->d_T);
            h->d_T = nullptr;
            h->d_T_cap = 0;
            HIPCHK(h, dalloc(&h->d_T, T.size()));
            h->d_T_cap = T.size();
        }
        if (!T.empty())
            HIPCHK(h, hipMemcpyAsync(h->d_T, T.data(), sizeof(double) * T.size(), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        p.T_tab = h->d_T;
        p.T_len = T.size();
        p.T_base = t0;
        p.T_zero_after = zero_after;
        return BISBM_OK;
    };
    const uint64_t total_sweeps = duration_steps / h->n;
    // (a table slice covers whole sweeps: at least one, however large the graph)
    const uint64_t tab_seg = std::max<uint64_t>(1, kTabCap / h->n);
    const bool tab_segments = fast && tabled && total_sweeps > tab_seg;
    if (tabled && !tab_segments) {
        int zero_after = 0;
        const std::vector<double> T = schedule_table(schedule, p.kw0, p.kw1, 0, fast ? total_sweeps * h->n : std::min(duration_steps, kTabCap), &zero_after);
        if (int rc = upload_table(T, 0, zero_after)) return rc;
    }
    // LDS plan.  eta goes to LDS when that still leaves room for four chains per CU (160 KiB / 4).
    const size_t K = h->K, D = (size_t)h->maxdeg + 1;
    const size_t eta_bytes = sizeof(uint32_t) * K * D;
    size_t lds;
    p.vlist_in_lds = 0;
    p.eta_w = p.eta_lo_a = p.eta_lo_b = 0;
    if (fast) {
        lds = sweep_fast_lds_bytes(h->ka, h->kb, h->maxdeg, false, 0);
        p.eta_in_lds = (lds + eta_bytes <= 40 * 1024) ? 1 : 0;
        if (!p.eta_in_lds) {
            // eta does not fit beside the rest (many blocks and / or long rows): the kernel keeps a window of it in LDS -- the
            // rows of the phase's own type, `eta_w` consecutive degrees -- placed per type where most nodes are (only rows of 1 to
            // 255 neighbours take the hot step at all); nodes of other degrees take the general step with eta in HBM
            const uint32_t kmax = std::max(h->ka, h->kb);
            const size_t room = lds < 40 * 1024 ? (40 * 1024 - lds) / (sizeof(uint32_t) * kmax) : 0;
            p.eta_w = (uint32_t)std::max<size_t>(1, std::min<size_t>(room, D));
            if (const char* w = getenv("BISBM_ETA_WINDOW")) p.eta_w = (uint32_t)std::max(1l, std::min<long>(atol(w), (long)D));  // (tests)
            for (int type = 0; type < 2; ++type) {
                const uint64_t* cnt = h->deg_count->data() + 257 * type;
                uint64_t in = 0, best = 0;
                uint32_t best_lo = 1;
                for (uint32_t d = 1; d <= 255; ++d) {  // window [d - eta_w + 1, d]
                    in += cnt[d];
                    if (d > p.eta_w) in -= cnt[d - p.eta_w];
                    const uint32_t lo = d >= p.eta_w ? d - p.eta_w + 1 : 1;
                    if (in > best) best = in, best_lo = lo;
                }
                (type ? p.eta_lo_b : p.eta_lo_a) = best_lo;
            }
        }
        lds = sweep_fast_lds_bytes(h->ka, h->kb, h->maxdeg, p.eta_in_lds != 0, p.eta_w);
    } else {
        // generic kernel: m quadrant (odd row stride), m_r, n_r, k_v histogram, staged rows; compat adds the
        // two mt19937 states and (small graphs) the visit list
        lds = generic_lds_base_bytes(h->ka, h->kb, h->wide, BISBM_RNG_PHILOX);
        p.eta_in_lds = (!h->wide && lds + eta_bytes <= 40 * 1024) ? 1 : 0;
        if (p.eta_in_lds) lds += eta_bytes;
        if (h->rng_mode == BISBM_RNG_MT19937_COMPAT) {
            lds += sizeof(uint32_t) * 624 * 4;  // two states and their tempered outputs
            if (sizeof(uint32_t) * h->n <= 48 * 1024 && lds + sizeof(uint32_t) * h->n <= 150 * 1024) {  // (wide mode: m_r / n_r of thousands of blocks come first)
                p.vlist_in_lds = 1;
                lds += sizeof(uint32_t) * h->n;
            }
        }
        lds = (lds + 15) & ~(size_t)15;
    }
    if (lds > 160 * 1024) return fail(h, BISBM_ERR_UNSUPPORTED, "chain state needs %zu B of LDS (> 160 KiB)", lds);

    p.simd_claims = nullptr;
    p.fixed_stepping_wave = 0;
    {
        const char* single = getenv("BISBM_SINGLE_STEPS");  // =1: one step per pass in every variant (A/B checks, tests)
        p.pair_steps = !single ? 3u : single[0] == '1' ? 0u : single[0] == '2' ? 1u : single[0] == '4' ? 2u : 3u;  // =2 / =4: at most two / four per pass
    }
    if (fast) {
        const char* fixed = getenv("BISBM_FIXED_ROLES");  // =1: wave 0 always steps, =2: wave 1 (A/B checks, tests)
        if (fixed && fixed[0] == '2') p.fixed_stepping_wave = 1;
        if (!(fixed && (fixed[0] == '1' || fixed[0] == '2'))) {
            if (!h->d_simd_claims) HIPCHK(h, dalloc(&h->d_simd_claims, kSimdClaims));
            HIPCHK(h, hipMemsetAsync(h->d_simd_claims, 0, sizeof(uint32_t) * kSimdClaims, h->stream));
            p.simd_claims = h->d_simd_claims;
        }
    }
    // One launch, or -- production kernel, at most 32 blocks of a type -- several launches of whole sweeps, so that the depth
    // of the passes can follow the chain.  Deep passes (four / eight steps) pay where few steps move or the blocks are many
    // enough for movers to miss each other (2.4 x on the reference's n_1000 data set, +17 % at 32 + 32 blocks near the mode,
    // +20 % in the cold part of a cooling schedule); from a random start on a large graph with few blocks nearly every step
    // moves, most followers clash, and two steps per pass are faster.  Which is which depends on the graph, the partition and
    // where the chain is, so it is MEASURED: every launch is timed, the depth with the best updates per ms so far runs, and
    // every sixteenth launch tries a neighbouring depth again (a chain leaves its burn-in, a schedule cools down).  The chain
    // is the same chain whatever runs (same Philox counters, bit-equal results).
    // (depth 1 = two steps per pass, 2 = four -- in 16-lane rows, two blocks per lane above 16 blocks of a type --, 3 = eight)
    const uint32_t max_depth = (!fast || p.pair_steps < 2u || h->ka > 32 || h->kb > 32) ? 0u
                               : std::min<uint32_t>(p.pair_steps, (h->ka <= 8 && h->kb <= 8) ? 3u : 2u);
    // Any schedule, any steps_await: what anneal() carries from sweep to sweep -- entropy_min_, the position of the last
    // minimum, the count of T < 1 steps, "this chain has returned" -- travels in the chain's scalars (SweepParams::resume, as for
    // the table slices above); a chain that has returned (steps_await == 0 at T >= 1: after its FIRST sweep, :96-98) is skipped
    // by the later launches, and the loop below ends when every chain has.  (A constant schedule at T = 0 runs general steps
    // only: nothing to choose.)
    // A production launch without anneal()'s early-stop bookkeeping does not keep the running sum of accepted dS (it costs every
    // pass ~10 instructions and nobody looks at it meanwhile): the call advances the sum by the change of the block-state part of
    // the description length instead -- evaluated before the first launch (or known from the call before) and after the last one.
    // (BISBM_KEEP_SUM=1: every launch keeps the sum of its own dS values, for the tests that compare exactly that with the change
    // of the description length)
    p.keep_sum = (getenv("BISBM_KEEP_SUM") && getenv("BISBM_KEEP_SUM")[0] == '1') ? 1u : 0u;
    const bool sum_from_entropy = fast && p.keep_sum == 0u && !sweep_fast_tracks_minimum(schedule, p.kw0, steps_await, duration_steps);
    if (sum_from_entropy) {
        if (!h->d_ent_prev) HIPCHK(h, dalloc(&h->d_ent_prev, h->n_chains));
        if (!h->ent_prev_valid)
            if (int rc = launch_block_entropy(h, h->d_ent_prev)) return rc;
    }
    h->ent_prev_valid = false;  // (until this call has ended the way it was meant to)
    const bool depth_segments = max_depth >= 2u && !(schedule == SCHED_CONSTANT && !(kwargs[0] > 0.f)) && total_sweeps >= 2;
    const bool segmented = depth_segments || tab_segments;
    std::vector<ChainScalars> sc(h->n_chains);
    std::vector<uint64_t> acc_sum(h->n_chains, 0), sweeps_sum(h->n_chains, 0);
    double ms_sum = 0;
    uint64_t updates = 0, sweeps_left = segmented ? total_sweeps : 0, sweeps_done = 0;
    // depth: >= 10^5 steps per chain and launch (tens of ms); table slices: what the table holds, or the depth's figure if smaller
    // (BISBM_LAUNCH_STEPS: tests cut calls into launches of single sweeps)
    const uint64_t launch_steps = getenv("BISBM_LAUNCH_STEPS") ? std::max<uint64_t>(1, strtoull(getenv("BISBM_LAUNCH_STEPS"), nullptr, 10)) : 100000;
    const uint64_t depth_seg = std::max<uint64_t>(1, (launch_steps + h->n - 1) / h->n);
    const uint64_t seg = !depth_segments ? tab_seg : tab_segments ? std::min(tab_seg, depth_seg) : depth_seg;
    // the table slice of the next launch is evaluated on the host while the current launch runs
    struct Slice {
        std::vector<double> T;
        int zero_after = 0;
    };
    auto slice_sweeps = [&](uint64_t left) { return (tab_segments || left >= 2 * seg) ? std::min(left, seg) : left; };
    std::future<Slice> next_slice;
    auto start_slice = [&](uint64_t first_sweep, uint64_t count) {
        const int sched = schedule;
        const float k0 = p.kw0, k1 = p.kw1;
        const uint64_t t0 = first_sweep * h->n, len = count * h->n;
        next_slice = std::async(std::launch::async, [sched, k0, k1, t0, len] {
            Slice sl;
            sl.T = schedule_table(sched, k0, k1, t0, len, &sl.zero_after);
            return sl;
        });
    };
    if (tab_segments) start_slice(0, slice_sweeps(sweeps_left));
    bool first = true;
    while (first || sweeps_left > 0) {
        if (segmented) {
            const uint64_t now = slice_sweeps(sweeps_left);
            p.duration = now * h->n;
            p.t_base = sweeps_done * h->n;
            p.resume = first ? 0u : 1u;
            if (tab_segments) {
                Slice sl;
                try {
                    sl = next_slice.get();
                } catch (...) {
                    return fail(h, BISBM_ERR_STATE, "temperature table: out of host memory");
                }
                if (int rc = upload_table(sl.T, p.t_base, sl.zero_after)) return rc;
            }
            sweeps_left -= now;
            sweeps_done += now;
        }
        first = false;
        uint32_t depth = max_depth;
        if (max_depth >= 2u) {
            // (bisbm_pass_policy.hpp: the incumbent depth, now and then a look at a neighbour, a switch only on a clear win)
            depth = h->passes.choose(max_depth, h->n <= 100000);
            if (const char* pd = getenv("BISBM_PASS_DEPTH"))  // diagnostic: 2 / 4 / 8 pins the depth of the passes
                depth = pd[0] == '2' ? 1u : pd[0] == '4' ? std::min(2u, max_depth) : pd[0] == '8' ? max_depth : depth;
        }
        p.pass_depth = depth;
        {  // (what launch_sweep_fast picks from these numbers)
            const uint32_t d = std::min(p.pair_steps, p.pass_depth);
            const bool cold = schedule == SCHED_CONSTANT && kwargs[0] == 0.f;  // T = 0 throughout: general steps only
            h->last_pass_steps = (!fast || cold) ? 1u
                                 : (h->ka <= 8 && h->kb <= 8 && d >= 3u) ? 8u
                                 : (h->ka <= 32 && h->kb <= 32 && d >= 2u) ? 4u
                                 : p.pair_steps != 0u ? 2u : 1u;
        }
        HIPCHK(h, hipEventRecord(h->ev0, h->stream));
        if (fast)
            HIPCHK(h, launch_sweep_fast(p, lds, h->stream));
        else
            HIPCHK(h, launch_sweep(p, h->rng_mode, lds, h->stream));
        HIPCHK(h, hipEventRecord(h->ev1, h->stream));
        if (tab_segments && sweeps_left > 0) start_slice(sweeps_done, slice_sweeps(sweeps_left));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        float ms = 0;
        HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
        ms_sum += ms;
        HIPCHK(h, hipMemcpy(sc.data(), h->d_scalars, sizeof(ChainScalars) * h->n_chains, hipMemcpyDeviceToHost));
        uint64_t upd = 0;
        bool all_stopped = true;
        for (uint32_t c = 0; c < h->n_chains; ++c) {
            acc_sum[c] += sc[c].last_accepted;
            sweeps_sum[c] += sc[c].last_sweeps;
            upd += sc[c].last_sweeps * h->n;
            all_stopped = all_stopped && sc[c].stopped != 0;
        }
        updates += upd;
        if (max_depth >= 2u && ms > 0.05f && upd > 0) {
            const double speed = (double)upd / ms;
            uint64_t acc_now = 0;
            for (uint32_t c = 0; c < h->n_chains; ++c) acc_now += sc[c].last_accepted;
            const double acc_frac = (double)acc_now / (double)upd;
            const double before[3] = {h->passes.figure(1), h->passes.figure(2), h->passes.figure(3)};
            const uint32_t incumbent = h->passes.current();
            h->passes.record(depth, speed, acc_frac);
            if (getenv("BISBM_PASS_LOG"))
                fprintf(stderr, "[bisbm passes] depth %u%s: %.3e updates/ms, accepted %.3f (before the launch: two %.3e, four %.3e, eight %.3e; incumbent %u -> %u)\n",
                        depth, depth != incumbent && incumbent != 0u && h->passes.settled() ? " (a look)" : "", speed, acc_frac, before[0], before[1], before[2],
                        incumbent, h->passes.current());
        }
        if (segmented && fast && all_stopped) {  // every chain has returned (:96-98)
            if (next_slice.valid()) next_slice.wait();
            break;
        }
    }
    const float ms = (float)ms_sum;
    h->last_kernel_ms = ms_sum;
    if (segmented) {  // the call's totals, as one launch would have left them
        for (uint32_t c = 0; c < h->n_chains; ++c) {
            sc[c].last_accepted = acc_sum[c];
            sc[c].last_sweeps = sweeps_sum[c];
            sc[c].last_rate = (fast && sc[c].stopped) ? (double)acc_sum[c] / (double)(sweeps_sum[c] * h->n)  // :97
                                                      : (double)acc_sum[c] / (double)duration_steps;          // :100
        }
        HIPCHK(h, hipMemcpy(h->d_scalars, sc.data(), sizeof(ChainScalars) * h->n_chains, hipMemcpyHostToDevice));
    }
    if (sum_from_entropy) {
        if (int rc = launch_block_entropy(h, h->d_tmp_f64)) return rc;
        HIPCHK(h, launch_sum_from_entropy(h->d_scalars, h->d_ent_prev, h->d_tmp_f64, h->n_chains, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->d_ent_prev, h->d_tmp_f64, sizeof(double) * h->n_chains, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        h->ent_prev_valid = true;
    }
    for (uint32_t c = 0; c < h->n_chains; ++c)
        if (acc_rate_out) acc_rate_out[c] = sc[c].last_rate;
    h->last_updates = updates;
    // BISBM_PLACEMENT_LOG=1 (diagnostic): how the dispatcher spread the launch over the chip.  A SIMD that hosts
    // the stepping waves of two chains runs both of them slower, and the launch takes as long as its slowest chain.
    if (fast) {
        const char* plog = getenv("BISBM_PLACEMENT_LOG");
        if (plog && plog[0] == '1') {
            std::map<uint32_t, int> main_per_simd, wg_per_cu;
            for (uint32_t c = 0; c < h->n_chains; ++c) {
                const uint32_t hw = sc[c].hw_id[0], cu = ((sc[c].xcc_id & 0xf) << 16) | (hw & 0xff00u);  // se, sh, cu ids
                ++main_per_simd[(cu << 2) | ((hw >> 4) & 3u)];
                ++wg_per_cu[cu];
            }
            int simd_hist[5] = {0, 0, 0, 0, 0}, cu_hist[9] = {0};
            for (auto& kv : main_per_simd) ++simd_hist[std::min(kv.second, 4)];
            for (auto& kv : wg_per_cu) ++cu_hist[std::min(kv.second, 8)];
            fprintf(stderr, "[bisbm placement] %.1f ms; CUs used %zu; workgroups per CU:", ms, wg_per_cu.size());
            for (int i = 1; i <= 8; ++i)
                if (cu_hist[i]) fprintf(stderr, " %dx%d", cu_hist[i], i);
            fprintf(stderr, "; stepping waves per SIMD: %d x1, %d x2, %d x3, %d x4+\n", simd_hist[1], simd_hist[2],
                    simd_hist[3], simd_hist[4]);
        }
    }
    return BISBM_OK;
}

}  // extern "C"
