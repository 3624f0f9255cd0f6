// bisbm_merge.hip -- block merges and splits between anneals, and what follows from them: chains of one handle
// in different shapes (sub-engines per shape).
//
// Reference lines cited as <file>:<line> relative to /root/reference/src.
#include "bisbm_engine.hpp"

using namespace bisbm;

// ---------------------------------------------------------------------------------------------
// Agglomerative merges between anneals: blockmodel_t::agg_merge x2, compute_b_adj_list,
// compute_dS(block_move_t), apply_block_moves, single_block_change (blockmodel.cc:109-288,335-372,567-611,
// 639-669).  K-scale work on the host, as in the reference, one chain after the other; the device supplies the
// first node of every label (the order in which the reference renumbers blocks) and applies the final relabelling
// to all chains at once, then rebuilds the block state.  Between proposal rounds the block matrix is merged on
// the host (m is additive over blocks), which is what the reference's full rebuild computes.
// mt19937-compat mode draws with libstdc++'s own std::mt19937 / uniform_real_distribution / discrete_distribution
// restored from the chain's device-side generator state -- the reference's draw sequence by construction.
// ---------------------------------------------------------------------------------------------
namespace {

void philox_host(uint64_t seed, uint32_t chain, uint32_t purpose, uint64_t idx, uint32_t out[4]) {
    uint32_t c0 = (uint32_t)idx, c1 = (uint32_t)(idx >> 32), c2 = chain, c3 = purpose;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0, out[1] = c1, out[2] = c2, out[3] = c3;
}
double u53_host(uint32_t hi, uint32_t lo) { return (double)((((uint64_t)hi << 32) | lo) >> 11) * 0x1.0p-53; }

constexpr uint32_t kPhxMergeA = 4, kPhxMergeB = 5;

struct MergeChain {
    // block state in the current numbering
    size_t K = 0, ka = 0, kb = 0, na = 0;
    std::vector<int> M;            // K x K, symmetric
    std::vector<int> m_r;          // row sums
    std::vector<uint32_t> first;   // lowest node id of each block
    std::vector<uint16_t> cmap;    // original label -> current label (one entry per original block, 0xffff = gone)
    // randomness
    bool compat = false;
    std::mt19937 engine, gen;
    std::uniform_real_distribution<> random_real;  // blockmodel.hh:16
    uint64_t seed = 0;
    uint32_t chain_gid = 0, epoch = 0;
    double epsilon = 0;
    const std::vector<double>* lg = nullptr;

    int at(size_t i, size_t j) const { return M[i * K + j]; }
    double lgamma_fast(long long x) const { return (*lg)[(size_t)x]; }  // table covers 2E+1 (bisbm_create)

    // single_block_change, blockmodel.cc:639-669 (ctr: index of the proposal inside the round, Philox mode)
    std::pair<size_t, size_t> propose(size_t src, uint64_t ctr) {
        if ((ka == 1 && src < ka) || (kb == 1 && src >= ka)) return {src, src};
        std::vector<size_t> badj;  // compute_b_adj_list, :274-288
        for (size_t t = 0; t < K; ++t)
            if (at(src, t) > 0) badj.push_back(t);
        size_t target;
        if (compat) {
            if (badj.empty()) {
                target = size_t(random_real(engine) * K);
            } else {
                const size_t t = badj[size_t(random_real(engine) * badj.size())];
                const double R_t = epsilon * K / (m_r[t] + epsilon * K);
                if (random_real(engine) < R_t) {
                    target = size_t(random_real(engine) * K);
                } else {
                    std::discrete_distribution<size_t> d(M.begin() + t * K, M.begin() + (t + 1) * K);
                    target = d(gen);  // drawn with `gen`, :656-657
                }
            }
        } else {
            uint32_t A[4], B[4];
            const uint64_t idx = ((uint64_t)epoch << 32) | ctr;
            philox_host(seed, chain_gid, kPhxMergeA, idx, A);
            philox_host(seed, chain_gid, kPhxMergeB, idx, B);
            const double u0 = u53_host(A[0], A[1]), u1 = u53_host(A[2], A[3]), u2 = u53_host(B[0], B[1]);
            if (badj.empty()) {
                target = std::min(size_t(u0 * (double)K), K - 1);
            } else {
                const size_t t = badj[std::min(size_t(u0 * (double)badj.size()), badj.size() - 1)];
                if (u1 * (m_r[t] + epsilon * (double)K) < epsilon * (double)K) {
                    target = std::min(size_t(u2 * (double)K), K - 1);
                } else {  // integer inverse CDF over row m[t][.]
                    const long long tot = m_r[t];
                    const long long x = std::min((long long)(u2 * (double)tot), tot - 1);
                    long long cum = 0;
                    target = K - 1;
                    for (size_t c = 0; c < K; ++c) {
                        cum += at(t, c);
                        if (cum > x) {
                            target = c;
                            break;
                        }
                    }
                }
            }
        }
        return src > target ? std::make_pair(src, target) : std::make_pair(target, src);  // higher index merges into lower
    }

    // compute_dS(const block_move_t&), blockmodel.cc:335-372
    double merge_dS(size_t r, size_t s) const {
        if (r == s || (r < ka && s >= ka) || (r >= ka && s < ka)) return std::numeric_limits<double>::infinity();
        double entropy0 = 0., entropy1 = 0.;
        for (size_t idx = 0; idx < K; ++idx) {
            const bool opposite = r < ka ? idx >= ka : idx < ka;
            if (opposite && m_r[idx] != 0) {
                entropy0 -= lgamma_fast(at(r, idx) + 1);
                entropy0 -= lgamma_fast(at(s, idx) + 1);
                entropy1 -= lgamma_fast(at(s, idx) + at(r, idx) + 1);
            }
        }
        entropy0 -= -lgamma_fast(m_r[r] + 1);
        entropy0 -= -lgamma_fast(m_r[s] + 1);
        entropy1 -= -lgamma_fast(m_r[r] + m_r[s] + 1);
        return entropy1 - entropy0;
    }

    using HeapItem = std::pair<double, size_t>;
    using Heap = std::priority_queue<HeapItem, std::vector<HeapItem>, std::greater<>>;

    // one proposal round: nm proposals for every block of [first_block, first_block + count), unique (source, target)
    // pairs keyed by dS (:147-159)
    void propose_round(size_t first_block, size_t count, int nm, std::vector<std::pair<size_t, size_t>>& moves, Heap& q) {
        std::set<std::pair<size_t, size_t>> seen;
        moves.clear();
        q = Heap();
        uint64_t ctr = 0;
        for (size_t b = first_block; b < first_block + count; ++b)
            for (int i = 0; i < nm; ++i) {
                const auto mv = propose(b, ctr++);
                if (seen.insert(mv).second) {
                    q.push({merge_dS(mv.first, mv.second), moves.size()});
                    moves.push_back(mv);
                }
            }
        ++epoch;
    }

    // bookkeeping of one accepted merge (:172-184)
    static void accept(std::set<size_t>& touched, std::vector<std::set<size_t>>& groups, size_t src, size_t tgt) {
        if (touched.count(src) == 0 && touched.count(tgt) == 0) {
            groups.push_back({src, tgt});
        } else {
            for (auto& g : groups)
                if (g.count(tgt) > 0 || g.count(src) > 0) {
                    g.insert({src, tgt});
                    break;
                }
        }
        touched.insert({src, tgt});
    }

    // apply_block_moves (:567-611) on the block level: merge the groups, renumber the blocks in the order of their
    // first node.  false = the reference's sanity check would fail.
    bool apply(const std::set<size_t>& touched, const std::vector<std::set<size_t>>& groups) {
        std::vector<size_t> to(K);
        for (size_t b = 0; b < K; ++b) {
            size_t mb = b;
            if (touched.count(mb) > 0)
                for (auto const& g : groups)
                    if (g.count(mb) > 0) mb = *g.begin();
            to[b] = mb;
        }
        std::vector<uint32_t> nfirst(K, 0xffffffffu);
        for (size_t b = 0; b < K; ++b) nfirst[to[b]] = std::min(nfirst[to[b]], first[b]);
        std::vector<size_t> order;  // surviving blocks by first node
        for (size_t b = 0; b < K; ++b)
            if (nfirst[b] != 0xffffffffu) order.push_back(b);
        std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return nfirst[x] < nfirst[y]; });
        std::vector<size_t> n2o(K, (size_t)-1);
        for (size_t i = 0; i < order.size(); ++i) n2o[order[i]] = i;
        const size_t nK = order.size();
        size_t nka = 0;
        for (size_t i = 0; i < nK; ++i) nka += nfirst[order[i]] < na;
        // labels of type-a nodes must occupy [0, nka): first nodes are sorted and type-a ids come first
        std::vector<int> nM(nK * nK, 0);
        for (size_t i = 0; i < K; ++i) {
            if (n2o[to[i]] == (size_t)-1) continue;  // empty block
            for (size_t j = 0; j < K; ++j)
                if (n2o[to[j]] != (size_t)-1) nM[n2o[to[i]] * nK + n2o[to[j]]] += at(i, j);
        }
        for (auto& c : cmap)
            if (c != 0xffff) c = n2o[to[c]] == (size_t)-1 ? 0xffff : (uint16_t)n2o[to[c]];
        std::vector<uint32_t> f2(nK);
        for (size_t i = 0; i < nK; ++i) f2[i] = nfirst[order[i]];
        first.swap(f2);
        M.swap(nM);
        K = nK;
        ka = nka;
        kb = nK - nka;
        m_r.assign(K, 0);
        for (size_t i = 0; i < K; ++i)
            for (size_t j = 0; j < K; ++j) m_r[i] += at(i, j);
        return ka >= 1 && kb >= 1;
    }

    // agg_merge(engine, diff_a, diff_b, nm), :109-206.  0 ok, -1 sanity, -3 cannot make progress
    int agg_merge(int diff_a, int diff_b, int nm) {
        for (int depth = 0; depth < 10000; ++depth) {
            if (diff_a + diff_b == 0) return 0;
            size_t first_block, count;
            if (diff_a > 0 && diff_b == 0)
                first_block = 0, count = ka;
            else if (diff_a == 0 && diff_b > 0)
                first_block = ka, count = kb;
            else
                first_block = 0, count = K;
            std::vector<std::pair<size_t, size_t>> moves;
            Heap q;
            propose_round(first_block, count, nm, moves, q);
            std::set<size_t> touched;
            std::vector<std::set<size_t>> groups;
            bool again = false;
            size_t merged = 0;
            while (diff_a + diff_b != 0 && !q.empty()) {
                if (q.top().first == std::numeric_limits<double>::infinity()) {  // :163-168: apply, then start over
                    again = true;
                    break;
                }
                const auto mv = moves[q.top().second];
                int* budget = (mv.first < ka && diff_a != 0) ? &diff_a : ((mv.first >= ka && diff_b != 0) ? &diff_b : nullptr);
                if (budget && !(touched.count(mv.first) > 0 && touched.count(mv.second) > 0)) {
                    *budget -= 1;
                    accept(touched, groups, mv.first, mv.second);
                    ++merged;
                }
                q.pop();
            }
            if (!apply(touched, groups)) return -1;
            if (!again) return 0;
            // the reference recurses without end when the remaining budget asks for merges in a type that is down to
            // one block
            if (merged == 0 && !((diff_a > 0 && ka > 1) || (diff_b > 0 && kb > 1))) return -3;
        }
        return -3;
    }

    // agg_merge(engine, diff, nm), :208-271
    int agg_merge_total(int diff, int nm) {
        if (diff == 0) return 0;
        const int DIFF = diff;
        std::set<size_t> touched;
        std::vector<std::set<size_t>> groups;
        std::vector<std::pair<size_t, size_t>> moves;
        Heap q;
        bool minS = true;
        for (int rounds = 0; minS; ++rounds) {
            if (rounds >= 10000) return -3;
            groups.clear();
            touched.clear();
            propose_round(0, K, nm, moves, q);
            while (diff != 0 && !q.empty()) {
                const auto mv = moves[q.top().second];
                if (!(touched.count(mv.first) > 0 && touched.count(mv.second) > 0)) {
                    diff -= 1;
                    accept(touched, groups, mv.first, mv.second);
                }
                minS = q.top().first == std::numeric_limits<double>::infinity();
                q.pop();
            }
            diff = DIFF;
        }
        return apply(touched, groups) ? 0 : -1;
    }
};

// blockmodel_t::agg_split(engine, type, nm), blockmodel.cc:505-565, in every chain (intended semantics: a node's position
// in its block's split vector is its rank within the block).  Device: ranks, the edge counts of every
// trial's cut (split_eval_kernel), the relabelling; host: the K-scale dS of every (block, trial) -- compute_dS(size_t,
// vector<bool>&), :374-424, serial sums in source order -- and the choice (lowest dS, strict <, blocks ascending,
// trials in order = lexicographic minimum of (dS, block, trial)).  mt19937-compat mode shuffles a real
// std::vector<bool> with the chain's std::mt19937 (the reference's draw sequence by construction) and uploads the cuts.
int run_split(bisbm_engine* h, int type, int nm) {
    if (!h->state_ready) return fail(h, BISBM_ERR_STATE, "call bisbm_init or bisbm_shuffle before bisbm_agg_merge");
    if (nm < 1 || nm > 65535) return fail(h, BISBM_ERR_INVALID_ARG, "nm must be in [1, 65535]");
    // Past 256 blocks (round 3): the handle is (or becomes) wide -- two-byte labels, the same kernels instantiated for them with
    // their per-block tables in HBM.  The new shape must be one wide mode serves (bisbm_check_shape).
    if (int rc = bisbm_check_shape(h->ka + (type ? 0u : 1u), h->kb + (type ? 1u : 0u), h->rng_mode)) {
        h->err = g_create_error;
        return rc;
    }
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const size_t C = h->n_chains, K = h->K, ka = h->ka, kb = h->kb;
    const size_t k_type = type ? kb : ka, k_oth = type ? ka : kb, b_lo = type ? ka : 0;
    const size_t n_type = type ? h->nb : h->na;
    const bool compat = h->rng_mode == BISBM_RNG_MT19937_COMPAT;

    std::vector<int32_t> n_r(C * K), m_r(C * K), quad(C * ka * kb);
    std::vector<ChainScalars> sc(C);
    HIPCHK(h, hipMemcpy(n_r.data(), h->d_n_r, sizeof(int32_t) * n_r.size(), hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(m_r.data(), h->d_m_r, sizeof(int32_t) * m_r.size(), hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(quad.data(), h->d_m, sizeof(int32_t) * quad.size(), hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(sc.data(), h->d_scalars, sizeof(ChainScalars) * C, hipMemcpyDeviceToHost));
    // (can every chain be split at all?  Asked before anything about the handle changes)
    for (size_t c = 0; c < C; ++c) {
        bool any = false;
        for (size_t b = 0; b < k_type; ++b) any |= n_r[c * K + b_lo + b] > 1;
        if (!any)
            return fail(h, BISBM_ERR_STATE, "chain %zu: no type-%c block has two nodes: nothing to split (the reference would add an empty block)",
                        c, type ? 'b' : 'a');
    }
    // 256 -> 257 blocks: the labels become two bytes first.  The byte buffers are kept until the split has been applied: a call
    // that fails before that puts them back, so a handle is wide exactly while it has more than 256 blocks (new_group's rule).
    uint8_t *narrow_labels = nullptr, *narrow_tmp = nullptr;
    bool applied = false;
    if (!h->wide && h->K + 1 > 256) {
        uint8_t *wide_labels = nullptr, *wide_tmp = nullptr;
        const size_t bytes = C * h->label_stride * 2;
        hipError_t e = dalloc(&wide_labels, bytes);
        if (e == hipSuccess && h->d_labels_tmp) e = dalloc(&wide_tmp, bytes);  // (Philox mode's snapshot buffer for shuffle_bisbm: sized for the label format)
        if (e == hipSuccess) e = hipMemsetAsync(wide_labels, 0, bytes, h->stream);  // (the padding [n, label_stride), as bisbm_create leaves it)
        if (e == hipSuccess) e = launch_labels_to_wide(h->d_labels, wide_labels, h->label_stride, (uint32_t)h->n, h->n_chains, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) {
            if (wide_labels) (void)hipFree(wide_labels);
            if (wide_tmp) (void)hipFree(wide_tmp);
            return fail(h, BISBM_ERR_HIP, "widening the labels: %s", hipGetErrorString(e));
        }
        narrow_labels = h->d_labels, narrow_tmp = h->d_labels_tmp;
        h->d_labels = wide_labels, h->d_labels_tmp = wide_tmp;
        h->wide = true;
    }

    uint32_t *d_rank = nullptr, *d_bits = nullptr, *d_chosen = nullptr, *d_rank_base = nullptr, *d_block_off = nullptr;
    int32_t *d_out_k = nullptr, *d_out_deg = nullptr;
    auto cleanup = [&]() {
        for (void* p : {(void*)d_rank, (void*)d_bits, (void*)d_chosen, (void*)d_out_k, (void*)d_out_deg, (void*)d_rank_base, (void*)d_block_off})
            if (p) (void)hipFree(p);
        if (narrow_labels) {
            if (!applied) {  // the split did not happen: back to byte labels
                (void)hipFree(h->d_labels);
                if (h->d_labels_tmp) (void)hipFree(h->d_labels_tmp);
                h->d_labels = narrow_labels, h->d_labels_tmp = narrow_tmp;
                h->wide = false;
            } else {
                (void)hipFree(narrow_labels);
                if (narrow_tmp) (void)hipFree(narrow_tmp);
            }
            narrow_labels = narrow_tmp = nullptr;
        }
    };
#define SCHK(expr)                                                                 \
    do {                                                                           \
        hipError_t e_ = (expr);                                                    \
        if (e_ != hipSuccess) {                                                    \
            cleanup();                                                             \
            return fail(h, BISBM_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
        }                                                                          \
    } while (0)
    SCHK(dalloc(&d_rank, C * n_type));
    SCHK(dalloc(&d_chosen, 2 * C));

    SplitParams sp{};
    sp.rowptr = h->d_rowptr;
    sp.col = h->d_col;
    sp.n = (uint32_t)h->n;
    sp.na = (uint32_t)h->na;
    sp.ka = h->ka;
    sp.kb = h->kb;
    sp.n_chains = h->n_chains;
    sp.first_chain_id = h->first_chain_id;
    sp.chain_gids = h->d_gids;
    sp.type = (uint32_t)type;
    sp.nm = (uint32_t)nm;
    sp.seed = h->seed;
    sp.labels = h->d_labels;
    sp.label_stride = h->label_stride;
    sp.n_r = h->d_n_r;
    sp.scalars = h->d_scalars;
    sp.rank = d_rank;
    sp.wide = h->wide ? 1u : 0u;
    if (h->wide) {
        SCHK(dalloc(&d_rank_base, C * K));
        SCHK(hipMemsetAsync(d_rank_base, 0, sizeof(uint32_t) * C * K, h->stream));
        sp.rank_base = d_rank_base;
        std::vector<uint32_t> off(C * k_type);  // (compat: where a block's bits start in a trial's cut)
        for (size_t c = 0; c < C; ++c) {
            uint32_t acc = 0;
            for (size_t b = 0; b < k_type; ++b) {
                off[c * k_type + b] = acc;
                acc += (uint32_t)n_r[c * K + b_lo + b];
            }
        }
        SCHK(dalloc(&d_block_off, off.size()));
        SCHK(hipMemcpy(d_block_off, off.data(), sizeof(uint32_t) * off.size(), hipMemcpyHostToDevice));
        sp.block_off = d_block_off;
    }
    SCHK(launch_split_rank(sp, h->stream));

    // mt19937-compat: the cuts come from std::shuffle on the chain's engine (:541-543), as bits at (block offset + rank)
    std::vector<uint32_t> mt_e;
    const size_t bit_words = (n_type + 31) / 32;
    if (compat) {
        if ((double)C * nm * bit_words * 4.0 > 2.0e9) {
            cleanup();
            return fail(h, BISBM_ERR_UNSUPPORTED, "mt19937-compat agg_split needs %.1f GB of cut bits; use fewer chains (compat is the parity path)",
                        (double)C * nm * bit_words * 4.0 / 1e9);
        }
        mt_e.resize(C * 624);
        SCHK(hipMemcpy(mt_e.data(), h->d_mt_engine, sizeof(uint32_t) * mt_e.size(), hipMemcpyDeviceToHost));
        std::vector<uint32_t> bits(C * (size_t)nm * bit_words, 0u);
        for (size_t c = 0; c < C; ++c) {
            std::mt19937 engine;
            {
                std::stringstream ss;
                for (int i = 0; i < 624; ++i) ss << mt_e[c * 624 + i] << ' ';
                ss << sc[c].engine_idx;
                ss >> engine;
            }
            size_t off = 0;
            for (size_t b = 0; b < k_type; ++b) {
                const size_t nb = (size_t)n_r[c * K + b_lo + b];
                if (nb > 1) {
                    std::vector<bool> splitter(nb, false);  // :532-537
                    for (size_t i = nb / 2; i < nb; ++i) splitter[i] = true;
                    std::shuffle(splitter.begin(), splitter.end(), engine);  // :541
                    for (int j = 0; j < nm; ++j) {
                        std::shuffle(splitter.begin(), splitter.end(), engine);  // :543
                        uint32_t* row = &bits[(c * (size_t)nm + (size_t)j) * bit_words];
                        for (size_t i = 0; i < nb; ++i)
                            if (splitter[i]) row[(off + i) >> 5] |= 1u << ((off + i) & 31);
                    }
                }
                off += nb;
            }
            std::stringstream ss;
            ss << engine;
            for (int i = 0; i < 624; ++i) ss >> mt_e[c * 624 + i];
            ss >> sc[c].engine_idx;
        }
        SCHK(dalloc(&d_bits, bits.size()));
        SCHK(hipMemcpy(d_bits, bits.data(), sizeof(uint32_t) * bits.size(), hipMemcpyHostToDevice));
        sp.bits = d_bits;
        sp.bit_words = (uint32_t)bit_words;
    }

    // trials in batches of at most ~256 MB of counts
    const size_t per_trial = C * k_type * k_oth * sizeof(int32_t);
    if (per_trial > ((size_t)2 << 30)) {
        cleanup();
        return fail(h, BISBM_ERR_UNSUPPORTED, "agg_split: the edge counts of one trial take %.1f GB (%zu chains x %zu x %zu blocks); use fewer chains",
                    (double)per_trial / 1e9, C, k_type, k_oth);
    }
    const size_t batch = std::max<size_t>(1, std::min<size_t>((size_t)nm, ((size_t)256 << 20) / std::max<size_t>(per_trial, 1)));
    SCHK(dalloc(&d_out_k, C * batch * k_type * k_oth));
    SCHK(dalloc(&d_out_deg, C * batch * k_type));
    sp.out_k = d_out_k;
    sp.out_deg = d_out_deg;
    std::vector<int32_t> out_k(C * batch * k_type * k_oth), out_deg(C * batch * k_type);
    struct Best {
        double dS = std::numeric_limits<double>::infinity();
        uint32_t block = 0, trial = 0;
        bool found = false;
    };
    std::vector<Best> best(C);
    const std::vector<double>& lg = h->tab->lg;
    for (size_t t0 = 0; t0 < (size_t)nm; t0 += batch) {
        const size_t nt = std::min(batch, (size_t)nm - t0);
        sp.trial0 = (uint32_t)t0;
        sp.n_trials = (uint32_t)nt;
        if (h->wide) {  // (counted with global atomics)
            SCHK(hipMemsetAsync(d_out_k, 0, sizeof(int32_t) * C * nt * k_type * k_oth, h->stream));
            SCHK(hipMemsetAsync(d_out_deg, 0, sizeof(int32_t) * C * nt * k_type, h->stream));
        }
        SCHK(launch_split_eval(sp, h->stream));
        SCHK(hipStreamSynchronize(h->stream));
        SCHK(hipMemcpy(out_k.data(), d_out_k, sizeof(int32_t) * C * nt * k_type * k_oth, hipMemcpyDeviceToHost));
        SCHK(hipMemcpy(out_deg.data(), d_out_deg, sizeof(int32_t) * C * nt * k_type, hipMemcpyDeviceToHost));
        auto eval_chain = [&](size_t c) {
            for (size_t b = 0; b < k_type; ++b) {
                if (n_r[c * K + b_lo + b] <= 1) continue;
                for (size_t j = 0; j < nt; ++j) {
                    const int32_t* k = &out_k[((c * nt + j) * k_type + b) * k_oth];
                    const int deg = out_deg[(c * nt + j) * k_type + b];
                    // compute_dS(size_t mb, vector<bool>&), :404-423
                    double entropy0 = 0., entropy1 = 0.;
                    for (size_t t = 0; t < k_oth; ++t) {
                        const int m_rt = type ? quad[(c * ka + t) * kb + b] : quad[(c * ka + b) * kb + t];
                        entropy0 -= lg[(size_t)(m_rt + 1)];
                        entropy1 -= lg[(size_t)(m_rt - k[t] + 1)];
                        entropy1 -= lg[(size_t)(k[t] + 1)];
                    }
                    const int m0r = m_r[c * K + b_lo + b];
                    entropy0 -= -lg[(size_t)(m0r + 1)];
                    entropy1 -= -lg[(size_t)(m0r - deg + 1)];
                    entropy1 -= -lg[(size_t)(deg + 1)];
                    const double dS = entropy1 - entropy0;
                    Best& B = best[c];
                    const uint32_t trial = (uint32_t)(t0 + j);
                    const bool better = dS < B.dS || (dS == B.dS && B.found && (b < B.block || (b == B.block && trial < B.trial)));
                    if (better) {
                        B.dS = dS;
                        B.block = (uint32_t)b;
                        B.trial = trial;
                        B.found = true;
                    }
                }
            }
        };
        if (!for_each_chain(C, eval_chain)) {  // chains are independent
            cleanup();
            return fail(h, BISBM_ERR_STATE, "agg_split: host-side evaluation failed (out of memory?)");
        }
    }
    std::vector<uint32_t> chosen(2 * C);
    for (size_t c = 0; c < C; ++c) {
        if (!best[c].found) {  // every dS was +inf or NaN: cannot happen with finite tables
            cleanup();
            return fail(h, BISBM_ERR_STATE, "chain %zu: no finite split dS", c);
        }
        chosen[2 * c] = best[c].block;
        chosen[2 * c + 1] = best[c].trial;
    }
    SCHK(hipMemcpy(d_chosen, chosen.data(), sizeof(uint32_t) * chosen.size(), hipMemcpyHostToDevice));
    sp.chosen = d_chosen;
    SCHK(launch_split_apply(sp, h->stream));
    applied = true;
    SCHK(hipStreamSynchronize(h->stream));
    for (size_t c = 0; c < C; ++c) sc[c].split_epoch += 1;
    SCHK(hipMemcpy(h->d_scalars, sc.data(), sizeof(ChainScalars) * C, hipMemcpyHostToDevice));
    if (compat) SCHK(hipMemcpy(h->d_mt_engine, mt_e.data(), sizeof(uint32_t) * mt_e.size(), hipMemcpyHostToDevice));
#undef SCHK
    cleanup();

    // one block more: the block-state arrays grow with K
    if (type)
        h->kb += 1;
    else
        h->ka += 1;
    h->K = h->ka + h->kb;
    h->state_ready = false;  // (until the rebuild below has run: a failed allocation must not leave a usable-looking handle)
    if (h->ka > h->cap_ka || h->kb > h->cap_kb) {
        h->cap_ka = std::max(h->cap_ka, h->ka);
        h->cap_kb = std::max(h->cap_kb, h->kb);
        const size_t capK = (size_t)h->cap_ka + h->cap_kb, D = (size_t)h->maxdeg + 1;
        for (void* p : {(void*)h->d_m, (void*)h->d_m_r, (void*)h->d_n_r, (void*)h->d_eta}) (void)hipFree(p);
        h->d_m = nullptr, h->d_m_r = nullptr, h->d_n_r = nullptr, h->d_eta = nullptr;
        HIPCHK(h, dalloc(&h->d_m, C * h->cap_ka * h->cap_kb));
        HIPCHK(h, dalloc(&h->d_m_r, C * capK));
        HIPCHK(h, dalloc(&h->d_n_r, C * capK));
        HIPCHK(h, dalloc(&h->d_eta, C * capK * D));
    }
    return rebuild_state(h);  // compute_n_r / k / m / m_r / eta_rk at the end of apply_split_moves (:454-458)
}

// which: 0 = agg_merge(diff_a, diff_b, nm), 1 = agg_merge(diff, nm)
constexpr int kDiverged = 1000;  // run_merges: the chains ended with different block counts (labels relabelled, state NOT rebuilt)

int run_merges(bisbm_engine* h, int which, int diff_a, int diff_b, int nm, std::vector<std::pair<size_t, size_t>>* ends_out = nullptr) {
    if (!h->state_ready) return fail(h, BISBM_ERR_STATE, "call bisbm_init or bisbm_shuffle before bisbm_agg_merge");
    if (nm < 1) return fail(h, BISBM_ERR_INVALID_ARG, "nm must be >= 1");
    if (which == 0) {  // blockmodel.cc:110-117: negative diffs are splits, one block at a time, type a first
        while (diff_a < 0) {
            const int rc = run_split(h, 0, nm);
            if (rc) return rc;
            ++diff_a;
        }
        while (diff_b < 0) {
            const int rc = run_split(h, 1, nm);
            if (rc) return rc;
            ++diff_b;
        }
        if (diff_a + diff_b == 0) return BISBM_OK;  // :118-120
    } else if (diff_a < 0) {
        return fail(h, BISBM_ERR_INVALID_ARG, "agg_merge(engine, diff, nm) takes diff >= 0 (blockmodel.cc:208-271 has no split branch)");
    }
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const size_t C = h->n_chains, K0 = h->K, ka0 = h->ka, kb0 = h->kb;

    // first node of every label, all chains
    // (label maps hold one entry per block before the call, padded to a multiple of 256; bytes, or two bytes when wide)
    const size_t L = (K0 + 255) & ~(size_t)255, lb = h->lbytes();
    uint8_t* d_map = nullptr;
    uint32_t* d_first = nullptr;
    HIPCHK(h, dalloc(&d_map, C * L * lb));
    HIPCHK(h, dalloc(&d_first, C * L));
    std::vector<uint8_t> ident(C * L * lb);
    for (size_t i = 0; i < C * L; ++i) {
        const uint16_t l = (uint16_t)(i % L);
        if (h->wide)
            std::memcpy(&ident[2 * i], &l, 2);
        else
            ident[i] = (uint8_t)l;
    }
    auto cleanup = [&]() {
        (void)hipFree(d_map);
        (void)hipFree(d_first);
    };
#define MCHK(expr)                                                                                     \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            cleanup();                                                                                 \
            return fail(h, BISBM_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));                     \
        }                                                                                              \
    } while (0)
    MCHK(hipMemcpy(d_map, ident.data(), ident.size(), hipMemcpyHostToDevice));
    // (on the handle's stream: it is a non-blocking stream, which a null-stream memset is not ordered with)
    MCHK(hipMemsetAsync(d_first, 0xff, sizeof(uint32_t) * C * L, h->stream));
    MCHK(launch_merge_first(h->d_labels, h->wide, h->label_stride, (uint32_t)h->n, h->n_chains, (uint32_t)L, d_map, d_first, h->stream));
    MCHK(hipStreamSynchronize(h->stream));
    std::vector<uint32_t> first(C * L);
    MCHK(hipMemcpy(first.data(), d_first, sizeof(uint32_t) * first.size(), hipMemcpyDeviceToHost));
    std::vector<int32_t> quad(C * ka0 * kb0);
    MCHK(hipMemcpy(quad.data(), h->d_m, sizeof(int32_t) * quad.size(), hipMemcpyDeviceToHost));
    std::vector<ChainScalars> sc(C);
    MCHK(hipMemcpy(sc.data(), h->d_scalars, sizeof(ChainScalars) * C, hipMemcpyDeviceToHost));
    std::vector<uint32_t> mt_e, mt_g;
    const bool compat = h->rng_mode == BISBM_RNG_MT19937_COMPAT;
    if (compat) {
        mt_e.resize(C * 624);
        mt_g.resize(C * 624);
        MCHK(hipMemcpy(mt_e.data(), h->d_mt_engine, sizeof(uint32_t) * mt_e.size(), hipMemcpyDeviceToHost));
        MCHK(hipMemcpy(mt_g.data(), h->d_mt_gen, sizeof(uint32_t) * mt_g.size(), hipMemcpyDeviceToHost));
    }
    auto load_mt = [](std::mt19937& g, const uint32_t* st, uint32_t pos) {  // libstdc++ textual state: 624 words, position
        std::stringstream ss;
        for (int i = 0; i < 624; ++i) ss << st[i] << ' ';
        ss << pos;
        ss >> g;
    };
    auto store_mt = [](const std::mt19937& g, uint32_t* st, uint32_t& pos) {
        std::stringstream ss;
        ss << g;
        for (int i = 0; i < 624; ++i) ss >> st[i];
        ss >> pos;
    };

    std::vector<uint16_t> fmap(C * L, 0);
    // the chains are independent: K-scale selection per chain, spread over the host's threads
    std::vector<int> rcs(C, 0);
    std::vector<std::pair<size_t, size_t>> ends(C);
    auto one_chain = [&](size_t c) {
        MergeChain mc;
        mc.K = K0, mc.ka = ka0, mc.kb = kb0, mc.na = (size_t)h->na;
        mc.M.assign(K0 * K0, 0);
        for (size_t a = 0; a < ka0; ++a)
            for (size_t b = 0; b < kb0; ++b) {
                const int v = quad[(c * ka0 + a) * kb0 + b];
                mc.M[a * K0 + ka0 + b] = v;
                mc.M[(ka0 + b) * K0 + a] = v;
            }
        mc.m_r.assign(K0, 0);
        for (size_t i = 0; i < K0; ++i)
            for (size_t j = 0; j < K0; ++j) mc.m_r[i] += mc.M[i * K0 + j];
        mc.first.assign(first.begin() + c * L, first.begin() + c * L + K0);
        mc.cmap.assign(L, 0xffff);
        for (size_t i = 0; i < K0; ++i) mc.cmap[i] = (uint16_t)i;
        mc.compat = compat;
        mc.seed = h->seed;
        mc.chain_gid = h->gid(c);
        mc.epoch = sc[c].merge_epoch;
        mc.epsilon = h->epsilon;
        mc.lg = &h->tab->lg;
        if (compat) {
            load_mt(mc.engine, &mt_e[c * 624], sc[c].engine_idx);
            load_mt(mc.gen, &mt_g[c * 624], sc[c].gen_idx);
        }
        // (the reference renumbers by first appearance on every apply_block_moves, also when nothing merged)
        rcs[c] = which == 0 ? mc.agg_merge(diff_a, diff_b, nm) : mc.agg_merge_total(diff_a, nm);
        if (rcs[c] != 0) return;
        ends[c] = {mc.ka, mc.kb};
        for (size_t i = 0; i < L; ++i) fmap[c * L + i] = mc.cmap[i] == 0xffff ? 0 : mc.cmap[i];
        sc[c].merge_epoch = mc.epoch;
        if (compat) {
            store_mt(mc.engine, &mt_e[c * 624], sc[c].engine_idx);
            store_mt(mc.gen, &mt_g[c * 624], sc[c].gen_idx);
        }
    };
    if (!for_each_chain(C, one_chain)) {
        cleanup();
        return fail(h, BISBM_ERR_STATE, "agg_merge: host-side selection failed (out of memory? a chain's merge state is K x K integers)");
    }
    for (size_t c = 0; c < C; ++c)
        if (rcs[c] != 0) {
            cleanup();
            return fail(h, BISBM_ERR_STATE,
                        rcs[c] == -3 ? "chain %zu: agg_merge cannot reach the requested block counts (the reference would recurse without end)"
                                     : "chain %zu: block renumbering inconsistent (the reference's sanity check, blockmodel.cc:605-609)",
                        c);
        }
    const size_t nka = ends[0].first, nkb = ends[0].second;
    bool diverged = false;
    for (size_t c = 1; c < C; ++c)
        if (ends[c] != ends[0]) {
            if (!ends_out) {
                cleanup();
                return fail(h, BISBM_ERR_STATE,
                            "chains ended with different block counts (chain 0: %zu+%zu, chain %zu: %zu+%zu); one (Ka,Kb) per handle",
                            nka, nkb, c, ends[c].first, ends[c].second);
            }
            diverged = true;
        }
    if (h->wide) {
        MCHK(hipMemcpy(d_map, fmap.data(), sizeof(uint16_t) * fmap.size(), hipMemcpyHostToDevice));
    } else {
        std::vector<uint8_t> fmap8(fmap.begin(), fmap.end());
        MCHK(hipMemcpy(d_map, fmap8.data(), fmap8.size(), hipMemcpyHostToDevice));
    }
    MCHK(launch_merge_relabel(h->d_labels, h->wide, h->label_stride, (uint32_t)h->n, h->n_chains, (uint32_t)L, d_map, h->stream));
    MCHK(hipMemcpy(h->d_scalars, sc.data(), sizeof(ChainScalars) * C, hipMemcpyHostToDevice));
    if (compat) {
        MCHK(hipMemcpy(h->d_mt_engine, mt_e.data(), sizeof(uint32_t) * mt_e.size(), hipMemcpyHostToDevice));
        MCHK(hipMemcpy(h->d_mt_gen, mt_g.data(), sizeof(uint32_t) * mt_g.size(), hipMemcpyHostToDevice));
    }
    MCHK(hipStreamSynchronize(h->stream));
#undef MCHK
    cleanup();
    if (diverged) {  // every chain's labels are in its own new numbering; the caller regroups the chains by shape
        *ends_out = ends;
        h->state_ready = false;
        return kDiverged;
    }
    h->ka = (uint32_t)nka;
    h->kb = (uint32_t)nkb;
    h->K = h->ka + h->kb;
    if (h->wide && h->K <= 256) {
        // the merges have brought the partition into the byte-label range: from here on the ordinary kernels run
        uint8_t* narrow = nullptr;
        HIPCHK(h, dalloc(&narrow, C * h->label_stride));
        hipError_t e = launch_labels_narrow(h->d_labels, narrow, h->label_stride, (uint32_t)h->n, h->n_chains, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) {
            (void)hipFree(narrow);
            return fail(h, BISBM_ERR_HIP, "labels_narrow: %s", hipGetErrorString(e));
        }
        (void)hipFree(h->d_labels);
        h->d_labels = narrow;
        h->wide = false;  // (d_labels_tmp, Philox mode, keeps its two-byte size: large enough for either format)
    }
    return rebuild_state(h);  // init_bisbm() at the end of apply_block_moves (:610)
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// chains with different block counts: sub-engines per shape (see bisbm_engine::groups)
// ---------------------------------------------------------------------------------------------
namespace {

// a sub-engine of `root` for `count` chains of shape (ka, kb): own per-chain arrays, stream and events; graph and tables borrowed
bisbm_engine* new_group(bisbm_engine* root, uint32_t ka, uint32_t kb, uint32_t count, std::string& err) {
    std::unique_ptr<bisbm_engine> gp(new bisbm_engine());
    bisbm_engine* g = gp.get();
    g->root = root;
    g->device = root->device;
    g->n = root->n, g->na = root->na, g->nb = root->nb, g->num_edges = root->num_edges, g->nnz = root->nnz;
    g->ka = ka, g->kb = kb, g->K = ka + kb, g->maxdeg = root->maxdeg, g->n_chains = count;
    g->cap_ka = ka, g->cap_kb = kb;
    g->wide = g->K > 256;
    g->epsilon = root->epsilon, g->rng_mode = root->rng_mode, g->seed = root->seed, g->gen_seed = root->gen_seed;
    g->label_stride = root->label_stride;
    forget_pass_speeds(g);  // (another shape: measured afresh)
    g->d_rowptr = root->d_rowptr, g->d_col = root->d_col, g->d_lgamma = root->d_lgamma, g->d_logtab = root->d_logtab, g->d_q = root->d_q;
    g->tab = root->tab, g->q_stride = root->q_stride, g->ent_deg = root->ent_deg, g->ent_multi = root->ent_multi;
    g->deg_count = root->deg_count;
    const size_t C = count, K = g->K, D = (size_t)g->maxdeg + 1;
    hipError_t e = hipStreamCreateWithFlags(&g->own_stream, hipStreamNonBlocking);
    g->stream = g->own_stream;
    if (e == hipSuccess) e = hipEventCreate(&g->ev0);
    if (e == hipSuccess) e = hipEventCreate(&g->ev1);
    if (e == hipSuccess) e = dalloc(&g->d_labels, C * g->label_stride * g->lbytes());
    if (e == hipSuccess) e = dalloc(&g->d_m, C * ka * kb);
    if (e == hipSuccess) e = dalloc(&g->d_m_r, C * K);
    if (e == hipSuccess) e = dalloc(&g->d_n_r, C * K);
    if (e == hipSuccess) e = dalloc(&g->d_eta, C * K * D);
    if (e == hipSuccess) e = dalloc(&g->d_scalars, C);
    if (e == hipSuccess) e = dalloc(&g->d_tmp_f64, C);
    if (e == hipSuccess) e = dalloc(&g->d_stage_u32, (size_t)g->n);
    if (e == hipSuccess) e = dalloc(&g->d_gids, C);
    if (e == hipSuccess && g->rng_mode == BISBM_RNG_MT19937_COMPAT) {
        e = dalloc(&g->d_vlist, C * g->n);
        if (e == hipSuccess) e = dalloc(&g->d_mt_engine, C * 624);
        if (e == hipSuccess) e = dalloc(&g->d_mt_gen, C * 624);
    } else if (e == hipSuccess) {
        e = dalloc(&g->d_labels_tmp, C * g->label_stride * g->lbytes());
    }
    if (e != hipSuccess) {
        err = std::string("sub-engine allocation: ") + hipGetErrorString(e);
        free_all(g);
        return nullptr;
    }
    return gp.release();
}

// The chains of `src` (labels already in each chain's own new numbering, shapes in `ends`) go to new sub-engines of
// `root`, one per distinct shape in order of first appearance; every chain keeps its generator state, counters and
// global id.  The new engines are appended to `out` with their state rebuilt.
int split_by_shape(bisbm_engine* root, bisbm_engine* src, const std::vector<std::pair<size_t, size_t>>& ends,
                   std::vector<bisbm_engine*>& out) {
    std::vector<std::pair<size_t, size_t>> shapes;
    for (auto const& e : ends)
        if (std::find(shapes.begin(), shapes.end(), e) == shapes.end()) shapes.push_back(e);
    const size_t n = (size_t)src->n;
    for (auto const& shape : shapes) {
        std::vector<uint32_t> members;
        for (size_t c = 0; c < ends.size(); ++c)
            if (ends[c] == shape) members.push_back((uint32_t)c);
        std::string err;
        bisbm_engine* g = new_group(root, (uint32_t)shape.first, (uint32_t)shape.second, (uint32_t)members.size(), err);
        if (!g) return fail(root, BISBM_ERR_HIP, "%s", err.c_str());
        out.push_back(g);
        for (size_t j = 0; j < members.size(); ++j) {
            const size_t c = members[j];
            g->gids.push_back(src->gid(c));
            g->ridx.push_back(src->ridx.empty() ? (uint32_t)c : src->ridx[c]);
            const uint8_t* from = src->d_labels + c * src->label_stride * src->lbytes();
            uint8_t* to = g->d_labels + j * g->label_stride * g->lbytes();
            hipError_t e;
            if (src->wide == g->wide)
                e = hipMemcpyAsync(to, from, n * g->lbytes(), hipMemcpyDeviceToDevice, g->stream);
            else  // (merges only lower K: a wide source, a byte-label destination)
                e = launch_labels_narrow(from, to, g->label_stride, (uint32_t)n, 1, g->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(g->d_scalars + j, src->d_scalars + c, sizeof(ChainScalars), hipMemcpyDeviceToDevice, g->stream);
            if (e == hipSuccess && g->rng_mode == BISBM_RNG_MT19937_COMPAT) {
                e = hipMemcpyAsync(g->d_mt_engine + j * 624, src->d_mt_engine + c * 624, sizeof(uint32_t) * 624, hipMemcpyDeviceToDevice, g->stream);
                if (e == hipSuccess) e = hipMemcpyAsync(g->d_mt_gen + j * 624, src->d_mt_gen + c * 624, sizeof(uint32_t) * 624, hipMemcpyDeviceToDevice, g->stream);
                if (e == hipSuccess) e = hipMemcpyAsync(g->d_vlist + j * n, src->d_vlist + c * n, sizeof(uint32_t) * n, hipMemcpyDeviceToDevice, g->stream);
            }
            if (e != hipSuccess) return fail(root, BISBM_ERR_HIP, "moving chain %zu to its group: %s", c, hipGetErrorString(e));
        }
        hipError_t e = hipMemcpyAsync(g->d_gids, g->gids.data(), sizeof(uint32_t) * g->gids.size(), hipMemcpyHostToDevice, g->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
        if (e != hipSuccess) return fail(root, BISBM_ERR_HIP, "group setup: %s", hipGetErrorString(e));
        const int rc = rebuild_state(g);
        if (rc) return fail(root, rc, "%s", g->err.c_str());
    }
    return BISBM_OK;
}

void remap_groups(bisbm_engine* root) {
    root->where.assign(root->n_chains, {0u, 0u});
    for (size_t gi = 0; gi < root->groups.size(); ++gi)
        for (size_t j = 0; j < root->groups[gi]->ridx.size(); ++j) root->where[root->groups[gi]->ridx[j]] = {(uint32_t)gi, (uint32_t)j};
}

// agg_merge(engine, diff, nm) on a handle whose chains may end (or already live) in different shapes
int merge_total_grouped(bisbm_engine* h, int diff, int nm) {
    std::vector<std::pair<size_t, size_t>> ends;
    if (h->groups.empty()) {
        const int rc = run_merges(h, 1, diff, 0, nm, &ends);
        if (rc != kDiverged) return rc;
        std::vector<bisbm_engine*> fresh;
        const int rc2 = split_by_shape(h, h, ends, fresh);
        if (rc2) {
            for (bisbm_engine* g : fresh) {
                free_all(g);
                delete g;
            }
            return rc2;
        }
        h->groups = fresh;
        free_chain_arrays(h);  // the chains live in the groups now
        h->state_ready = true;
        remap_groups(h);
        return BISBM_OK;
    }
    for (bisbm_engine* g : h->groups)  // (all or nothing, as in bisbm_agg_merge)
        if (diff > (int)g->ka + (int)g->kb - 2)
            return fail(h, BISBM_ERR_STATE, "agg_merge(%d): a chain of this handle has %u + %u blocks", diff, g->ka, g->kb);
    std::vector<bisbm_engine*> next;
    int rc_all = BISBM_OK;
    for (bisbm_engine* g : h->groups) {
        if (rc_all) {
            next.push_back(g);
            continue;
        }
        const int rc = run_merges(g, 1, diff, 0, nm, &ends);
        if (rc == BISBM_OK) {
            next.push_back(g);
        } else if (rc == kDiverged) {
            // the new groups first; `g` goes only once every one of its chains has a new home.  If that fails part-way (an
            // allocation), the half-made groups are dropped and `g` stays in the handle -- its chains hold their merged labels
            // under the old shape -- with its block state marked stale, so nothing addresses a chain that no longer exists
            std::vector<bisbm_engine*> fresh;
            rc_all = split_by_shape(h, g, ends, fresh);
            if (rc_all == BISBM_OK) {
                next.insert(next.end(), fresh.begin(), fresh.end());
                free_all(g);
                delete g;
            } else {
                for (bisbm_engine* f : fresh) {
                    free_all(f);
                    delete f;
                }
                g->state_ready = false;
                next.push_back(g);
            }
        } else {
            h->err = g->err;
            rc_all = rc;
            next.push_back(g);
        }
    }
    h->groups = next;
    remap_groups(h);
    return rc_all;
}

}  // namespace

int bisbm_agg_merge(bisbm_handle h, int diff_a, int diff_b, int nm) {
    if (!h) return BISBM_ERR_INVALID_ARG;
    if (!h->devs.empty()) {
        // (all or nothing, as for one device: a request a chain of some device cannot meet is refused before any device changes)
        for (bisbm_engine* d : h->devs)
            for (bisbm_engine* g : d->groups.empty() ? std::vector<bisbm_engine*>{d} : d->groups)
                if (diff_a >= (int)g->ka || diff_b >= (int)g->kb)
                    return fail(h, BISBM_ERR_STATE, "agg_merge(%d, %d): a chain of this handle has %u + %u blocks", diff_a, diff_b, g->ka, g->kb);
        const int rc = on_devices(h, [&](bisbm_engine* d, size_t) { return bisbm_agg_merge(d, diff_a, diff_b, nm); });
        (void)multi_common_shape(h, nullptr, nullptr);
        return rc;
    }
    if (!h->groups.empty()) {  // the same change of counts in every group: each keeps one shape
        for (bisbm_engine* g : h->groups)  // (all or nothing: a request no chain of some group can meet is refused before any group changes)
            if (diff_a >= (int)g->ka || diff_b >= (int)g->kb)
                return fail(h, BISBM_ERR_STATE, "agg_merge(%d, %d): a chain of this handle has %u + %u blocks", diff_a, diff_b, g->ka, g->kb);
        for (bisbm_engine* g : h->groups) {
            const int rc = run_merges(g, 0, diff_a, diff_b, nm);
            if (rc) {
                h->err = g->err;
                return rc;
            }
        }
        return BISBM_OK;
    }
    return run_merges(h, 0, diff_a, diff_b, nm);
}

int bisbm_agg_merge_total(bisbm_handle h, int diff, int nm) {
    if (!h) return BISBM_ERR_INVALID_ARG;
    if (!h->devs.empty()) {
        for (bisbm_engine* d : h->devs)
            for (bisbm_engine* g : d->groups.empty() ? std::vector<bisbm_engine*>{d} : d->groups)
                if (diff > (int)g->ka + (int)g->kb - 2)
                    return fail(h, BISBM_ERR_STATE, "agg_merge(%d): a chain of this handle has %u + %u blocks", diff, g->ka, g->kb);
        const int rc = on_devices(h, [&](bisbm_engine* d, size_t) { return bisbm_agg_merge_total(d, diff, nm); });
        (void)multi_common_shape(h, nullptr, nullptr);
        return rc;
    }
    return merge_total_grouped(h, diff, nm);
}

int bisbm_get_ka_kb_chain(bisbm_handle h, uint32_t chain, uint32_t* ka, uint32_t* kb) {
    if (!h) return BISBM_ERR_INVALID_ARG;
    if (chain >= h->n_chains) return fail(h, BISBM_ERR_INVALID_ARG, "chain out of range");
    if (!h->devs.empty()) {
        uint32_t local;
        bisbm_engine* d = h->devs[dev_of_chain(h, chain, &local)];
        return bisbm_get_ka_kb_chain(d, local, ka, kb);
    }
    const bisbm_engine* e = h->groups.empty() ? h : h->groups[h->where[chain].first];
    if (ka) *ka = e->ka;
    if (kb) *kb = e->kb;
    return BISBM_OK;
}
