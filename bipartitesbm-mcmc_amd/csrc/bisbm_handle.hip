// bisbm_handle.hip -- host side of the C ABI (include/bisbm.h): the handle, device memory, labels in and out, state build,
// getters, entropy().  No CPU compute path exists here: every operation on chain state runs in a HIP kernel
// (bisbm_kernels.hip), and bisbm_create fails without a device.
//
// Reference lines cited as <file>:<line> relative to /root/reference/src.
#include "bisbm_engine.hpp"

using namespace bisbm;

namespace bisbm {

thread_local std::string g_create_error;

int fail(bisbm_engine* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (h)
        h->err = buf;
    else
        g_create_error = buf;
    return code;
}

void free_chain_arrays(bisbm_engine* h) {
    void** ptrs[] = {(void**)&h->d_labels, (void**)&h->d_labels_tmp, (void**)&h->d_vlist, (void**)&h->d_m, (void**)&h->d_m_r,
                     (void**)&h->d_n_r, (void**)&h->d_eta, (void**)&h->d_scalars, (void**)&h->d_mt_engine, (void**)&h->d_mt_gen,
                     (void**)&h->d_tmp_f64, (void**)&h->d_counts, (void**)&h->d_gids, (void**)&h->d_ent_prev};
    h->ent_prev_valid = false;
    for (void** p : ptrs)
        if (*p) {
            (void)hipFree(*p);
            *p = nullptr;
        }
}

void free_all(bisbm_engine* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    for (bisbm_engine* g : h->groups) {
        free_all(g);
        delete g;
    }
    h->groups.clear();
    free_chain_arrays(h);
    if (h->root) {  // a sub-engine: the graph and the tables belong to the handle it serves
        h->d_rowptr = nullptr, h->d_col = nullptr, h->d_lgamma = nullptr, h->d_logtab = nullptr, h->d_q = nullptr;
    }
    void* ptrs[] = {h->d_rowptr, h->d_col, h->d_lgamma, h->d_logtab, h->d_q, h->d_T, h->d_stage_u32, h->d_simd_claims};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
}

void mt_seed_host(uint32_t* mt, uint64_t seed) {  // std::mt19937(seed): seed mod 2^32
    mt[0] = (uint32_t)seed;
    for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
}

int rebuild_state(bisbm_engine* h) {
    BuildParams bp{};
    bp.rowptr = h->d_rowptr;
    bp.col = h->d_col;
    bp.n = (uint32_t)h->n;
    bp.na = (uint32_t)h->na;
    bp.ka = h->ka;
    bp.kb = h->kb;
    bp.maxdeg = h->maxdeg;
    bp.n_chains = h->n_chains;
    bp.labels = h->d_labels;
    bp.label_stride = h->label_stride;
    bp.wide = h->wide ? 1u : 0u;
    bp.m = h->d_m;
    bp.m_r = h->d_m_r;
    bp.n_r = h->d_n_r;
    bp.eta = h->d_eta;
    HIPCHK(h, launch_state_build(bp, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->state_ready = true;
    h->ent_prev_valid = false;
    // (a partition put in place from outside -- init, shuffle, merges, splits: the pass depths are measured afresh, see bisbm_anneal)
    forget_pass_speeds(h);
    return BISBM_OK;
}

bool common_shape(bisbm_engine* h) {
    for (bisbm_engine* g : h->groups)
        if (g->ka != h->groups[0]->ka || g->kb != h->groups[0]->kb) return false;
    h->ka = h->groups[0]->ka, h->kb = h->groups[0]->kb, h->K = h->ka + h->kb;
    return true;
}

size_t generic_lds_base_bytes(uint32_t ka, uint32_t kb, bool wide, int rng_mode) {
    const size_t K = (size_t)ka + kb, S = kb | 1u;
    size_t lds = sizeof(int32_t) * ((wide ? 0 : (size_t)ka * S) + 2 * K + std::max<uint32_t>(std::max(ka, kb), 64)) + sizeof(uint32_t) * 64 * 64;
    if (rng_mode == BISBM_RNG_MT19937_COMPAT) lds += sizeof(uint32_t) * 624 * 4;
    return lds;
}

int launch_block_entropy(bisbm_engine* h, double* d_out) {
    EntropyParams ep{};
    ep.ka = h->ka;
    ep.kb = h->kb;
    ep.maxdeg = h->maxdeg;
    ep.n_chains = h->n_chains;
    ep.m = h->d_m;
    ep.m_r = h->d_m_r;
    ep.n_r = h->d_n_r;
    ep.eta = h->d_eta;
    ep.lgamma_tab = h->d_lgamma;
    ep.lgamma_size = h->tab->lg.size();
    ep.q_tab = h->d_q;
    ep.q_stride = h->q_stride;
    ep.log_tab = h->d_logtab;
    ep.out = d_out;
    HIPCHK(h, launch_entropy(ep, h->stream));
    return BISBM_OK;
}

}  // namespace bisbm

extern "C" {

int bisbm_abi_version(void) { return BISBM_ABI_VERSION; }

int bisbm_check_shape(uint32_t ka, uint32_t kb, int rng_mode) {
    if (ka == 0 || kb == 0) return fail(nullptr, BISBM_ERR_INVALID_ARG, "ka and kb must be >= 1");
    if (rng_mode != BISBM_RNG_PHILOX && rng_mode != BISBM_RNG_MT19937_COMPAT)
        return fail(nullptr, BISBM_ERR_INVALID_ARG, "unknown rng_mode %d", rng_mode);
    if ((uint64_t)ka + kb > 65535) return fail(nullptr, BISBM_ERR_UNSUPPORTED, "ka + kb = %llu > 65535 (labels are at most two bytes)", (unsigned long long)ka + kb);
    const bool wide = ka + kb > 256;
    const size_t lds = (generic_lds_base_bytes(ka, kb, wide, rng_mode) + 15) & ~(size_t)15;
    if (wide && lds > kLdsPerCu)
        return fail(nullptr, BISBM_ERR_UNSUPPORTED,
                    "%u + %u blocks: above 256 blocks m_r, n_r and the k_v histogram of a chain stay in LDS and need %zu B here (a CU has %zu); "
                    "the limit is about %u blocks for an even split in this RNG mode", ka, kb, lds, kLdsPerCu,
                    (unsigned)((kLdsPerCu - sizeof(uint32_t) * 64 * 64 - (rng_mode == BISBM_RNG_MT19937_COMPAT ? sizeof(uint32_t) * 624 * 4 : 0)) / 10));
    return BISBM_OK;
}

const char* bisbm_last_error(bisbm_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int bisbm_create(bisbm_handle* out, uint64_t n, uint64_t na, uint64_t nb, const uint64_t* rowptr,
                 const uint32_t* col, uint32_t ka, uint32_t kb, double epsilon, uint32_t n_chains,
                 uint32_t first_chain_id, int device, int rng_mode, uint64_t seed, uint64_t gen_seed) {
    if (!out) return fail(nullptr, BISBM_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (!rowptr || (!col && rowptr[n] != 0)) return fail(nullptr, BISBM_ERR_INVALID_ARG, "rowptr/col is NULL");
    if (n == 0 || na + nb != n) return fail(nullptr, BISBM_ERR_INVALID_ARG, "na + nb must equal n > 0");
    if (ka == 0 || kb == 0) return fail(nullptr, BISBM_ERR_INVALID_ARG, "ka and kb must be >= 1");
    if ((uint64_t)ka + kb > 65535) return fail(nullptr, BISBM_ERR_UNSUPPORTED, "ka + kb = %llu > 65535 (labels are at most two bytes)", (unsigned long long)ka + kb);
    if (rng_mode != BISBM_RNG_PHILOX && rng_mode != BISBM_RNG_MT19937_COMPAT)
        return fail(nullptr, BISBM_ERR_INVALID_ARG, "unknown rng_mode %d", rng_mode);
    if (int rc = bisbm_check_shape(ka, kb, rng_mode)) return rc;  // (the LDS plan of wide mode: refuse here, not at the first anneal)
    if (ka > na || kb > nb) return fail(nullptr, BISBM_ERR_INVALID_ARG, "more blocks than nodes of a type (ka %u / na %llu, kb %u / nb %llu)", ka, (unsigned long long)na, kb, (unsigned long long)nb);
    if (n_chains == 0) return fail(nullptr, BISBM_ERR_INVALID_ARG, "n_chains must be >= 1");
    if (rng_mode != BISBM_RNG_PHILOX && rng_mode != BISBM_RNG_MT19937_COMPAT)
        return fail(nullptr, BISBM_ERR_INVALID_ARG, "unknown rng_mode %d", rng_mode);
    if (n >= 0xFFFFFFFFull || rowptr[n] >= 0xFFFFFFFFull)
        return fail(nullptr, BISBM_ERR_UNSUPPORTED, "more than 2^32-1 nodes or adjacency entries");
    if (rowptr[0] != 0) return fail(nullptr, BISBM_ERR_INVALID_ARG, "rowptr[0] != 0");
    const uint64_t nnz = rowptr[n];
    if (nnz % 2) return fail(nullptr, BISBM_ERR_INVALID_ARG, "odd number of adjacency entries");
    uint32_t maxdeg = 0;
    auto deg_count = std::make_shared<std::vector<uint64_t>>(2 * 257, 0);
    for (uint64_t v = 0; v < n; ++v) {
        if (rowptr[v + 1] < rowptr[v]) return fail(nullptr, BISBM_ERR_INVALID_ARG, "rowptr is not monotone");
        maxdeg = std::max<uint32_t>(maxdeg, (uint32_t)(rowptr[v + 1] - rowptr[v]));
        const bool vb = v >= na;
        ++(*deg_count)[(vb ? 257 : 0) + std::min<uint64_t>(rowptr[v + 1] - rowptr[v], 256)];
        for (uint64_t e = rowptr[v]; e < rowptr[v + 1]; ++e) {
            if (col[e] >= n) return fail(nullptr, BISBM_ERR_NOT_BIPARTITE, "neighbour id %u >= n", col[e]);
            if ((col[e] >= na) == vb)
                return fail(nullptr, BISBM_ERR_NOT_BIPARTITE, "edge (%llu,%u) joins two nodes of one type",
                            (unsigned long long)v, col[e]);
        }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, BISBM_ERR_NO_DEVICE, "no HIP device: this engine has no CPU path");
    if (device < 0 || device >= ndev) return fail(nullptr, BISBM_ERR_NO_DEVICE, "device %d out of range (%d devices)", device, ndev);

    std::unique_ptr<bisbm_engine> hp(new bisbm_engine());
    bisbm_engine* h = hp.get();
    h->device = device;
    h->n = n;
    h->na = na;
    h->nb = nb;
    h->nnz = nnz;
    h->num_edges = nnz / 2;
    h->ka = ka;
    h->kb = kb;
    h->K = ka + kb;
    h->maxdeg = maxdeg;
    h->deg_count = deg_count;
    h->n_chains = n_chains;
    h->first_chain_id = first_chain_id;
    h->epsilon = epsilon;
    h->rng_mode = rng_mode;
    h->seed = seed;
    h->gen_seed = gen_seed;
    h->label_stride = (n + 255) & ~(uint64_t)255;
    h->cap_ka = ka;
    h->cap_kb = kb;
    h->wide = ka + kb > 256;

    auto bail = [&](int code) {
        g_create_error = h->err;
        free_all(h);
        return code;
    };
#define CCHK(expr)                                                                      \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) {                                                         \
            fail(h, BISBM_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));             \
            return bail(BISBM_ERR_HIP);                                                 \
        }                                                                               \
    } while (0)

    CCHK(hipSetDevice(device));
    CCHK(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    h->stream = h->own_stream;
    CCHK(hipEventCreate(&h->ev0));
    CCHK(hipEventCreate(&h->ev1));

    // tables: lgamma covers every index the sweep and entropy() can touch
    // (cache.cc:86-91 sizes it 2E+1 and grows on demand; the values are lgamma(i) either way)
    uint64_t lg_size = 2 * h->num_edges + 2;
    lg_size = std::max<uint64_t>(lg_size, n + 3);
    lg_size = std::max<uint64_t>(lg_size, (uint64_t)ka * kb + h->num_edges + 2);
    if (lg_size >= (1ull << 28)) return fail(nullptr, BISBM_ERR_UNSUPPORTED, "graph needs a %llu-entry lgamma table (>= 2^28)", (unsigned long long)lg_size);  // kernels address the tables with 32-bit byte offsets
    const uint32_t kcap = (uint32_t)std::min<uint64_t>(kQNmax, std::max(na, nb) + 1);
    h->tab = get_tables(lg_size, std::max<uint32_t>(kcap, 2));
    h->q_stride = h->tab->kcap + 1;

    const size_t C = n_chains, K = h->K, D = (size_t)maxdeg + 1;
    CCHK(dalloc(&h->d_rowptr, n + 1));
    CCHK(dalloc(&h->d_col, nnz + 4));  // four spare entries: the production kernel reads a row's ids 16 bytes at a time
    CCHK(hipMemset(h->d_col, 0, sizeof(uint32_t) * (nnz + 4)));
    CCHK(dalloc(&h->d_labels, C * h->label_stride * h->lbytes()));
    CCHK(dalloc(&h->d_m, C * ka * kb));
    CCHK(dalloc(&h->d_m_r, C * K));
    CCHK(dalloc(&h->d_n_r, C * K));
    CCHK(dalloc(&h->d_eta, C * K * D));
    CCHK(dalloc(&h->d_scalars, C));
    CCHK(dalloc(&h->d_lgamma, h->tab->lg.size()));
    CCHK(dalloc(&h->d_logtab, h->tab->lo.size()));
    CCHK(dalloc(&h->d_q, h->tab->q.size()));
    CCHK(dalloc(&h->d_tmp_f64, C));
    CCHK(dalloc(&h->d_stage_u32, n));
    if (rng_mode == BISBM_RNG_MT19937_COMPAT) {
        CCHK(dalloc(&h->d_vlist, C * n));
        CCHK(dalloc(&h->d_mt_engine, C * 624));
        CCHK(dalloc(&h->d_mt_gen, C * 624));
    } else {
        CCHK(dalloc(&h->d_labels_tmp, C * h->label_stride * h->lbytes()));
    }

    {
        std::vector<uint32_t> rp32(n + 1);
        for (uint64_t v = 0; v <= n; ++v) rp32[v] = (uint32_t)rowptr[v];
        CCHK(hipMemcpy(h->d_rowptr, rp32.data(), sizeof(uint32_t) * (n + 1), hipMemcpyHostToDevice));
        if (nnz) CCHK(hipMemcpy(h->d_col, col, sizeof(uint32_t) * nnz, hipMemcpyHostToDevice));
        CCHK(hipMemcpy(h->d_lgamma, h->tab->lg.data(), sizeof(double) * h->tab->lg.size(), hipMemcpyHostToDevice));
        CCHK(hipMemcpy(h->d_logtab, h->tab->lo.data(), sizeof(double) * h->tab->lo.size(), hipMemcpyHostToDevice));
        CCHK(hipMemcpy(h->d_q, h->tab->q.data(), sizeof(double) * h->tab->q.size(), hipMemcpyHostToDevice));
        CCHK(hipMemset(h->d_labels, 0, C * h->label_stride * h->lbytes()));
        std::vector<ChainScalars> sc(C);
        for (auto& s : sc) {
            std::memset(&s, 0, sizeof(s));
            s.engine_idx = 624;
            s.gen_idx = 624;
        }
        CCHK(hipMemcpy(h->d_scalars, sc.data(), sizeof(ChainScalars) * C, hipMemcpyHostToDevice));
        if (rng_mode == BISBM_RNG_MT19937_COMPAT) {
            std::vector<uint32_t> st(C * 624), vl(C * n);
            for (size_t c = 0; c < C; ++c) mt_seed_host(&st[c * 624], seed + first_chain_id + c);
            CCHK(hipMemcpy(h->d_mt_engine, st.data(), sizeof(uint32_t) * st.size(), hipMemcpyHostToDevice));
            for (size_t c = 0; c < C; ++c) mt_seed_host(&st[c * 624], gen_seed + first_chain_id + c);
            CCHK(hipMemcpy(h->d_mt_gen, st.data(), sizeof(uint32_t) * st.size(), hipMemcpyHostToDevice));
            for (size_t c = 0; c < C; ++c)
                for (uint64_t v = 0; v < n; ++v) vl[c * n + v] = (uint32_t)v;  // blockmodel.cc:41
            CCHK(hipMemcpy(h->d_vlist, vl.data(), sizeof(uint32_t) * vl.size(), hipMemcpyHostToDevice));
        }
    }
#undef CCHK

    // chain-independent terms of entropy() (blockmodel.cc:755-757,772-779), reference order
    {
        double ent = 0;
        for (uint64_t v = 0; v < n; ++v) ent -= h_lgamma_fast(*h->tab, (rowptr[v + 1] - rowptr[v]) + 1);
        h->ent_deg = ent;
        double mul = 0;
        std::vector<uint32_t> tmp;
        for (uint64_t y = 0; y < n; ++y) {
            const uint64_t d = rowptr[y + 1] - rowptr[y];
            if (d < 2) continue;
            tmp.assign(col + rowptr[y], col + rowptr[y + 1]);
            std::sort(tmp.begin(), tmp.end());
            for (size_t i = 0; i < tmp.size();) {
                size_t j = i;
                while (j < tmp.size() && tmp[j] == tmp[i]) ++j;
                if (j - i > 1 && y > tmp[i]) mul += h_lgamma_fast(*h->tab, (j - i) + 1);
                i = j;
            }
        }
        h->ent_multi = mul;
    }
    *out = hp.release();
    return BISBM_OK;
}

int bisbm_destroy(bisbm_handle h) {
    if (!h) return BISBM_ERR_INVALID_ARG;
    if (!h->devs.empty()) multi_free(h);
    free_all(h);
    delete h;
    return BISBM_OK;
}

int bisbm_set_stream(bisbm_handle h, void* hip_stream) {
    if (!h) return BISBM_ERR_INVALID_ARG;
    if (!h->devs.empty()) return fail(h, BISBM_ERR_UNSUPPORTED, "a handle over several devices runs every device on a stream of its own");
    for (bisbm_engine* g : h->groups) g->stream = hip_stream ? (hipStream_t)hip_stream : g->own_stream;
    h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    return BISBM_OK;
}

int bisbm_set_memberships(bisbm_handle h, int64_t chain, const uint32_t* labels) {
    if (!h) return BISBM_ERR_INVALID_ARG;
    if (!labels) return fail(h, BISBM_ERR_INVALID_ARG, "labels is NULL");
    if (chain != BISBM_ALL_CHAINS && (chain < 0 || chain >= (int64_t)h->n_chains))
        return fail(h, BISBM_ERR_INVALID_ARG, "chain %lld out of range", (long long)chain);
    if (!h->devs.empty()) {
        if (chain == BISBM_ALL_CHAINS) return on_devices(h, [&](bisbm_engine* d, size_t) { return bisbm_set_memberships(d, BISBM_ALL_CHAINS, labels); });
        uint32_t local;
        bisbm_engine* d = h->devs[dev_of_chain(h, (uint32_t)chain, &local)];
        const int rc = bisbm_set_memberships(d, local, labels);
        if (rc) h->err = d->err;
        return rc;
    }
    if (!h->groups.empty()) {  // (the labels must name blocks of the chain's own shape)
        if (chain == BISBM_ALL_CHAINS) return each_group(h, [&](bisbm_engine* g) { return bisbm_set_memberships(g, BISBM_ALL_CHAINS, labels); });
        bisbm_engine* g = h->groups[h->where[chain].first];
        const int rc = bisbm_set_memberships(g, h->where[chain].second, labels);
        if (rc) h->err = g->err;
        return rc;
    }
    for (uint64_t v = 0; v < h->n; ++v) {
        const uint32_t b = labels[v];
        const bool ok = v < h->na ? b < h->ka : (b >= h->ka && b < h->K);
        if (!ok) return fail(h, BISBM_ERR_INVALID_ARG, "label %u of node %llu is not a block of the node's type", b, (unsigned long long)v);
    }
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpyAsync(h->d_stage_u32, labels, sizeof(uint32_t) * h->n, hipMemcpyHostToDevice, h->stream));
    const uint32_t first = chain == BISBM_ALL_CHAINS ? 0 : (uint32_t)chain;
    const uint32_t cnt = chain == BISBM_ALL_CHAINS ? h->n_chains : 1;
    HIPCHK(h, launch_labels_broadcast(h->d_stage_u32, h->d_labels, h->wide, h->label_stride, (uint32_t)h->n, first, cnt, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->state_ready = false;
    return BISBM_OK;
}

int bisbm_init(bisbm_handle h) {
    if (!h) return BISBM_ERR_INVALID_ARG;
    if (!h->devs.empty()) return on_devices(h, [](bisbm_engine* d, size_t) { return bisbm_init(d); });
    if (!h->groups.empty()) return each_group(h, [](bisbm_engine* g) { return bisbm_init(g); });
    HIPCHK(h, hipSetDevice(h->device));
    return rebuild_state(h);
}

int bisbm_shuffle(bisbm_handle h) {
    if (!h) return BISBM_ERR_INVALID_ARG;
    if (!h->devs.empty()) return on_devices(h, [](bisbm_engine* d, size_t) { return bisbm_shuffle(d); });
    if (!h->groups.empty()) return each_group(h, [](bisbm_engine* g) { return bisbm_shuffle(g); });
    HIPCHK(h, hipSetDevice(h->device));
    ShuffleParams sp{};
    sp.n = (uint32_t)h->n;
    sp.na = (uint32_t)h->na;
    sp.nb = (uint32_t)h->nb;
    sp.n_chains = h->n_chains;
    sp.first_chain_id = h->first_chain_id;
    sp.chain_gids = h->d_gids;
    sp.seed = h->seed;
    sp.labels = h->d_labels;
    sp.labels_old = h->d_labels_tmp;
    sp.label_stride = h->label_stride;
    sp.scalars = h->d_scalars;
    sp.mt_engine = h->d_mt_engine;
    sp.wide = h->wide ? 1u : 0u;
    if (h->rng_mode == BISBM_RNG_PHILOX)
        HIPCHK(h, hipMemcpyAsync(h->d_labels_tmp, h->d_labels, (size_t)h->n_chains * h->label_stride * h->lbytes(),
                                 hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, launch_shuffle(sp, h->rng_mode, h->stream));
    return rebuild_state(h);
}

int bisbm_get_memberships(bisbm_handle h, uint32_t chain, uint32_t* labels_out) {
    if (!h) return BISBM_ERR_INVALID_ARG;
    if (!labels_out || chain >= h->n_chains) return fail(h, BISBM_ERR_INVALID_ARG, "bad chain or NULL output");
    if (!h->devs.empty()) {
        uint32_t local;
        bisbm_engine* d = h->devs[dev_of_chain(h, chain, &local)];
        const int rc = bisbm_get_memberships(d, local, labels_out);
        if (rc) h->err = d->err;
        return rc;
    }
    if (!h->groups.empty()) {
        bisbm_engine* g = h->groups[h->where[chain].first];
        const int rc = bisbm_get_memberships(g, h->where[chain].second, labels_out);
        if (rc) h->err = g->err;
        return rc;
    }
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, launch_labels_widen(h->d_labels + (size_t)chain * h->label_stride * h->lbytes(), h->wide, h->d_stage_u32, (uint32_t)h->n, h->stream));
    HIPCHK(h, hipMemcpyAsync(labels_out, h->d_stage_u32, sizeof(uint32_t) * h->n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return BISBM_OK;
}

int bisbm_get_block_state(bisbm_handle h, uint32_t chain, int32_t* m, int32_t* m_r, int32_t* n_r, uint32_t* eta) {
    if (!h) return BISBM_ERR_INVALID_ARG;
    if (chain >= h->n_chains) return fail(h, BISBM_ERR_INVALID_ARG, "chain out of range");
    if (!h->devs.empty()) {
        uint32_t local;
        bisbm_engine* d = h->devs[dev_of_chain(h, chain, &local)];
        const int rc = bisbm_get_block_state(d, local, m, m_r, n_r, eta);
        if (rc) h->err = d->err;
        return rc;
    }
    if (!h->groups.empty()) {  // (array sizes follow the chain's own shape: bisbm_get_ka_kb_chain)
        bisbm_engine* g = h->groups[h->where[chain].first];
        const int rc = bisbm_get_block_state(g, h->where[chain].second, m, m_r, n_r, eta);
        if (rc) h->err = g->err;
        return rc;
    }
    if (!h->state_ready) return fail(h, BISBM_ERR_STATE, "block state not built yet");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const size_t K = h->K, D = (size_t)h->maxdeg + 1;
    if (m) {
        std::vector<int32_t> quad((size_t)h->ka * h->kb);
        HIPCHK(h, hipMemcpy(quad.data(), h->d_m + (size_t)chain * h->ka * h->kb, sizeof(int32_t) * quad.size(), hipMemcpyDeviceToHost));
        std::memset(m, 0, sizeof(int32_t) * K * K);
        for (uint32_t a = 0; a < h->ka; ++a)
            for (uint32_t b = 0; b < h->kb; ++b) {
                m[a * K + (h->ka + b)] = quad[(size_t)a * h->kb + b];
                m[(h->ka + b) * K + a] = quad[(size_t)a * h->kb + b];
            }
    }
    if (m_r) HIPCHK(h, hipMemcpy(m_r, h->d_m_r + (size_t)chain * K, sizeof(int32_t) * K, hipMemcpyDeviceToHost));
    if (n_r) HIPCHK(h, hipMemcpy(n_r, h->d_n_r + (size_t)chain * K, sizeof(int32_t) * K, hipMemcpyDeviceToHost));
    if (eta) HIPCHK(h, hipMemcpy(eta, h->d_eta + (size_t)chain * K * D, sizeof(uint32_t) * K * D, hipMemcpyDeviceToHost));
    return BISBM_OK;
}

int bisbm_get_cum_dS(bisbm_handle h, double* out) {
    if (!h) return BISBM_ERR_INVALID_ARG;
    if (h && !h->devs.empty() && out) return on_devices(h, [&](bisbm_engine* d, size_t i) { return bisbm_get_cum_dS(d, out + h->dev_first[i]); });
    if (!out) return fail(h, BISBM_ERR_INVALID_ARG, "out is NULL");
    if (!h->groups.empty()) return gather_groups<double>(h, out, [](bisbm_engine* g, double* o) { return bisbm_get_cum_dS(g, o); });
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    std::vector<ChainScalars> sc(h->n_chains);
    HIPCHK(h, hipMemcpy(sc.data(), h->d_scalars, sizeof(ChainScalars) * h->n_chains, hipMemcpyDeviceToHost));
    for (uint32_t c = 0; c < h->n_chains; ++c) out[c] = sc[c].cum_dS;
    return BISBM_OK;
}

int bisbm_get_last_counts(bisbm_handle h, uint64_t* accepted, uint64_t* sweeps) {
    if (!h) return BISBM_ERR_INVALID_ARG;
    if (!h->devs.empty())
        return on_devices(h, [&](bisbm_engine* d, size_t i) {
            return bisbm_get_last_counts(d, accepted ? accepted + h->dev_first[i] : nullptr, sweeps ? sweeps + h->dev_first[i] : nullptr);
        });
    if (!h->groups.empty()) {
        const int rc = gather_groups<uint64_t>(h, accepted, [](bisbm_engine* g, uint64_t* o) { return bisbm_get_last_counts(g, o, nullptr); });
        return rc ? rc : gather_groups<uint64_t>(h, sweeps, [](bisbm_engine* g, uint64_t* o) { return bisbm_get_last_counts(g, nullptr, o); });
    }
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    std::vector<ChainScalars> sc(h->n_chains);
    HIPCHK(h, hipMemcpy(sc.data(), h->d_scalars, sizeof(ChainScalars) * h->n_chains, hipMemcpyDeviceToHost));
    for (uint32_t c = 0; c < h->n_chains; ++c) {
        if (accepted) accepted[c] = sc[c].last_accepted;
        if (sweeps) sweeps[c] = sc[c].last_sweeps;
    }
    return BISBM_OK;
}

int bisbm_entropy(bisbm_handle h, double* out) {
    if (!h) return BISBM_ERR_INVALID_ARG;
    if (!out) return fail(h, BISBM_ERR_INVALID_ARG, "out is NULL");
    if (!h->devs.empty()) return on_devices(h, [&](bisbm_engine* d, size_t i) { return bisbm_entropy(d, out + h->dev_first[i]); });
    if (!h->groups.empty()) return gather_groups<double>(h, out, [](bisbm_engine* g, double* o) { return bisbm_entropy(g, o); });
    if (!h->state_ready) return fail(h, BISBM_ERR_STATE, "block state not built yet");
    HIPCHK(h, hipSetDevice(h->device));
    if (int rc = launch_block_entropy(h, h->d_tmp_f64)) return rc;
    std::vector<double> part(h->n_chains);
    HIPCHK(h, hipMemcpyAsync(part.data(), h->d_tmp_f64, sizeof(double) * h->n_chains, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const HostTables& t = *h->tab;
    for (uint32_t c = 0; c < h->n_chains; ++c) {  // blockmodel.cc:753-787, statement order kept
        double ent = h->ent_deg;
        ent += part[c];
        ent += h->ent_multi;
        ent += h_lbinom_fast(t, (uint64_t)h->ka * h->kb + h->num_edges - 1, h->num_edges);
        ent += h_lbinom_fast(t, h->na - 1, h->ka - 1);
        ent += h_lbinom_fast(t, h->nb - 1, h->kb - 1);
        ent += (h->na * h->nb == 0) ? 0. : std::log((double)(h->na * h->nb));  // safelog, without the na*nb table (F5)
        ent += h_lgamma_fast(t, h->na + 1);
        ent += h_lgamma_fast(t, h->nb + 1);
        out[c] = ent;
    }
    return BISBM_OK;
}

int bisbm_get_ka_kb(bisbm_handle h, uint32_t* ka, uint32_t* kb) {
    if (!h) return BISBM_ERR_INVALID_ARG;
    if (!h->devs.empty()) return multi_common_shape(h, ka, kb);
    if (!h->groups.empty() && !common_shape(h))
        return fail(h, BISBM_ERR_STATE, "the chains of this handle have different block counts: ask per chain (bisbm_get_ka_kb_chain)");
    if (ka) *ka = h->ka;
    if (kb) *kb = h->kb;
    return BISBM_OK;
}

int bisbm_get_sizes(bisbm_handle h, uint64_t* n, uint64_t* num_edges, uint32_t* max_degree, uint32_t* n_chains) {
    if (!h) return BISBM_ERR_INVALID_ARG;
    if (n) *n = h->n;
    if (num_edges) *num_edges = h->num_edges;
    if (max_degree) *max_degree = h->maxdeg;
    if (n_chains) *n_chains = h->n_chains;
    return BISBM_OK;
}

int bisbm_last_sweep_timing(bisbm_handle h, double* kernel_ms, uint64_t* node_updates) {
    if (!h) return BISBM_ERR_INVALID_ARG;
    if (kernel_ms) *kernel_ms = h->last_kernel_ms;
    if (node_updates) *node_updates = h->last_updates;
    return BISBM_OK;
}

int bisbm_last_pass_steps(bisbm_handle h, uint32_t* steps_per_pass) {
    if (!h) return BISBM_ERR_INVALID_ARG;
    if (steps_per_pass) *steps_per_pass = h->last_pass_steps;
    return BISBM_OK;
}

int bisbm_debug_log_q(bisbm_handle h, const int32_t* n, const int32_t* k, size_t count, int fast, double* out) {
    if (!h) return BISBM_ERR_INVALID_ARG;
    if (!h->devs.empty()) {
        const int rc = bisbm_debug_log_q(h->devs[0], n, k, count, fast, out);
        if (rc) h->err = h->devs[0]->err;
        return rc;
    }
    if (!n || !k || !out) return fail(h, BISBM_ERR_INVALID_ARG, "NULL argument");
    if (count == 0) return BISBM_OK;
    HIPCHK(h, hipSetDevice(h->device));
    int32_t *dn = nullptr, *dk = nullptr;
    double* dout = nullptr;
    HIPCHK(h, dalloc(&dn, count));
    HIPCHK(h, dalloc(&dk, count));
    HIPCHK(h, dalloc(&dout, count));
    HIPCHK(h, hipMemcpy(dn, n, sizeof(int32_t) * count, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(dk, k, sizeof(int32_t) * count, hipMemcpyHostToDevice));
    Tables tab{h->d_lgamma, h->tab->lg.size(), h->d_q, h->q_stride, h->d_logtab};
    HIPCHK(h, launch_log_q_probe(tab, dn, dk, count, dout, fast, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(out, dout, sizeof(double) * count, hipMemcpyDeviceToHost));
    (void)hipFree(dn);
    (void)hipFree(dk);
    (void)hipFree(dout);
    return BISBM_OK;
}

}  // extern "C"
