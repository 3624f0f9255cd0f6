// bisbm_runtime.hip -- host side of the C ABI (include/bisbm.h): handle, device memory, host-built
// numeric tables, kernel launches.  No CPU compute path exists here: every operation on chain state
// runs in a HIP kernel (bisbm_kernels.hip), and bisbm_create fails without a device.
//
// Reference lines cited as <file>:<line> relative to /root/reference/src.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <map>
#include <memory>
#include <mutex>
#include <queue>
#include <random>
#include <set>
#include <sstream>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/bisbm.h"
#include "bisbm_kernels.hpp"

using namespace bisbm;

namespace {

thread_local std::string g_create_error;

// ------------------------------------------------------------------------------------------
// host-built tables (the reference builds the same tables on the host at construction:
// blockmodel.cc:47-48 -> support/cache.cc:64-91, support/int_part.cc:34-51)
// ------------------------------------------------------------------------------------------
struct HostTables {
    std::vector<double> lg;  // lg[i] = lgamma(i), lg[0] = +inf
    std::vector<double> lo;  // lo[i] = log(i), lo[0] = 0 (safelog, cache.hh:38-44)
    std::vector<double> q;   // (10001) x (kcap+1)
    uint32_t kcap = 0;
};

std::mutex g_tab_mu;
std::map<std::pair<uint64_t, uint32_t>, std::shared_ptr<HostTables>> g_tab_cache;

double log_sum(double a, double b) {  // int_part.cc:30-32
    return std::max(a, b) + std::log1p(std::exp(-std::fabs(a - b)));
}

void fill_lgamma(std::vector<double>& lg, std::vector<double>& lo) {  // cache.cc:64-79, :25-37
    const size_t n = lg.size();
    lg[0] = INFINITY;
    lo[0] = 0.0;
    unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    if (n < (1u << 16)) nt = 1;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) {
        th.emplace_back([&, t] {
            int sign;
            for (size_t i = 1 + t; i < n; i += nt) {
                lg[i] = lgamma_r((double)i, &sign);
                lo[i] = std::log((double)i);
            }
        });
    }
    for (auto& x : th) x.join();
}

// int_part.cc:34-51 with the columns cut at kcap (column k only depends on columns <= k).  Cells
// the reference never writes (k > n) stay -inf and are read as such by the recurrence.
// Rows are processed in blocks: inside a block columns < B are filled row by row (a cell (n,k)
// reads (n-k,k), which can lie in the same block only when k < B); the remaining columns of the
// block's rows are independent of each other and are split over threads.  The evaluation order per
// cell is unchanged, so the values equal the serial recurrence bit for bit.
void fill_q(std::vector<double>& q, uint32_t kcap) {
    const size_t stride = (size_t)kcap + 1;
    std::fill(q.begin(), q.end(), -INFINITY);
    const size_t B = 128;
    auto cell = [&](size_t n, size_t k) {
        double* row = q.data() + n * stride;
        row[k] = log_sum(row[k], row[k - 1]);
        if (n > k) row[k] = log_sum(row[k], q[(n - k) * stride + k]);
    };
    unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    if (kcap < 2048) nt = 1;
    for (size_t n0 = 1; n0 <= (size_t)kQNmax; n0 += B) {
        const size_t n1 = std::min<size_t>(n0 + B, (size_t)kQNmax + 1);
        for (size_t n = n0; n < n1; ++n) {
            q[n * stride + 1] = 0;
            const size_t kmax = std::min<size_t>(std::min<size_t>(n, kcap), B - 1);
            for (size_t k = 2; k <= kmax; ++k) cell(n, k);
        }
        auto rest = [&](size_t lo, size_t hi) {
            for (size_t n = lo; n < hi; ++n) {
                const size_t kmax = std::min<size_t>(n, kcap);
                for (size_t k = B; k <= kmax; ++k) cell(n, k);
            }
        };
        if (nt == 1 || n1 <= B) {
            rest(n0, n1);
        } else {
            std::vector<std::thread> th;
            const size_t per = (n1 - n0 + nt - 1) / nt;
            for (unsigned t = 0; t < nt; ++t) {
                const size_t lo = n0 + t * per, hi = std::min(n1, lo + per);
                if (lo < hi) th.emplace_back(rest, lo, hi);
            }
            for (auto& x : th) x.join();
        }
    }
}

std::shared_ptr<HostTables> get_tables(uint64_t lg_size, uint32_t kcap) {
    std::lock_guard<std::mutex> lk(g_tab_mu);
    for (auto& kv : g_tab_cache)
        if (kv.first.first >= lg_size && kv.first.second >= kcap && kv.first.second <= 2 * kcap + 64 &&
            kv.first.first <= 2 * lg_size + 4096)
            return kv.second;
    auto t = std::make_shared<HostTables>();
    t->lg.resize(lg_size);
    t->lo.resize(lg_size);
    fill_lgamma(t->lg, t->lo);
    t->kcap = kcap;
    t->q.resize((size_t)(kQNmax + 1) * ((size_t)kcap + 1));
    fill_q(t->q, kcap);
    if (g_tab_cache.size() > 4) g_tab_cache.clear();
    g_tab_cache[{lg_size, kcap}] = t;
    return t;
}

double h_lgamma_fast(const HostTables& t, uint64_t x) {  // cache.hh:82-93
    if (x < t.lg.size()) return t.lg[x];
    if (x == 0) return INFINITY;
    int sign;
    return lgamma_r((double)x, &sign);
}

double h_lbinom_fast(const HostTables& t, uint64_t N, uint64_t k) {  // util.hh:41-47
    if (N == 0 || k == 0 || k > N) return 0;
    return (h_lgamma_fast(t, N + 1) - h_lgamma_fast(t, k + 1)) - h_lgamma_fast(t, N - k + 1);
}

// metropolis_hasting.cc:10-13,20-23 evaluated with the host libm (the reference's own values) for steps t0 .. t0 + len - 1 of a
// call, on up to 16 threads.  The exponential schedule's table ends with its first exact zero (pow has underflowed: it is
// monotone for 0 <= kw1 < 1, so every later step is 0 as well; *zero_after says so).
std::vector<double> schedule_table(int schedule, float kw0, float kw1, uint64_t t0, uint64_t len, int* zero_after) {
    std::vector<double> T((size_t)len);
    *zero_after = 0;
    auto fill = [&](uint64_t lo, uint64_t hi) {
        for (uint64_t j = lo; j < hi; ++j) {
            const uint64_t t = t0 + j;
            if (schedule == SCHED_EXPONENTIAL) {
                T[j] = (double)kw0 * std::pow((double)kw1, (double)t);
            } else {
                const float x = (float)t + kw1;
                const size_t i = (size_t)x;
                T[j] = (double)kw0 / (i == 0 ? 0. : std::log((double)i));
            }
        }
    };
    const unsigned nt = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency())), len >> 14));
    if (nt <= 1) {
        fill(0, len);
    } else {
        std::vector<std::thread> th;
        for (unsigned i = 0; i < nt; ++i) th.emplace_back(fill, len * i / nt, len * (i + 1) / nt);
        for (auto& x : th) x.join();
    }
    if (schedule == SCHED_EXPONENTIAL && kw1 < 1.f && kw1 >= 0.f)
        for (uint64_t j = 0; j < len; ++j)
            if (T[j] == 0.) {
                T.resize((size_t)j + 1);
                *zero_after = 1;
                break;
            }
    return T;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// the handle
// ------------------------------------------------------------------------------------------
struct bisbm_engine {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;
    // shape
    uint64_t n = 0, na = 0, nb = 0, num_edges = 0, nnz = 0;
    uint32_t ka = 0, kb = 0, K = 0, maxdeg = 0, n_chains = 0, first_chain_id = 0;
    double epsilon = 0;
    int rng_mode = 0;
    uint64_t seed = 0, gen_seed = 0;
    bool state_ready = false;
    // device memory
    uint32_t* d_rowptr = nullptr;
    uint32_t* d_col = nullptr;
    uint8_t* d_labels = nullptr;      // [chain][label_stride] labels: bytes, or two bytes each while `wide`
    uint8_t* d_labels_tmp = nullptr;
    size_t label_stride = 0;          // in labels
    bool wide = false;                // KA + KB > 256 (a --merge run starts at one block per node): generic kernel only,
                                      // two-byte labels, m read and updated in HBM; back to bytes once K <= 256
    size_t lbytes() const { return wide ? 2 : 1; }
    uint32_t* d_vlist = nullptr;
    int32_t* d_m = nullptr;
    int32_t* d_m_r = nullptr;
    int32_t* d_n_r = nullptr;
    uint32_t* d_eta = nullptr;
    ChainScalars* d_scalars = nullptr;
    uint32_t* d_simd_claims = nullptr;  // production kernel: stepping-wave claims per SIMD, zeroed before every launch
    uint32_t* d_mt_engine = nullptr;
    uint32_t* d_mt_gen = nullptr;
    double* d_lgamma = nullptr;
    double* d_logtab = nullptr;
    double* d_q = nullptr;
    double* d_T = nullptr;
    size_t d_T_cap = 0;
    double* d_tmp_f64 = nullptr;  // n_chains doubles
    uint32_t* d_stage_u32 = nullptr;  // n uint32 staging
    uint32_t* d_counts = nullptr;     // internal marginal buffer n*kmax
    uint32_t counts_kmax = 0;         // columns d_counts was sized for
    uint32_t counts_cols = 0;         // columns of the histogram it currently holds (max(KA, KB) at the last reset)
    uint32_t cap_ka = 0, cap_kb = 0;  // block counts d_m / d_m_r / d_n_r / d_eta are allocated for
    std::shared_ptr<HostTables> tab;
    uint32_t q_stride = 0;
    // chain-independent part of entropy()
    double ent_deg = 0, ent_multi = 0;
    // nodes of every degree 0..256 per type (256: all longer rows), shared with the sub-engines: what the production kernel's
    // eta window is placed by (bisbm_anneal)
    std::shared_ptr<std::vector<uint64_t>> deg_count;
    // last sweep timing
    double last_kernel_ms = 0;
    uint64_t last_updates = 0;
    uint32_t last_pass_steps = 0;  // steps per pass of the last sweep launch (1, 2, 4, 8)
    // production kernel, both block counts <= 16: which depth of pass (1 / 2 / 3 = two / four / eight steps) runs how fast
    // HERE (updates per ms of the launches so far, 0 = not tried yet), and how many launches ago another one was tried
    // measured speed (updates per ms) of the pass depths: the last two launches of each depth with their accepted fractions
    // ([d][0]: the latest), how many launches measured it since the partition was put in place, and the figure they stand for
    double pass_sample[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    double pass_sample_acc[4][2] = {{-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}};
    uint32_t pass_n[4] = {0, 0, 0, 0};
    double pass_speed[4] = {0, 0, 0, 0};
    double last_acc = -1;                    // accepted fraction of the last launch (any depth)
    uint32_t pass_launches = 0;
    bool pass_up = false;
    // Chains with different block counts (after a one-argument agg_merge, blockmodel.cc:208-271: every run ends where it
    // ends).  Kernels are launched for one (KA, KB), so the handle then becomes a CONTAINER: its chains live in
    // sub-engines, one per distinct shape (`groups`), which borrow the graph and the tables from it (`root`); chain c of
    // the handle is chain where[c].second of group where[c].first.  A sub-engine knows the global id of each of its
    // chains (`gids`, the key of the Philox streams) and its index in the handle (`ridx`).
    std::vector<bisbm_engine*> groups;
    std::vector<std::pair<uint32_t, uint32_t>> where;
    bisbm_engine* root = nullptr;
    std::vector<uint32_t> gids, ridx;
    uint32_t* d_gids = nullptr;
    uint32_t gid(size_t c) const { return gids.empty() ? first_chain_id + (uint32_t)c : gids[c]; }
    // Several devices behind one handle (bisbm_create_multi): the handle is a container of one full engine per device
    // (`devs`; graph and tables replicated, one stream and one host thread per device); chains dev_first[i] ..
    // dev_first[i + 1] - 1 of the handle live in devs[i], in order, so global chain ids -- the keys of the random streams --
    // do not depend on the number of devices.  `pool`: what the pooling of the marginal histogram over the devices needs.
    std::vector<bisbm_engine*> devs;
    std::vector<uint32_t> dev_first;
    struct DevicePool* pool = nullptr;
    uint64_t counts_rows = 0;  // rows of the internal marginal buffer (n, or n rounded up to a multiple of the device count)
};

namespace {

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

#define HIPCHK(h, expr)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) return fail((h), BISBM_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

template <class T>
hipError_t dalloc(T** p, size_t count) {
    return hipMalloc((void**)p, sizeof(T) * std::max<size_t>(count, 1));
}

void free_chain_arrays(bisbm_engine* h) {
    void** ptrs[] = {(void**)&h->d_labels, (void**)&h->d_labels_tmp, (void**)&h->d_vlist, (void**)&h->d_m, (void**)&h->d_m_r,
                     (void**)&h->d_n_r, (void**)&h->d_eta, (void**)&h->d_scalars, (void**)&h->d_mt_engine, (void**)&h->d_mt_gen,
                     (void**)&h->d_tmp_f64, (void**)&h->d_counts, (void**)&h->d_gids};
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

void forget_pass_speeds(bisbm_engine* h) {
    for (int d = 0; d < 4; ++d) {
        h->pass_sample[d][0] = h->pass_sample[d][1] = h->pass_speed[d] = 0;
        h->pass_sample_acc[d][0] = h->pass_sample_acc[d][1] = -1;
        h->pass_n[d] = 0;
    }
    h->last_acc = -1;
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
    // (a partition put in place from outside -- init, shuffle, merges, splits: the pass depths are measured afresh, see bisbm_anneal)
    forget_pass_speeds(h);
    h->pass_launches = 0;
    return BISBM_OK;
}

// container handles (bisbm_engine::groups): run `f` on every group, first error wins
template <class F>
int each_group(bisbm_engine* h, F f) {
    for (bisbm_engine* g : h->groups) {
        const int rc = f(g);
        if (rc) {
            h->err = g->err;
            return rc;
        }
    }
    return BISBM_OK;
}
// container handles: do all groups have one shape (again)?  If so the handle's own ka / kb / K follow it.
bool common_shape(bisbm_engine* h) {
    for (bisbm_engine* g : h->groups)
        if (g->ka != h->groups[0]->ka || g->kb != h->groups[0]->kb) return false;
    h->ka = h->groups[0]->ka, h->kb = h->groups[0]->kb, h->K = h->ka + h->kb;
    return true;
}
// ... and gather one value per chain from the groups into the handle's chain order
template <class T, class F>
int gather_groups(bisbm_engine* h, T* out, F f) {
    return each_group(h, [&](bisbm_engine* g) {
        std::vector<T> tmp(g->n_chains);
        const int rc = f(g, tmp.data());
        if (rc == BISBM_OK && out)
            for (size_t j = 0; j < tmp.size(); ++j) out[g->ridx[j]] = tmp[j];
        return rc;
    });
}

// LDS of the generic kernel without the optional parts (eta, the visit list): the a x b quadrant of m (odd row stride; in
// HBM while wide), m_r, n_r, the k_v histogram, staged rows; compat mode adds two mt19937 states and their tempered outputs.
// Wide mode (KA + KB > 256) therefore ends where m_r / n_r / the histogram leave the 160 KiB of a CU -- about 11 000 to
// 18 000 blocks depending on the split and the RNG mode -- well below what two-byte labels could name.
size_t generic_lds_base_bytes(uint32_t ka, uint32_t kb, bool wide, int rng_mode) {
    const size_t K = (size_t)ka + kb, S = kb | 1u;
    size_t lds = sizeof(int32_t) * ((wide ? 0 : (size_t)ka * S) + 2 * K + std::max<uint32_t>(std::max(ka, kb), 64)) + sizeof(uint32_t) * 64 * 64;
    if (rng_mode == BISBM_RNG_MT19937_COMPAT) lds += sizeof(uint32_t) * 624 * 4;
    return lds;
}
constexpr size_t kLdsPerCu = 160 * 1024;

// fn(c) for every chain, on up to 16 host threads when there are enough chains.  An exception that left a worker thread
// (std::bad_alloc from a chain's host-side merge state) would end the process through std::terminate, and one that left
// the calling thread would cross the C boundary: both are caught here and reported as `false`.
template <class F>
bool for_each_chain(size_t C, F&& fn) {
    std::atomic<bool> ok{true};
    auto guarded = [&](size_t c) {
        try {
            fn(c);
        } catch (...) {
            ok = false;
        }
    };
    const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min<size_t>(16, std::thread::hardware_concurrency()), C / 4));
    if (nt <= 1) {
        for (size_t c = 0; c < C; ++c) guarded(c);
    } else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t)
            th.emplace_back([&, t] {
                for (size_t c = t; c < C; c += nt) guarded(c);
            });
        for (auto& x : th) x.join();
    }
    return ok;
}

}  // namespace

#include "bisbm_multi.hpp"

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
            // A measurement belongs to a regime: a depth whose figure was taken at an accepted fraction more than 0.1 away from
            // the last launch's (a schedule cooling down, a chain leaving its burn-in) counts as not tried -- the speed of a deep
            // pass depends on how many steps move far more than that of a two-steps pass does, so the running depth's own speed
            // does not tell.
            // And a measurement can be an outlier, always to the slow side (the first launches of a process are up to 20 % slow
            // while the device comes up to its clocks): a depth's figure is the BEST of its last two launches in the regime, and
            // after a new partition every depth is launched twice before any is trusted.
            auto in_regime = [&](uint32_t d, int i) {
                return h->pass_sample[d][i] > 0 && !(h->last_acc >= 0 && std::fabs(h->last_acc - h->pass_sample_acc[d][i]) > 0.1);
            };
            for (uint32_t d = 1; d <= max_depth; ++d)
                h->pass_speed[d] = std::max(in_regime(d, 0) ? h->pass_sample[d][0] : 0., in_regime(d, 1) ? h->pass_sample[d][1] : 0.);
            // (one launch is enough for a depth that comes out more than 25 % behind a depth measured twice: outliers are smaller)
            double twice = 0;
            for (uint32_t d = 1; d <= max_depth; ++d)
                if (h->pass_n[d] >= 2u) twice = std::max(twice, h->pass_speed[d]);
            auto fresh = [&](uint32_t d) { return h->pass_speed[d] > 0 && (h->pass_n[d] >= 2u || h->pass_speed[d] < 0.75 * twice); };
            uint32_t best = 0;
            for (uint32_t d = 1; d <= max_depth; ++d)
                if (fresh(d) && (best == 0 || h->pass_speed[d] > h->pass_speed[best])) best = d;
            // not tried yet (or not in this regime): on small graphs the deepest first (there it won in every regime measured, and
            // a call that is too short to be split runs its one launch with the first choice); on large ones
            // the shallowest first (from a random start it is the faster one, and the next launches look further)
            bool all_tried = true;
            for (uint32_t d = 1; d <= max_depth; ++d) all_tried = all_tried && fresh(d);
            if (h->n <= 100000) {
                for (uint32_t d = 1; d <= max_depth; ++d)
                    if (!fresh(d)) best = d;
            } else {
                for (uint32_t d = max_depth; d >= 1; --d)
                    if (!fresh(d)) best = d;
            }
            depth = best ? best : max_depth;
            if (all_tried && ++h->pass_launches >= 16u) {  // look again at a neighbour of the best
                h->pass_launches = 0;
                h->pass_up = !h->pass_up;  // (one side, then the other)
                depth = (h->pass_up && best < max_depth) || best == 1u ? best + 1 : best - 1;
            }
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
            h->pass_sample[depth][1] = h->pass_sample[depth][0];
            h->pass_sample_acc[depth][1] = h->pass_sample_acc[depth][0];
            h->pass_sample[depth][0] = speed;
            h->pass_sample_acc[depth][0] = h->last_acc = acc_frac;
            h->pass_n[depth] += 1;
            if (getenv("BISBM_PASS_LOG"))
                fprintf(stderr, "[bisbm passes] depth %u: %.3e updates/ms, accepted %.3f (before the launch: two %.3e, four %.3e, eight %.3e)\n", depth, speed,
                        acc_frac, h->pass_speed[1], h->pass_speed[2], h->pass_speed[3]);
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
    ep.out = h->d_tmp_f64;
    HIPCHK(h, launch_entropy(ep, h->stream));
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

int bisbm_marginals_reset(bisbm_handle h) {
    if (!h) return BISBM_ERR_INVALID_ARG;
    if (!h->devs.empty()) return on_devices(h, [](bisbm_engine* d, size_t) { return bisbm_marginals_reset(d); });
    if (!h->groups.empty() && !common_shape(h))
        return fail(h, BISBM_ERR_STATE, "the chains of this handle have different block counts: no common marginal histogram");
    HIPCHK(h, hipSetDevice(h->device));
    const uint32_t kmax = std::max(h->ka, h->kb);
    const size_t cnt = (size_t)std::max<uint64_t>(h->n, h->counts_rows) * kmax;  // (rows past n stay zero: see DevicePool)
    if (h->d_counts && h->counts_kmax < kmax) {
        (void)hipFree(h->d_counts);
        h->d_counts = nullptr;
    }
    if (!h->d_counts) {
        HIPCHK(h, dalloc(&h->d_counts, cnt));
        h->counts_kmax = kmax;
    }
    h->counts_cols = kmax;
    HIPCHK(h, hipMemsetAsync(h->d_counts, 0, sizeof(uint32_t) * cnt, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return BISBM_OK;
}

int bisbm_marginals_accumulate(bisbm_handle h, uint32_t* device_counts) {
    if (!h) return BISBM_ERR_INVALID_ARG;
    if (!h->devs.empty()) {
        if (device_counts) return fail(h, BISBM_ERR_UNSUPPORTED, "a handle over several devices accumulates into its own buffers (device_counts must be NULL); bisbm_marginals_map pools them");
        if (int rc = multi_common_shape(h, nullptr, nullptr)) return rc;
        return on_devices(h, [](bisbm_engine* d, size_t) { return bisbm_marginals_accumulate(d, nullptr); });
    }
    if (!h->groups.empty() && !common_shape(h))
        return fail(h, BISBM_ERR_STATE, "the chains of this handle have different block counts: no common marginal histogram");
    HIPCHK(h, hipSetDevice(h->device));
    if (!device_counts) {
        // (a histogram made before a merge / split changed max(KA, KB) has another row length: start afresh)
        if (!h->d_counts || h->counts_cols != std::max(h->ka, h->kb)) {
            int rc = bisbm_marginals_reset(h);
            if (rc) return rc;
        }
        device_counts = h->d_counts;
    }
    if (!h->groups.empty())  // groups that have come to one shape again: every group adds its chains to the same histogram
        return each_group(h, [&](bisbm_engine* g) { return bisbm_marginals_accumulate(g, device_counts); });
    MarginalParams mp{};
    mp.n = (uint32_t)h->n;
    mp.na = (uint32_t)h->na;
    mp.ka = h->ka;
    mp.kmax = std::max(h->ka, h->kb);
    mp.n_chains = h->n_chains;
    mp.labels = h->d_labels;
    mp.label_stride = h->label_stride;
    mp.wide = h->wide ? 1u : 0u;
    mp.counts = device_counts;
    HIPCHK(h, launch_marginals(mp, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return BISBM_OK;
}

int bisbm_marginals_get(bisbm_handle h, uint32_t* counts_out) {
    if (!h) return BISBM_ERR_INVALID_ARG;
    if (!counts_out) return fail(h, BISBM_ERR_INVALID_ARG, "counts_out is NULL");
    if (!h->devs.empty()) return multi_marginals_get(h, counts_out);
    if (!h->groups.empty() && !common_shape(h))
        return fail(h, BISBM_ERR_STATE, "the chains of this handle have different block counts: no common marginal histogram");
    if (!h->d_counts) return fail(h, BISBM_ERR_STATE, "no internal marginal buffer yet");
    if (h->counts_cols != std::max(h->ka, h->kb))
        return fail(h, BISBM_ERR_STATE, "the block counts changed since the histogram was made (%u columns then, %u now)", h->counts_cols,
                    std::max(h->ka, h->kb));
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(counts_out, h->d_counts, sizeof(uint32_t) * (size_t)h->n * std::max(h->ka, h->kb), hipMemcpyDeviceToHost));
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


int bisbm_create_multi(bisbm_handle* out, uint64_t n, uint64_t na, uint64_t nb, const uint64_t* rowptr, const uint32_t* col,
                       uint32_t ka, uint32_t kb, double epsilon, uint32_t n_chains, uint32_t first_chain_id, const int* devices,
                       int n_devices, int rng_mode, uint64_t seed, uint64_t gen_seed) {
    if (!out) return fail(nullptr, BISBM_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (!devices || n_devices < 1) return fail(nullptr, BISBM_ERR_INVALID_ARG, "devices is NULL or empty");
    if ((uint32_t)n_devices > n_chains) return fail(nullptr, BISBM_ERR_INVALID_ARG, "%d devices for %u chains: every device needs at least one chain", n_devices, n_chains);
    std::unique_ptr<bisbm_engine> hp(new bisbm_engine());
    bisbm_engine* h = hp.get();
    const size_t nd = (size_t)n_devices;
    h->device = devices[0];
    h->n = n, h->na = na, h->nb = nb, h->ka = ka, h->kb = kb, h->K = ka + kb;
    h->n_chains = n_chains, h->first_chain_id = first_chain_id, h->epsilon = epsilon, h->rng_mode = rng_mode, h->seed = seed, h->gen_seed = gen_seed;
    h->counts_rows = (n + nd - 1) / nd * nd;  // node ranges of equal size for the reduce-scatter (rows past n stay zero)
    // contiguous chain ranges, the first n_chains % n_devices devices one chain more (the split bench.py and
    // distributed.shard_chains use)
    h->devs.assign(nd, nullptr);
    h->dev_first.assign(nd + 1, 0);
    for (size_t i = 0; i < nd; ++i) h->dev_first[i + 1] = h->dev_first[i] + n_chains / (uint32_t)nd + (i < n_chains % nd ? 1u : 0u);
    // the devices are set up side by side (graph upload, table upload); the host tables are built once and shared
    std::vector<int> rcs(nd, BISBM_OK);
    std::vector<std::string> errs(nd);
    std::mutex err_mu;
    {
        std::vector<std::thread> th;
        for (size_t i = 0; i < nd; ++i)
            th.emplace_back([&, i] {
                bisbm_handle d = nullptr;
                rcs[i] = bisbm_create(&d, n, na, nb, rowptr, col, ka, kb, epsilon, h->dev_first[i + 1] - h->dev_first[i],
                                      first_chain_id + h->dev_first[i], devices[i], rng_mode, seed, gen_seed);
                if (rcs[i]) {
                    std::lock_guard<std::mutex> lk(err_mu);  // (the message of a failed create is a process-wide string)
                    errs[i] = g_create_error;
                }
                h->devs[i] = d;
            });
        for (auto& t : th) t.join();
    }
    for (size_t i = 0; i < nd; ++i)
        if (rcs[i]) {
            const int rc = rcs[i];
            const std::string msg = "device " + std::to_string(devices[i]) + ": " + errs[i];
            for (bisbm_engine*& d : h->devs)
                if (d) {
                    free_all(d);
                    delete d;
                    d = nullptr;
                }
            h->devs.clear();
            return fail(nullptr, rc, "%s", msg.c_str());
        }
    for (bisbm_engine* d : h->devs) d->counts_rows = h->counts_rows;
    h->num_edges = h->devs[0]->num_edges, h->nnz = h->devs[0]->nnz, h->maxdeg = h->devs[0]->maxdeg;
    *out = hp.release();
    return BISBM_OK;
}

int bisbm_device_count(bisbm_handle h, int* n_devices, int* devices, uint32_t* first_chain) {
    if (!h) return BISBM_ERR_INVALID_ARG;
    const size_t nd = h->devs.empty() ? 1 : h->devs.size();
    if (n_devices) *n_devices = (int)nd;
    for (size_t i = 0; i < nd; ++i) {
        if (devices) devices[i] = h->devs.empty() ? h->device : h->devs[i]->device;
        if (first_chain) first_chain[i] = h->devs.empty() ? 0u : h->dev_first[i];
    }
    return BISBM_OK;
}

int bisbm_marginals_map(bisbm_handle h, uint32_t* labels_out) {
    if (!h) return BISBM_ERR_INVALID_ARG;
    if (!labels_out) return fail(h, BISBM_ERR_INVALID_ARG, "labels_out is NULL");
    return h->devs.empty() ? single_marginals_map(h, labels_out) : multi_marginals_map(h, labels_out);
}

}  // extern "C"


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
    if (!h->wide && h->K + 1 > 256) {  // 256 -> 257 blocks: the labels become two bytes first
        uint8_t* wide_labels = nullptr;
        HIPCHK(h, dalloc(&wide_labels, (size_t)h->n_chains * h->label_stride * 2));
        hipError_t e = launch_labels_to_wide(h->d_labels, wide_labels, h->label_stride, (uint32_t)h->n, h->n_chains, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) {
            (void)hipFree(wide_labels);
            return fail(h, BISBM_ERR_HIP, "widening the labels: %s", hipGetErrorString(e));
        }
        (void)hipFree(h->d_labels);
        h->d_labels = wide_labels;
        h->wide = true;
        if (h->d_labels_tmp) {  // (Philox mode's snapshot buffer for shuffle_bisbm: sized for the label format)
            (void)hipFree(h->d_labels_tmp);
            h->d_labels_tmp = nullptr;
            HIPCHK(h, dalloc(&h->d_labels_tmp, (size_t)h->n_chains * h->label_stride * 2));
        }
    }
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
    for (size_t c = 0; c < C; ++c) {
        bool any = false;
        for (size_t b = 0; b < k_type; ++b) any |= n_r[c * K + b_lo + b] > 1;
        if (!any)
            return fail(h, BISBM_ERR_STATE, "chain %zu: no type-%c block has two nodes: nothing to split (the reference would add an empty block)",
                        c, type ? 'b' : 'a');
    }

    uint32_t *d_rank = nullptr, *d_bits = nullptr, *d_chosen = nullptr, *d_rank_base = nullptr, *d_block_off = nullptr;
    int32_t *d_out_k = nullptr, *d_out_deg = nullptr;
    auto cleanup = [&]() {
        for (void* p : {(void*)d_rank, (void*)d_bits, (void*)d_chosen, (void*)d_out_k, (void*)d_out_deg, (void*)d_rank_base, (void*)d_block_off})
            if (p) (void)hipFree(p);
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
