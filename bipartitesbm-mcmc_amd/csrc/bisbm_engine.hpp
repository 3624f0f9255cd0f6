// bisbm_engine.hpp -- what the translation units of the host side share (internal; the C ABI is include/bisbm.h): the handle,
// the host-built tables, error / allocation helpers and the dispatch templates of container handles (chains of several shapes:
// `groups`; several devices: `devs`).
//
//   bisbm_tables.cpp     host-built numeric tables, temperature tables (no HIP)
//   bisbm_handle.hip     create / destroy / labels in and out / init / shuffle / getters / entropy
//   bisbm_anneal.hip     bisbm_anneal: LDS plan, launch slicing (table slices, pass depth), bookkeeping across launches
//   bisbm_marginals.hip  per-node label histogram, MAP labels of one engine
//   bisbm_multi.hip      several devices behind one handle: creation, dispatch, pooling (RCCL / peer copies)
//   bisbm_merge.hip      agg_merge / agg_split between anneals, chains of one handle in different shapes
//
// Reference lines cited as <file>:<line> relative to /root/reference/src.
#pragma once

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
#include "bisbm_pass_policy.hpp"

namespace bisbm {

// ------------------------------------------------------------------------------------------
// host-built tables (the reference builds the same tables on the host at construction:
// blockmodel.cc:47-48 -> support/cache.cc:64-91, support/int_part.cc:34-51); bisbm_tables.cpp
// ------------------------------------------------------------------------------------------
struct HostTables {
    std::vector<double> lg;  // lg[i] = lgamma(i), lg[0] = +inf
    std::vector<double> lo;  // lo[i] = log(i), lo[0] = 0 (safelog, cache.hh:38-44)
    std::vector<double> q;   // (10001) x (kcap+1)
    uint32_t kcap = 0;
};
std::shared_ptr<HostTables> get_tables(uint64_t lg_size, uint32_t kcap);
double h_lgamma_fast(const HostTables& t, uint64_t x);                // cache.hh:82-93
double h_lbinom_fast(const HostTables& t, uint64_t N, uint64_t k);    // util.hh:41-47
// metropolis_hasting.cc:10-13,20-23 with the host libm for steps t0 .. t0 + len - 1 of a call
std::vector<double> schedule_table(int schedule, float kw0, float kw1, uint64_t t0, uint64_t len, int* zero_after);

}  // namespace bisbm

// ------------------------------------------------------------------------------------------
// the handle
// ------------------------------------------------------------------------------------------
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
    bisbm::ChainScalars* d_scalars = nullptr;
    uint32_t* d_simd_claims = nullptr;  // production kernel: stepping-wave claims per SIMD, zeroed before every launch
    uint32_t* d_mt_engine = nullptr;
    uint32_t* d_mt_gen = nullptr;
    double* d_lgamma = nullptr;
    double* d_logtab = nullptr;
    double* d_q = nullptr;
    double* d_T = nullptr;
    size_t d_T_cap = 0;
    double* d_tmp_f64 = nullptr;  // n_chains doubles
    // block-state part of the description length of every chain as the last production launch without early-stop bookkeeping
    // left it (such launches do not keep the running sum of accepted dS: bisbm_anneal advances it by the change of this,
    // sweep_fast_tracks_minimum); valid while nothing else has changed the block state since
    double* d_ent_prev = nullptr;
    bool ent_prev_valid = false;
    uint32_t* d_stage_u32 = nullptr;  // n uint32 staging
    uint32_t* d_counts = nullptr;     // internal marginal buffer n*kmax
    uint32_t counts_kmax = 0;         // columns d_counts was sized for
    uint32_t counts_cols = 0;         // columns of the histogram it currently holds (max(KA, KB) at the last reset)
    uint32_t cap_ka = 0, cap_kb = 0;  // block counts d_m / d_m_r / d_n_r / d_eta are allocated for
    std::shared_ptr<bisbm::HostTables> tab;
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
    // which depth of pass (two / four / eight steps) the next production launch runs: chosen from the timed launches so far
    // (bisbm_pass_policy.hpp); belongs to a partition: init / shuffle / merges / splits reset it
    bisbm::PassDepthPolicy passes;
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

namespace bisbm {

extern thread_local std::string g_create_error;  // message of a failed bisbm_create (there is no handle to hold it)

int fail(bisbm_engine* h, int code, const char* fmt, ...) __attribute__((format(printf, 3, 4)));

#define HIPCHK(h, expr)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) return fail((h), BISBM_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

template <class T>
hipError_t dalloc(T** p, size_t count) {
    return hipMalloc((void**)p, sizeof(T) * std::max<size_t>(count, 1));
}

void free_chain_arrays(bisbm_engine* h);
void free_all(bisbm_engine* h);
void mt_seed_host(uint32_t* mt, uint64_t seed);  // std::mt19937(seed): seed mod 2^32
inline void forget_pass_speeds(bisbm_engine* h) { h->passes.reset(); }
int rebuild_state(bisbm_engine* h);
// block-state part of entropy() of every chain into d_out (n_chains doubles on the device), on the handle's stream, no sync
int launch_block_entropy(bisbm_engine* h, double* d_out);

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
bool common_shape(bisbm_engine* h);
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
size_t generic_lds_base_bytes(uint32_t ka, uint32_t kb, bool wide, int rng_mode);
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

// ---- several devices behind one handle (bisbm_multi.hip) ------------------------------------------------------------------
inline uint32_t dev_of_chain(const bisbm_engine* h, uint32_t chain, uint32_t* local) {
    uint32_t i = 0;
    while (i + 1 < h->devs.size() && chain >= h->dev_first[i + 1]) ++i;
    *local = chain - h->dev_first[i];
    return i;
}

// fn(engine of device i, i) on one host thread per device
template <class F>
int on_devices(bisbm_engine* h, F&& fn) {
    const size_t nd = h->devs.size();
    std::vector<int> rcs(nd, BISBM_OK);
    if (nd == 1) {
        rcs[0] = fn(h->devs[0], (size_t)0);
    } else {
        std::vector<std::thread> th;
        for (size_t i = 0; i < nd; ++i)
            th.emplace_back([&, i] {
                try {
                    rcs[i] = fn(h->devs[i], i);
                } catch (...) {
                    rcs[i] = BISBM_ERR_STATE;
                    h->devs[i]->err = "out of host memory";
                }
            });
        for (auto& t : th) t.join();
    }
    // the first failing device's code is returned; the message names every device that failed (the others have done their
    // part of the call: see bisbm_agg_merge in include/bisbm.h for what that means for calls that change state)
    int rc = BISBM_OK;
    std::string msg;
    for (size_t i = 0; i < nd; ++i)
        if (rcs[i]) {
            if (!rc) rc = rcs[i];
            msg += (msg.empty() ? "" : "; ") + ("device " + std::to_string(h->devs[i]->device) + ": " + h->devs[i]->err);
        }
    if (rc) h->err = msg;
    return rc;
}

int multi_common_shape(bisbm_engine* h, uint32_t* ka, uint32_t* kb);
int multi_anneal(bisbm_engine* h, int schedule, const float kwargs[2], uint64_t duration_steps, uint64_t steps_await, double* acc_rate_out);
int multi_marginals_get(bisbm_engine* h, uint32_t* counts_out);
int multi_marginals_map(bisbm_engine* h, uint32_t* labels_out);
void multi_free(bisbm_engine* h);
// MAP labels from the internal histogram of one engine (no pooling); bisbm_marginals.hip
int single_marginals_map(bisbm_engine* h, uint32_t* labels_out);

}  // namespace bisbm
