// bisbm_multi.hip -- several devices behind one handle (bisbm_create_multi).
//
// SURVEY 8(e) / BASELINE north_star: chains are independent, so they shard over the GPUs of a node as contiguous chain
// ranges -- graph and tables replicated per device, Philox streams keyed by the GLOBAL chain id, no exchange during
// sweeps.  The handle is a container of one full engine per device; every call of the C ABI is dispatched to the
// engine(s) that own the chains it names, all-chain calls on one host thread per device.  The only exchange is the pooling
// of the per-node marginal histogram: reduce-scatter by node range -> argmax on the owner -> all-gather of the labels,
// through RCCL (ncclReduceScatter / ncclAllGather over xGMI, one communicator per device in this process), resolved at
// run time from librccl.so so that single-device users never load it.  Where RCCL cannot serve (the same device listed
// twice -- the one-GPU rehearsal of `--devices 0,0` --, the library missing, BISBM_POOL=p2p) the same exchange runs as
// peer copies (hipMemcpyPeerAsync of every other device's slice to the owner of the node range + an add kernel): on the fully
// connected xGMI topology that is the same traffic pattern, one slice per link.
//
// Reference lines cited as <file>:<line> relative to /root/reference/src.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "bisbm_engine.hpp"

using namespace bisbm;

namespace {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

RcclApi& rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.lib) break;
        }
        if (!api.lib) return;
        auto sym = [&](const char* n) { return dlsym(api.lib, n); };
        api.CommInitAll = (decltype(api.CommInitAll))sym("ncclCommInitAll");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
        api.ReduceScatter = (decltype(api.ReduceScatter))sym("ncclReduceScatter");
        api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
        api.ok = api.CommInitAll && api.CommDestroy && api.GroupStart && api.GroupEnd && api.ReduceScatter && api.AllGather && api.GetErrorString;
    });
    return api;
}

}  // namespace

// What pooling the marginal histogram over the devices of a handle needs: per device the reduced slice of its node range,
// a staging slice (peer-copy path), the labels of its range and of all nodes; RCCL communicators when that path serves.
struct DevicePool {
    std::vector<ncclComm_t> comms;  // empty: peer-copy path
    std::string why_not_rccl;
    std::vector<uint32_t*> d_red, d_stage;
    std::vector<uint16_t*> d_lab, d_all;
    uint64_t per = 0;      // nodes per device range (the last ranges may reach past n: those rows are zero)
    uint32_t kmax_cap = 0;  // columns the slices are allocated for (0: nothing usable allocated)
    uint32_t peer_direct = 0;  // peer-copy path: ordered device pairs with direct peer access enabled
};

namespace bisbm {

namespace {

// the calling thread's current device, put back when the call returns (the pooling calls visit every device of the handle on
// the caller's thread; a torch caller's later "current device" allocations must not move with them)
struct DeviceGuard {
    int saved = -1;
    DeviceGuard() {
        if (hipGetDevice(&saved) != hipSuccess) saved = -1;
    }
    ~DeviceGuard() {
        if (saved >= 0) (void)hipSetDevice(saved);
    }
};

void pool_free(bisbm_engine* h) {
    DevicePool* P = h->pool;
    if (!P) return;
    DeviceGuard guard;
    for (size_t i = 0; i < h->devs.size(); ++i) {
        (void)hipSetDevice(h->devs[i]->device);
        if (i < P->comms.size() && P->comms[i]) (void)rccl_api().CommDestroy(P->comms[i]);
        for (void* p : {(void*)(i < P->d_red.size() ? P->d_red[i] : nullptr), (void*)(i < P->d_stage.size() ? P->d_stage[i] : nullptr),
                        (void*)(i < P->d_lab.size() ? P->d_lab[i] : nullptr), (void*)(i < P->d_all.size() ? P->d_all[i] : nullptr)})
            if (p) (void)hipFree(p);
    }
    delete P;
    h->pool = nullptr;
}

// Peer-copy path: device i reads its node range out of every other device's histogram.  Without peer access enabled
// hipMemcpyPeerAsync stages through host memory; with it the copy goes over the xGMI link between the two devices.  (Asked for
// once per ordered pair; "already enabled" is fine; a pair that cannot have it keeps the staged copy, which is still correct.)
void enable_peer_access(bisbm_engine* h, DevicePool* P) {
    const size_t nd = h->devs.size();
    P->peer_direct = 0;
    for (size_t i = 0; i < nd; ++i) {
        if (hipSetDevice(h->devs[i]->device) != hipSuccess) continue;
        for (size_t j = 0; j < nd; ++j) {
            const int a = h->devs[i]->device, b = h->devs[j]->device;
            if (a == b) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, a, b) != hipSuccess || !can) continue;
            const hipError_t e = hipDeviceEnablePeerAccess(b, 0);
            if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) P->peer_direct += 1;
            (void)hipGetLastError();  // ("already enabled" stays behind as the thread's last error otherwise)
        }
    }
}

int pool_prepare(bisbm_engine* h, uint32_t kmax) {
    const size_t nd = h->devs.size();
    if (!h->pool) {
        h->pool = new DevicePool();
        DevicePool* P = h->pool;
        P->per = h->counts_rows / nd;
        // RCCL wants one rank per device: distinct ordinals.  BISBM_POOL=p2p forces the peer-copy path.
        std::set<int> distinct;
        for (bisbm_engine* d : h->devs) distinct.insert(d->device);
        const char* mode = getenv("BISBM_POOL");
        if (mode && !strcmp(mode, "p2p")) {
            P->why_not_rccl = "BISBM_POOL=p2p";
        } else if (distinct.size() != nd) {
            P->why_not_rccl = "a device is listed more than once";
        } else if (!rccl_api().ok) {
            P->why_not_rccl = "librccl.so could not be loaded";
        } else {
            std::vector<int> ids;
            for (bisbm_engine* d : h->devs) ids.push_back(d->device);
            P->comms.assign(nd, nullptr);
            const ncclResult_t r = rccl_api().CommInitAll(P->comms.data(), (int)nd, ids.data());
            if (r != ncclSuccess) {
                P->why_not_rccl = std::string("ncclCommInitAll: ") + rccl_api().GetErrorString(r);
                P->comms.clear();
            }
        }
        if (P->comms.empty()) enable_peer_access(h, P);
        if (getenv("BISBM_POOL_LOG")) {
            if (P->comms.empty())
                fprintf(stderr, "[bisbm pool] peer-copy path (%s); direct peer access on %u of %zu device pairs\n", P->why_not_rccl.c_str(),
                        P->peer_direct, nd * (nd - 1));
            else
                fprintf(stderr, "[bisbm pool] RCCL path: %zu communicator(s) in this process (ncclCommInitAll)\n", nd);
        }
    }
    DevicePool* P = h->pool;
    if (P->kmax_cap >= kmax) return BISBM_OK;
    // (re)allocation: nothing of the old size counts any more from here on -- a failure part-way leaves kmax_cap = 0, so the
    // next call starts over instead of handing half-made buffers to the collectives
    P->kmax_cap = 0;
    auto release = [&](auto& vec) {
        for (size_t i = 0; i < vec.size(); ++i)
            if (vec[i]) {
                (void)hipSetDevice(h->devs[i]->device);
                (void)hipFree(vec[i]);
                vec[i] = nullptr;
            }
        vec.assign(nd, nullptr);
    };
    release(P->d_red);
    release(P->d_stage);
    release(P->d_lab);
    release(P->d_all);
    for (size_t i = 0; i < nd; ++i) {
        HIPCHK(h, hipSetDevice(h->devs[i]->device));
        HIPCHK(h, dalloc(&P->d_lab[i], (size_t)P->per));
        HIPCHK(h, dalloc(&P->d_all[i], (size_t)P->per * nd));
        HIPCHK(h, dalloc(&P->d_red[i], (size_t)P->per * kmax));
        if (P->comms.empty()) HIPCHK(h, dalloc(&P->d_stage[i], (size_t)P->per * kmax));
    }
    P->kmax_cap = kmax;
    return BISBM_OK;
}

}  // namespace

// ---- the calls of the C ABI on a container ------------------------------------------------------------------------------
int multi_common_shape(bisbm_engine* h, uint32_t* ka, uint32_t* kb) {
    uint32_t a0 = 0, b0 = 0;
    for (size_t i = 0; i < h->devs.size(); ++i) {
        uint32_t a, b;
        if (bisbm_get_ka_kb(h->devs[i], &a, &b) != BISBM_OK || (i > 0 && (a != a0 || b != b0)))
            return fail(h, BISBM_ERR_STATE, "the chains of this handle have different block counts: ask per chain (bisbm_get_ka_kb_chain)");
        a0 = a, b0 = b;
    }
    h->ka = a0, h->kb = b0, h->K = a0 + b0;
    if (ka) *ka = a0;
    if (kb) *kb = b0;
    return BISBM_OK;
}

int multi_anneal(bisbm_engine* h, int schedule, const float kwargs[2], uint64_t duration_steps, uint64_t steps_await, double* acc_rate_out) {
    const int rc = on_devices(h, [&](bisbm_engine* d, size_t i) {
        return bisbm_anneal(d, schedule, kwargs, duration_steps, steps_await, acc_rate_out ? acc_rate_out + h->dev_first[i] : nullptr);
    });
    h->last_kernel_ms = 0;
    h->last_updates = 0;
    for (bisbm_engine* d : h->devs) {  // the devices run side by side: the call lasted as long as the slowest one
        h->last_kernel_ms = std::max(h->last_kernel_ms, d->last_kernel_ms);
        h->last_updates += d->last_updates;
        h->last_pass_steps = std::max(d == h->devs[0] ? 0u : h->last_pass_steps, d->last_pass_steps);
    }
    return rc;
}

int multi_marginals_get(bisbm_engine* h, uint32_t* counts_out) {
    uint32_t ka, kb;
    if (int rc = multi_common_shape(h, &ka, &kb)) return rc;
    const size_t cnt = (size_t)h->n * std::max(ka, kb);
    std::memset(counts_out, 0, sizeof(uint32_t) * cnt);
    std::vector<uint32_t> part(cnt);
    for (bisbm_engine* d : h->devs) {  // (the whole histogram on the host is the expensive way to look at it: bisbm_marginals_map pools on the devices)
        if (int rc = bisbm_marginals_get(d, part.data())) {
            h->err = d->err;
            return rc;
        }
        for (size_t i = 0; i < cnt; ++i) counts_out[i] += part[i];
    }
    return BISBM_OK;
}

int multi_marginals_map(bisbm_engine* h, uint32_t* labels_out) {
    uint32_t ka, kb;
    if (int rc = multi_common_shape(h, &ka, &kb)) return rc;
    const uint32_t kmax = std::max(ka, kb);
    const size_t nd = h->devs.size();
    for (bisbm_engine* d : h->devs)
        if (!d->d_counts || d->counts_cols != kmax) return fail(h, BISBM_ERR_STATE, "no marginal histogram of the present block counts yet");
    DeviceGuard guard;
    if (int rc = pool_prepare(h, kmax)) return rc;
    DevicePool* P = h->pool;
    const uint64_t per = P->per;
    const size_t slice = (size_t)per * kmax;
    if (!P->comms.empty()) {
        // reduce-scatter by node range, argmax on the owner, all-gather of the labels: one group call each, every
        // device on its own stream.  A group that has been opened is always closed, whatever happens inside it -- an open
        // group would swallow every later RCCL call of the process.
        RcclApi& R = rccl_api();
        hipError_t he = hipSuccess;
        ncclResult_t r = R.GroupStart();
        if (r != ncclSuccess) return fail(h, BISBM_ERR_HIP, "ncclGroupStart: %s", R.GetErrorString(r));
        for (size_t i = 0; i < nd && r == ncclSuccess && he == hipSuccess; ++i) {
            he = hipSetDevice(h->devs[i]->device);
            if (he == hipSuccess) r = R.ReduceScatter(h->devs[i]->d_counts, P->d_red[i], slice, ncclUint32, ncclSum, P->comms[i], h->devs[i]->stream);
        }
        ncclResult_t re = R.GroupEnd();
        if (he != hipSuccess) return fail(h, BISBM_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(he));
        if (r != ncclSuccess || re != ncclSuccess) return fail(h, BISBM_ERR_HIP, "ncclReduceScatter: %s", R.GetErrorString(r != ncclSuccess ? r : re));
        for (size_t i = 0; i < nd; ++i) {
            HIPCHK(h, hipSetDevice(h->devs[i]->device));
            HIPCHK(h, launch_marginal_map(P->d_red[i], (uint32_t)per, kmax, (uint32_t)(i * per), (uint32_t)h->n, (uint32_t)h->na, ka, P->d_lab[i], h->devs[i]->stream));
        }
        r = R.GroupStart();
        if (r != ncclSuccess) return fail(h, BISBM_ERR_HIP, "ncclGroupStart: %s", R.GetErrorString(r));
        for (size_t i = 0; i < nd && r == ncclSuccess && he == hipSuccess; ++i) {
            he = hipSetDevice(h->devs[i]->device);
            if (he == hipSuccess) r = R.AllGather(P->d_lab[i], P->d_all[i], (size_t)per * sizeof(uint16_t), ncclUint8, P->comms[i], h->devs[i]->stream);
        }
        re = R.GroupEnd();
        if (he != hipSuccess) return fail(h, BISBM_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(he));
        if (r != ncclSuccess || re != ncclSuccess) return fail(h, BISBM_ERR_HIP, "ncclAllGather: %s", R.GetErrorString(r != ncclSuccess ? r : re));
        std::vector<uint16_t> lab((size_t)per * nd);
        HIPCHK(h, hipSetDevice(h->devs[0]->device));
        HIPCHK(h, hipMemcpyAsync(lab.data(), P->d_all[0], sizeof(uint16_t) * lab.size(), hipMemcpyDeviceToHost, h->devs[0]->stream));
        for (size_t i = 0; i < nd; ++i) {
            HIPCHK(h, hipSetDevice(h->devs[i]->device));
            HIPCHK(h, hipStreamSynchronize(h->devs[i]->stream));
        }
        for (uint64_t v = 0; v < h->n; ++v) labels_out[v] = lab[v];
        return BISBM_OK;
    }
    // peer-copy path: the owner of a node range starts from its own slice, pulls every other device's slice of the range and
    // adds it; the owners work side by side (one host thread and one stream each)
    std::vector<uint16_t> lab((size_t)per * nd);
    const int rc = on_devices(h, [&](bisbm_engine* d, size_t i) -> int {
        HIPCHK(d, hipSetDevice(d->device));
        HIPCHK(d, hipMemcpyAsync(P->d_red[i], d->d_counts + i * slice, sizeof(uint32_t) * slice, hipMemcpyDeviceToDevice, d->stream));
        for (size_t j = 0; j < nd; ++j) {
            if (j == i) continue;
            HIPCHK(d, hipMemcpyPeerAsync(P->d_stage[i], d->device, h->devs[j]->d_counts + i * slice, h->devs[j]->device, sizeof(uint32_t) * slice, d->stream));
            HIPCHK(d, launch_counts_add(P->d_red[i], P->d_stage[i], slice, d->stream));
        }
        HIPCHK(d, launch_marginal_map(P->d_red[i], (uint32_t)per, kmax, (uint32_t)(i * per), (uint32_t)h->n, (uint32_t)h->na, ka, P->d_lab[i], d->stream));
        HIPCHK(d, hipMemcpyAsync(lab.data() + i * per, P->d_lab[i], sizeof(uint16_t) * per, hipMemcpyDeviceToHost, d->stream));
        HIPCHK(d, hipStreamSynchronize(d->stream));
        return BISBM_OK;
    });
    if (rc) return rc;
    for (uint64_t v = 0; v < h->n; ++v) labels_out[v] = lab[v];
    return BISBM_OK;
}

void multi_free(bisbm_engine* h) {
    DeviceGuard guard;
    pool_free(h);
    for (bisbm_engine* d : h->devs) {
        free_all(d);
        delete d;
    }
    h->devs.clear();
}

}  // namespace bisbm

extern "C" {

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

}  // extern "C"
