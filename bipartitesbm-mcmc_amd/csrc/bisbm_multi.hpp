// bisbm_multi.hpp -- several devices behind one handle (bisbm_create_multi): included once by bisbm_runtime.hip, after the
// definition of bisbm_engine.
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
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>

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
    uint32_t kmax_cap = 0;  // columns the slices are allocated for
};

namespace {

uint32_t dev_of_chain(const bisbm_engine* h, uint32_t chain, uint32_t* local) {
    uint32_t i = 0;
    while (i + 1 < h->devs.size() && chain >= h->dev_first[i + 1]) ++i;
    *local = chain - h->dev_first[i];
    return i;
}

// fn(engine of device i, i) on one host thread per device; the first failing device's code and message win
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
    for (size_t i = 0; i < nd; ++i)
        if (rcs[i]) {
            h->err = "device " + std::to_string(h->devs[i]->device) + ": " + h->devs[i]->err;
            return rcs[i];
        }
    return BISBM_OK;
}

void pool_free(bisbm_engine* h) {
    DevicePool* P = h->pool;
    if (!P) return;
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
        if (!P->why_not_rccl.empty() && getenv("BISBM_POOL_LOG"))
            fprintf(stderr, "[bisbm pool] peer-copy path (%s)\n", P->why_not_rccl.c_str());
    }
    DevicePool* P = h->pool;
    if (P->kmax_cap >= kmax) return BISBM_OK;
    for (size_t i = 0; i < nd; ++i) {
        HIPCHK(h, hipSetDevice(h->devs[i]->device));
        for (void* p : {(void*)(i < P->d_red.size() ? P->d_red[i] : nullptr), (void*)(i < P->d_stage.size() ? P->d_stage[i] : nullptr)})
            if (p) (void)hipFree(p);
    }
    P->d_red.assign(nd, nullptr);
    P->d_stage.assign(nd, nullptr);
    if (P->d_lab.empty()) {
        P->d_lab.assign(nd, nullptr);
        P->d_all.assign(nd, nullptr);
        for (size_t i = 0; i < nd; ++i) {
            HIPCHK(h, hipSetDevice(h->devs[i]->device));
            HIPCHK(h, dalloc(&P->d_lab[i], (size_t)P->per));
            HIPCHK(h, dalloc(&P->d_all[i], (size_t)P->per * nd));
        }
    }
    for (size_t i = 0; i < nd; ++i) {
        HIPCHK(h, hipSetDevice(h->devs[i]->device));
        HIPCHK(h, dalloc(&P->d_red[i], (size_t)P->per * kmax));
        if (P->comms.empty()) HIPCHK(h, dalloc(&P->d_stage[i], (size_t)P->per * kmax));
    }
    P->kmax_cap = kmax;
    return BISBM_OK;
}

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

// MAP labels from the internal histogram of one engine (no pooling): argmax kernel + copy
int single_marginals_map(bisbm_engine* h, uint32_t* labels_out) {
    if (!h->groups.empty() && !common_shape(h))
        return fail(h, BISBM_ERR_STATE, "the chains of this handle have different block counts: no common marginal histogram");
    if (!h->d_counts || h->counts_cols != std::max(h->ka, h->kb)) return fail(h, BISBM_ERR_STATE, "no marginal histogram of the present block counts yet");
    HIPCHK(h, hipSetDevice(h->device));
    uint16_t* d_lab = nullptr;
    HIPCHK(h, dalloc(&d_lab, (size_t)h->n));
    std::vector<uint16_t> lab((size_t)h->n);
    hipError_t e = launch_marginal_map(h->d_counts, (uint32_t)h->n, h->counts_cols, 0, (uint32_t)h->n, (uint32_t)h->na, h->ka, d_lab, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(lab.data(), d_lab, sizeof(uint16_t) * h->n, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(d_lab);
    if (e != hipSuccess) return fail(h, BISBM_ERR_HIP, "marginal MAP labels: %s", hipGetErrorString(e));
    for (uint64_t v = 0; v < h->n; ++v) labels_out[v] = lab[v];
    return BISBM_OK;
}

int multi_marginals_map(bisbm_engine* h, uint32_t* labels_out) {
    uint32_t ka, kb;
    if (int rc = multi_common_shape(h, &ka, &kb)) return rc;
    const uint32_t kmax = std::max(ka, kb);
    const size_t nd = h->devs.size();
    for (bisbm_engine* d : h->devs)
        if (!d->d_counts || d->counts_cols != kmax) return fail(h, BISBM_ERR_STATE, "no marginal histogram of the present block counts yet");
    if (int rc = pool_prepare(h, kmax)) return rc;
    DevicePool* P = h->pool;
    const uint64_t per = P->per;
    const size_t slice = (size_t)per * kmax;
    if (!P->comms.empty()) {
        // reduce-scatter by node range, argmax on the owner, all-gather of the labels: one group call each, every
        // device on its own stream
        RcclApi& R = rccl_api();
        ncclResult_t r = R.GroupStart();
        for (size_t i = 0; i < nd && r == ncclSuccess; ++i) {
            HIPCHK(h, hipSetDevice(h->devs[i]->device));
            r = R.ReduceScatter(h->devs[i]->d_counts, P->d_red[i], slice, ncclUint32, ncclSum, P->comms[i], h->devs[i]->stream);
        }
        const ncclResult_t r2 = R.GroupEnd();
        if (r != ncclSuccess || r2 != ncclSuccess) return fail(h, BISBM_ERR_HIP, "ncclReduceScatter: %s", R.GetErrorString(r != ncclSuccess ? r : r2));
        for (size_t i = 0; i < nd; ++i) {
            HIPCHK(h, hipSetDevice(h->devs[i]->device));
            HIPCHK(h, launch_marginal_map(P->d_red[i], (uint32_t)per, kmax, (uint32_t)(i * per), (uint32_t)h->n, (uint32_t)h->na, ka, P->d_lab[i], h->devs[i]->stream));
        }
        r = R.GroupStart();
        for (size_t i = 0; i < nd && r == ncclSuccess; ++i) {
            HIPCHK(h, hipSetDevice(h->devs[i]->device));
            r = R.AllGather(P->d_lab[i], P->d_all[i], (size_t)per * sizeof(uint16_t), ncclUint8, P->comms[i], h->devs[i]->stream);
        }
        const ncclResult_t r3 = R.GroupEnd();
        if (r != ncclSuccess || r3 != ncclSuccess) return fail(h, BISBM_ERR_HIP, "ncclAllGather: %s", R.GetErrorString(r != ncclSuccess ? r : r3));
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
    pool_free(h);
    for (bisbm_engine* d : h->devs) {
        free_all(d);
        delete d;
    }
    h->devs.clear();
}

}  // namespace
