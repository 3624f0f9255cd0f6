// bisbm_marginals.hip -- the per-node label histogram over chains and its MAP labels for one engine; pooling
// over the devices of a handle is bisbm_multi.hip.
#include "bisbm_engine.hpp"

using namespace bisbm;

namespace bisbm {

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

}  // namespace bisbm

extern "C" {

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

int bisbm_marginals_map(bisbm_handle h, uint32_t* labels_out) {
    if (!h) return BISBM_ERR_INVALID_ARG;
    if (!labels_out) return fail(h, BISBM_ERR_INVALID_ARG, "labels_out is NULL");
    return h->devs.empty() ? single_marginals_map(h, labels_out) : multi_marginals_map(h, labels_out);
}

}  // extern "C"
