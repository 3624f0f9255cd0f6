// bisbm.hpp -- C++ mirror of the reference's class API over the C ABI (include/bisbm.h).
//
// A caller written against the reference (src/mcmc_main.cc) keeps its shape: the same type aliases
// (types.hh:8-29), `blockmodel_t` with the constructor of blockmodel.hh:22-23, `metropolis_hasting`
// with `anneal` (metropolis_hasting.hh:48-53) taking one of the five `*_schedule` functions, the
// loaders of graph_utilities.hh and `output_vec` of output_functions.hh.  All chain state lives in
// the HIP library; this header only forwards.  Errors of the library are thrown as std::runtime_error
// (the reference is noexcept-and-terminate).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <iostream>
#include <random>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/bisbm.h"
#include "../../include/bisbm_io.h"

namespace bisbm_host {

using edge_t = std::pair<size_t, size_t>;  // types.hh:8-12
using edge_list_t = std::vector<edge_t>;
using uint_vec_t = std::vector<unsigned int>;
using int_vec_t = std::vector<int>;
using float_vec_t = std::vector<float>;
using int_mat_t = std::vector<std::vector<int>>;
using uint_mat_t = std::vector<std::vector<unsigned int>>;

// adj_list_t of the reference (vector<vector<size_t>>) in CSR form: row v = neighbours of v in edge-file order
struct adj_list_t {
    std::vector<uint64_t> rowptr;
    std::vector<uint32_t> col;
    size_t size() const { return rowptr.empty() ? 0 : rowptr.size() - 1; }
};

// ---- graph_utilities.hh:10-17 ----
inline bool load_edge_list(edge_list_t& edge_list, const std::string& path) {
    edge_list.clear();
    uint64_t *a = nullptr, *b = nullptr;
    const long n = bisbm_io_read_edge_list(path.c_str(), &a, &b);
    if (n < 0) return false;
    edge_list.reserve((size_t)n);
    for (long i = 0; i < n; ++i) edge_list.emplace_back((size_t)a[i], (size_t)b[i]);
    bisbm_io_free(a);
    bisbm_io_free(b);
    return true;
}

inline bool load_memberships(uint_vec_t& memberships, const std::string& path) {
    memberships.clear();
    uint32_t* p = nullptr;
    const long n = bisbm_io_read_memberships(path.c_str(), &p);
    if (n < 0) return false;
    memberships.assign(p, p + n);
    bisbm_io_free(p);
    return true;
}

inline adj_list_t edge_to_adj(const edge_list_t& edge_list, size_t num_vertices = 0) {
    size_t n = num_vertices;
    for (auto const& e : edge_list) n = std::max(n, std::max(e.first, e.second) + 1);  // graph_utilities.cc:39-44
    std::vector<uint64_t> a(edge_list.size()), b(edge_list.size());
    for (size_t i = 0; i < edge_list.size(); ++i) {
        a[i] = edge_list[i].first;
        b[i] = edge_list[i].second;
    }
    adj_list_t adj;
    adj.rowptr.assign(n + 1, 0);
    adj.col.assign(2 * edge_list.size() + 1, 0);
    bisbm_io_edges_to_csr(a.data(), b.data(), a.size(), n, adj.rowptr.data(), adj.col.data());
    adj.col.resize(2 * edge_list.size());
    return adj;
}

// load_edge_list + edge_to_adj with the optional binary CSR cache of include/bisbm_io.h (num_vertices as in edge_to_adj)
inline bool load_adj_cached(adj_list_t& adj, const std::string& path, size_t num_vertices, bool use_cache, bool* hit = nullptr) {
    uint64_t* rp = nullptr;
    uint32_t* cl = nullptr;
    uint64_t ne = 0;
    int h = 0;
    if (bisbm_io_load_csr(path.c_str(), num_vertices, use_cache ? 1 : 0, &rp, &cl, &ne, &h) != 0) return false;
    adj.rowptr.assign(rp, rp + num_vertices + 1);
    adj.col.assign(cl, cl + 2 * ne);
    bisbm_io_free(rp);
    bisbm_io_free(cl);
    if (hit) *hit = h != 0;
    return true;
}

// Ingest-time renumbering for ids without structure (include/bisbm_io.h): new_id[v] = id of node v in the engine's graph
inline std::vector<uint32_t> locality_order(const adj_list_t& adj, size_t na) {
    std::vector<uint32_t> new_id(adj.size());
    if (bisbm_io_locality_order(adj.size(), na, adj.rowptr.data(), adj.col.data(), new_id.data()) != 0)
        throw std::runtime_error("locality_order: bad graph");
    return new_id;
}
inline adj_list_t permute_adj(const adj_list_t& adj, const std::vector<uint32_t>& new_id) {
    adj_list_t out;
    out.rowptr.assign(adj.rowptr.size(), 0);
    out.col.assign(adj.col.size() + 1, 0);
    if (bisbm_io_permute_csr(adj.size(), adj.rowptr.data(), adj.col.data(), new_id.data(), out.rowptr.data(), out.col.data()) != 0)
        throw std::runtime_error("permute_adj: bad permutation");
    out.col.resize(adj.col.size());
    return out;
}

// ---- output_functions.hh:20-29 ----
template <typename T>
void output_vec(const T& vec, std::ostream& stream = std::clog) {
    for (auto it = vec.begin(); it != vec.end(); ++it) stream << *it << " ";
    stream << "\n";
}

// ---- cooling schedules, metropolis_hasting.hh:13-21 / metropolis_hasting.cc:10-37 ----
// The functions exist so that `&exponential_schedule` etc. can be passed to anneal() as in the reference;
// the kernels evaluate the same expressions (pow/log ones from a host table built with these very calls).
inline double exponential_schedule(size_t t, float_vec_t kw) noexcept { return kw[0] * std::pow(kw[1], t); }
inline double linear_schedule(size_t t, float_vec_t kw) noexcept { return kw[0] - kw[1] * t; }
inline double logarithmic_schedule(size_t t, float_vec_t kw) noexcept {
    const float x = t + kw[1];
    const size_t i = (size_t)x;
    return kw[0] / (i == 0 ? 0. : std::log((double)i));
}
inline double constant_schedule(size_t, float_vec_t kw) noexcept { return kw[0]; }
inline double abrupt_cool_schedule(size_t t, float_vec_t kw) noexcept { return t < kw[0] ? 1. : 0.; }
using schedule_fn = double (*)(size_t, float_vec_t);

// The stage plan of the merge drivers, support/util.hh:99-145: the side with the larger drop goes down geometrically
// (floor(start / ratio^i) until <= end), the other side gets the same number of points with the ratio
// pow(start / end [integer division], 1 / (n - 1)).
inline std::pair<std::vector<int>, std::vector<int>> geospace(long start_a_in, long end_a_in, long start_b_in,
                                                              long end_b_in, double ratio) {
    if (ratio <= 1.) return {std::vector<int>{0}, std::vector<int>{0}};
    int start_a = (int)start_a_in, end_a = (int)end_a_in, start_b = (int)start_b_in, end_b = (int)end_b_in;
    const bool reverse = start_a - end_a < start_b - end_b;
    if (reverse) {
        std::swap(start_a, start_b);
        std::swap(end_a, end_b);
    }
    std::vector<int> ga, gb;
    int d = start_a;
    for (size_t i = 1; d > end_a; ++i) {
        ga.push_back(d);
        d = (int)std::floor(start_a / std::pow(ratio, (double)i));
    }
    ga.push_back(end_a);
    const size_t n = ga.size();
    const double r_ = std::pow((double)(start_b / end_b), 1. / (double)(n - 1));
    for (size_t idx = 0; idx + 1 < n; ++idx) gb.push_back((int)std::floor(start_b / std::pow(r_, (double)idx)));
    gb.push_back(end_b);
    return reverse ? std::make_pair(gb, ga) : std::make_pair(ga, gb);
}

struct engine_options {  // what the reference does not have: chains, device, RNG definition
    uint32_t n_chains = 1;
    uint32_t first_chain_id = 0;
    int device = 0;
    std::vector<int> devices;  // more than one entry: the chains are spread over these devices behind one handle (bisbm_create_multi)
    int rng_mode = BISBM_RNG_MT19937_COMPAT;
    uint64_t seed = 0;      // std::mt19937 engine(seed) of mcmc_main.cc:242, or the Philox key
    uint64_t gen_seed = 0;  // the hidden blockmodel_t::gen (blockmodel.hh:17-18); the reference seeds it from random_device
};

class blockmodel_t {
public:
    // blockmodel.hh:22-23; `g` is accepted and unused exactly like in the reference
    blockmodel_t(const uint_vec_t& memberships, uint_vec_t types, size_t /*g*/, size_t KA, size_t KB, double epsilon,
                 const adj_list_t* adj_list_ptr, const engine_options& opt = engine_options())
        : KA_(KA), KB_(KB), n_chains_(opt.n_chains) {
        size_t na = 0, nb = 0;
        for (auto t : types) (t == 0 ? na : nb) += 1;
        n_ = na + nb;
        const int rc = opt.devices.size() > 1
                           ? bisbm_create_multi(&h_, n_, na, nb, adj_list_ptr->rowptr.data(), adj_list_ptr->col.data(), (uint32_t)KA,
                                                (uint32_t)KB, epsilon, opt.n_chains, opt.first_chain_id, opt.devices.data(),
                                                (int)opt.devices.size(), opt.rng_mode, opt.seed, opt.gen_seed)
                           : bisbm_create(&h_, n_, na, nb, adj_list_ptr->rowptr.data(), adj_list_ptr->col.data(), (uint32_t)KA,
                                          (uint32_t)KB, epsilon, opt.n_chains, opt.first_chain_id,
                                          opt.devices.empty() ? opt.device : opt.devices[0], opt.rng_mode, opt.seed, opt.gen_seed);
        if (rc != BISBM_OK) throw std::runtime_error(std::string("bisbm_create: ") + bisbm_last_error(nullptr));
        check(bisbm_set_memberships(h_, BISBM_ALL_CHAINS, memberships.data()));
    }
    ~blockmodel_t() {
        if (h_) bisbm_destroy(h_);
    }
    blockmodel_t(const blockmodel_t&) = delete;
    blockmodel_t& operator=(const blockmodel_t&) = delete;

    void init_bisbm() { check(bisbm_init(h_)); }  // blockmodel.cc:682-688
    // blockmodel.cc:672-680; the engine lives in the library, the arguments keep the reference's signature
    void shuffle_bisbm(std::mt19937& /*engine*/, size_t /*NA*/, size_t /*NB*/) { check(bisbm_shuffle(h_)); }
    void shuffle_bisbm() { check(bisbm_shuffle(h_)); }

    const uint_vec_t* get_memberships(uint32_t chain = 0) {  // blockmodel.cc:87
        memberships_.resize(n_);
        check(bisbm_get_memberships(h_, chain, memberships_.data()));
        return &memberships_;
    }
    // blockmodel.cc:109-206 and :208-271 (call sites mcmc_main.cc:365,385,429,434,446); the engine lives in the
    // library.  Negative diffs (agg_split) throw: not provided, see include/bisbm.h.
    void agg_merge(std::mt19937& /*engine*/, int diff_a, int diff_b, int nm) { agg_merge(diff_a, diff_b, nm); }
    void agg_merge(std::mt19937& /*engine*/, int diff, int nm) { agg_merge(diff, nm); }
    void agg_merge(int diff_a, int diff_b, int nm) {
        check(bisbm_agg_merge(h_, diff_a, diff_b, nm));
        refresh_k();
    }
    void agg_merge(int diff, int nm) {
        check(bisbm_agg_merge_total(h_, diff, nm));
        refresh_k();
    }
    // block counts of one chain (the one-argument agg_merge lets every chain end with its own, blockmodel.cc:208-271)
    std::pair<size_t, size_t> ka_kb(uint32_t chain) const {
        uint32_t ka = 0, kb = 0;
        check(bisbm_get_ka_kb_chain(h_, chain, &ka, &kb));
        return {ka, kb};
    }
    size_t get_KA() const noexcept { return KA_; }
    size_t get_KB() const noexcept { return KB_; }
    int get_num_edges() const {
        uint64_t e = 0;
        bisbm_get_sizes(h_, nullptr, &e, nullptr, nullptr);
        return (int)e;
    }
    double get_entropy(uint32_t chain = 0) {  // blockmodel.cc:91
        std::vector<double> v(n_chains_);
        check(bisbm_get_cum_dS(h_, v.data()));
        return v[chain];
    }
    std::vector<double> entropy_all() {
        std::vector<double> v(n_chains_);
        check(bisbm_entropy(h_, v.data()));
        return v;
    }
    double entropy(uint32_t chain = 0) { return entropy_all()[chain]; }  // blockmodel.cc:753-787
    void summary(uint32_t chain = 0) {                                    // blockmodel.cc:748-751
        std::clog << "(Ka, Kb) = (" << KA_ << ", " << KB_ << ") \n";
        std::clog << "entropy: " << entropy(chain) << "\n";
    }
    // Marginal histogram over samples and chains (what README.md:49-53 describes for "marginalization" and the reference's
    // code never does): reset, add the present labels of every chain, read the MAP label of every node
    // (most frequent block, ties -> lowest index) in the reference's block numbering.
    void marginals_reset() { check(bisbm_marginals_reset(h_)); }
    void marginals_accumulate() { check(bisbm_marginals_accumulate(h_, nullptr)); }
    // the marginal estimate of README.md:49-53: every node's most frequent block, pooled over the handle's devices on the
    // devices (bisbm_marginals_map)
    uint_vec_t marginal_map_labels(size_t /*NA*/) {
        std::vector<uint32_t> lab(n_);
        check(bisbm_marginals_map(h_, lab.data()));
        return uint_vec_t(lab.begin(), lab.end());
    }
    bisbm_handle handle() const { return h_; }
    uint32_t n_chains() const { return n_chains_; }

private:
    void check(int rc) const {
        if (rc != BISBM_OK) throw std::runtime_error(std::string("bisbm: ") + bisbm_last_error(h_));
    }
    void refresh_k() {
        uint32_t ka = 0, kb = 0;
        check(bisbm_get_ka_kb_chain(h_, 0, &ka, &kb));  // (chain 0's: after agg_merge(diff, nm) chains may differ, see ka_kb())
        KA_ = ka;
        KB_ = kb;
    }
    bisbm_handle h_ = nullptr;
    size_t KA_, KB_, n_ = 0;
    uint32_t n_chains_;
    uint_vec_t memberships_;
};

class metropolis_hasting {
public:
    // metropolis_hasting.hh:48-53.  Returns the acceptance rate of chain 0; rates() has all chains.
    double anneal(blockmodel_t& blockmodel, schedule_fn cooling_schedule, const float_vec_t& kwargs, size_t duration,
                  size_t steps_await, std::mt19937& /*engine*/) {
        return anneal(blockmodel, cooling_schedule, kwargs, duration, steps_await);
    }
    double anneal(blockmodel_t& blockmodel, schedule_fn cooling_schedule, const float_vec_t& kwargs, size_t duration,
                  size_t steps_await) {
        int id;
        if (cooling_schedule == &exponential_schedule)
            id = BISBM_SCHED_EXPONENTIAL;
        else if (cooling_schedule == &linear_schedule)
            id = BISBM_SCHED_LINEAR;
        else if (cooling_schedule == &logarithmic_schedule)
            id = BISBM_SCHED_LOGARITHMIC;
        else if (cooling_schedule == &constant_schedule)
            id = BISBM_SCHED_CONSTANT;
        else if (cooling_schedule == &abrupt_cool_schedule)
            id = BISBM_SCHED_ABRUPT_COOL;
        else
            throw std::runtime_error("anneal: pass one of the five *_schedule functions");
        float kw[2] = {kwargs.size() > 0 ? kwargs[0] : 0.f, kwargs.size() > 1 ? kwargs[1] : 0.f};
        rates_.assign(blockmodel.n_chains(), 0.);
        const int rc = bisbm_anneal(blockmodel.handle(), id, kw, duration, steps_await, rates_.data());
        if (rc != BISBM_OK) throw std::runtime_error(std::string("bisbm_anneal: ") + bisbm_last_error(blockmodel.handle()));
        return rates_[0];
    }
    const std::vector<double>& rates() const { return rates_; }

private:
    std::vector<double> rates_;
};

}  // namespace bisbm_host
