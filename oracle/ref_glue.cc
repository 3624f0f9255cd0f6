// ref_glue.cc -- extern "C" entry points over the three Boost-free translation units of the
// reference, compiled from where they lie under /root/reference (see Makefile, target ref).
// TEST INFRASTRUCTURE ONLY: used to pin oracle/bisbm_oracle.c; never linked by the product.
// The other reference TUs (blockmodel.cc, metropolis_hasting.cc, support/cache.cc,
// support/int_part.cc, mcmc_main.cc) include Boost headers, which this image lacks, so they are
// unbuildable here (DESIGN.md "Oracle and pinning").
#include <cstdint>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#include "graph_utilities.hh"   // /root/reference/src/graph_utilities.hh:5-17
#include "output_functions.hh"  // /root/reference/src/output_functions.hh:20-29

double spence(double);  // /root/reference/src/support/spence.cc:108

extern "C" {

double ref_spence(double x) { return spence(x); }

// load_edge_list + edge_to_adj, flattened to CSR.  Returns number of edges, -1 on open failure.
// Call once with rowptr == nullptr to get sizes (n_out = adjacency size, nnz_out = entries).
long ref_edge_list_to_csr(const char* path, size_t num_vertices, uint64_t* rowptr, uint32_t* col,
                          size_t* n_out, size_t* nnz_out) {
    edge_list_t el;
    if (!load_edge_list(el, path)) return -1;
    adj_list_t adj = edge_to_adj(el, num_vertices);
    size_t nnz = 0;
    for (auto const& row : adj) nnz += row.size();
    *n_out = adj.size();
    *nnz_out = nnz;
    if (rowptr) {
        size_t p = 0;
        for (size_t i = 0; i < adj.size(); ++i) {
            rowptr[i] = p;
            for (auto nb : adj[i]) col[p++] = (uint32_t)nb;
        }
        rowptr[adj.size()] = p;
    }
    return (long)el.size();
}

long ref_edge_list_raw(const char* path, uint64_t* a, uint64_t* b, size_t cap) {
    edge_list_t el;
    if (!load_edge_list(el, path)) return -1;
    for (size_t i = 0; i < el.size() && i < cap; ++i) {
        a[i] = el[i].first;
        b[i] = el[i].second;
    }
    return (long)el.size();
}

long ref_load_memberships(const char* path, uint32_t* out, size_t cap) {
    uint_vec_t mb;
    if (!load_memberships(mb, path)) return -1;
    for (size_t i = 0; i < mb.size() && i < cap; ++i) out[i] = mb[i];
    return (long)mb.size();
}

size_t ref_output_vec(const uint32_t* v, size_t n, char* out, size_t cap) {
    uint_vec_t vec(v, v + n);
    std::ostringstream os;
    output_vec<uint_vec_t>(vec, os);
    std::string s = os.str();
    if (s.size() + 1 <= cap) std::memcpy(out, s.c_str(), s.size() + 1);
    return s.size();
}
}
