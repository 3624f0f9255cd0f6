// bisbm_io.cpp -- edge list / membership text I/O with the reference's exact line semantics
// (include/bisbm_io.h).  Host only; a buffered byte scanner instead of one stringstream per line.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include <algorithm>
#include <cmath>
#include <numeric>
#include <thread>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/bisbm_io.h"

namespace {

struct LineReader {
    FILE* f;
    std::vector<char> buf;
    size_t pos = 0, len = 0;
    bool eof = false;
    explicit LineReader(FILE* f_) : f(f_), buf(1 << 20) {}
    int get() {
        if (pos == len) {
            if (eof) return -1;
            len = fread(buf.data(), 1, buf.size(), f);
            pos = 0;
            if (len == 0) {
                eof = true;
                return -1;
            }
        }
        return (unsigned char)buf[pos++];
    }
    // std::getline: false only when nothing at all could be read
    bool line(std::string& out) {
        out.clear();
        bool any = false;
        int c;
        while ((c = get()) >= 0) {
            any = true;
            if (c == '\n') break;
            out.push_back((char)c);
        }
        return any;
    }
};

inline bool is_blank(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; }

// `stream >> size_t`: 1 = value read; 0 = parse failure (C++11 writes 0 and sets failbit);
// -1 = nothing but blanks left (sentry fails, the variable is untouched)
int extract(const char*& p, uint64_t& v) {
    while (is_blank(*p)) ++p;
    if (*p == '\0') return -1;
    const char* q = p;
    bool neg = false;
    if (*q == '+' || *q == '-') {
        neg = *q == '-';
        ++q;
    }
    if (*q < '0' || *q > '9') return 0;
    uint64_t x = 0;
    while (*q >= '0' && *q <= '9') x = x * 10 + (uint64_t)(*q++ - '0');
    v = neg ? (uint64_t)0 - x : x;
    p = q;
    return 1;
}

template <class T>
T* to_malloc(const std::vector<T>& v) {
    T* p = (T*)std::malloc(sizeof(T) * (v.empty() ? 1 : v.size()));
    if (p && !v.empty()) std::memcpy(p, v.data(), sizeof(T) * v.size());
    return p;
}

}  // namespace

extern "C" {

long bisbm_io_read_edge_list(const char* path, uint64_t** a, uint64_t** b) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return -1;
    LineReader rd(f);
    std::vector<uint64_t> va, vb;
    uint64_t node_a = 0, node_b = 0;  // declared outside the loop in the reference (:23)
    std::string line;
    while (rd.line(line)) {
        const char* p = line.c_str();
        uint64_t v = 0;
        const int first = extract(p, v);
        if (first == 1) {
            node_a = v;
            const int second = extract(p, v);
            if (second == 1)
                node_b = v;
            else if (second == 0)
                node_b = 0;
        } else if (first == 0) {
            node_a = 0;  // failbit is now set: the second extraction does nothing
        }
        va.push_back(node_a);
        vb.push_back(node_b);
    }
    std::fclose(f);
    *a = to_malloc(va);
    *b = to_malloc(vb);
    return (long)va.size();
}

long bisbm_io_read_memberships(const char* path, uint32_t** labels) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return -1;
    LineReader rd(f);
    std::vector<uint32_t> out;
    uint64_t membership = 0;
    std::string line;
    while (rd.line(line)) {
        const char* p = line.c_str();
        uint64_t v = 0;
        const int r = extract(p, v);
        if (r == 1)
            membership = v;
        else if (r == 0)
            membership = 0;
        out.push_back((uint32_t)membership);
    }
    std::fclose(f);
    *labels = to_malloc(out);
    return (long)out.size();
}

int bisbm_io_edges_to_csr(const uint64_t* a, const uint64_t* b, size_t n_edges, uint64_t n, uint64_t* rowptr,
                          uint32_t* col) {
    std::memset(rowptr, 0, sizeof(uint64_t) * (n + 1));
    for (size_t e = 0; e < n_edges; ++e) {
        if (a[e] >= n || b[e] >= n) return -1;
        ++rowptr[a[e] + 1];
        ++rowptr[b[e] + 1];
    }
    for (uint64_t v = 0; v < n; ++v) rowptr[v + 1] += rowptr[v];
    std::vector<uint64_t> cursor(rowptr, rowptr + n);
    for (size_t e = 0; e < n_edges; ++e) {  // push_back(b) on row a, then push_back(a) on row b (:45-46)
        col[cursor[a[e]]++] = (uint32_t)b[e];
        col[cursor[b[e]]++] = (uint32_t)a[e];
    }
    return 0;
}

namespace {
struct CsrCacheHeader {
    char magic[8];
    uint32_t version, reserved;
    uint64_t src_size;
    int64_t src_mtime_ns;
    uint64_t n, n_edges;
};
static_assert(sizeof(CsrCacheHeader) == 48, "cache header layout");
constexpr char kCsrMagic[8] = {'B', 'I', 'S', 'B', 'M', 'C', 'S', 'R'};

// rowptr starts at 0 and never decreases, every neighbour id is a node: what the walks below index with
bool csr_is_sound(uint64_t n, const uint64_t* rowptr, const uint32_t* col) {
    if (rowptr[0] != 0) return false;
    for (uint64_t v = 0; v < n; ++v)
        if (rowptr[v + 1] < rowptr[v]) return false;
    if (rowptr[n] && !col) return false;
    for (uint64_t e = 0; e < rowptr[n]; ++e)
        if (col[e] >= n) return false;
    return true;
}
}  // namespace

int bisbm_io_load_csr(const char* path, uint64_t n, int use_cache, uint64_t** rowptr, uint32_t** col, uint64_t* n_edges,
                      int* cache_hit) {
    if (cache_hit) *cache_hit = 0;
    *rowptr = nullptr;
    *col = nullptr;
    struct stat st;
    if (stat(path, &st) != 0) return -1;
    const std::string cpath = std::string(path) + ".bisbm_csr";
    const int64_t mtime_ns = (int64_t)st.st_mtim.tv_sec * 1000000000ll + (int64_t)st.st_mtim.tv_nsec;
    if (use_cache) {
        if (FILE* f = std::fopen(cpath.c_str(), "rb")) {
            CsrCacheHeader h;
            struct stat cs;
            bool ok = std::fread(&h, sizeof(h), 1, f) == 1 && std::memcmp(h.magic, kCsrMagic, 8) == 0 && h.version == 1 &&
                      h.src_size == (uint64_t)st.st_size && h.src_mtime_ns == mtime_ns && h.n == n && fstat(fileno(f), &cs) == 0 &&
                      (uint64_t)cs.st_size == sizeof(h) + sizeof(uint64_t) * (n + 1) + sizeof(uint32_t) * 2 * h.n_edges;
            if (ok) {
                uint64_t* rp = (uint64_t*)std::malloc(sizeof(uint64_t) * (n + 1));
                uint32_t* cl = (uint32_t*)std::malloc(sizeof(uint32_t) * (2 * h.n_edges + 1));
                ok = rp && cl && std::fread(rp, sizeof(uint64_t), n + 1, f) == n + 1 &&
                     (h.n_edges == 0 || std::fread(cl, sizeof(uint32_t), 2 * h.n_edges, f) == 2 * h.n_edges) &&
                     rp[n] == 2 * h.n_edges && csr_is_sound(n, rp, cl);  // a damaged body falls back to the text
                if (ok) {
                    std::fclose(f);
                    *rowptr = rp;
                    *col = cl;
                    *n_edges = h.n_edges;
                    if (cache_hit) *cache_hit = 1;
                    return 0;
                }
                std::free(rp);
                std::free(cl);
            }
            std::fclose(f);
        }
    }
    uint64_t *a = nullptr, *b = nullptr;
    const long ne = bisbm_io_read_edge_list(path, &a, &b);
    if (ne < 0) return -1;
    uint64_t* rp = (uint64_t*)std::malloc(sizeof(uint64_t) * (n + 1));
    uint32_t* cl = (uint32_t*)std::malloc(sizeof(uint32_t) * (2 * (size_t)ne + 1));
    const int rc = bisbm_io_edges_to_csr(a, b, (size_t)ne, n, rp, cl);
    std::free(a);
    std::free(b);
    if (rc != 0) {
        std::free(rp);
        std::free(cl);
        return -2;
    }
    *rowptr = rp;
    *col = cl;
    *n_edges = (uint64_t)ne;
    if (use_cache) {  // best effort: a read-only directory just means no cache
        const std::string tmp = cpath + ".tmp." + std::to_string((long)getpid());
        if (FILE* f = std::fopen(tmp.c_str(), "wb")) {
            CsrCacheHeader h;
            std::memcpy(h.magic, kCsrMagic, 8);
            h.version = 1;
            h.reserved = 0;
            h.src_size = (uint64_t)st.st_size;
            h.src_mtime_ns = mtime_ns;
            h.n = n;
            h.n_edges = (uint64_t)ne;
            const bool ok = std::fwrite(&h, sizeof(h), 1, f) == 1 && std::fwrite(rp, sizeof(uint64_t), n + 1, f) == n + 1 &&
                            (ne == 0 || std::fwrite(cl, sizeof(uint32_t), 2 * (size_t)ne, f) == 2 * (size_t)ne);
            const bool closed = std::fclose(f) == 0;
            if (!(ok && closed && std::rename(tmp.c_str(), cpath.c_str()) == 0)) std::remove(tmp.c_str());
        }
    }
    return 0;
}

namespace {
#ifndef BISBM_EMBED_DIM
#define BISBM_EMBED_DIM 4
#endif
#ifndef BISBM_EMBED_ROUNDS
#define BISBM_EMBED_ROUNDS 8
#endif
constexpr int kEmbedDim = BISBM_EMBED_DIM;
constexpr int kEmbedRounds = BISBM_EMBED_ROUNDS;
constexpr size_t kOrderLeaf = 4096;  // nodes of one type per kd cell, about: one tile of the kernel's visit order

// y[v] = mean over the neighbours u of v of x[u], rows [lo, hi)
void neighbour_means(const uint64_t* rowptr, const uint32_t* col, const float* x, float* y, uint64_t lo, uint64_t hi) {
    for (uint64_t v = lo; v < hi; ++v) {
        float acc[kEmbedDim] = {0};
        const uint64_t e0 = rowptr[v], e1 = rowptr[v + 1];
        for (uint64_t e = e0; e < e1; ++e) {
            const float* xu = x + (size_t)col[e] * kEmbedDim;
            for (int j = 0; j < kEmbedDim; ++j) acc[j] += xu[j];
        }
        const float inv = e1 > e0 ? 1.0f / (float)(e1 - e0) : 0.f;
        for (int j = 0; j < kEmbedDim; ++j) y[(size_t)v * kEmbedDim + j] = acc[j] * inv;
    }
}

void parallel_rows(uint64_t lo, uint64_t hi, const std::function<void(uint64_t, uint64_t)>& f) {
    const unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    if (hi - lo < (1u << 15) || nt == 1) {
        f(lo, hi);
        return;
    }
    std::vector<std::thread> th;
    const uint64_t per = (hi - lo + nt - 1) / nt;
    for (unsigned t = 0; t < nt; ++t) {
        const uint64_t a = lo + t * per, b = std::min(hi, a + per);
        if (a < b) th.emplace_back(f, a, b);
    }
    for (auto& t : th) t.join();
}

// centre the columns of rows [lo, hi) and orthonormalise them (modified Gram-Schmidt, double accumulators)
void orthonormalise(float* x, uint64_t lo, uint64_t hi) {
    const size_t cnt = (size_t)(hi - lo);
    if (cnt == 0) return;
    for (int j = 0; j < kEmbedDim; ++j) {
        double mean = 0;
        for (size_t i = 0; i < cnt; ++i) mean += x[(lo + i) * kEmbedDim + j];
        mean /= (double)cnt;
        for (size_t i = 0; i < cnt; ++i) x[(lo + i) * kEmbedDim + j] -= (float)mean;
        for (int k = 0; k < j; ++k) {
            double dot = 0;
            for (size_t i = 0; i < cnt; ++i) dot += (double)x[(lo + i) * kEmbedDim + j] * x[(lo + i) * kEmbedDim + k];
            for (size_t i = 0; i < cnt; ++i) x[(lo + i) * kEmbedDim + j] -= (float)dot * x[(lo + i) * kEmbedDim + k];
        }
        double nrm = 0;
        for (size_t i = 0; i < cnt; ++i) nrm += (double)x[(lo + i) * kEmbedDim + j] * x[(lo + i) * kEmbedDim + j];
        const float inv = nrm > 0 ? (float)(1.0 / std::sqrt(nrm)) : 0.f;
        for (size_t i = 0; i < cnt; ++i) x[(lo + i) * kEmbedDim + j] *= inv;
    }
}

struct KdOrder {
    const float* x;
    uint64_t na;
    std::vector<uint32_t>* out_a;
    std::vector<uint32_t>* out_b;
    // ids[lo, hi): nodes of both types in this cell; split at the median of coordinate `depth % dim`
    void run(std::vector<uint32_t>& ids, size_t lo, size_t hi, int depth) {
        if (hi - lo <= 2 * kOrderLeaf || depth > 48) {
            std::sort(ids.begin() + lo, ids.begin() + hi);  // inside a cell: the caller's order (stable, deterministic)
            for (size_t i = lo; i < hi; ++i) (ids[i] < na ? out_a : out_b)->push_back(ids[i]);
            return;
        }
        const int dim = depth % kEmbedDim;
        const size_t mid = lo + (hi - lo) / 2;
        std::nth_element(ids.begin() + lo, ids.begin() + mid, ids.begin() + hi, [&](uint32_t u, uint32_t v) {
            const float xu = x[(size_t)u * kEmbedDim + dim], xv = x[(size_t)v * kEmbedDim + dim];
            return xu < xv || (xu == xv && u < v);
        });
        run(ids, lo, mid, depth + 1);
        run(ids, mid, hi, depth + 1);
    }
};
}  // namespace

int bisbm_io_locality_order(uint64_t n, uint64_t na, const uint64_t* rowptr, const uint32_t* col, uint32_t* new_id) {
    if (!rowptr || !new_id || na > n || n >= 0xffffffffull || !csr_is_sound(n, rowptr, col)) return -1;
    std::vector<float> x((size_t)n * kEmbedDim), y((size_t)n * kEmbedDim);
    uint64_t state = 0x9E3779B97F4A7C15ull;  // fixed start: splitmix64
    for (size_t i = 0; i < x.size(); ++i) {
        uint64_t z = (state += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        x[i] = (float)((double)(z >> 11) * 0x1.0p-53 - 0.5);
    }
    orthonormalise(x.data(), na, n);
    for (int round = 0; round < kEmbedRounds; ++round) {
        // type a from type b, then type b from type a (one step of the power iteration on D_b^-1 A^T D_a^-1 A)
        parallel_rows(0, na, [&](uint64_t lo, uint64_t hi) { neighbour_means(rowptr, col, x.data(), y.data(), lo, hi); });
        std::copy(y.begin(), y.begin() + (size_t)na * kEmbedDim, x.begin());
        parallel_rows(na, n, [&](uint64_t lo, uint64_t hi) { neighbour_means(rowptr, col, x.data(), y.data(), lo, hi); });
        std::copy(y.begin() + (size_t)na * kEmbedDim, y.end(), x.begin() + (size_t)na * kEmbedDim);
        orthonormalise(x.data(), na, n);
    }
    // both types in the coordinates of type b's vectors: type a = means over its neighbours
    parallel_rows(0, na, [&](uint64_t lo, uint64_t hi) { neighbour_means(rowptr, col, x.data(), y.data(), lo, hi); });
    std::copy(y.begin(), y.begin() + (size_t)na * kEmbedDim, x.begin());
    parallel_rows(na, n, [&](uint64_t lo, uint64_t hi) { neighbour_means(rowptr, col, x.data(), y.data(), lo, hi); });
    std::copy(y.begin() + (size_t)na * kEmbedDim, y.end(), x.begin() + (size_t)na * kEmbedDim);
    std::vector<uint32_t> ids(n), oa, ob;
    std::iota(ids.begin(), ids.end(), 0u);
    oa.reserve(na);
    ob.reserve(n - na);
    KdOrder kd{x.data(), na, &oa, &ob};
    kd.run(ids, 0, (size_t)n, 0);
    for (size_t i = 0; i < oa.size(); ++i) new_id[oa[i]] = (uint32_t)i;
    for (size_t i = 0; i < ob.size(); ++i) new_id[ob[i]] = (uint32_t)(na + i);
    // Refinement: the cell order already lays related cells of the two types side by side, so the MEDIAN position of a
    // node's neighbours (robust against the edges that leave its group) says where the node belongs; re-sort each type
    // by it, a few times.  Positions are fractions of the other type's range.
#ifndef BISBM_ORDER_REFINE
#define BISBM_ORDER_REFINE 4
#endif
    std::vector<float> key(n);
    std::vector<uint32_t> idx;
    for (int pass = 0; pass < 2 * BISBM_ORDER_REFINE; ++pass) {
        const bool type_b = pass & 1;
        const uint64_t lo = type_b ? na : 0, hi = type_b ? n : na, olo = type_b ? 0 : na, ocnt = type_b ? na : n - na;
        if (hi == lo || ocnt == 0) continue;
        parallel_rows(lo, hi, [&](uint64_t r0, uint64_t r1) {
            std::vector<uint32_t> tmp;
            for (uint64_t v = r0; v < r1; ++v) {
                const uint64_t e0 = rowptr[v], e1 = rowptr[v + 1];
                if (e1 == e0) {
                    key[v] = (float)(new_id[v] - lo) / (float)(hi - lo);  // isolated: stays where it is
                    continue;
                }
                tmp.clear();
                for (uint64_t e = e0; e < e1; ++e) tmp.push_back(new_id[col[e]]);
                std::nth_element(tmp.begin(), tmp.begin() + tmp.size() / 2, tmp.end());
                key[v] = (float)(tmp[tmp.size() / 2] - olo) / (float)ocnt;
            }
        });
        idx.resize(hi - lo);
        std::iota(idx.begin(), idx.end(), (uint32_t)lo);
        std::sort(idx.begin(), idx.end(), [&](uint32_t u, uint32_t v) { return key[u] < key[v] || (key[u] == key[v] && new_id[u] < new_id[v]); });
        std::vector<uint32_t> fresh(hi - lo);
        for (size_t i = 0; i < idx.size(); ++i) fresh[idx[i] - lo] = (uint32_t)(lo + i);
        std::copy(fresh.begin(), fresh.end(), new_id + lo);
    }
    return 0;
}

int bisbm_io_permute_csr(uint64_t n, const uint64_t* rowptr, const uint32_t* col, const uint32_t* new_id, uint64_t* rowptr_out,
                         uint32_t* col_out) {
    if (!rowptr || !new_id || !rowptr_out || n >= 0xffffffffull || !csr_is_sound(n, rowptr, col) || (rowptr[n] && !col_out))
        return -1;
    std::vector<uint32_t> old_of(n, 0xffffffffu);
    for (uint64_t v = 0; v < n; ++v) {
        if (new_id[v] >= n || old_of[new_id[v]] != 0xffffffffu) return -1;  // not a bijection of [0, n)
        old_of[new_id[v]] = (uint32_t)v;
    }
    rowptr_out[0] = 0;
    for (uint64_t w = 0; w < n; ++w) rowptr_out[w + 1] = rowptr_out[w] + (rowptr[old_of[w] + 1] - rowptr[old_of[w]]);
    for (uint64_t w = 0; w < n; ++w) {
        const uint64_t v = old_of[w];
        uint64_t o = rowptr_out[w];
        for (uint64_t e = rowptr[v]; e < rowptr[v + 1]; ++e) col_out[o++] = new_id[col[e]];
    }
    return 0;
}

size_t bisbm_io_format_labels(const uint32_t* labels, size_t n, char* out, size_t cap) {
    size_t w = 0;
    char tmp[16];
    for (size_t i = 0; i < n; ++i) {
        const int k = std::snprintf(tmp, sizeof(tmp), "%u ", labels[i]);
        if (out && w + (size_t)k < cap) std::memcpy(out + w, tmp, (size_t)k);
        w += (size_t)k;
    }
    if (out && w + 1 < cap) {
        out[w] = '\n';
        out[w + 1] = '\0';
    }
    return w + 1;
}

void bisbm_io_free(void* p) { std::free(p); }
}
