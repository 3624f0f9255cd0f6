// bisbm_io.cpp -- edge list / membership text I/O with the reference's exact line semantics
// (include/bisbm_io.h).  Host only; a buffered byte scanner instead of one stringstream per line.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

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
                     (h.n_edges == 0 || std::fread(cl, sizeof(uint32_t), 2 * h.n_edges, f) == 2 * h.n_edges) && rp[0] == 0 &&
                     rp[n] == 2 * h.n_edges;
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
