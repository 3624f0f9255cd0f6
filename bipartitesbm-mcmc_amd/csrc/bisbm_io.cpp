// bisbm_io.cpp -- edge list / membership text I/O with the reference's exact line semantics
// (include/bisbm_io.h).  Host only; a buffered byte scanner instead of one stringstream per line.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

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
