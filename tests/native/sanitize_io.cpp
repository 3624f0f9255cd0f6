// Sanitizer driver for the host I/O of the product library (bisbm_io.cpp compiled by g++ with
// -fsanitize=address,undefined; see tests/test_sanitizers.py).  Feeds the text scanners malformed files, the CSR cache
// damaged cache files, and the renumbering random graphs (isolated nodes, multi-edges, one-node classes); checks the
// results that do not need a second implementation (the cache never changes what a load returns, the renumbering is a
// bijection inside each class, the permuted CSR is the same multigraph).  Exit code 0 = no finding.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../include/bisbm_io.h"

static int fails = 0;
#define CHECK(cond)                                                       \
    do {                                                                  \
        if (!(cond)) {                                                    \
            std::fprintf(stderr, "CHECK failed line %d: %s\n", __LINE__, #cond); \
            ++fails;                                                      \
        }                                                                 \
    } while (0)

static void write_file(const std::string& path, const std::string& data) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) std::exit(3);
    std::fwrite(data.data(), 1, data.size(), f);
    std::fclose(f);
}

static std::string read_file(const std::string& path) {
    std::string out;
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return out;
    char buf[4096];
    size_t k;
    while ((k = std::fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, k);
    std::fclose(f);
    return out;
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const std::string dir = argv[1];
    std::mt19937_64 rng(std::strtoull(argv[2], nullptr, 10));
    auto below = [&](uint64_t k) { return (uint64_t)(rng() % k); };
    const char* tokens[] = {"", "x", "-3", "+7", "12abc", "1e3", "0007", "18446744073709551615", "99999999999999999999999", "4", "17", "0"};
    const char* seps[] = {" ", "\t", "  ", ",", ""};

    // ---- text scanners on junk ----
    for (int it = 0; it < 400; ++it) {
        std::string text;
        const int lines = (int)below(12);
        for (int l = 0; l < lines; ++l) {
            text += tokens[below(12)];
            text += seps[below(5)];
            text += tokens[below(12)];
            if (below(4) == 0) text += " trailing";
            if (below(5) == 0) text += "\r";
            if (l + 1 < lines || below(2)) text += "\n";
        }
        if (below(10) == 0) text.append(1, '\0');  // an embedded NUL
        const std::string p = dir + "/junk.txt";
        write_file(p, text);
        uint64_t *a = nullptr, *b = nullptr;
        const long ne = bisbm_io_read_edge_list(p.c_str(), &a, &b);
        CHECK(ne >= 0);
        for (long i = 0; i < ne; ++i) (void)(a[i] + b[i]);  // every element is readable
        bisbm_io_free(a);
        bisbm_io_free(b);
        uint32_t* lab = nullptr;
        const long nl = bisbm_io_read_memberships(p.c_str(), &lab);
        CHECK(nl >= 0);
        for (long i = 0; i < nl; ++i) (void)lab[i];
        bisbm_io_free(lab);
    }
    {
        uint64_t *a = nullptr, *b = nullptr;
        CHECK(bisbm_io_read_edge_list((dir + "/does-not-exist").c_str(), &a, &b) == -1);
        uint32_t* lab = nullptr;
        CHECK(bisbm_io_read_memberships((dir + "/does-not-exist").c_str(), &lab) == -1);
    }

    // ---- CSR cache: intact, then damaged in every header field and truncated ----
    for (int it = 0; it < 60; ++it) {
        const uint64_t na = 1 + below(30), nb = 1 + below(30), n = na + nb;
        const size_t ne = (size_t)below(150);
        std::string text;
        for (size_t e = 0; e < ne; ++e) text += std::to_string(below(na)) + "\t" + std::to_string(na + below(nb)) + "\n";
        const std::string p = dir + "/g.el", cache = p + ".bisbm_csr";
        std::remove(cache.c_str());
        write_file(p, text);
        uint64_t *rp0 = nullptr, *rp = nullptr, nedges = 0;
        uint32_t *cl0 = nullptr, *cl = nullptr;
        int hit = -1;
        CHECK(bisbm_io_load_csr(p.c_str(), n, 1, &rp0, &cl0, &nedges, &hit) == 0 && hit == 0 && nedges == ne);
        CHECK(bisbm_io_load_csr(p.c_str(), n, 1, &rp, &cl, &nedges, &hit) == 0 && hit == 1 && nedges == ne);
        CHECK(std::memcmp(rp, rp0, (n + 1) * 8) == 0 && (ne == 0 || std::memcmp(cl, cl0, 2 * ne * 4) == 0));
        bisbm_io_free(rp);
        bisbm_io_free(cl);
        const std::string good = read_file(cache);
        CHECK(good.size() == 48 + (n + 1) * 8 + 2 * ne * 4);
        for (int dmg = 0; dmg < 12 && !good.empty(); ++dmg) {
            std::string bad = good;
            if (dmg < 6)
                bad[(size_t)dmg * 8 + below(8)] ^= (char)(1 + below(255));  // one byte of one header word
            else if (dmg == 6)
                bad.resize(below(bad.size()));  // truncated
            else if (dmg == 7)
                bad.append(13, 'z');  // too long
            else if (bad.size() > 48)
                bad[48 + below(bad.size() - 48)] ^= (char)(1 + below(255));  // one byte of the body: see below
            write_file(cache, bad);
            rp = nullptr;
            cl = nullptr;
            const int rc = bisbm_io_load_csr(p.c_str(), n, 1, &rp, &cl, &nedges, &hit);
            CHECK(rc == 0 && nedges == ne);  // a damaged cache is ignored and rewritten, never trusted
            if (rc == 0 && dmg >= 8) {
                // a flipped body byte under an intact header cannot be told from data (no checksum: the cache is the
                // caller's own file), but what comes back is always a CSR the walks can index: ids < n, offsets monotone
                bool sound = rp[0] == 0 && rp[n] == 2 * ne;
                for (uint64_t v = 0; v < n && sound; ++v) sound = rp[v + 1] >= rp[v];
                for (uint64_t e = 0; e < 2 * ne && sound; ++e) sound = cl[e] < n;
                CHECK(sound);
                std::vector<uint32_t> tmp_id(n);
                CHECK(bisbm_io_locality_order(n, na, rp, cl, tmp_id.data()) == 0);
                bisbm_io_free(rp);
                bisbm_io_free(cl);
                std::remove(cache.c_str());
            } else if (rc == 0) {
                CHECK(std::memcmp(rp, rp0, (n + 1) * 8) == 0 && (ne == 0 || std::memcmp(cl, cl0, 2 * ne * 4) == 0));
                bisbm_io_free(rp);
                bisbm_io_free(cl);
            }
        }
        // an id outside [0, n): refused, nothing returned
        write_file(p, text + std::to_string(n + below(5)) + " 0\n");
        rp = nullptr;
        cl = nullptr;
        CHECK(bisbm_io_load_csr(p.c_str(), n, (int)below(2), &rp, &cl, &nedges, &hit) == -2);

        // ---- renumbering and the permuted CSR ----
        std::vector<uint32_t> new_id(n);
        CHECK(bisbm_io_locality_order(n, na, rp0, cl0, new_id.data()) == 0);
        std::vector<uint32_t> seen(n, 0);
        for (uint64_t v = 0; v < n; ++v) {
            CHECK(new_id[v] < n && (new_id[v] < na) == (v < na));
            if (new_id[v] < n) ++seen[new_id[v]];
        }
        CHECK(std::all_of(seen.begin(), seen.end(), [](uint32_t c) { return c == 1; }));
        std::vector<uint64_t> rp2(n + 1);
        std::vector<uint32_t> cl2(2 * ne + 1);
        CHECK(bisbm_io_permute_csr(n, rp0, cl0, new_id.data(), rp2.data(), cl2.data()) == 0);
        for (uint64_t v = 0; v < n; ++v) {
            const uint64_t w = new_id[v];
            CHECK(rp2[w + 1] - rp2[w] == rp0[v + 1] - rp0[v]);
            for (uint64_t j = 0; j < rp0[v + 1] - rp0[v]; ++j) CHECK(cl2[rp2[w] + j] == new_id[cl0[rp0[v] + j]]);
        }
        CHECK(bisbm_io_locality_order(n, n + 1, rp0, cl0, new_id.data()) == -1);  // na > n
        if (n >= 2) {  // not a bijection: refused, nothing written past the arrays
            std::vector<uint32_t> dup(new_id);
            dup[below(n)] = dup[(below(n - 1) + 1) % n];
            std::vector<uint32_t> sorted_dup(dup);
            std::sort(sorted_dup.begin(), sorted_dup.end());
            if (std::adjacent_find(sorted_dup.begin(), sorted_dup.end()) != sorted_dup.end())
                CHECK(bisbm_io_permute_csr(n, rp0, cl0, dup.data(), rp2.data(), cl2.data()) == -1);
        }
        bisbm_io_free(rp0);
        bisbm_io_free(cl0);
    }

    // ---- output_vec sizing ----
    for (int it = 0; it < 50; ++it) {
        std::vector<uint32_t> lab(below(40));
        for (auto& x : lab) x = (uint32_t)rng();
        const size_t need = bisbm_io_format_labels(lab.data(), lab.size(), nullptr, 0);
        std::vector<char> buf(need + 1, '#');
        CHECK(bisbm_io_format_labels(lab.data(), lab.size(), buf.data(), buf.size()) == need);
        CHECK(need >= 1 && buf[need - 1] == '\n' && buf[need] == '\0');
    }
    if (fails) std::fprintf(stderr, "%d checks failed\n", fails);
    return fails ? 1 : 0;
}
