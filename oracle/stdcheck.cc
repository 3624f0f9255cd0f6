// stdcheck.cc -- the REAL libstdc++ <random>/<algorithm> behind a C ABI, so tests can check the
// hand restatement in bisbm_oracle.c draw for draw.  TEST INFRASTRUCTURE ONLY.
// These are the third-party algorithms the reference calls at metropolis_hasting.cc:57,80 and
// blockmodel.cc:617-628,673-674; they are pinned to libstdc++ 11 (g++ 11.4 in this image).
#include <algorithm>
#include <cstdint>
#include <random>
#include <vector>

extern "C" {

void std_mt_raw(uint64_t seed, size_t n, uint32_t* out) {
    std::mt19937 g(seed);
    for (size_t i = 0; i < n; ++i) out[i] = (uint32_t)g();
}

void std_canonical(uint64_t seed, size_t n, double* out) {
    std::mt19937 g(seed);
    std::uniform_real_distribution<> d(0, 1);
    for (size_t i = 0; i < n; ++i) out[i] = d(g);
}

// shuffle iota(n) `reps` times in place (as anneal does with vlist); then one canonical draw so the
// engine position is checked too
double std_shuffle(uint64_t seed, size_t n, size_t reps, uint32_t* out) {
    std::mt19937 g(seed);
    std::vector<unsigned int> v(n);
    for (size_t i = 0; i < n; ++i) v[i] = (unsigned)i;
    for (size_t r = 0; r < reps; ++r) std::shuffle(v.begin(), v.end(), g);
    for (size_t i = 0; i < n; ++i) out[i] = v[i];
    std::uniform_real_distribution<> d(0, 1);
    return d(g);
}

// shuffle_bisbm shape: pointer-range shuffle of two segments (blockmodel.cc:673-674)
void std_shuffle_two(uint64_t seed, uint32_t* v, size_t na, size_t nb) {
    std::mt19937 g(seed);
    std::shuffle(&v[0], &v[na], g);
    std::shuffle(&v[na], &v[na + nb], g);
}

// agg_split's splitter_ (blockmodel.cc:532-543): vector<bool> of floor(n/2) false then true, shuffled `reps` times
void std_shuffle_bool(uint64_t seed, size_t n, size_t reps, uint8_t* out) {
    std::mt19937 g(seed);
    std::vector<bool> v(n, false);
    for (size_t i = n / 2; i < n; ++i) v[i] = true;
    for (size_t r = 0; r < reps; ++r) std::shuffle(v.begin(), v.end(), g);
    for (size_t i = 0; i < n; ++i) out[i] = v[i];
}

void std_discrete(uint64_t seed, const int* w, size_t n, size_t draws, uint64_t* out) {
    std::mt19937 g(seed);
    std::vector<int> wv(w, w + n);
    for (size_t i = 0; i < draws; ++i) {
        std::discrete_distribution<size_t> d(wv.begin(), wv.end());
        out[i] = d(g);
    }
}

void std_uniform_int(uint64_t seed, uint32_t range, size_t n, uint32_t* out) {
    std::mt19937 g(seed);
    for (size_t i = 0; i < n; ++i) {
        std::uniform_int_distribution<unsigned long> d(0, range - 1);
        out[i] = (uint32_t)d(g);
    }
}
}
