/*
 * bisbm_oracle.c -- CPU restatement of the reference's node-label MH sweep.
 * TEST INFRASTRUCTURE ONLY (see bisbm_oracle.h for who may use it and how it is pinned).
 *
 * Two RNG modes:
 *   ORC_RNG_COMPAT  the reference's exact draw sequence: std::mt19937 `engine`, the hidden
 *                   second engine `gen` (blockmodel.hh:17-18), libstdc++-11 shuffle /
 *                   generate_canonical / discrete_distribution restated by hand,
 *                   serial FP64 summation in source order.  This is what is compared with the
 *                   reference outputs recorded in SURVEY.md.
 *   ORC_RNG_PHILOX  the production definition the HIP kernels implement: the same Markov chain
 *                   (same proposal distribution, same dS, same accept rule, same bookkeeping)
 *                   driven by Philox4x32-10 counters, a Feistel visit order per sweep and type
 *                   (all type-a nodes, then all type-b nodes), an integer inverse-CDF draw, and a
 *                   fixed 64-leaf butterfly (levels 32,1,2,4,8,16) for the three FP64 sums (dS, accu0, accu1).
 *                   The GPU must match this mode bit-for-bit on integers.
 *
 * Paths cited below are relative to /root/reference/src.
 */
#include "bisbm_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * numerics
 * ---------------------------------------------------------------------------------------- */

static double *g_lgamma = NULL; /* support/cache.cc:21-23 (__lgamma_cache) */
static size_t g_lgamma_size = 0;
static double *g_safelog = NULL; /* __safelog_cache */
static size_t g_safelog_size = 0;
static double *g_q = NULL; /* support/int_part.cc:28 (__q_cache), columns truncated at g_q_kcap */
static size_t g_q_kcap = 0;
#define ORC_Q_NMAX 10000 /* blockmodel.cc:48 init_q_cache(10000) */

/* support/int_part.cc:30-32 */
static double log_sum(double a, double b) {
    double mx = a > b ? a : b;
    return mx + log1p(exp(-fabs(a - b)));
}

/* support/int_part.cc:34-51.  The reference fills a (nmax+1)^2 table; only columns k <= n_r+1 can
 * ever be read, and column k depends on columns <= k only, so the columns are cut at kcap.  Cells
 * the reference never writes (k > n) stay -inf and are read as such by the recurrence (F7). */
static void build_q_cache(size_t kcap) {
    size_t stride = kcap + 1;
    free(g_q);
    g_q = (double *)malloc(sizeof(double) * (ORC_Q_NMAX + 1) * stride);
    for (size_t i = 0; i < (ORC_Q_NMAX + 1) * stride; ++i) g_q[i] = -INFINITY;
    for (size_t n = 1; n <= ORC_Q_NMAX; ++n) {
        double *row = g_q + n * stride;
        row[1] = 0;
        size_t kmax = n < kcap ? n : kcap;
        for (size_t k = 2; k <= kmax; ++k) {
            row[k] = log_sum(row[k], row[k - 1]);
            if (n > k) row[k] = log_sum(row[k], g_q[(n - k) * stride + k]);
        }
    }
    g_q_kcap = kcap;
}

/* support/cache.cc:64-79 (init_lgamma) and :25-37 (init_safelog) */
void orc_init_tables(size_t lgamma_size, size_t q_kcap) {
    if (lgamma_size > g_lgamma_size) {
        g_lgamma = (double *)realloc(g_lgamma, sizeof(double) * lgamma_size);
        if (g_lgamma_size == 0) g_lgamma[0] = INFINITY;
        for (size_t i = g_lgamma_size > 1 ? g_lgamma_size : 1; i < lgamma_size; ++i)
            g_lgamma[i] = lgamma((double)i);
        g_lgamma_size = lgamma_size;
    }
    if (lgamma_size > g_safelog_size) {
        g_safelog = (double *)realloc(g_safelog, sizeof(double) * lgamma_size);
        for (size_t i = g_safelog_size; i < lgamma_size; ++i)
            g_safelog[i] = i == 0 ? 0.0 : log((double)i);
        g_safelog_size = lgamma_size;
    }
    if (q_kcap > ORC_Q_NMAX) q_kcap = ORC_Q_NMAX;
    if (q_kcap < 2) q_kcap = 2;
    if (q_kcap > g_q_kcap) build_q_cache(q_kcap);
}

/* support/cache.hh:82-93.  Beyond the table the reference grows it with lgamma(i); the value is
 * the same, so it is computed in place here. */
double orc_lgamma_fast(size_t x) {
    if (x < g_lgamma_size) return g_lgamma[x];
    if (x == 0) return INFINITY;
    return lgamma((double)x);
}

/* support/cache.hh:46-57 */
double orc_safelog_fast(size_t x) {
    if (x < g_safelog_size) return g_safelog[x];
    if (x == 0) return 0.0;
    return log((double)x);
}

const double *orc_lgamma_table(size_t *size) {
    *size = g_lgamma_size;
    return g_lgamma;
}

const double *orc_q_table(size_t *rows, size_t *stride) {
    *rows = ORC_Q_NMAX + 1;
    *stride = g_q_kcap + 1;
    return g_q;
}

/* support/util.hh:41-47 */
double orc_lbinom_fast(size_t N, size_t k) {
    if (N == 0 || k == 0 || k > N) return 0;
    return (orc_lgamma_fast(N + 1) - orc_lgamma_fast(k + 1)) - orc_lgamma_fast(N - k + 1);
}

/* support/spence.cc:24-47,91-154 (Cephes dilogarithm) */
static const double SP_A[8] = {
    4.65128586073990045278E-5, 7.31589045238094711071E-3, 1.33847639578309018650E-1,
    8.79691311754530315341E-1, 2.71149851196553469920E0,  4.25697156008121755724E0,
    3.29771340985225106936E0,  1.00000000000000000126E0,
};
static const double SP_B[8] = {
    6.90990488912553276999E-4, 2.54043763932544379113E-2, 2.82974860602568089943E-1,
    1.41172597751831069617E0,  3.63800533345137075418E0,  5.03278880143316990390E0,
    3.54771340985225096217E0,  9.99999999999999998740E-1,
};

static double polevl(double x, const double *coef, int N) {
    double ans = coef[0];
    for (int i = 1; i <= N; ++i) ans = ans * x + coef[i];
    return ans;
}

double orc_spence(double x) {
    double w, y, z;
    int flag = 0;
    if (x < 0.0) return NAN;
    if (x == 1.0) return 0.0;
    if (x == 0.0) return M_PI * M_PI / 6.0;
    if (x > 2.0) {
        x = 1.0 / x;
        flag |= 2;
    }
    if (x > 1.5) {
        w = (1.0 / x) - 1.0;
        flag |= 2;
    } else if (x < 0.5) {
        w = -x;
        flag |= 1;
    } else
        w = x - 1.0;
    y = -w * polevl(w, SP_A, 7) / polevl(w, SP_B, 7);
    if (flag & 1) y = (M_PI * M_PI) / 6.0 - log(x) * log1p(-x) - y;
    if (flag & 2) {
        z = log(x);
        y = -0.5 * z * z - y;
    }
    return y;
}

/* support/int_part.cc:77-87 */
static double get_v(double u) {
    const double epsilon = 1e-8;
    double v = u;
    double delta = 1;
    while (delta > epsilon) {
        double n_v = u * sqrt(orc_spence(exp(-v)));
        delta = fabs(n_v - v);
        v = n_v;
    }
    return v;
}

/* support/int_part.cc:73-75,89-98 */
double orc_log_q_approx(size_t n, size_t k) {
    if ((double)k < pow((double)n, 1 / 4.)) /* log_q_approx_small */
        return orc_lbinom_fast(n - 1, k - 1) - orc_lgamma_fast(k + 1);
    double u = (double)k / sqrt((double)n);
    double v = get_v(u);
    double lf = log(v) - log1p(-exp(-v) * (1 + u * u / 2)) / 2 - log(2) * 3 / 2. - log(u) - log(M_PI);
    double g = 2 * v / u - u * log1p(-exp(-v));
    return lf - log((double)n) + sqrt((double)n) * g;
}

/* Philox mode (production definition): the same formulas with get_v iterated to convergence
 * (|dv| <= 1e-14 instead of the reference's 1e-8) for k^2 >= 6.25 n, i.e. u >= 2.5 -- the range the HIP kernels
 * evaluate by Newton steps on the series of spence (bisbm_device.hpp: log_q_closed / log_q_mid / log_q_low). */
double orc_log_q_approx_philox(size_t n, size_t k) {
    if ((double)k < pow((double)n, 1 / 4.)) return orc_log_q_approx(n, k);
    if (4.0 * ((double)k * (double)k) < 25.0 * (double)n) return orc_log_q_approx(n, k);
    double u = (double)k / sqrt((double)n);
    double v = u, delta = 1;
    for (int guard = 0; delta > 1e-14 && guard < 1000; ++guard) {
        double n_v = u * sqrt(orc_spence(exp(-v)));
        delta = fabs(n_v - v);
        v = n_v;
    }
    double lf = log(v) - log1p(-exp(-v) * (1 + u * u / 2)) / 2 - log(2) * 3 / 2. - log(u) - log(M_PI);
    double g = 2 * v / u - u * log1p(-exp(-v));
    return lf - log((double)n) + sqrt((double)n) * g;
}

double orc_q_cache_at(size_t n, size_t k) {
    if (n > ORC_Q_NMAX || k > g_q_kcap) return NAN;
    return g_q[n * (g_q_kcap + 1) + k];
}

/* support/int_part.hh:27-37 */
double orc_log_q(int n, int k) {
    if (n <= 0 || k < 1) return 0;
    if (k > n) k = n;
    if (n < ORC_Q_NMAX + 1) {
        if ((size_t)k > g_q_kcap) orc_init_tables(0, (size_t)k);
        return g_q[(size_t)n * (g_q_kcap + 1) + (size_t)k];
    }
    return orc_log_q_approx((size_t)n, (size_t)k);
}

double orc_log_q_philox(int n, int k) {
    if (n <= 0 || k < 1) return 0;
    if (k > n) k = n;
    if (n < ORC_Q_NMAX + 1) return orc_log_q(n, k);
    return orc_log_q_approx_philox((size_t)n, (size_t)k);
}

/* ------------------------------------------------------------------------------------------
 * cooling schedules -- metropolis_hasting.cc:10-37.  kwargs are float (types.hh:16); the mixed
 * float/double arithmetic below is what the C++ expressions promote to.
 * ---------------------------------------------------------------------------------------- */
double orc_schedule(int schedule, uint64_t t, float kw0, float kw1) {
    switch (schedule) {
    case ORC_SCHED_EXPONENTIAL: /* :10-13  float * std::pow(float, size_t) -> double pow */
        return (double)kw0 * pow((double)kw1, (double)t);
    case ORC_SCHED_LINEAR: /* :15-18  float - float * size_t : size_t -> float, all FP32 */
        return (double)(kw0 - kw1 * (float)t);
    case ORC_SCHED_LOGARITHMIC: { /* :20-23  t + kw1 is float, truncated to the table index */
        float x = (float)t + kw1;
        return (double)kw0 / orc_safelog_fast((size_t)x);
    }
    case ORC_SCHED_CONSTANT: /* :25-28 */
        return (double)kw0;
    case ORC_SCHED_ABRUPT_COOL: /* :30-37  size_t < float compares as float */
        return ((float)t < kw0) ? 1. : 0.;
    }
    return NAN;
}

/* ------------------------------------------------------------------------------------------
 * libstdc++-11 compatible RNG (/usr/include/c++/11/bits/random.tcc,
 * uniform_int_dist.h, stl_algo.h).  Checked against the real library by oracle/stdcheck.cc.
 * ---------------------------------------------------------------------------------------- */
void orc_mt_seed(orc_mt19937 *g, uint64_t seed) {
    g->mt[0] = (uint32_t)seed; /* seed mod 2^32 */
    for (int i = 1; i < 624; ++i)
        g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}

uint32_t orc_mt_next(orc_mt19937 *g) {
    if (g->idx >= 624) {
        uint32_t *mt = g->mt;
        for (int k = 0; k < 624; ++k) {
            uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
            mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

/* uniform_real_distribution<double>(0,1) -> generate_canonical<double,53>: two 32-bit draws */
double orc_mt_canonical(orc_mt19937 *g) {
    double x0 = (double)orc_mt_next(g);
    double x1 = (double)orc_mt_next(g);
    double r = (x0 + x1 * 4294967296.0) / 18446744073709551616.0;
    if (r >= 1.0) r = nextafter(1.0, 0.0);
    return r;
}

/* uniform_int_distribution downscale for a 32-bit URBG: Lemire, uniform_int_dist.h:245-272 */
uint32_t orc_mt_lemire(orc_mt19937 *g, uint32_t range) {
    uint64_t product = (uint64_t)orc_mt_next(g) * (uint64_t)range;
    uint32_t low = (uint32_t)product;
    if (low < range) {
        uint32_t threshold = (uint32_t)(-range) % range;
        while (low < threshold) {
            product = (uint64_t)orc_mt_next(g) * (uint64_t)range;
            low = (uint32_t)product;
        }
    }
    return (uint32_t)(product >> 32);
}

/* std::shuffle, stl_algo.h:3729-3793 */
/* std::shuffle of a vector<bool> (blockmodel.cc:541-543): the same swaps as for any random-access range */
void orc_mt_shuffle_u8(orc_mt19937 *g, uint8_t *v, size_t n) {
    if (n == 0) return;
    uint8_t tmp;
#define ORC_SWAP8(i, j) (tmp = v[i], v[i] = v[j], v[j] = tmp)
    if (0xFFFFFFFFull / n >= n) {
        size_t i = 1;
        if ((n % 2) == 0) {
            size_t j = orc_mt_lemire(g, 2);
            ORC_SWAP8(i, j);
            ++i;
        }
        while (i < n) {
            uint64_t s = i + 1;
            uint32_t x = orc_mt_lemire(g, (uint32_t)(s * (s + 1)));
            size_t p0 = x / (s + 1), p1 = x % (s + 1);
            ORC_SWAP8(i, p0);
            ++i;
            ORC_SWAP8(i, p1);
            ++i;
        }
        return;
    }
    for (size_t i = 1; i < n; ++i) {
        size_t j = orc_mt_lemire(g, (uint32_t)(i + 1));
        ORC_SWAP8(i, j);
    }
#undef ORC_SWAP8
}

void orc_mt_shuffle_u32(orc_mt19937 *g, uint32_t *v, size_t n) {
    if (n == 0) return;
    uint32_t tmp;
#define ORC_SWAP(i, j) (tmp = v[i], v[i] = v[j], v[j] = tmp)
    if (0xFFFFFFFFull / n >= n) {
        size_t i = 1;
        if ((n % 2) == 0) {
            size_t j = orc_mt_lemire(g, 2);
            ORC_SWAP(i, j);
            ++i;
        }
        while (i < n) {
            uint64_t s = i + 1; /* __swap_range */
            uint32_t x = orc_mt_lemire(g, (uint32_t)(s * (s + 1)));
            size_t p0 = x / (s + 1), p1 = x % (s + 1);
            ORC_SWAP(i, p0);
            ++i;
            ORC_SWAP(i, p1);
            ++i;
        }
        return;
    }
    for (size_t i = 1; i < n; ++i) {
        size_t j = orc_mt_lemire(g, (uint32_t)(i + 1));
        ORC_SWAP(i, j);
    }
#undef ORC_SWAP
}

/* std::discrete_distribution<size_t>(w.begin(), w.end())(g): random.tcc:2656-2678,2697-2714 */
size_t orc_mt_discrete(orc_mt19937 *g, const int *w, size_t n) {
    if (n < 2) return 0; /* no draw */
    double sum = 0.0;
    for (size_t i = 0; i < n; ++i) sum += (double)w[i];
    double *cp = (double *)malloc(sizeof(double) * n);
    double acc = 0.0;
    for (size_t i = 0; i < n; ++i) {
        double p = (double)w[i] / sum;
        acc = (i == 0) ? p : acc + p;
        cp[i] = acc;
    }
    cp[n - 1] = 1.0;
    double u = orc_mt_canonical(g);
    size_t lo = 0, len = n; /* std::lower_bound: first cp[i] >= u */
    while (len > 0) {
        size_t half = len >> 1;
        if (cp[lo + half] < u) {
            lo = lo + half + 1;
            len = len - half - 1;
        } else
            len = half;
    }
    free(cp);
    return lo;
}

/* ------------------------------------------------------------------------------------------
 * Philox4x32-10 and the production-mode draw definitions
 * ---------------------------------------------------------------------------------------- */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0;
    out[1] = c1;
    out[2] = c2;
    out[3] = c3;
}

enum { PHX_STEP_A = 0, PHX_STEP_B = 1, PHX_SWEEP_KEY = 2, PHX_INIT_SHUFFLE = 3, PHX_MERGE_A = 4, PHX_MERGE_B = 5, PHX_SPLIT = 6 };

static void phx_draw(uint64_t seed, uint32_t chain, uint32_t purpose, uint64_t idx, uint32_t out[4]) {
    uint32_t ctr[4] = {(uint32_t)idx, (uint32_t)(idx >> 32), chain, purpose};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    orc_philox4x32_10(ctr, key, out);
}

static double u53(uint32_t hi, uint32_t lo) {
    uint64_t b = ((uint64_t)hi << 32) | lo;
    return (double)(b >> 11) * 0x1.0p-53;
}

static uint32_t mix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}

/* 4-round alternating Feistel network on b = max(2, bitlen(n-1)) bits with cycle walking:
 * a bijection of [0, n). */
static uint32_t feistel_perm(const uint32_t keys[4], uint32_t n, uint32_t i) {
    if (n <= 1) return 0;
    uint32_t b = 0;
    while (((uint64_t)1 << b) < (uint64_t)n) ++b;
    if (b < 2) b = 2;
    uint32_t wa = b / 2, wb = b - wa;
    uint32_t x = i;
    do {
        uint32_t A = x >> wb, B = x & ((1u << wb) - 1u);
        uint32_t cwa = wa, cwb = wb;
        for (int r = 0; r < 4; ++r) {
            uint32_t F = mix32(B ^ keys[r]) & ((1u << cwa) - 1u);
            uint32_t nA = B, nB = A ^ F;
            A = nA;
            B = nB;
            uint32_t t = cwa;
            cwa = cwb;
            cwb = t;
        }
        x = (A << wb) | B; /* after 4 rounds the widths are (wa, wb) again */
    } while (x >= n);
    return x;
}

/* Visit order of a Philox-mode phase, local in node ids: tiles of 4096 consecutive ids in a keyed order, inside a
 * tile its 64 cells of 64 ids in a keyed order, inside a cell the ids in a keyed order (consecutive steps gather
 * labels from the same neighbourhoods and read adjacent CSR rows); cycle walking over the padded domain. */
static uint32_t tiled_perm(const uint32_t keys[4], uint32_t n, uint32_t i) {
    const uint32_t ntiles = (n + 4095u) >> 12;
    /* a class that fits one tile has only the cells it needs (ceil(n / 64) of them): the padded domain is then less than
     * 64 ids larger than the class instead of 4096 (the reference's own data sets have 14..500 nodes per type) */
    const uint32_t ncells = ntiles == 1 ? (n + 63u) >> 6 : 64u;
    uint32_t x = i;
    do {
        uint32_t t = feistel_perm(keys, ntiles, x >> 12);
        uint32_t k2[4] = {keys[1] ^ (t * 0x9E3779B9u), keys[2], keys[3], keys[0]};
        uint32_t c = feistel_perm(k2, ncells, (x >> 6) & 63u);
        uint32_t k3[4] = {keys[2] ^ (((t << 6) | c) * 0x85EBCA6Bu), keys[3], keys[0], keys[1]};
        x = (t << 12) | (c << 6) | feistel_perm(k3, 64u, x & 63u);
    } while (x >= n);
    return x;
}

/* Philox-mode visit order of one sweep: first every type-a node, then every type-b node (the two colour
 * classes of the bipartite graph: within a phase no visited node is a neighbour of another, so the
 * neighbour labels a step reads are frozen for the whole phase), each class in its own keyed
 * permutation.  Position i in [0, na+nb). */
uint32_t orc_philox_visit(uint64_t seed, uint32_t chain, uint64_t sweep, uint32_t na, uint32_t nb, uint32_t i) {
    uint32_t keys[4];
    if (i < na) {
        phx_draw(seed, chain, PHX_SWEEP_KEY, 2 * sweep, keys);
        return tiled_perm(keys, na, i);
    }
    phx_draw(seed, chain, PHX_SWEEP_KEY, 2 * sweep + 1, keys);
    return na + tiled_perm(keys, nb, i - na);
}

/* ------------------------------------------------------------------------------------------
 * text I/O -- graph_utilities.cc
 * ---------------------------------------------------------------------------------------- */

/* `std::stringstream(line) >> size_t`: skip blanks, parse digits.  Returns 1 on success,
 * 0 = extraction failure that writes 0 (C++11), -1 = sentry failure (nothing written). */
static int parse_size(const char **pp, uint64_t *out) {
    const char *p = *pp;
    while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\v' || *p == '\f' || *p == '\n') ++p;
    if (*p == '\0') {
        *pp = p;
        return -1;
    }
    const char *q = p;
    int neg = 0;
    if (*q == '+' || *q == '-') {
        neg = (*q == '-');
        ++q;
    }
    if (*q < '0' || *q > '9') {
        *pp = p;
        return 0;
    }
    uint64_t v = 0;
    while (*q >= '0' && *q <= '9') {
        v = v * 10 + (uint64_t)(*q - '0');
        ++q;
    }
    *out = neg ? (uint64_t)(-(int64_t)v) : v;
    *pp = q;
    return 1;
}

static char *read_line(FILE *f, char **buf, size_t *cap) {
    size_t len = 0;
    int c;
    int any = 0;
    while ((c = fgetc(f)) != EOF) {
        any = 1;
        if (c == '\n') break;
        if (len + 2 > *cap) {
            *cap = *cap ? *cap * 2 : 256;
            *buf = (char *)realloc(*buf, *cap);
        }
        (*buf)[len++] = (char)c;
    }
    if (!any) return NULL;
    if (len + 1 > *cap) {
        *cap = len + 16;
        *buf = (char *)realloc(*buf, *cap);
    }
    (*buf)[len] = '\0';
    return *buf;
}

/* graph_utilities.cc:20-34.  node_a/node_b live outside the loop, so a blank line re-pushes the
 * previous pair and a non-numeric line pushes (0, previous b). */
long orc_load_edge_list(const char *path, uint64_t **a, uint64_t **b) {
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    size_t n = 0, cap = 1024;
    *a = (uint64_t *)malloc(sizeof(uint64_t) * cap);
    *b = (uint64_t *)malloc(sizeof(uint64_t) * cap);
    uint64_t node_a = 0, node_b = 0;
    char *buf = NULL;
    size_t bcap = 0;
    while (read_line(f, &buf, &bcap)) {
        const char *p = buf;
        uint64_t v;
        int ok = parse_size(&p, &v);
        if (ok == 1)
            node_a = v;
        else if (ok == 0)
            node_a = 0;
        if (ok == 1) { /* a failed first extraction leaves failbit set: the second is skipped */
            int ok2 = parse_size(&p, &v);
            if (ok2 == 1)
                node_b = v;
            else if (ok2 == 0)
                node_b = 0;
        }
        if (n == cap) {
            cap *= 2;
            *a = (uint64_t *)realloc(*a, sizeof(uint64_t) * cap);
            *b = (uint64_t *)realloc(*b, sizeof(uint64_t) * cap);
        }
        (*a)[n] = node_a;
        (*b)[n] = node_b;
        ++n;
    }
    free(buf);
    fclose(f);
    return (long)n;
}

/* graph_utilities.cc:5-18 */
long orc_load_memberships(const char *path, uint32_t **labels) {
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    size_t n = 0, cap = 1024;
    *labels = (uint32_t *)malloc(sizeof(uint32_t) * cap);
    uint64_t membership = 0;
    char *buf = NULL;
    size_t bcap = 0;
    while (read_line(f, &buf, &bcap)) {
        const char *p = buf;
        uint64_t v;
        int ok = parse_size(&p, &v);
        if (ok == 1)
            membership = v;
        else if (ok == 0)
            membership = 0;
        if (n == cap) {
            cap *= 2;
            *labels = (uint32_t *)realloc(*labels, sizeof(uint32_t) * cap);
        }
        (*labels)[n++] = (uint32_t)membership;
    }
    free(buf);
    fclose(f);
    return (long)n;
}

/* graph_utilities.cc:36-49 */
int orc_edge_to_csr(const uint64_t *a, const uint64_t *b, size_t n_edges, size_t n,
                    uint64_t *rowptr, uint32_t *col) {
    memset(rowptr, 0, sizeof(uint64_t) * (n + 1));
    for (size_t e = 0; e < n_edges; ++e) {
        if (a[e] >= n || b[e] >= n) return -1;
        rowptr[a[e] + 1]++;
        rowptr[b[e] + 1]++;
    }
    for (size_t i = 0; i < n; ++i) rowptr[i + 1] += rowptr[i];
    uint64_t *fill = (uint64_t *)malloc(sizeof(uint64_t) * (n + 1));
    memcpy(fill, rowptr, sizeof(uint64_t) * (n + 1));
    for (size_t e = 0; e < n_edges; ++e) {
        col[fill[a[e]]++] = (uint32_t)b[e];
        col[fill[b[e]]++] = (uint32_t)a[e];
    }
    free(fill);
    return 0;
}

/* output_functions.hh:20-29 */
size_t orc_format_vec(const uint32_t *v, size_t n, char *out, size_t cap) {
    size_t w = 0;
    for (size_t i = 0; i < n; ++i) {
        int k = snprintf(out + w, w < cap ? cap - w : 0, "%u ", v[i]);
        w += (size_t)k;
    }
    if (w + 1 < cap) {
        out[w] = '\n';
        out[w + 1] = '\0';
    }
    return w + 1;
}

/* ------------------------------------------------------------------------------------------
 * model state -- blockmodel.{hh,cc}
 * ---------------------------------------------------------------------------------------- */
struct orc_model {
    size_t n, na, nb, ka, kb, K;
    size_t num_edges, max_degree;
    double epsilon;
    double entropy; /* blockmodel.hh:100 "not true entropy": running sum of accepted dS */
    uint64_t *rowptr;
    uint32_t *col;
    int *deg;
    uint32_t *labels; /* memberships_ */
    uint32_t *vlist;  /* blockmodel.cc:41, persisted and re-shuffled in place */
    int *m;           /* K*K, symmetric (blockmodel.cc:702-714) */
    int *m_r;
    int *n_r;
    uint32_t *eta; /* K*(max_degree+1) */
    int *kv;       /* k_[v] of the node being moved, recomputed from CSR + labels */
    /* MH object state (metropolis_hasting.hh:26-28) */
    double entropy_min;
    double accu_r;
    double phx_accu0, phx_accu1; /* Philox mode: the two Hastings sums of the last transition_ratio */
    /* rng */
    int rng_mode;
    orc_mt19937 engine, gen;
    uint64_t phx_seed;
    uint32_t phx_chain;
    uint64_t sweeps_total;  /* Philox counters: sweeps executed over the model's lifetime */
    uint32_t shuffle_epoch; /* Philox counters: number of shuffle_bisbm calls so far */
    uint32_t merge_epoch;   /* Philox counters: proposal rounds of agg_merge so far */
    uint32_t split_epoch;   /* Philox counters: agg_split calls so far */
    size_t cap_K;           /* blocks the state arrays are allocated for */
    double last_split_dS;   /* dS of the cut the last agg_split applied */
    uint64_t last_accepted, last_sweeps;
};

size_t orc_n(const orc_model *m) { return m->n; }
size_t orc_k(const orc_model *m) { return m->K; }
size_t orc_num_edges(const orc_model *m) { return m->num_edges; }
size_t orc_max_degree(const orc_model *m) { return m->max_degree; }
uint64_t orc_last_accepted(const orc_model *m) { return m->last_accepted; }
uint64_t orc_last_sweeps(const orc_model *m) { return m->last_sweeps; }
uint64_t orc_total_sweeps(const orc_model *m) { return m->sweeps_total; }
double orc_get_entropy(const orc_model *m) { return m->entropy; }

void orc_get_memberships(const orc_model *m, uint32_t *out) { memcpy(out, m->labels, sizeof(uint32_t) * m->n); }
void orc_get_m(const orc_model *m, int32_t *out) { memcpy(out, m->m, sizeof(int) * m->K * m->K); }
void orc_get_m_r(const orc_model *m, int32_t *out) { memcpy(out, m->m_r, sizeof(int) * m->K); }
void orc_get_n_r(const orc_model *m, int32_t *out) { memcpy(out, m->n_r, sizeof(int) * m->K); }
void orc_get_eta(const orc_model *m, uint32_t *out) {
    memcpy(out, m->eta, sizeof(uint32_t) * m->K * (m->max_degree + 1));
}
void orc_get_vlist(const orc_model *m, uint32_t *out) { memcpy(out, m->vlist, sizeof(uint32_t) * m->n); }

/* blockmodel.cc:15-75 */
orc_model *orc_create(size_t n, size_t na, size_t nb, const uint64_t *rowptr, const uint32_t *col,
                      size_t ka, size_t kb, double epsilon, const uint32_t *labels) {
    if (na + nb != n) return NULL;
    orc_model *m = (orc_model *)calloc(1, sizeof(orc_model));
    m->n = n;
    m->na = na;
    m->nb = nb;
    m->ka = ka;
    m->kb = kb;
    m->K = ka + kb;
    m->epsilon = epsilon;
    m->rowptr = (uint64_t *)malloc(sizeof(uint64_t) * (n + 1));
    memcpy(m->rowptr, rowptr, sizeof(uint64_t) * (n + 1));
    size_t nnz = rowptr[n];
    m->col = (uint32_t *)malloc(sizeof(uint32_t) * (nnz ? nnz : 1));
    memcpy(m->col, col, sizeof(uint32_t) * nnz);
    m->deg = (int *)malloc(sizeof(int) * n);
    m->labels = (uint32_t *)malloc(sizeof(uint32_t) * n);
    m->vlist = (uint32_t *)malloc(sizeof(uint32_t) * n);
    memcpy(m->labels, labels, sizeof(uint32_t) * n);
    size_t maxdeg = 0;
    for (size_t j = 0; j < n; ++j) {
        m->deg[j] = (int)(rowptr[j + 1] - rowptr[j]);
        if ((size_t)m->deg[j] > maxdeg) maxdeg = (size_t)m->deg[j];
        m->vlist[j] = (uint32_t)j;
    }
    m->num_edges = nnz / 2;
    m->max_degree = maxdeg;
    size_t K = m->K;
    m->m = (int *)calloc(K * K, sizeof(int));
    m->m_r = (int *)calloc(K, sizeof(int));
    m->n_r = (int *)calloc(K, sizeof(int));
    m->eta = (uint32_t *)calloc(K * (maxdeg + 1), sizeof(uint32_t));
    m->kv = (int *)calloc(K, sizeof(int));
    m->cap_K = K;
    m->entropy_min = INFINITY;
    m->accu_r = 0.;
    m->rng_mode = ORC_RNG_COMPAT;
    orc_mt_seed(&m->engine, 5489u);
    orc_mt_seed(&m->gen, 5489u);
    /* blockmodel.cc:47-48: init_cache(E) -> tables of 2E+1; init_q_cache(10000) */
    size_t lg = 2 * m->num_edges + 2;
    if (lg < n + 3) lg = n + 3;
    size_t kc = (na > nb ? na : nb) + 1;
    orc_init_tables(lg, kc);
    return m;
}

void orc_destroy(orc_model *m) {
    if (!m) return;
    free(m->rowptr);
    free(m->col);
    free(m->deg);
    free(m->labels);
    free(m->vlist);
    free(m->m);
    free(m->m_r);
    free(m->n_r);
    free(m->eta);
    free(m->kv);
    free(m);
}

void orc_seed_compat(orc_model *m, uint64_t engine_seed, uint64_t gen_seed) {
    m->rng_mode = ORC_RNG_COMPAT;
    orc_mt_seed(&m->engine, engine_seed);
    orc_mt_seed(&m->gen, gen_seed);
}

void orc_seed_philox(orc_model *m, uint64_t seed, uint32_t chain_id) {
    m->rng_mode = ORC_RNG_PHILOX;
    m->phx_seed = seed;
    m->phx_chain = chain_id;
    m->sweeps_total = 0;
    m->shuffle_epoch = 0;
}

void orc_set_memberships(orc_model *m, const uint32_t *labels) {
    memcpy(m->labels, labels, sizeof(uint32_t) * m->n);
}

/* blockmodel.cc:682-688 with compute_n_r :740-746, compute_m :702-714, compute_m_r :716-727,
 * compute_eta_rk :729-738.  (compute_k :691-700 is replaced by per-step recomputation.) */
void orc_init_bisbm(orc_model *m) {
    size_t K = m->K, D = m->max_degree + 1;
    memset(m->n_r, 0, sizeof(int) * K);
    memset(m->m, 0, sizeof(int) * K * K);
    memset(m->m_r, 0, sizeof(int) * K);
    memset(m->eta, 0, sizeof(uint32_t) * K * D);
    for (size_t v = 0; v < m->n; ++v) m->n_r[m->labels[v]]++;
    for (size_t v = 0; v < m->n; ++v) {
        uint32_t r = m->labels[v];
        for (uint64_t e = m->rowptr[v]; e < m->rowptr[v + 1]; ++e) m->m[r * K + m->labels[m->col[e]]]++;
    }
    for (size_t r = 0; r < K; ++r) {
        size_t s = 0;
        for (size_t c = 0; c < K; ++c) s += (size_t)m->m[r * K + c];
        m->m_r[r] = (int)s;
    }
    for (size_t v = 0; v < m->n; ++v) m->eta[m->labels[v] * D + (size_t)m->deg[v]]++;
}

/* blockmodel.cc:672-680.  Compat: two std::shuffle calls on `engine`.  Philox: each type's label
 * array is gathered through a keyed Feistel permutation (block sizes preserved, fully parallel). */
void orc_shuffle_bisbm(orc_model *m) {
    if (m->rng_mode == ORC_RNG_COMPAT) {
        orc_mt_shuffle_u32(&m->engine, m->labels, m->na);
        orc_mt_shuffle_u32(&m->engine, m->labels + m->na, m->nb);
    } else {
        uint32_t *old = (uint32_t *)malloc(sizeof(uint32_t) * m->n);
        memcpy(old, m->labels, sizeof(uint32_t) * m->n);
        for (uint32_t ty = 0; ty < 2; ++ty) {
            uint32_t keys[4];
            uint64_t idx = ((uint64_t)ty << 32) | m->shuffle_epoch;
            phx_draw(m->phx_seed, m->phx_chain, PHX_INIT_SHUFFLE, idx, keys);
            size_t base = ty ? m->na : 0, cnt = ty ? m->nb : m->na;
            for (size_t i = 0; i < cnt; ++i)
                m->labels[base + i] = old[base + feistel_perm(keys, (uint32_t)cnt, (uint32_t)i)];
        }
        m->shuffle_epoch++;
        free(old);
    }
    orc_init_bisbm(m);
}

/* k_[v] (blockmodel.cc:691-700), recomputed from adjacency and labels */
static void compute_kv(orc_model *m, size_t v) {
    memset(m->kv, 0, sizeof(int) * m->K);
    for (uint64_t e = m->rowptr[v]; e < m->rowptr[v + 1]; ++e) m->kv[m->labels[m->col[e]]]++;
}

/* blockmodel.cc:753-787 */
double orc_entropy(orc_model *m) {
    size_t K = m->K, D = m->max_degree + 1;
    double ent = 0;
    for (size_t v = 0; v < m->n; ++v) ent -= orc_lgamma_fast((size_t)m->deg[v] + 1);
    for (size_t r = 0; r < K; ++r) {
        for (size_t s = r + 1; s < K; ++s) ent -= orc_lgamma_fast((size_t)m->m[r * K + s] + 1);
        for (size_t d = 0; d < D; ++d) ent -= orc_lgamma_fast((size_t)m->eta[r * D + d] + 1);
        ent += orc_lgamma_fast((size_t)m->m_r[r] + 1);
        ent += orc_log_q(m->m_r[r], m->n_r[r]);
    }
    /* adj_map_ multiplicities (blockmodel.cc:62-74,772-779): for node y, neighbours in ascending
     * id order (std::map), multiplicity > 1 and y > neighbour. */
    uint32_t *tmp = (uint32_t *)malloc(sizeof(uint32_t) * (m->max_degree + 1));
    for (size_t y = 0; y < m->n; ++y) {
        size_t d = (size_t)m->deg[y];
        if (d < 2) continue;
        memcpy(tmp, m->col + m->rowptr[y], sizeof(uint32_t) * d);
        for (size_t i = 1; i < d; ++i) { /* insertion sort: rows are short */
            uint32_t x = tmp[i];
            size_t j = i;
            while (j > 0 && tmp[j - 1] > x) {
                tmp[j] = tmp[j - 1];
                --j;
            }
            tmp[j] = x;
        }
        size_t i = 0;
        while (i < d) {
            size_t j = i;
            while (j < d && tmp[j] == tmp[i]) ++j;
            size_t mult = j - i;
            if (mult > 1 && y > tmp[i]) ent += orc_lgamma_fast(mult + 1);
            i = j;
        }
    }
    free(tmp);
    ent += orc_lbinom_fast(m->ka * m->kb + m->num_edges - 1, m->num_edges);
    ent += orc_lbinom_fast(m->na - 1, m->ka - 1);
    ent += orc_lbinom_fast(m->nb - 1, m->kb - 1);
    ent += orc_safelog_fast(m->na * m->nb); /* the reference grows a na*nb table here (F5) */
    ent += orc_lgamma_fast(m->na + 1);
    ent += orc_lgamma_fast(m->nb + 1);
    return ent;
}

/* blockmodel.cc:290-333 (dead code in the reference; used as an RNG-free known answer) */
double orc_compute_dS_vertex(orc_model *m, size_t v, size_t r, size_t s) {
    if (r == s) return INFINITY;
    size_t K = m->K;
    double entropy0 = 0., entropy1 = 0.;
    compute_kv(m, v);
    int deg = m->deg[v];
    for (size_t t = 0; t < K; ++t) {
        int k = m->kv[t];
        int crit = (r < m->ka) ? (t >= m->ka) : (t < m->ka);
        if (crit && k != 0) {
            entropy0 -= orc_lgamma_fast((size_t)(m->m[r * K + t] + 1));
            entropy0 -= orc_lgamma_fast((size_t)(m->m[s * K + t] + 1));
            entropy1 -= orc_lgamma_fast((size_t)(m->m[r * K + t] - k + 1));
            entropy1 -= orc_lgamma_fast((size_t)(m->m[s * K + t] + k + 1));
        }
    }
    entropy0 -= -orc_lgamma_fast((size_t)(m->m_r[r] + 1));
    entropy0 -= -orc_lgamma_fast((size_t)(m->m_r[s] + 1));
    entropy1 -= -orc_lgamma_fast((size_t)(m->m_r[r] - deg + 1));
    entropy1 -= -orc_lgamma_fast((size_t)(m->m_r[s] + deg + 1));
    return entropy1 - entropy0;
}

/* fixed 64-leaf xor butterflies: the summation trees the wave uses.  dS: levels 32, then 1,2,4,8,16 (level 32 first: a lane that
 * holds two leaves adds its own pair before the cross-lane levels); the two Hastings sums: levels 32, 16, then 1,2,4,8 (level 16
 * second: one row swap folds both sums of a step at once).  With fewer leaves in use the levels that only add +0.0 drop out. */
static double butterfly64_levels(double *x, const int *levels) {
    double y[64];
    for (int j = 0; j < 6; ++j) {
        const int lvl = levels[j];
        for (int i = 0; i < 64; ++i) y[i] = x[i] + x[i ^ lvl];
        memcpy(x, y, sizeof(y));
    }
    return x[0];
}
static double butterfly64(double *x) {
    static const int levels[6] = {32, 1, 2, 4, 8, 16};
    return butterfly64_levels(x, levels);
}
static double butterfly64_accu(double *x) {
    static const int levels[6] = {32, 16, 1, 2, 4, 8};
    return butterfly64_levels(x, levels);
}

/* metropolis_hasting.cc:103-192.  Requires m->kv == k_[v].  Returns dS and sets m->accu_r. */
static double transition_ratio(orc_model *m, size_t v, size_t r, size_t s) {
    if (r == s) { /* :109-112 */
        m->accu_r = 1.;
        m->phx_accu0 = m->phx_accu1 = 1.;
        return 0.;
    }
    size_t KA = m->ka, K_ = m->K, D = m->max_degree + 1;
    double K = (double)(m->ka + m->kb); /* :120 */
    if ((r < KA && s >= KA) || (r >= KA && s < KA)) return INFINITY; /* :121-123, accu_r stale */
    double epsilon = m->epsilon;
    int deg = m->deg[v];
    const int *m_row_r = m->m + r * K_, *m_row_s = m->m + s * K_;
    int n_r_r = m->n_r[r], n_r_s = m->n_r[s];
    int eta_r = (int)m->eta[r * D + (size_t)deg], eta_s = (int)m->eta[s * D + (size_t)deg];
    int m0r = m->m_r[r], m1r = m0r - deg;
    int m0s = m->m_r[s], m1s = m0s + deg;
    double accu0 = 0., accu1 = 0., entropy0 = 0., entropy1 = 0.;
    size_t t_lo = (r < KA) ? KA : 0, t_hi = (r < KA) ? K_ : KA; /* criterion, :149 */
    if (m->rng_mode == ORC_RNG_COMPAT) {
        for (size_t t = t_lo; t < t_hi; ++t) { /* :150-163, ascending index, serial sums */
            int k = m->kv[t];
            if (k != 0) {
                accu0 += k * (m_row_s[t] + epsilon) / (m->m_r[t] + epsilon * K) / deg;
                accu1 += k * (m_row_r[t] - k + epsilon) / (m->m_r[t] + epsilon * K) / deg;
                entropy0 -= orc_lgamma_fast((size_t)(m_row_r[t] + 1));
                entropy0 -= orc_lgamma_fast((size_t)(m_row_s[t] + 1));
                entropy1 -= orc_lgamma_fast((size_t)(m_row_r[t] - k + 1));
                entropy1 -= orc_lgamma_fast((size_t)(m_row_s[t] + k + 1));
            }
        }
    } else {
        /* Production (Philox-mode) arithmetic: the same quantities in a cheaper, wave-shaped form.
         *   leaf l (0..63) collects the opposite-type blocks j with j mod 64 == l:
         *     a0_l += k (m_st + eps) * inv_t,  a1_l += k (m_rt - k + eps) * inv_t,  inv_t = 1/(m_r[t] + eps K)
         *       (the common factor 1/deg of :153-154 cancels in accu1/accu0 and is dropped)
         *     d_l  += (lg(m_rt+1) + lg(m_st+1)) - (lg(m_rt-k+1) + lg(m_st+k+1))        (:155-158 as S1 - S0)
         *   the eight scalar lgamma terms (:164-177) are folded into leaves 0..7 and the four log_q
         *   terms (:179-183) into leaves 0..3, then each array is summed by the 64-leaf butterfly. */
        double a0[64] = {0}, a1[64] = {0}, d[64] = {0};
        for (size_t t = t_lo; t < t_hi; ++t) {
            int k = m->kv[t];
            size_t leaf = (t - t_lo) & 63;
            if (k != 0) {
                double inv = 1.0 / (m->m_r[t] + epsilon * K);
                a0[leaf] += k * (m_row_s[t] + epsilon) * inv;
                a1[leaf] += k * (m_row_r[t] - k + epsilon) * inv;
                double L1 = orc_lgamma_fast((size_t)(m_row_r[t] + 1));
                double L2 = orc_lgamma_fast((size_t)(m_row_s[t] + 1));
                double L3 = orc_lgamma_fast((size_t)(m_row_r[t] - k + 1));
                double L4 = orc_lgamma_fast((size_t)(m_row_s[t] + k + 1));
                d[leaf] += (L1 + L2) - (L3 + L4);
            }
        }
        d[0] = d[0] + -orc_lgamma_fast((size_t)(m0r + 1));
        d[1] = d[1] + -orc_lgamma_fast((size_t)(m0s + 1));
        d[2] = d[2] + orc_lgamma_fast((size_t)(m1r + 1));
        d[3] = d[3] + orc_lgamma_fast((size_t)(m1s + 1));
        d[4] = d[4] + orc_lgamma_fast((size_t)(eta_r + 1));
        d[5] = d[5] + orc_lgamma_fast((size_t)(eta_s + 1));
        d[6] = d[6] + -orc_lgamma_fast((size_t)(eta_r - 1 + 1));
        d[7] = d[7] + -orc_lgamma_fast((size_t)(eta_s + 1 + 1));
        d[0] = d[0] + -orc_log_q_philox(m0r, n_r_r);
        d[1] = d[1] + -orc_log_q_philox(m0s, n_r_s);
        d[2] = d[2] + orc_log_q_philox(m1r, n_r_r - 1);
        d[3] = d[3] + orc_log_q_philox(m1s, n_r_s + 1);
        m->phx_accu0 = deg == 0 ? 1. : butterfly64_accu(a0);
        m->phx_accu1 = deg == 0 ? 1. : butterfly64_accu(a1);
        m->accu_r = m->phx_accu1 / m->phx_accu0;
        return butterfly64(d);
    }
    entropy0 -= -orc_lgamma_fast((size_t)(m0r + 1)); /* :164-168 */
    entropy0 -= -orc_lgamma_fast((size_t)(m0s + 1));
    entropy1 -= -orc_lgamma_fast((size_t)(m1r + 1));
    entropy1 -= -orc_lgamma_fast((size_t)(m1s + 1));
    entropy0 += -orc_lgamma_fast((size_t)(eta_r + 1)); /* :173-177 */
    entropy0 += -orc_lgamma_fast((size_t)(eta_s + 1));
    entropy1 += -orc_lgamma_fast((size_t)(eta_r - 1 + 1));
    entropy1 += -orc_lgamma_fast((size_t)(eta_s + 1 + 1));
    entropy0 += orc_log_q(m0r, n_r_r); /* :179-183 */
    entropy0 += orc_log_q(m0s, n_r_s);
    entropy1 += orc_log_q(m1r, n_r_r - 1);
    entropy1 += orc_log_q(m1s, n_r_s + 1);
    if (deg == 0) /* :185-189 */
        m->accu_r = 1;
    else
        m->accu_r = accu1 / accu0;
    return entropy1 - entropy0;
}

double orc_transition_ratio(orc_model *m, size_t v, size_t s, double *accu_r) {
    compute_kv(m, v);
    double dS = transition_ratio(m, v, m->labels[v], s);
    if (accu_r) *accu_r = m->accu_r;
    return dS;
}

/* blockmodel.cc:461-503 (k_ bookkeeping :492-495 dropped: k_[v] is recomputed per step) */
static int apply_mcmc_move(orc_model *m, size_t v, size_t r, size_t s, double dS) {
    size_t K = m->K, D = m->max_degree + 1;
    --m->n_r[r];
    if (m->n_r[r] == 0) { /* :467-471 veto after the accept draw */
        ++m->n_r[r];
        return 0;
    }
    ++m->n_r[s];
    --m->eta[r * D + (size_t)m->deg[v]];
    ++m->eta[s * D + (size_t)m->deg[v]];
    for (size_t i = 0; i < K; ++i) { /* :479-487 */
        int k = m->kv[i];
        if (k != 0) {
            m->m[r * K + i] -= k;
            m->m[s * K + i] += k;
            m->m[i * K + r] = m->m[r * K + i];
            m->m[i * K + s] = m->m[s * K + i];
        }
    }
    m->m_r[r] -= m->deg[v];
    m->m_r[s] += m->deg[v];
    m->labels[v] = (uint32_t)s;
    m->entropy += dS;
    return 1;
}

/* blockmodel.cc:613-637, compat draw order */
static size_t propose_compat(orc_model *m, size_t v) {
    size_t ty = v < m->na ? 0 : 1;
    if ((ty == 0 && m->ka == 1) || (ty == 1 && m->kb == 1)) return m->labels[v];
    size_t d = (size_t)m->deg[v];
    if (d == 0) return (size_t)(orc_mt_canonical(&m->engine) * (double)m->K);
    size_t which = (size_t)(orc_mt_canonical(&m->engine) * (double)d);
    size_t j = m->col[m->rowptr[v] + which];
    size_t t = m->labels[j];
    double R_t = m->epsilon * (double)m->K / (m->m_r[t] + m->epsilon * (double)m->K);
    if (orc_mt_canonical(&m->engine) < R_t) return (size_t)(orc_mt_canonical(&m->engine) * (double)m->K);
    return orc_mt_discrete(&m->gen, m->m + t * m->K, m->K); /* drawn with `gen`, :627-628 */
}

/* Production definition of the same proposal: u_idx, u_R, u_tgt are the step's Philox uniforms. */
static size_t propose_philox(orc_model *m, size_t v, double u_idx, double u_R, double u_tgt) {
    size_t ty = v < m->na ? 0 : 1;
    if ((ty == 0 && m->ka == 1) || (ty == 1 && m->kb == 1)) return m->labels[v];
    size_t d = (size_t)m->deg[v];
    size_t K = m->K;
    if (d == 0) {
        size_t s = (size_t)(u_idx * (double)K);
        return s >= K ? K - 1 : s;
    }
    size_t which = (size_t)(u_idx * (double)d);
    if (which >= d) which = d - 1;
    size_t j = m->col[m->rowptr[v] + which];
    size_t t = m->labels[j];
    /* u < eps K / (m_r[t] + eps K)  <=>  u (m_r[t] + eps K) < eps K   (blockmodel.cc:622-624, no division) */
    if (u_R * (m->m_r[t] + m->epsilon * (double)K) < m->epsilon * (double)K) {
        size_t s = (size_t)(u_tgt * (double)K);
        return s >= K ? K - 1 : s;
    }
    /* integer inverse CDF over row m[t][.] restricted to v's own type (the only non-zero part) */
    int64_t tot = m->m_r[t];
    int64_t x = (int64_t)(u_tgt * (double)tot);
    if (x >= tot) x = tot - 1;
    size_t lo = ty == 0 ? 0 : m->ka, hi = ty == 0 ? m->ka : K;
    int64_t cum = 0;
    for (size_t s = lo; s < hi; ++s) {
        cum += m->m[t * K + s];
        if (cum > x) return s;
    }
    return hi - 1;
}

/* test hook: the Philox-mode proposal for given uniforms (tests enumerate them to get the exact target distribution
 * and compare it with blockmodel.cc:619-628's R_t/K + (1 - R_t) m[t][s]/m_r[t]) */
size_t orc_propose_philox(orc_model *m, size_t v, double u_idx, double u_R, double u_tgt) {
    return propose_philox(m, v, u_idx, u_R, u_tgt);
}

/* metropolis_hasting.cc:42-62 */
static int step_compat(orc_model *m, size_t v, double temperature) {
    size_t s = propose_compat(m, v);
    size_t r = m->labels[v];
    compute_kv(m, v);
    double dS = transition_ratio(m, v, r, s);
    if (temperature == 0.) {
        if (dS < 0) return apply_mcmc_move(m, v, r, s, dS);
        return 0;
    }
    double a = -1. / temperature * dS + log(m->accu_r);
    if (a > 0.) return apply_mcmc_move(m, v, r, s, dS);
    if (orc_mt_canonical(&m->engine) < exp(a)) return apply_mcmc_move(m, v, r, s, dS);
    return 0;
}

static int step_philox(orc_model *m, size_t v, double temperature, uint64_t gstep) {
    uint32_t A[4], B[4];
    phx_draw(m->phx_seed, m->phx_chain, PHX_STEP_A, gstep, A);
    phx_draw(m->phx_seed, m->phx_chain, PHX_STEP_B, gstep, B);
    double u_idx = u53(A[0], A[1]), u_R = u53(A[2], A[3]);
    double u_tgt = u53(B[0], B[1]), u_acc = u53(B[2], B[3]);
    size_t s = propose_philox(m, v, u_idx, u_R, u_tgt);
    size_t r = m->labels[v];
    compute_kv(m, v);
    double dS = transition_ratio(m, v, r, s);
    int cross = (r != s) && ((r < m->ka) != (s < m->ka));
    if (cross) return 0; /* dS = +inf: never accepted (:121-123) */
    if (temperature == 0.) {
        if (dS < 0) return apply_mcmc_move(m, v, r, s, dS);
        return 0;
    }
    /* u < exp(-dS/T) accu1/accu0 (:54-57) written without the log and the quotient */
    if (u_acc * m->phx_accu0 < m->phx_accu1 * exp(-dS * (1.0 / temperature))) return apply_mcmc_move(m, v, r, s, dS);
    return 0;
}

/* Diagnostic ("two steps per wave"): runs `sweeps` Philox-mode sweeps at constant T exactly as
 * orc_anneal does and, along the way, counts how often step q+1 of a 64-node chunk could have been evaluated against
 * the state BEFORE step q without changing its outcome: step q not applied (rejected, vetoed or r == s), or applied
 * with {r,s} disjoint from {r',s'} and k_q[t'] == 0 (t' = block of step q+1's pivot neighbour: column t' of m feeds its
 * proposal).  out: {steps, pair evaluations, pairs whose second step stood, first steps applied, conflicts by row,
 * conflicts by column only}. */
void orc_pair_probe(orc_model *m, uint64_t sweeps, double temperature, uint64_t out[6]) {
    size_t n = m->n;
    int *kv_a = (int *)malloc(sizeof(int) * m->K);
    memset(out, 0, sizeof(uint64_t) * 6);
    for (uint64_t sw = 0; sw < sweeps; ++sw) {
        uint32_t keys[2][4];
        phx_draw(m->phx_seed, m->phx_chain, PHX_SWEEP_KEY, 2 * m->sweeps_total, keys[0]);
        phx_draw(m->phx_seed, m->phx_chain, PHX_SWEEP_KEY, 2 * m->sweeps_total + 1, keys[1]);
        for (int ph = 0; ph < 2; ++ph) {
            size_t n_own = ph ? m->nb : m->na, base = ph ? m->na : 0;
            for (size_t c0 = 0; c0 < n_own; c0 += 64) {
                size_t cnt = n_own - c0 < 64 ? n_own - c0 : 64;
                size_t q = 0;
                while (q < cnt) {
                    size_t vi = base + c0 + q;
                    size_t v = base + tiled_perm(keys[ph], (uint32_t)n_own, (uint32_t)(c0 + q));
                    uint64_t gs = m->sweeps_total * n + vi;
                    size_t r = m->labels[v];
                    int have_b = q + 1 < cnt;
                    size_t rb = 0, sb = 0, tb = 0;
                    int b_uses_column = 0;
                    if (have_b) { /* step q+1's proposal on the state before step q */
                        size_t vb = base + tiled_perm(keys[ph], (uint32_t)n_own, (uint32_t)(c0 + q + 1));
                        uint32_t A[4], B[4];
                        phx_draw(m->phx_seed, m->phx_chain, PHX_STEP_A, gs + 1, A);
                        phx_draw(m->phx_seed, m->phx_chain, PHX_STEP_B, gs + 1, B);
                        rb = m->labels[vb];
                        sb = propose_philox(m, vb, u53(A[0], A[1]), u53(A[2], A[3]), u53(B[0], B[1]));
                        size_t d = (size_t)m->deg[vb];
                        if (d) {
                            size_t which = (size_t)(u53(A[0], A[1]) * (double)d);
                            if (which >= d) which = d - 1;
                            tb = m->labels[m->col[m->rowptr[vb] + which]];
                            b_uses_column = 1;
                        }
                    }
                    compute_kv(m, v);
                    memcpy(kv_a, m->kv, sizeof(int) * m->K);
                    int ok = step_philox(m, v, temperature, gs);
                    size_t s = m->labels[v];
                    int changed = ok && s != r;
                    out[0]++;
                    out[3] += (uint64_t)changed;
                    if (!have_b) {
                        q += 1;
                        continue;
                    }
                    out[1]++;
                    int row = changed && (rb == r || rb == s || sb == r || sb == s);
                    /* column t' of m changes where k_q[t'] != 0, but only rows r and s of it: the running sums of the
                     * inverse CDF move only for blocks in [min(r,s), max(r,s)), so a target outside that span stands */
                    size_t lo_rs = r < s ? r : s, hi_rs = r < s ? s : r;
                    int col = changed && b_uses_column && kv_a[tb] != 0 && sb > lo_rs && sb < hi_rs;
                    if (row)
                        out[4]++;
                    else if (col)
                        out[5]++;
                    if (row || col) {
                        q += 1;
                    } else {
                        size_t vb = base + tiled_perm(keys[ph], (uint32_t)n_own, (uint32_t)(c0 + q + 1));
                        { /* the claim itself: the proposal evaluated after step q is the one evaluated before it */
                            uint32_t A[4], B[4];
                            phx_draw(m->phx_seed, m->phx_chain, PHX_STEP_A, gs + 1, A);
                            phx_draw(m->phx_seed, m->phx_chain, PHX_STEP_B, gs + 1, B);
                            if (propose_philox(m, vb, u53(A[0], A[1]), u53(A[2], A[3]), u53(B[0], B[1])) != sb) abort();
                        }
                        step_philox(m, vb, temperature, gs + 1);
                        out[0]++;
                        out[2]++;
                        q += 2;
                    }
                }
            }
        }
        m->sweeps_total++;
    }
    free(kv_a);
}

/* Diagnostic, like orc_pair_probe but for passes of up to `depth` (<= 8) steps: step q+j of a pass stands when every
 * step before it in the pass that moved its node left its inputs alone (rows: block sets disjoint; column t': no edges
 * to it, or the target outside the moved span); a pass ends at the first step that does not stand.  out[0] = steps,
 * out[1] = passes, out[2 + j] = passes that committed exactly j + 1 steps.  The chain is anneal()'s. */
void orc_depth_probe(orc_model *m, uint64_t sweeps, double temperature, int depth, uint64_t out[12]) {
    size_t n = m->n, K = m->K;
    if (depth < 1) depth = 1;
    if (depth > 8) depth = 8;
    int *kv_hist = (int *)malloc(sizeof(int) * K * 8);
    memset(out, 0, sizeof(uint64_t) * 12);
    for (uint64_t sw = 0; sw < sweeps; ++sw) {
        uint32_t keys[2][4];
        phx_draw(m->phx_seed, m->phx_chain, PHX_SWEEP_KEY, 2 * m->sweeps_total, keys[0]);
        phx_draw(m->phx_seed, m->phx_chain, PHX_SWEEP_KEY, 2 * m->sweeps_total + 1, keys[1]);
        for (int ph = 0; ph < 2; ++ph) {
            size_t n_own = ph ? m->nb : m->na, base = ph ? m->na : 0;
            for (size_t c0 = 0; c0 < n_own; c0 += 64) {
                size_t cnt = n_own - c0 < 64 ? n_own - c0 : 64;
                size_t q = 0;
                while (q < cnt) {
                    size_t d = cnt - q < (size_t)depth ? cnt - q : (size_t)depth;
                    size_t rj[8], sj[8], tj[8];
                    int uses_col[8];
                    for (size_t j = 0; j < d; ++j) { /* every step of the pass proposed on the state before the pass */
                        size_t vj = base + tiled_perm(keys[ph], (uint32_t)n_own, (uint32_t)(c0 + q + j));
                        uint64_t gs = m->sweeps_total * n + base + c0 + q + j;
                        uint32_t A[4], B[4];
                        phx_draw(m->phx_seed, m->phx_chain, PHX_STEP_A, gs, A);
                        phx_draw(m->phx_seed, m->phx_chain, PHX_STEP_B, gs, B);
                        rj[j] = m->labels[vj];
                        sj[j] = propose_philox(m, vj, u53(A[0], A[1]), u53(A[2], A[3]), u53(B[0], B[1]));
                        size_t dg = (size_t)m->deg[vj];
                        uses_col[j] = dg != 0;
                        tj[j] = 0;
                        if (dg) {
                            size_t which = (size_t)(u53(A[0], A[1]) * (double)dg);
                            if (which >= dg) which = dg - 1;
                            tj[j] = m->labels[m->col[m->rowptr[vj] + which]];
                        }
                    }
                    size_t committed = 0;
                    size_t mr[8], ms[8];
                    int moved[8];
                    for (size_t j = 0; j < d; ++j) {
                        int stands = 1;
                        for (size_t i = 0; i < j && stands; ++i) {
                            if (!moved[i]) continue;
                            size_t lo = mr[i] < ms[i] ? mr[i] : ms[i], hi = mr[i] < ms[i] ? ms[i] : mr[i];
                            if (rj[j] == mr[i] || rj[j] == ms[i] || sj[j] == mr[i] || sj[j] == ms[i]) stands = 0;
                            else if (uses_col[j] && kv_hist[i * K + tj[j]] != 0 && sj[j] > lo && sj[j] < hi) stands = 0;
                        }
                        if (!stands) break;
                        size_t vj = base + tiled_perm(keys[ph], (uint32_t)n_own, (uint32_t)(c0 + q + j));
                        uint64_t gs = m->sweeps_total * n + base + c0 + q + j;
                        compute_kv(m, vj);
                        memcpy(kv_hist + j * K, m->kv, sizeof(int) * K);
                        size_t r0 = m->labels[vj];
                        if (j > 0) { /* the claim: the proposal made before the pass is the proposal now */
                            uint32_t A[4], B[4];
                            phx_draw(m->phx_seed, m->phx_chain, PHX_STEP_A, gs, A);
                            phx_draw(m->phx_seed, m->phx_chain, PHX_STEP_B, gs, B);
                            if (propose_philox(m, vj, u53(A[0], A[1]), u53(A[2], A[3]), u53(B[0], B[1])) != sj[j] || r0 != rj[j]) abort();
                        }
                        int ok = step_philox(m, vj, temperature, gs);
                        moved[j] = ok && m->labels[vj] != r0;
                        mr[j] = r0;
                        ms[j] = m->labels[vj];
                        ++committed;
                    }
                    out[0] += committed;
                    out[1] += 1;
                    out[2 + committed - 1] += 1;
                    q += committed;
                }
            }
        }
        m->sweeps_total++;
    }
    free(kv_hist);
}

/* metropolis_hasting.cc:64-101 */
double orc_anneal(orc_model *m, int schedule, float kw0, float kw1, uint64_t duration,
                  uint64_t steps_await) {
    size_t num_nodes = m->n;
    uint64_t accepted_steps = 0;
    uint64_t u = 0;
    m->entropy_min = INFINITY;
    uint64_t all_sweeps = duration / num_nodes;
    double temperature = 1;
    m->last_sweeps = 0;
    for (uint64_t sweep = 0; sweep < all_sweeps; ++sweep) {
        uint32_t keys_a[4], keys_b[4];
        if (m->rng_mode == ORC_RNG_COMPAT)
            orc_mt_shuffle_u32(&m->engine, m->vlist, num_nodes);
        else {
            phx_draw(m->phx_seed, m->phx_chain, PHX_SWEEP_KEY, 2 * m->sweeps_total, keys_a);
            phx_draw(m->phx_seed, m->phx_chain, PHX_SWEEP_KEY, 2 * m->sweeps_total + 1, keys_b);
        }
        uint64_t current_step = num_nodes * sweep;
        for (size_t vi = 0; vi < num_nodes; ++vi) {
            temperature = orc_schedule(schedule, current_step + vi, kw0, kw1);
            int ok;
            if (m->rng_mode == ORC_RNG_COMPAT)
                ok = step_compat(m, m->vlist[vi], temperature);
            else {
                size_t v = vi < m->na ? tiled_perm(keys_a, (uint32_t)m->na, (uint32_t)vi)
                                      : m->na + tiled_perm(keys_b, (uint32_t)m->nb, (uint32_t)(vi - m->na));
                ok = step_philox(m, v, temperature, m->sweeps_total * num_nodes + vi);
            }
            if (ok) {
                ++accepted_steps;
                if (m->entropy < m->entropy_min) {
                    m->entropy_min = m->entropy;
                    u = 0;
                }
            }
            if (temperature < 1.) ++u;
        }
        m->sweeps_total++;
        m->last_sweeps = sweep + 1;
        if (u >= steps_await) {
            m->last_accepted = accepted_steps;
            return (double)accepted_steps / (double)((sweep + 1) * num_nodes);
        }
    }
    m->last_accepted = accepted_steps;
    return (double)accepted_steps / (double)duration;
}

/* ------------------------------------------------------------------------------------------
 * Agglomerative merges and splits between anneals, blockmodel.cc:109-288,335-459,505-611,639-669.
 * agg_split follows the INTENDED semantics (the reference's split dS indexes its split vector
 * with a counter over all nodes, an out-of-range read): the position of a node is its rank within its block.
 * ---------------------------------------------------------------------------------------- */
size_t orc_ka(const orc_model *m) { return m->ka; }
size_t orc_kb(const orc_model *m) { return m->kb; }

typedef struct {
    size_t source, target;
} orc_block_move;

/* single_block_change, blockmodel.cc:639-669.  badj: blocks t with m[src][t] > 0, ascending (:274-288).
 * Philox mode: the same proposal from the counter-based uniforms (u0, u1, u2) of proposal `ctr` of the
 * current merge epoch, random-target test and target draw written as in the vertex proposal. */
static orc_block_move single_block_change(orc_model *m, size_t src, uint64_t ctr) {
    size_t K = m->K, KA = m->ka, KB = m->kb;
    orc_block_move mv;
    if ((KA == 1 && src < KA) || (KB == 1 && src >= KA)) {
        mv.source = mv.target = src;
        return mv;
    }
    size_t nadj = 0;
    for (size_t t = 0; t < K; ++t) nadj += m->m[src * K + t] > 0;
    size_t target;
    if (m->rng_mode == ORC_RNG_COMPAT) {
        if (nadj == 0) {
            target = (size_t)(orc_mt_canonical(&m->engine) * (double)K);
        } else {
            size_t which = (size_t)(orc_mt_canonical(&m->engine) * (double)nadj);
            size_t t = 0, seen = 0;
            for (; t < K; ++t)
                if (m->m[src * K + t] > 0 && seen++ == which) break;
            double R_t = m->epsilon * (double)K / (m->m_r[t] + m->epsilon * (double)K);
            if (orc_mt_canonical(&m->engine) < R_t)
                target = (size_t)(orc_mt_canonical(&m->engine) * (double)K);
            else
                target = orc_mt_discrete(&m->gen, m->m + t * K, K);
        }
    } else {
        uint32_t A[4], B[4];
        uint64_t idx = ((uint64_t)m->merge_epoch << 32) | ctr;
        phx_draw(m->phx_seed, m->phx_chain, PHX_MERGE_A, idx, A);
        phx_draw(m->phx_seed, m->phx_chain, PHX_MERGE_B, idx, B);
        double u0 = u53(A[0], A[1]), u1 = u53(A[2], A[3]), u2 = u53(B[0], B[1]);
        if (nadj == 0) {
            target = (size_t)(u0 * (double)K);
            if (target >= K) target = K - 1;
        } else {
            size_t which = (size_t)(u0 * (double)nadj);
            if (which >= nadj) which = nadj - 1;
            size_t t = 0, seen = 0;
            for (; t < K; ++t)
                if (m->m[src * K + t] > 0 && seen++ == which) break;
            if (u1 * (m->m_r[t] + m->epsilon * (double)K) < m->epsilon * (double)K) {
                target = (size_t)(u2 * (double)K);
                if (target >= K) target = K - 1;
            } else { /* integer inverse CDF over row m[t][.] */
                int64_t tot = m->m_r[t];
                int64_t x = (int64_t)(u2 * (double)tot);
                if (x >= tot) x = tot - 1;
                int64_t cum = 0;
                target = K - 1;
                for (size_t c = 0; c < K; ++c) {
                    cum += m->m[t * K + c];
                    if (cum > x) {
                        target = c;
                        break;
                    }
                }
            }
        }
    }
    if (src > target) {
        mv.source = src;
        mv.target = target;
    } else {
        mv.source = target;
        mv.target = src;
    }
    return mv;
}

/* compute_dS(const block_move_t&), blockmodel.cc:335-372 */
double orc_merge_dS(const orc_model *m, size_t r, size_t s) {
    size_t K = m->K, KA = m->ka;
    if (r == s || (r < KA && s >= KA) || (r >= KA && s < KA)) return INFINITY;
    double entropy0 = 0., entropy1 = 0.;
    for (size_t idx = 0; idx < K; ++idx) {
        int crit = (r < KA) ? (idx >= KA) : (idx < KA);
        if (crit && m->m_r[idx] != 0) {
            entropy0 -= orc_lgamma_fast((size_t)(m->m[r * K + idx] + 1));
            entropy0 -= orc_lgamma_fast((size_t)(m->m[s * K + idx] + 1));
            entropy1 -= orc_lgamma_fast((size_t)(m->m[s * K + idx] + m->m[r * K + idx] + 1));
        }
    }
    entropy0 -= -orc_lgamma_fast((size_t)(m->m_r[r] + 1));
    entropy0 -= -orc_lgamma_fast((size_t)(m->m_r[s] + 1));
    entropy1 -= -orc_lgamma_fast((size_t)(m->m_r[r] + m->m_r[s] + 1));
    return entropy1 - entropy0;
}

typedef struct {
    double dS;
    size_t ii;
} orc_heap_item;
static int heap_cmp(const void *a, const void *b) { /* std::pair<double,size_t> ascending = pop order of the min-heap */
    const orc_heap_item *x = (const orc_heap_item *)a, *y = (const orc_heap_item *)b;
    if (x->dS < y->dS) return -1;
    if (x->dS > y->dS) return 1;
    return x->ii < y->ii ? -1 : (x->ii > y->ii ? 1 : 0);
}

/* the groups of accepted merges: vector<set<size_t>> in creation order (blockmodel.cc:169-200) */
typedef struct {
    size_t *group_of_slot; /* members, grouped: members[g*K .. ] */
    size_t *count;
    size_t n_groups, K;
} orc_groups;
static int group_has(const orc_groups *G, size_t g, size_t x) {
    for (size_t i = 0; i < G->count[g]; ++i)
        if (G->group_of_slot[g * G->K + i] == x) return 1;
    return 0;
}
static void group_add(orc_groups *G, size_t g, size_t x) {
    if (!group_has(G, g, x)) G->group_of_slot[g * G->K + G->count[g]++] = x;
}
static size_t group_min(const orc_groups *G, size_t g) {
    size_t mn = G->group_of_slot[g * G->K];
    for (size_t i = 1; i < G->count[g]; ++i)
        if (G->group_of_slot[g * G->K + i] < mn) mn = G->group_of_slot[g * G->K + i];
    return mn;
}
/* one accepted move (source, target): :172-184 */
static void accept_move(orc_groups *G, unsigned char *in_e, size_t src, size_t tgt) {
    if (!in_e[src] && !in_e[tgt]) {
        size_t g = G->n_groups++;
        G->count[g] = 0;
        group_add(G, g, src);
        group_add(G, g, tgt);
    } else {
        for (size_t g = 0; g < G->n_groups; ++g)
            if (group_has(G, g, tgt) || group_has(G, g, src)) {
                group_add(G, g, src);
                group_add(G, g, tgt);
                break;
            }
    }
    in_e[src] = in_e[tgt] = 1;
}

/* apply_block_moves, blockmodel.cc:567-611.  Returns 0, or -1 on the reference's sanity failure (exit(0) there). */
static int apply_block_moves(orc_model *m, const unsigned char *in_e, const orc_groups *G) {
    size_t n = m->n;
    for (size_t v = 0; v < n; ++v) {
        uint32_t mb = m->labels[v];
        if (in_e[mb]) {
            for (size_t g = 0; g < G->n_groups; ++g)
                if (group_has(G, g, mb)) mb = (uint32_t)group_min(G, g); /* (keeps scanning with the new label) */
        }
        m->labels[v] = mb;
    }
    int *n2o = (int *)malloc(sizeof(int) * (m->K + 1));
    for (size_t i = 0; i <= m->K; ++i) n2o[i] = -1;
    size_t KA = 0, KB = 0, cnt = 0;
    for (size_t v = 0; v < n; ++v) {
        uint32_t mb = m->labels[v];
        if (n2o[mb] == -1) n2o[mb] = (int)cnt++;
        mb = (uint32_t)n2o[mb];
        m->labels[v] = mb;
        if (v < m->na) {
            if (mb > KA) KA = mb;
        } else {
            if (mb > KB) KB = mb;
        }
    }
    free(n2o);
    KB -= KA;
    KA += 1;
    m->ka = KA;
    m->kb = KB;
    m->K = KA + KB;
    if (cnt != m->K) return -1;
    orc_init_bisbm(m);
    return 0;
}

/* the proposal round shared by both agg_merge overloads (:147-159 / :231-243): nm proposals per listed
 * block, de-duplicated on "source>target", keyed by dS */
static size_t propose_round(orc_model *m, size_t first, size_t count, int nm, orc_block_move *moves,
                            orc_heap_item *heap) {
    size_t K = m->K, ii = 0;
    unsigned char *seen = (unsigned char *)calloc(K * K, 1);
    uint64_t ctr = 0;
    for (size_t b = first; b < first + count; ++b)
        for (int i = 0; i < nm; ++i) {
            orc_block_move mv = single_block_change(m, b, ctr++);
            if (!seen[mv.source * K + mv.target]) {
                seen[mv.source * K + mv.target] = 1;
                moves[ii] = mv;
                heap[ii].dS = orc_merge_dS(m, mv.source, mv.target);
                heap[ii].ii = ii;
                ++ii;
            }
        }
    free(seen);
    m->merge_epoch++;
    qsort(heap, ii, sizeof(orc_heap_item), heap_cmp);
    return ii;
}

/* room for one more block in the state arrays */
static void grow_blocks(orc_model *m, size_t K) {
    if (K <= m->cap_K) return;
    size_t D = m->max_degree + 1;
    m->m = (int *)realloc(m->m, sizeof(int) * K * K);
    m->m_r = (int *)realloc(m->m_r, sizeof(int) * K);
    m->n_r = (int *)realloc(m->n_r, sizeof(int) * K);
    m->eta = (uint32_t *)realloc(m->eta, sizeof(uint32_t) * K * D);
    m->kv = (int *)realloc(m->kv, sizeof(int) * K);
    m->cap_K = K;
}

/* compute_dS(size_t mb, vector<bool>& split_move), blockmodel.cc:374-424, with the position of a node in the split
 * vector being its RANK WITHIN THE BLOCK (ascending node id).  The reference advances its counter for every node of
 * the graph (:402 sits outside the `if (_mb == r_)`), which reads past the vector unless the block's nodes are the
 * first ids; agg_split's own pass (:554-561) indexes by rank, and that is the meaning restated here.
 * k[t]: edges from the moved nodes to opposite-type block t; deg: their degree sum. */
static double split_dS(const orc_model *m, size_t r, const int *k, int deg) {
    size_t K = m->K;
    double entropy0 = 0., entropy1 = 0.;
    size_t t_lo = (r < m->ka) ? m->ka : 0, t_hi = (r < m->ka) ? K : m->ka; /* criterion, :380 */
    for (size_t t = t_lo; t < t_hi; ++t) {                                  /* :409-417 */
        entropy0 -= orc_lgamma_fast((size_t)(m->m[r * K + t] + 1));
        entropy1 -= orc_lgamma_fast((size_t)(m->m[r * K + t] - k[t] + 1));
        entropy1 -= orc_lgamma_fast((size_t)(k[t] + 1));
    }
    entropy0 -= -orc_lgamma_fast((size_t)(m->m_r[r] + 1)); /* :418-421 */
    entropy1 -= -orc_lgamma_fast((size_t)(m->m_r[r] - deg + 1));
    entropy1 -= -orc_lgamma_fast((size_t)(deg + 1));
    return entropy1 - entropy0;
}

/* agg_split(engine, type, nm), blockmodel.cc:505-565 + apply_split_moves :428-459: every block of the type with more
 * than one node is cut nm times into a random floor(n/2) / ceil(n/2) partition (`splitter_` shuffled once, then once
 * more before every trial); the cut with the lowest dS over all blocks and trials (strict <, blocks ascending, trials
 * in order) is applied: the marked nodes form a new block -- label KA for type a (every label >= KA moves up by
 * one first, :434-443), label K for type b -- and the block state is rebuilt.
 * compat: the shuffles are std::shuffle on `engine`.  Philox: node of rank i is marked in trial j of block b iff
 * feistel_perm(key(b, j), n_b, i) >= floor(n_b / 2) (a keyed random bijection: the same uniform cut, no serial
 * shuffle), key = Philox(seed, chain, PHX_SPLIT, split_epoch << 32 | b << 16 | j).
 * Returns 0, -3 when no block of the type has two nodes (the reference would apply an empty move list and still
 * count a block more), -4 when the label format would overflow (K + 1 > 256 is the product's limit, not the oracle's). */
int orc_agg_split(orc_model *m, int type, int nm) {
    size_t K = m->K, n = m->n;
    size_t b_lo = type ? m->ka : 0, b_hi = type ? K : m->ka;
    size_t v_lo = type ? m->na : 0, v_hi = type ? n : m->na;
    if (nm < 1 || nm > 65535) return -5;
    /* nodes of every block in ascending id: rank -> node */
    size_t *off = (size_t *)calloc(K + 1, sizeof(size_t));
    for (size_t b = 0; b < K; ++b) off[b + 1] = off[b] + (size_t)m->n_r[b];
    uint32_t *members = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    size_t *fill = (size_t *)calloc(K, sizeof(size_t));
    for (size_t v = 0; v < n; ++v) members[off[m->labels[v]] + fill[m->labels[v]]++] = (uint32_t)v;
    free(fill);
    double best = INFINITY;
    size_t best_block = 0;
    uint8_t *best_cut = NULL;
    size_t best_n = 0;
    int *k = (int *)malloc(sizeof(int) * K);
    for (size_t b = b_lo; b < b_hi; ++b) {
        size_t nb = (size_t)m->n_r[b];
        if (nb <= 1) continue;
        size_t unchange = nb / 2;
        uint8_t *cut = (uint8_t *)malloc(nb);
        for (size_t i = 0; i < nb; ++i) cut[i] = i >= unchange;
        if (m->rng_mode == ORC_RNG_COMPAT) orc_mt_shuffle_u8(&m->engine, cut, nb); /* :541 */
        for (int j = 0; j < nm; ++j) {
            if (m->rng_mode == ORC_RNG_COMPAT) {
                orc_mt_shuffle_u8(&m->engine, cut, nb); /* :543 */
            } else {
                uint32_t keys[4];
                uint64_t idx = ((uint64_t)m->split_epoch << 32) | ((uint64_t)b << 16) | (uint64_t)j;
                phx_draw(m->phx_seed, m->phx_chain, PHX_SPLIT, idx, keys);
                for (size_t i = 0; i < nb; ++i) cut[i] = feistel_perm(keys, (uint32_t)nb, (uint32_t)i) >= unchange;
            }
            memset(k, 0, sizeof(int) * K);
            int deg = 0;
            for (size_t i = 0; i < nb; ++i) {
                if (!cut[i]) continue;
                size_t v = members[off[b] + i];
                for (uint64_t e = m->rowptr[v]; e < m->rowptr[v + 1]; ++e) k[m->labels[m->col[e]]]++;
                deg += m->deg[v];
            }
            double dS = split_dS(m, b, k, deg);
            if (dS < best) { /* :545-550 */
                best = dS;
                best_block = b;
                best_n = nb;
                free(best_cut);
                best_cut = (uint8_t *)malloc(nb);
                memcpy(best_cut, cut, nb);
            }
        }
        free(cut);
    }
    free(k);
    m->split_epoch++;
    m->last_split_dS = best;
    if (!best_cut) {
        free(off), free(members);
        return -3;
    }
    /* apply_split_moves, :428-459 */
    if (!type) {
        for (size_t v = 0; v < n; ++v)
            if (m->labels[v] >= m->ka) m->labels[v]++;
        for (size_t i = 0; i < best_n; ++i)
            if (best_cut[i]) m->labels[members[off[best_block] + i]] = (uint32_t)m->ka;
        m->ka++;
    } else {
        for (size_t i = 0; i < best_n; ++i)
            if (best_cut[i]) m->labels[members[off[best_block] + i]] = (uint32_t)K;
        m->kb++;
    }
    (void)v_lo, (void)v_hi;
    m->K = K + 1;
    grow_blocks(m, m->K);
    free(best_cut), free(off), free(members);
    orc_init_bisbm(m);
    return 0;
}

double orc_last_split_dS(const orc_model *m) { return m->last_split_dS; }

/* agg_merge(engine, diff_a, diff_b, nm), blockmodel.cc:109-206.  Negative diffs split first (:110-117).  0 ok,
 * -1 sanity failure, -3 no progress possible (the reference would recurse without end or split nothing). */
int orc_agg_merge(orc_model *m, int diff_a, int diff_b, int nm) {
    while (diff_a < 0) {
        int rc = orc_agg_split(m, 0, nm);
        if (rc) return rc;
        diff_a++;
    }
    while (diff_b < 0) {
        int rc = orc_agg_split(m, 1, nm);
        if (rc) return rc;
        diff_b++;
    }
    for (int depth = 0; depth < 10000; ++depth) {
        if (diff_a + diff_b == 0) return 0;
        size_t K = m->K, first, count;
        if (diff_a > 0 && diff_b == 0) {
            first = 0;
            count = m->ka;
        } else if (diff_a == 0 && diff_b > 0) {
            first = m->ka;
            count = m->kb;
        } else {
            first = 0;
            count = K;
        }
        orc_block_move *moves = (orc_block_move *)malloc(sizeof(orc_block_move) * (size_t)nm * count + 1);
        orc_heap_item *heap = (orc_heap_item *)malloc(sizeof(orc_heap_item) * (size_t)nm * count + 1);
        size_t nq = propose_round(m, first, count, nm, moves, heap);
        unsigned char *in_e = (unsigned char *)calloc(K, 1);
        orc_groups G = {(size_t *)malloc(sizeof(size_t) * K * K), (size_t *)calloc(K, sizeof(size_t)), 0, K};
        int recurse = 0;
        size_t merged = 0;
        for (size_t qi = 0; qi < nq && diff_a + diff_b != 0; ++qi) {
            if (heap[qi].dS == INFINITY) { /* :163-168 */
                recurse = 1;
                break;
            }
            orc_block_move mv = moves[heap[qi].ii];
            if (mv.source < m->ka && diff_a != 0) {
                if (!(in_e[mv.source] && in_e[mv.target])) {
                    diff_a -= 1;
                    accept_move(&G, in_e, mv.source, mv.target);
                    ++merged;
                }
            } else if (mv.source >= m->ka && diff_b != 0) {
                if (!(in_e[mv.source] && in_e[mv.target])) {
                    diff_b -= 1;
                    accept_move(&G, in_e, mv.source, mv.target);
                    ++merged;
                }
            }
        }
        int rc = apply_block_moves(m, in_e, &G);
        free(moves);
        free(heap);
        free(in_e);
        free(G.group_of_slot);
        free(G.count);
        if (rc) return rc;
        if (!recurse) return 0;
        /* the reference recurses without end when the remaining budget asks for merges in a type that is down
         * to one block */
        if (merged == 0 && !((diff_a > 0 && m->ka > 1) || (diff_b > 0 && m->kb > 1))) return -3;
    }
    return -3;
}

/* agg_merge(engine, diff, nm), blockmodel.cc:208-271 (--nature) */
int orc_agg_merge_total(orc_model *m, int diff, int nm) {
    if (diff == 0) return 0;
    if (diff < 0) return -2;
    size_t K = m->K;
    orc_block_move *moves = (orc_block_move *)malloc(sizeof(orc_block_move) * (size_t)nm * K + 1);
    orc_heap_item *heap = (orc_heap_item *)malloc(sizeof(orc_heap_item) * (size_t)nm * K + 1);
    unsigned char *in_e = (unsigned char *)calloc(K, 1);
    orc_groups G = {(size_t *)malloc(sizeof(size_t) * K * K), (size_t *)calloc(K, sizeof(size_t)), 0, K};
    const int DIFF = diff;
    int minS = 1, rounds = 0;
    while (minS) {
        if (++rounds > 10000) {
            free(moves), free(heap), free(in_e), free(G.group_of_slot), free(G.count);
            return -3;
        }
        G.n_groups = 0;
        memset(in_e, 0, K);
        size_t nq = propose_round(m, 0, K, nm, moves, heap);
        for (size_t qi = 0; qi < nq && diff != 0; ++qi) {
            orc_block_move mv = moves[heap[qi].ii];
            if (!(in_e[mv.source] && in_e[mv.target])) {
                diff -= 1;
                accept_move(&G, in_e, mv.source, mv.target);
            }
            minS = heap[qi].dS == INFINITY;
        }
        diff = DIFF;
    }
    int rc = apply_block_moves(m, in_e, &G);
    free(moves), free(heap), free(in_e), free(G.group_of_slot), free(G.count);
    return rc;
}

/* geospace, support/util.hh:99-145 (ints).  Writes at most cap entries per side, returns the count. */
size_t orc_geospace(long start_a_in, long end_a_in, long start_b_in, long end_b_in, double ratio, int *out_a,
                    int *out_b, size_t cap) {
    if (ratio <= 1.) {
        out_a[0] = out_b[0] = 0;
        return 1;
    }
    int reverse = 0;
    int start_a = (int)start_a_in, end_a = (int)end_a_in, start_b = (int)start_b_in, end_b = (int)end_b_in;
    if (start_a - end_a < start_b - end_b) {
        start_a = (int)start_b_in, end_a = (int)end_b_in;
        start_b = (int)start_a_in, end_b = (int)end_a_in;
        reverse = 1;
    }
    int *ga = reverse ? out_b : out_a, *gb = reverse ? out_a : out_b;
    size_t n = 0, i = 0;
    int d = start_a;
    while (d > end_a && n + 1 < cap) {
        ga[n++] = d;
        i += 1;
        d = (int)floor(start_a / pow(ratio, (double)i));
    }
    ga[n++] = end_a;
    double r_ = pow((double)(start_b / end_b), 1. / (double)(n - 1)); /* integer division, :134 */
    for (size_t idx = 0; idx + 1 < n; ++idx) gb[idx] = (int)floor(start_b / pow(r_, (double)idx));
    gb[n - 1] = end_b;
    return n;
}
