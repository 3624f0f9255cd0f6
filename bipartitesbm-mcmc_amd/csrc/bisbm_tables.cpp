// bisbm_tables.cpp -- host-built numeric tables of the engine (plain C++, no HIP): lgamma / log / log_q as the reference
// builds them at construction, and the temperature tables of the pow / log schedules (glibc values, the reference's own).
//
// Reference lines cited as <file>:<line> relative to /root/reference/src.
#include "bisbm_engine.hpp"

namespace bisbm {

namespace {

std::mutex g_tab_mu;
std::map<std::pair<uint64_t, uint32_t>, std::shared_ptr<HostTables>> g_tab_cache;

double log_sum(double a, double b) {  // int_part.cc:30-32
    return std::max(a, b) + std::log1p(std::exp(-std::fabs(a - b)));
}

void fill_lgamma(std::vector<double>& lg, std::vector<double>& lo) {  // cache.cc:64-79, :25-37
    const size_t n = lg.size();
    lg[0] = INFINITY;
    lo[0] = 0.0;
    unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    if (n < (1u << 16)) nt = 1;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) {
        th.emplace_back([&, t] {
            int sign;
            for (size_t i = 1 + t; i < n; i += nt) {
                lg[i] = lgamma_r((double)i, &sign);
                lo[i] = std::log((double)i);
            }
        });
    }
    for (auto& x : th) x.join();
}

// int_part.cc:34-51 with the columns cut at kcap (column k only depends on columns <= k).  Cells
// the reference never writes (k > n) stay -inf and are read as such by the recurrence.
// Rows are processed in blocks: inside a block columns < B are filled row by row (a cell (n,k)
// reads (n-k,k), which can lie in the same block only when k < B); the remaining columns of the
// block's rows are independent of each other and are split over threads.  The evaluation order per
// cell is unchanged, so the values equal the serial recurrence bit for bit.
void fill_q(std::vector<double>& q, uint32_t kcap) {
    const size_t stride = (size_t)kcap + 1;
    std::fill(q.begin(), q.end(), -INFINITY);
    const size_t B = 128;
    auto cell = [&](size_t n, size_t k) {
        double* row = q.data() + n * stride;
        row[k] = log_sum(row[k], row[k - 1]);
        if (n > k) row[k] = log_sum(row[k], q[(n - k) * stride + k]);
    };
    unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    if (kcap < 2048) nt = 1;
    for (size_t n0 = 1; n0 <= (size_t)kQNmax; n0 += B) {
        const size_t n1 = std::min<size_t>(n0 + B, (size_t)kQNmax + 1);
        for (size_t n = n0; n < n1; ++n) {
            q[n * stride + 1] = 0;
            const size_t kmax = std::min<size_t>(std::min<size_t>(n, kcap), B - 1);
            for (size_t k = 2; k <= kmax; ++k) cell(n, k);
        }
        auto rest = [&](size_t lo, size_t hi) {
            for (size_t n = lo; n < hi; ++n) {
                const size_t kmax = std::min<size_t>(n, kcap);
                for (size_t k = B; k <= kmax; ++k) cell(n, k);
            }
        };
        if (nt == 1 || n1 <= B) {
            rest(n0, n1);
        } else {
            std::vector<std::thread> th;
            const size_t per = (n1 - n0 + nt - 1) / nt;
            for (unsigned t = 0; t < nt; ++t) {
                const size_t lo = n0 + t * per, hi = std::min(n1, lo + per);
                if (lo < hi) th.emplace_back(rest, lo, hi);
            }
            for (auto& x : th) x.join();
        }
    }
}

}  // namespace

std::shared_ptr<HostTables> get_tables(uint64_t lg_size, uint32_t kcap) {
    std::lock_guard<std::mutex> lk(g_tab_mu);
    for (auto& kv : g_tab_cache)
        if (kv.first.first >= lg_size && kv.first.second >= kcap && kv.first.second <= 2 * kcap + 64 &&
            kv.first.first <= 2 * lg_size + 4096)
            return kv.second;
    auto t = std::make_shared<HostTables>();
    t->lg.resize(lg_size);
    t->lo.resize(lg_size);
    fill_lgamma(t->lg, t->lo);
    t->kcap = kcap;
    t->q.resize((size_t)(kQNmax + 1) * ((size_t)kcap + 1));
    fill_q(t->q, kcap);
    if (g_tab_cache.size() > 4) g_tab_cache.clear();
    g_tab_cache[{lg_size, kcap}] = t;
    return t;
}

double h_lgamma_fast(const HostTables& t, uint64_t x) {  // cache.hh:82-93
    if (x < t.lg.size()) return t.lg[x];
    if (x == 0) return INFINITY;
    int sign;
    return lgamma_r((double)x, &sign);
}

double h_lbinom_fast(const HostTables& t, uint64_t N, uint64_t k) {  // util.hh:41-47
    if (N == 0 || k == 0 || k > N) return 0;
    return (h_lgamma_fast(t, N + 1) - h_lgamma_fast(t, k + 1)) - h_lgamma_fast(t, N - k + 1);
}

// metropolis_hasting.cc:10-13,20-23 evaluated with the host libm (the reference's own values) for steps t0 .. t0 + len - 1 of a
// call, on up to 16 threads.  The exponential schedule's table ends with its first exact zero (pow has underflowed: it is
// monotone for 0 <= kw1 < 1, so every later step is 0 as well; *zero_after says so).
std::vector<double> schedule_table(int schedule, float kw0, float kw1, uint64_t t0, uint64_t len, int* zero_after) {
    std::vector<double> T((size_t)len);
    *zero_after = 0;
    auto fill = [&](uint64_t lo, uint64_t hi) {
        for (uint64_t j = lo; j < hi; ++j) {
            const uint64_t t = t0 + j;
            if (schedule == SCHED_EXPONENTIAL) {
                T[j] = (double)kw0 * std::pow((double)kw1, (double)t);
            } else {
                const float x = (float)t + kw1;
                const size_t i = (size_t)x;
                T[j] = (double)kw0 / (i == 0 ? 0. : std::log((double)i));
            }
        }
    };
    const unsigned nt = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency())), len >> 14));
    if (nt <= 1) {
        fill(0, len);
    } else {
        std::vector<std::thread> th;
        for (unsigned i = 0; i < nt; ++i) th.emplace_back(fill, len * i / nt, len * (i + 1) / nt);
        for (auto& x : th) x.join();
    }
    if (schedule == SCHED_EXPONENTIAL && kw1 < 1.f && kw1 >= 0.f)
        for (uint64_t j = 0; j < len; ++j)
            if (T[j] == 0.) {
                T.resize((size_t)j + 1);
                *zero_after = 1;
                break;
            }
    return T;
}

}  // namespace bisbm
