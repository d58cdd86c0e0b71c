// The bucket map of a Float64 sort column (rust_dataframe_amd/csrc/rdf_sort_map.h) on the CPU: the very function the kernels
// run (os_map_value) under the very planner the host runs (os_plan_f64).  A radix sort over value buckets is correct iff the map
// is monotone — x <= y  =>  bucket(x) <= bucket(y) — and stays inside its 2^bits buckets; it is FAST iff the fullest bucket stays
// inside the LDS finish (<= 4096 rows).  Both are held here on columns the planner was written for and on ones written against it.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <limits>
#include <random>

#include "mini_test.hpp"
#include "../../rust_dataframe_amd/csrc/rdf_sort_map.h"

using namespace rdfk;

namespace {

struct Mapped { OsPlan plan; std::vector<uint32_t> bucket; uint32_t fullest = 0; };

// what os_column_passes does: a sample of 8192 keys at hashed places of equal strides, the plan, then every key's bucket
std::vector<double> sample_of(const std::vector<double>& col, int& outside) {
    const int64_t n = (int64_t)col.size();
    const int S = (int)std::min<int64_t>(8192, n);
    const int64_t stride = std::max<int64_t>(1, n / S);
    std::vector<double> xs;
    outside = 0;
    for (int i = 0; i < S; ++i) {
        uint64_t h = (uint64_t)i * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
        int64_t pos = (int64_t)i * stride + (int64_t)(h % (uint64_t)stride);
        if (pos >= n) pos = n - 1;
        const double x = col[(size_t)pos];
        if (std::isfinite(x)) xs.push_back(x); else ++outside;
    }
    return xs;
}
uint32_t bucket_of(double x, const OsPlan& p) {
    uint64_t b; memcpy(&b, &x, 8);
    return os_map_value(x, (b >> 63) != 0, p.fb, reinterpret_cast<const OsSegXY*>(p.segs.data()));
}
// total order of the sort: by the order-preserving key bits (negative NaNs first, then -inf ... +inf, positive NaNs last)
uint64_t key_bits(double x) { uint64_t b; memcpy(&b, &x, 8); return (b >> 63) ? ~b : (b ^ 0x8000000000000000ull); }

Mapped map_column(std::vector<double> col, bool expect_plan) {
    Mapped m;
    int outside = 0;
    std::vector<double> xs = sample_of(col, outside);
    double lo = -HUGE_VAL, hi = HUGE_VAL;
    {
        const auto mm = std::minmax_element(col.begin(), col.end(), [](double a, double b) { return key_bits(a) < key_bits(b); });
        if (*mm.first == *mm.first && *mm.second == *mm.second) { lo = *mm.first; hi = *mm.second; }
    }
    os_plan_f64(xs.data(), xs.size(), outside, (int64_t)col.size(), lo, hi, 0, m.plan);
    CHECK_EQ(m.plan.sampled, expect_plan);
    if (!m.plan.sampled) return m;
    CHECK(m.plan.fb.bits >= 12 && m.plan.fb.bits <= 24);
    const uint32_t nb = 1u << m.plan.fb.bits;
    if (!m.plan.fb.flat) {     // the shares tile [tail, nb - tail) exactly
        uint32_t next = (uint32_t)m.plan.fb.tail;
        for (const OsSeg& s : m.plan.segs) { CHECK_EQ(s.base, next); CHECK(s.share >= 1); next += s.share; }
        CHECK_EQ(next, nb - (uint32_t)m.plan.fb.tail);
    }
    std::sort(col.begin(), col.end(), [](double a, double b) { return key_bits(a) < key_bits(b); });
    m.bucket.resize(col.size());
    std::vector<uint32_t> fill(nb, 0);
    uint32_t prev = 0;
    for (size_t i = 0; i < col.size(); ++i) {
        const uint32_t k = bucket_of(col[i], m.plan);
        CHECK(k < nb);
        CHECK(k >= prev);                       // monotone over the whole sorted column
        prev = k;
        m.bucket[i] = k;
        m.fullest = std::max(m.fullest, ++fill[k]);
    }
    return m;
}

std::vector<double> draw(size_t n, unsigned seed, const std::function<double(std::mt19937_64&)>& f) {
    std::mt19937_64 g(seed);
    std::vector<double> v(n);
    for (double& x : v) x = f(g);
    return v;
}
const size_t N = 2000000;

}  // namespace

TEST(uniform_columns_get_one_linear_map_between_the_tails) {
    std::uniform_real_distribution<double> u(0.0, 1.0);
    Mapped m = map_column(draw(N, 1, [&](std::mt19937_64& g) { return u(g); }), true);
    CHECK(m.plan.fb.flat == 1);
    CHECK(m.fullest <= 512);                  // its size class: the smallest
    std::uniform_real_distribution<double> w(-3e9, 7e9);
    m = map_column(draw(N, 2, [&](std::mt19937_64& g) { return w(g); }), true);
    CHECK(m.plan.fb.flat == 1 && m.fullest <= 512);
}

TEST(bell_shaped_and_heavy_tailed_columns_fill_their_buckets_evenly) {
    std::normal_distribution<double> nd(7.0, 1e3);
    Mapped m = map_column(draw(N, 3, [&](std::mt19937_64& g) { return nd(g); }), true);
    CHECK(m.plan.fb.flat == 0);
    CHECK(m.fullest <= 1500);                 // planned for <= 1000 expected (at 2e6 rows a bucket is two sample values wide); buckets over [min, max] put 4-5 x the average in the densest
    std::lognormal_distribution<double> ln(0.0, 1.0);
    m = map_column(draw(N, 4, [&](std::mt19937_64& g) { return ln(g); }), true);
    CHECK(m.fullest <= 1500);
    std::exponential_distribution<double> ex(1.0);
    m = map_column(draw(N, 5, [&](std::mt19937_64& g) { return ex(g); }), true);
    CHECK(m.fullest <= 1500);
    std::student_t_distribution<double> st(2.0);         // tails far beyond the sample's range: the geometric buckets take them
    m = map_column(draw(N, 6, [&](std::mt19937_64& g) { return st(g); }), true);
    CHECK(m.fullest <= 4096);
    std::cauchy_distribution<double> ca(0.0, 1.0);       // no moments at all
    m = map_column(draw(N, 7, [&](std::mt19937_64& g) { return ca(g); }), true);
    CHECK(m.fullest <= 4096);
}

TEST(a_few_far_outliers_infinities_and_nans_leave_the_plan_alone) {
    std::normal_distribution<double> nd(100.0, 3.0);
    std::vector<double> col = draw(N, 8, [&](std::mt19937_64& g) { return nd(g); });
    const double inf = std::numeric_limits<double>::infinity(), nan = std::numeric_limits<double>::quiet_NaN();
    const double specials[] = {inf, -inf, nan, -nan, 1e300, -1e300, 1e12, -4e9, 5e-324, -0.0, 0.0, std::numeric_limits<double>::max(), std::numeric_limits<double>::lowest()};
    std::mt19937_64 g(9);
    for (int i = 0; i < 40; ++i) col[g() % N] = specials[i % 13];
    Mapped m = map_column(col, true);
    CHECK(m.fullest <= 1500);
    // every special lands in a tail, on its own side
    const uint32_t nb = 1u << m.plan.fb.bits, T = (uint32_t)m.plan.fb.tail;
    CHECK(bucket_of(inf, m.plan) == nb - 1 && bucket_of(nan, m.plan) == nb - 1 && bucket_of(-inf, m.plan) == 0 && bucket_of(-nan, m.plan) == 0);
    CHECK(bucket_of(1e300, m.plan) >= nb - T && bucket_of(-1e300, m.plan) < T && bucket_of(1e12, m.plan) >= nb - T && bucket_of(1e12, m.plan) <= bucket_of(1e300, m.plan) &&
          bucket_of(m.plan.fb.hi + 3.0, m.plan) >= nb - T && bucket_of(m.plan.fb.hi + 3.0, m.plan) < bucket_of(1e12, m.plan));     // (16 doublings of the range's half width per 128 tail buckets)
}

TEST(columns_no_map_can_split_are_refused_before_a_pass_is_spent) {
    std::uniform_real_distribution<double> u(0.0, 1.0);
    std::vector<double> col = draw(N, 10, [&](std::mt19937_64& g) { return u(g); });
    std::mt19937_64 g(11);
    for (size_t i = 0; i < N; ++i) if (g() % 20 == 0) col[i] = 0.25;             // 5 % of the rows hold one value
    map_column(col, false);
    std::exponential_distribution<double> ex(1.0);
    map_column(draw(N, 12, [&](std::mt19937_64& g2) { const double e = ex(g2); return e * e * e * e * e * e; }), false);   // concentrates however far one zooms in
    col = draw(N, 13, [&](std::mt19937_64& g2) { return std::floor(u(g2) * 1000.0); });                                   // 1000 distinct values
    map_column(col, false);
    col = draw(N, 14, [&](std::mt19937_64& g2) { return u(g2); });
    for (size_t i = 0; i < N; ++i) if (g() % 50 == 0) col[i] = std::numeric_limits<double>::infinity();                   // 2 % infinities
    map_column(col, false);
    map_column(std::vector<double>(N, 3.5), false);                                                                         // one value
}

TEST(the_map_is_monotone_across_every_seam) {
    // around every segment boundary, the range's two ends and the tails' bucket edges: neighbours in ulps, in order
    std::normal_distribution<double> nd(-40.0, 0.01);
    Mapped m = map_column(draw(N, 15, [&](std::mt19937_64& g) { return nd(g); }), true);
    const OsBucket& f = m.plan.fb;
    std::vector<double> probes;
    for (int c = 0; c <= f.nseg; ++c) {
        double x = f.lo + (double)c / f.scale;
        for (int k = 0; k < 6; ++k) x = std::nextafter(x, -HUGE_VAL);
        for (int k = 0; k < 12; ++k) { probes.push_back(x); x = std::nextafter(x, HUGE_VAL); }
    }
    for (double d = 1e-12; d < 1e300; d *= 1.7) { probes.push_back(f.hi + d); probes.push_back(f.lo - d); }
    std::sort(probes.begin(), probes.end());
    uint32_t prev = 0;
    for (double x : probes) { const uint32_t k = bucket_of(x, m.plan); CHECK(k >= prev); CHECK(k < (1u << f.bits)); prev = k; }
}

int main() { return run_all(); }
