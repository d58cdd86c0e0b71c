// rdf_sort_map.h — the bucket map of a Float64 sort column (rdf_sort.hip, os_column_passes in rdf_capi.cpp): value -> bucket,
// monotone, planned from a sample of the keys.  Shared by the kernels (device), the host planner, and a CPU test that holds the
// map to its one obligation — x <= y  =>  bucket(x) <= bucket(y), for every plan and every double — on the code the device runs
// (tests/cpp/test_sort_map.cpp).  No HIP types in here.
#pragma once
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <cmath>
#include <vector>

#if defined(__HIPCC__)
#define RDF_SORT_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define RDF_SORT_HD inline
#endif

namespace rdfk {

// One segment of the piecewise-linear part: the value range is cut into nseg equal-width segments, segment c owns the buckets
// base .. base + share - 1, as many as its share of the sample asks for (an equi-depth map: bell-shaped and heavy-tailed columns
// fill their buckets as evenly as uniform ones).  bucket = base + floor(frac(t) * share), t = (x - lo) * scale.
struct OsSeg { uint32_t base, share; };
struct OsBucket {
    double lo, scale; int32_t bits, flip;            // flip: the stored keys are ~(key bits) (descending)
    // tail == 0: one linear map, bucket = floor((x - lo) * scale), clamped.  Else (x - lo) * scale = the segment of [lo, hi)
    // the key lies in, the first / last `tail` buckets take the keys below lo / from hi on in GEOMETRIC steps (eight buckets per
    // doubling of the distance from the range, in units of 1 / tinv): far outliers, infinities and the thin ends of heavy-tailed
    // columns spread over them instead of piling up in one end bucket
    const OsSeg* seg; int32_t nseg, tail; double hi, tinv;
    int32_t flat, pad; double flat_scale;            // flat: every segment has the same share — bucket = tail + floor(t * flat_scale), no table
};
constexpr int kOsSegs = 256;                         // segments of a plan (the kernels keep the table in LDS)

RDF_SORT_HD uint64_t os_bits_of(double v) {
    return __builtin_bit_cast(uint64_t, v);
}

// The bucket of value x (before `flip`); negative = the sign bit of the value's bit pattern (where a NaN goes: NaNs order
// below / above everything as their bits do).  segs: the plan's table (LDS on the device), unused when flat or tail == 0.
template <class SegPtr>
RDF_SORT_HD uint32_t os_map_value(double x, bool negative, const OsBucket& f, SegPtr segs) {
    const double t = (x - f.lo) * f.scale;
    const uint32_t top = (1u << f.bits) - 1;
    if (t != t) return negative ? 0u : top;
    if (!f.tail) return t <= 0.0 ? 0u : (t >= (double)top ? top : (uint32_t)t);
    // t is monotone in x (a subtraction of and a multiplication by constants round monotonically); its integer part picks the
    // segment, its fraction the bucket among the segment's own
    const uint32_t T = (uint32_t)f.tail;
    if (t < 0.0 || t >= (double)f.nseg) {
        // a tail: y = 1 + distance from the range; the bits of a double >= 1 are a piecewise-linear log2 of it
        const bool low = t < 0.0;
        double y = (low ? f.lo - x : x - f.hi) * f.tinv + 1.0;
        y = y >= 1.0 ? y : 1.0;                           // (a key the rounding of t put outside by an ulp)
        const uint64_t g64 = (os_bits_of(y) - 0x3FF0000000000000ull) >> 49;
        const uint32_t g = g64 < (uint64_t)(T - 1) ? (uint32_t)g64 : T - 1;
        return low ? T - 1 - g : top - (T - 1) + g;
    }
    if (f.flat) {                                         // the sample found the column evenly spread: one linear map between the tails
        const uint32_t mid = top + 1 - 2 * T;
        const uint32_t w = (uint32_t)(t * f.flat_scale);
        return T + (w < mid ? w : mid - 1);
    }
    const int c = (int)t;
    const uint32_t sbase = segs[c].x, share = segs[c].y;
    const uint32_t w = (uint32_t)((t - (double)c) * (double)share);          // the place inside the segment: frac(t), monotone in t
    return sbase + (w < share ? w : share - 1);
}

// ---- the host's plan ---------------------------------------------------------------------------------------------------------
// From a sample of the column (xs: its finite values; outside: how many sampled keys were infinite / NaN) and the key range's
// ends [lo, hi] (-inf / +inf when not finite): the range the ROWS lie in — extended by half its width on either side (the tails
// of 1e9 draws of a bell curve reach 1.6 x as far as those of 8192); beyond it, the geometric tails —, the share of each of 256
// equal-width segments of it, the bucket bits, and the warnings that no map helps: a value met twice (held by thousands of rows),
// a region that keeps concentrating however far one zooms in (x^6 of an exponential), infinities / NaNs by the thousand.
struct OsSegXY { uint32_t x, y; };                       // what os_map_value reads: {base, share}
struct OsPlan {
    bool sampled = false;            // false: no value map for this column (the byte passes sort any order)
    int bits = 0, most_equal = 1;
    double within = 1.0, smin = 0.0, smax = 0.0;
    OsBucket fb;                     // seg left null: the caller uploads `segs` unless fb.flat
    std::vector<OsSeg> segs;
};
inline void os_plan_f64(const double* xs, size_t nxs, int outside, int64_t n, double lo, double hi, int flip, OsPlan& p) {
    p = OsPlan();
    memset(&p.fb, 0, sizeof p.fb);
    const size_t taken = nxs + (size_t)outside;
    if (nxs < 1024 || taken == 0) return;
    const double rows_per_sample = (double)n / (double)taken;
    {   // the same value twice in the sample = a value held by ~2 n / S rows, which no bucket bits can split
        std::vector<uint64_t> seen(16384, 0);
        std::vector<uint16_t> times(16384, 0);
        for (size_t i = 0; i < nxs; ++i) {
            const uint64_t k = os_bits_of(xs[i]);
            size_t h = (size_t)((k * 0x9E3779B97F4A7C15ull) >> 50);
            while (times[h] && seen[h] != k) h = (h + 1) & 16383;
            seen[h] = k;
            if (times[h] < 65535) ++times[h];
            p.most_equal = std::max<int>(p.most_equal, times[h]);
        }
    }
    if (p.most_equal >= 2 && (double)p.most_equal * rows_per_sample > 3000.0) return;
    if ((double)outside * rows_per_sample > 1500.0) return;       // the non-finite keys end in the last buckets of the tails: a few fit
    double smin = HUGE_VAL, smax = -HUGE_VAL;
    for (size_t i = 0; i < nxs; ++i) { smin = std::min(smin, xs[i]); smax = std::max(smax, xs[i]); }
    p.smin = smin; p.smax = smax;
    if (!(smax > smin)) return;
    const double ext = 0.5 * (smax - smin);
    const double lo2 = std::max(lo, smin - ext), hi2 = std::min(hi, smax + ext);
    const int nseg = kOsSegs;
    const double cs = (double)nseg / (hi2 - lo2);
    if (!std::isfinite(cs) || !std::isfinite(1.0 / ext)) return;
    std::vector<int> cnt((size_t)nseg, 0);
    auto seg_of = [&](double x) { return std::min(nseg - 1, std::max(0, (int)((x - lo2) * cs))); };
    for (size_t i = 0; i < nxs; ++i) ++cnt[(size_t)seg_of(xs[i])];
    // how uneven the fullest segment is inside: 8 sub-bins, again inside the fullest of those, ... while the sample still has
    // 512 values there (smooth columns never do: 32 per segment on average)
    double within = 1.0;
    {
        const int at0 = (int)(std::max_element(cnt.begin(), cnt.end()) - cnt.begin());
        double blo = lo2 + at0 / cs, bhi = lo2 + (at0 + 1) / cs;
        std::vector<double> cur, nxt;
        for (size_t i = 0; i < nxs; ++i) if (seg_of(xs[i]) == at0) cur.push_back(xs[i]);
        for (int level = 0; level < 6 && cur.size() >= 512; ++level) {
            int c8[8] = {0};
            const double s8 = 8.0 / (bhi - blo);
            if (!std::isfinite(s8)) { within = 1e30; break; }
            auto sub = [&](double x) { return std::min(7, std::max(0, (int)((x - blo) * s8))); };
            for (double x : cur) ++c8[sub(x)];
            const int at = (int)(std::max_element(c8, c8 + 8) - c8);
            within *= 8.0 * (double)c8[at] / (double)cur.size();
            nxt.clear();
            for (double x : cur) if (sub(x) == at) nxt.push_back(x);
            cur.swap(nxt);
            const double w = (bhi - blo) / 8.0;
            blo += at * w; bhi = blo + w;
        }
    }
    p.within = within;
    // bucket bits: the fewest that leave <= 1000 rows expected in the fullest bucket, with 30 % for the noise of a segment's count
    // (32 sample values on average) and 30 % for the slope inside a segment; 1/16 of the buckets are the two geometric tails.
    // (Measured at 5e7 bell-shaped keys: 17 bits with a tenth of the buckets in the 1024-row LDS class sort in 3.65 ms, 18 bits
    // with every bucket <= 512 rows in 3.83.)  Up to 2500 expected rows are accepted at 24 bits.
    auto fullest = [&](int bits_) { return (double)n / ((double)((int64_t)1 << bits_) * (15.0 / 16.0)) * 1.69 * within; };
    int bits = 12;
    while (bits < 24 && fullest(bits) > 1000.0) ++bits;
    p.bits = bits;
    if (fullest(bits) > 2500.0) return;
    p.sampled = true;
    int B = bits;
    const int64_t T = ((int64_t)1 << B) / 32;
    const int64_t nb = ((int64_t)1 << B) - 2 * T;
    p.segs.assign((size_t)nseg, OsSeg{0, 0});
    // shares by an UPPER estimate of a segment's rows, count + 2 sqrt(count) + 2: a segment that drew 2 of the sample's values
    // may well hold the rows of 8 (the thin ends of a bell curve are many such segments, and the fullest bucket is the maximum
    // over all of them), one that drew 100 holds the rows of 80 .. 120; the margins beyond the sample's extremes are not empty
    // either.  At least one bucket each, the rounding's remainder dealt to the fullest segments.
    std::vector<double> weight((size_t)nseg);
    double total = 0.0;
    for (int c = 0; c < nseg; ++c) { const double k = (double)cnt[(size_t)c]; weight[(size_t)c] = k + 2.0 * std::sqrt(k) + 2.0; total += weight[(size_t)c]; }
    std::vector<int64_t> share((size_t)nseg);
    int64_t given = 0;
    for (int c = 0; c < nseg; ++c) { share[(size_t)c] = std::max<int64_t>(1, (int64_t)((double)nb * weight[(size_t)c] / total)); given += share[(size_t)c]; }
    std::vector<int> order((size_t)nseg);
    for (int c = 0; c < nseg; ++c) order[(size_t)c] = c;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return cnt[(size_t)a] > cnt[(size_t)b]; });
    for (int r = 0; given != nb; r = (r + 1) % nseg) {
        const int c = order[(size_t)r];
        if (given < nb) { ++share[(size_t)c]; ++given; }
        else if (share[(size_t)c] > 1) { --share[(size_t)c]; --given; }
    }
    int64_t base = T;
    for (int c = 0; c < nseg; ++c) {
        p.segs[(size_t)c].base = (uint32_t)base;
        p.segs[(size_t)c].share = (uint32_t)share[(size_t)c];
        base += share[(size_t)c];
    }
    OsBucket& fb = p.fb;
    fb.lo = lo2; fb.scale = cs; fb.bits = B; fb.flip = flip; fb.seg = nullptr; fb.nseg = nseg;
    fb.hi = hi2; fb.tail = (int32_t)T; fb.tinv = 1.0 / ext;
    // evenly spread between the sample's extremes (chi-square over the segments there, 3 sigma)?  Then no table: one linear map
    // that hugs the sample's range (no shares to starve the empty margins of buckets) — uniform columns keep the cost of one map
    const int c0 = seg_of(smin) + 1, c1 = seg_of(smax) - 1;
    if (c1 - c0 >= 16) {
        double tot = 0.0, chi = 0.0;
        for (int c = c0; c <= c1; ++c) tot += cnt[(size_t)c];
        const double e = tot / (double)(c1 - c0 + 1);
        for (int c = c0; c <= c1; ++c) chi += ((double)cnt[(size_t)c] - e) * ((double)cnt[(size_t)c] - e) / e;
        const double dof = (double)(c1 - c0);
        if (e >= 8.0 && chi < dof + 3.0 * std::sqrt(2.0 * dof)) {
            const double m = 0.02 * (smax - smin), lo3 = std::max(lo, smin - m), hi3 = std::min(hi, smax + m);
            fb.flat = 1;
            fb.lo = lo3; fb.hi = hi3; fb.scale = (double)nseg / (hi3 - lo3);
            // Poisson noise only: the fewest bits with avg + 5 sqrt(avg) <= 512 rows (the smallest class of the LDS finish)
            B = 12;
            for (;; ++B) {
                const double avg = (double)n / ((double)((int64_t)1 << B) * (15.0 / 16.0));
                if (avg + 5.0 * std::sqrt(avg) <= 512.0 || B == 24) break;
            }
            const int64_t T2 = ((int64_t)1 << B) / 32;
            fb.bits = B; fb.tail = (int32_t)T2; fb.flat_scale = (double)(((int64_t)1 << B) - 2 * T2) / (double)nseg;
            p.bits = B;
        }
    }
}

}  // namespace rdfk
