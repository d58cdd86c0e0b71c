// rdf_common.hip.h — device-side helpers shared by the kernel translation units.
#pragma once
#include "rdf_device.h"

namespace rdfk {

__device__ __forceinline__ uint64_t uniform64(uint64_t v) {
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
// wave index inside the block as a scalar (the compiler cannot prove threadIdx.x >> 6 uniform)
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// Pointers read out of chunk tables in memory are generic ("flat") to the compiler: flat loads count in lgkmcnt
// as well as vmcnt, so the next scalar-load wait (s_waitcnt lgkmcnt(0)) also drains every vector load in flight.
// Everything the ABI hands us lives in global memory: say so.
template <class T> using GlobalPtr = const T __attribute__((address_space(1)))*;
template <class T> using GlobalMutPtr = T __attribute__((address_space(1)))*;
template <class T> __device__ __forceinline__ GlobalPtr<T> as_global(const void* p) { return (GlobalPtr<T>)p; }
// Host-built lookup tables (chunk descriptors, tile prefix sums) are never written by a kernel: read through the
// CONSTANT address space they become scalar loads (s_load, lgkmcnt) even in kernels that also store / do atomics.  As
// plain global loads the compiler has to assume a clobber and issues them on the vector memory path, where waiting for a
// descriptor (vmcnt(0)) also waits for every data load in flight — the prefetch pipeline of the streaming kernels then
// degenerates to one exposed memory latency per batch.
template <class T> using ConstPtr = const T __attribute__((address_space(4)))*;
template <class T> __device__ __forceinline__ ConstPtr<T> as_const(const void* p) { return (ConstPtr<T>)p; }
__device__ __forceinline__ DevChunkCol const_col(const DevChunkCol* table, int64_t i) {   // table[i] through the constant address space
    const ConstPtr<DevChunkCol> t = as_const<DevChunkCol>(table);
    DevChunkCol c;
    c.values = t[i].values; c.validity = t[i].validity; c.offset = t[i].offset;
    return c;
}
template <class T> __device__ __forceinline__ GlobalMutPtr<T> as_global_mut(void* p) { return (GlobalMutPtr<T>)p; }

__device__ __forceinline__ int clamp64(int64_t v) { return v <= 0 ? 0 : (v >= 64 ? 64 : (int)v); }

// NW consecutive 64-bit windows of an LSB-first bitmap starting at bit `bitpos` (wave-uniform), rows
// past `nbits` cleared.  Scalar-load variant (the specialised kernels' default):
// all NW+1 aligned words are fetched with independent scalar loads (indices
// clamped to the last word that holds a requested bit, so nothing outside the ABI's "readable to the
// next 8-byte boundary" is touched) and funnel-shifted on the scalar unit: no branches between the
// loads, no VALU work.
template <int NW>
__device__ __forceinline__ void load_windows_s(const uint8_t* base, int64_t bitpos, int64_t nbits, uint64_t (&win)[NW]) {
    if (nbits <= 0) {
#pragma unroll
        for (int j = 0; j < NW; ++j) win[j] = 0;
        return;
    }
    const uint64_t addr = uniform64((uint64_t)(uintptr_t)base + (uint64_t)(bitpos >> 3));
    const GlobalPtr<uint64_t> w = (GlobalPtr<uint64_t>)(uintptr_t)(addr & ~7ull);
    const int sh = __builtin_amdgcn_readfirstlane((int)(addr & 7) * 8 + (int)(bitpos & 7));
    const int64_t want = nbits < (int64_t)64 * NW ? nbits : (int64_t)64 * NW;
    const int last = __builtin_amdgcn_readfirstlane((int)((sh + want - 1) >> 6));  // index of the last needed word
    uint64_t word[NW + 1];
#pragma unroll
    for (int i = 0; i <= NW; ++i) word[i] = w[i < last ? i : last];
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        uint64_t r = word[j] >> sh;
        if (sh) r |= word[j + 1] << (64 - sh);
        const int64_t left = nbits - (int64_t)64 * j;
        if (left < 64) r = left <= 0 ? 0 : (r & ((1ull << left) - 1));
        win[j] = r;
    }
}

// The common case of the above: all 64 * NW rows exist (a full tile).  No end-of-chunk masks — as 64-bit shifts and selects
// on wave-uniform values they were ~100 vector-ALU instructions per tile (SALU has no 64-bit relational compare, so the compiler
// moves such arithmetic to the vector unit), a quarter of what the headline kernel may spend per 512-row tile at the HBM rate —,
// NW consecutive words in fixed places and the one word beyond them only when the window is not word-aligned.
template <int NW>
__device__ __forceinline__ void load_windows_full_s(const uint8_t* base, int64_t bitpos, uint64_t (&win)[NW]) {
    const uint64_t addr = uniform64((uint64_t)(uintptr_t)base + (uint64_t)(bitpos >> 3));
    const GlobalPtr<uint64_t> w = (GlobalPtr<uint64_t>)(uintptr_t)(addr & ~7ull);
    const int sh = __builtin_amdgcn_readfirstlane((int)(addr & 7) * 8 + (int)(bitpos & 7));
    uint64_t word[NW + 1];
#pragma unroll
    for (int i = 0; i < NW; ++i) word[i] = w[i];
    word[NW] = w[sh ? NW : NW - 1];
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        uint64_t r = word[j] >> sh;
        if (sh) r |= word[j + 1] << (64 - sh);
        win[j] = r;
    }
}

// Same result, but the aligned words travel through the VECTOR memory path (one global_load_dwordx2 by lanes 0..NW, then
// v_readlane into scalars).  With block-wide tiles (round 1) this was the faster way to stream a bitmap every wave touches
// once (0.80 of peak vs 0.69-0.74 with scalar loads); with wave-granular tiles and the next tile located under the
// current tile's loads the scalar variant is level or ahead (1.30 vs 1.355 ms per 1e9 rows) and is the default; kept as
// the A/B (rdf_set_option("vec_bitmap", 1)) and for the kernels that have every lane busy anyway.
// Must be called with all lanes of the wave active.
template <int NW>
__device__ __forceinline__ void load_windows(const uint8_t* base, int64_t bitpos, int64_t nbits, uint64_t (&win)[NW]) {
    static_assert(NW < 64, "one lane per word");
    if (nbits <= 0) {
#pragma unroll
        for (int j = 0; j < NW; ++j) win[j] = 0;
        return;
    }
    const uint64_t addr = uniform64((uint64_t)(uintptr_t)base + (uint64_t)(bitpos >> 3));
    const GlobalPtr<uint64_t> w = (GlobalPtr<uint64_t>)(uintptr_t)(addr & ~7ull);
    const int sh = __builtin_amdgcn_readfirstlane((int)(addr & 7) * 8 + (int)(bitpos & 7));
    const int64_t want = nbits < (int64_t)64 * NW ? nbits : (int64_t)64 * NW;
    const int last = __builtin_amdgcn_readfirstlane((int)((sh + want - 1) >> 6));
    const int lane = threadIdx.x & 63;
    uint64_t mine = 0;
    if (lane <= NW) mine = w[lane < last ? lane : last];
    uint64_t word[NW + 1];
#pragma unroll
    for (int i = 0; i <= NW; ++i) {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mine, i);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mine >> 32), i);
        word[i] = ((uint64_t)hi << 32) | lo;
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        uint64_t r = word[j] >> sh;
        if (sh) r |= word[j + 1] << (64 - sh);
        const int64_t left = nbits - (int64_t)64 * j;
        if (left < 64) r = left <= 0 ? 0 : (r & ((1ull << left) - 1));
        win[j] = r;
    }
}

// single window (kept for call sites that need just one)
__device__ __forceinline__ uint64_t load_bits64(const uint8_t* base, int64_t bitpos, int nbits) {
    uint64_t w[1];
    load_windows<1>(base, bitpos, nbits, w);
    return w[0];
}

__device__ __forceinline__ uint64_t shfl_xor64(uint64_t v, int m) {
    uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, m);
    uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), m);
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ double u2d(uint64_t v) { return __longlong_as_double((long long)v); }
__device__ __forceinline__ uint64_t d2u(double v) { return (uint64_t)__double_as_longlong(v); }
__device__ __forceinline__ float u2f(uint64_t v) { return __uint_as_float((uint32_t)v); }
__device__ __forceinline__ uint64_t f2u(float v) { return (uint64_t)__float_as_uint(v); }

// Integers live in the accumulator sign-/zero-extended to 64 bits.
__device__ __forceinline__ uint64_t normalize_int(int dt, uint64_t x) {
    switch (dt) {
        case RDF_I8: return (uint64_t)(int64_t)(int8_t)x;
        case RDF_I16: return (uint64_t)(int64_t)(int16_t)x;
        case RDF_I32: return (uint64_t)(int64_t)(int32_t)x;
        case RDF_U8: return x & 0xFFull;
        case RDF_U16: return x & 0xFFFFull;
        case RDF_U32: return x & 0xFFFFFFFFull;
        default: return x;
    }
}
__device__ __forceinline__ bool dt_is_signed(int dt) { return dt <= RDF_I64; }

// ---- aggregate combine by class (F64: ieee add / NaN-ignoring min,max; ints: wrapping add) ----
__device__ __forceinline__ void agg_init(int cls, uint64_t& sum, uint64_t& mn, uint64_t& mx, int64_t& cnt) {
    cnt = 0;
    if (cls == CLS_F64) { sum = d2u(0.0); mn = mx = 0x7FF8000000000000ull; }
    else if (cls == CLS_SIGNED) { sum = 0; mn = (uint64_t)INT64_MAX; mx = (uint64_t)INT64_MIN; }
    else { sum = 0; mn = ~0ull; mx = 0; }
}
__device__ __forceinline__ void agg_merge(int cls, uint64_t& sum, uint64_t& mn, uint64_t& mx, int64_t& cnt,
                                          uint64_t s2, uint64_t mn2, uint64_t mx2, int64_t c2) {
    cnt += c2;
    if (cls == CLS_F64) {
        sum = d2u(u2d(sum) + u2d(s2));
        mn = d2u(fmin(u2d(mn), u2d(mn2)));
        mx = d2u(fmax(u2d(mx), u2d(mx2)));
    } else if (cls == CLS_SIGNED) {
        sum += s2;
        mn = (uint64_t)((int64_t)mn2 < (int64_t)mn ? (int64_t)mn2 : (int64_t)mn);
        mx = (uint64_t)((int64_t)mx2 > (int64_t)mx ? (int64_t)mx2 : (int64_t)mx);
    } else {
        sum += s2;
        mn = mn2 < mn ? mn2 : mn;
        mx = mx2 > mx ? mx2 : mx;
    }
}

// Block-wide reduction of one aggregate: wave butterfly (fixed order => deterministic), then the 4
// wave results folded in wave order by thread 0.
__device__ __forceinline__ void block_reduce_agg(int cls, uint64_t sum, uint64_t mn, uint64_t mx, int64_t cnt,
                                                 AggPartial* lds4, AggPartial* out) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        uint64_t s2 = shfl_xor64(sum, m), mn2 = shfl_xor64(mn, m), mx2 = shfl_xor64(mx, m);
        int64_t c2 = (int64_t)shfl_xor64((uint64_t)cnt, m);
        agg_merge(cls, sum, mn, mx, cnt, s2, mn2, mx2, c2);
    }
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { lds4[wave].sum = sum; lds4[wave].mn = mn; lds4[wave].mx = mx; lds4[wave].cnt = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t s = lds4[0].sum, a = lds4[0].mn, b = lds4[0].mx;
        int64_t c = lds4[0].cnt;
        for (int w = 1; w < kBlock / 64; ++w) agg_merge(cls, s, a, b, c, lds4[w].sum, lds4[w].mn, lds4[w].mx, lds4[w].cnt);
        out->sum = s; out->mn = a; out->mx = b; out->cnt = c;
    }
}

// f64 sin / cos / tan for the fused kernels.  The device libm's versions cost ~100 lane-cycles per element (the C1-shape
// kernel sum(sin(x + c)) ran at 0.40 of the HBM peak, VALU-bound); these are the classic two-step evaluation written
// out directly: Cody-Waite reduction by pi/2 in four FMA steps (the first three constants hold 33 bits each, so
// k * constant is exact for |k| < 2^16 and the chain keeps the RELATIVE accuracy of r even next to a zero of the
// function), then a polynomial kernel.  tan divides the two degree-13 / degree-12 minimax kernels of fdlibm's k_sin.c /
// k_cos.c on |r| <= pi/4.  Checked on the host against glibc over 2e7 points incl. the neighbourhood of every multiple of
// pi/2 below 1e5: <= 2 ulp, relative error <= 3.2e-16.  Arguments with |x| >= 1e5, infinities and NaNs take the libm path.
__device__ __forceinline__ void trig_reduce(double x, double& r, int& q) {
    const double kd = rint(x * 6.36619772367581382433e-01);   // 2 / pi
    q = (int)kd;
    double t = fma(-kd, 1.57079632673412561417e+00, x);       // pi/2, bits 1..33
    t = fma(-kd, 6.07710050630396597660e-11, t);              // bits 34..66
    t = fma(-kd, 2.02226624871116645580e-21, t);              // bits 67..99
    r = fma(-kd, 8.47842766036889956997e-32, t);              // the tail
}
__device__ __forceinline__ double trig_ksin(double r) {
    const double z = r * r, v = z * r;
    const double p = fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06),
                                -1.98412698298579493134e-04), 8.33333333332248946124e-03);
    return fma(v, fma(z, p, -1.66666666666666324348e-01), r);
}
__device__ __forceinline__ double trig_kcos(double r) {
    const double z = r * r;
    const double p = z * fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07),
                                            2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    return 1.0 - (0.5 * z - z * p);
}
// sin / cos use ONE odd polynomial on [-pi/2, pi/2] instead (x = m * pi/2 + r with m even for sin, odd for cos; the sign is
// the parity of m / 2): sin r = r + r z P(z), P of degree 7 in z = r^2 fitted at Chebyshev nodes in 60-digit arithmetic
// (relative error 3.6e-17 before rounding).  About half the arithmetic of evaluating both quarter-pi kernels and
// selecting; same host check: <= 2 ulp, relative error <= 2.7e-16, cos(0) == 1, sin(-0.0) == -0.0.
__device__ __forceinline__ double trig_reduce_m(double x, double md) {
    double t = fma(-md, 1.57079632673412561417e+00, x);
    t = fma(-md, 6.07710050630396597660e-11, t);
    t = fma(-md, 2.02226624871116645580e-21, t);
    return fma(-md, 8.47842766036889956997e-32, t);
}
__device__ __forceinline__ double trig_psin(double r, int k) {
    const double z = r * r;
    double p = 2.73143472074881618678e-15;
    p = fma(p, z, -7.64396949104287201694e-13);
    p = fma(p, z, 1.60589772872252839761e-10);
    p = fma(p, z, -2.50521076166068844386e-08);
    p = fma(p, z, 2.75573192191601764039e-06);
    p = fma(p, z, -1.98412698412549632883e-04);
    p = fma(p, z, 8.33333333333331587045e-03);
    p = fma(p, z, -1.66666666666666657415e-01);
    const double v = fma(r * z, p, r);
    return u2d(d2u(v) ^ ((uint64_t)(k & 1) << 63));
}
__device__ __forceinline__ double rdf_sin(double x) {
    const double ax = fabs(x);
    if (!(ax < 1.0e5)) return sin(x);
    if (ax < 0x1p-26) return x;            // also keeps -0.0
    const double kd = rint(x * 3.18309886183790671538e-01);   // 1 / pi
    return trig_psin(trig_reduce_m(x, 2.0 * kd), (int)kd);
}
__device__ __forceinline__ double rdf_cos(double x) {
    if (!(fabs(x) < 1.0e5)) return cos(x);
    const double kd = rint(fma(x, 3.18309886183790671538e-01, -0.5));
    return trig_psin(trig_reduce_m(x, fma(2.0, kd, 1.0)), (int)kd + 1);
}
__device__ __forceinline__ double rdf_tan(double x) {
    const double ax = fabs(x);
    if (!(ax < 1.0e5)) return tan(x);
    if (ax < 0x1p-26) return x;
    double r; int q;
    trig_reduce(x, r, q);
    const double s = trig_ksin(r), c = trig_kcos(r);
    return (q & 1) ? -c / s : s / c;
}

// R rows of one lane at once.  rdf_sin / rdf_cos / rdf_tan above test every argument for the libm path (|x| >= 1e5, inf,
// NaN) and for tiny values with their own branches: per row two compare + exec-mask + branch sequences, and the branches
// fence each row's ~20-deep FMA chain off from its neighbours.  Here ONE wave-wide test covers all R rows (a single lane
// holding one large argument sends the wave's R rows down the per-row path — rare, and still correct), and the common path
// is branch-free: R independent chains the scheduler interleaves, the tiny-argument result picked by a select.
// KIND 0 = sin, 1 = cos, 2 = tan.  Same arithmetic per element as the scalar functions: bit-identical results.
// (the per-row path is a real call: inlined R times, libm's Payne-Hanek reduction set the kernel's register count — 117
// VGPRs, 4 waves per SIMD — for a path that almost never runs)
template <int KIND>
__device__ __attribute__((noinline)) double rdf_trig_any(double x) { return KIND == 0 ? rdf_sin(x) : KIND == 1 ? rdf_cos(x) : rdf_tan(x); }
template <int KIND, int R>
__device__ __forceinline__ void rdf_trig_rows(const double (&a)[R], double (&out)[R]) {
    bool big = false;
#pragma unroll
    for (int r = 0; r < R; ++r) big |= !(fabs(a[r]) < 1.0e5);
    if (__ballot(big) != 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) out[r] = rdf_trig_any<KIND>(a[r]);
        return;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const double x = a[r];
        if (KIND == 0) {
            const double kd = rint(x * 3.18309886183790671538e-01);
            // trig_reduce_m(x, 2 kd) with the factor 2 folded into the constants (a power of two: the same roundings)
            double t = fma(-kd, 2.0 * 1.57079632673412561417e+00, x);
            t = fma(-kd, 2.0 * 6.07710050630396597660e-11, t);
            t = fma(-kd, 2.0 * 2.02226624871116645580e-21, t);
            t = fma(-kd, 2.0 * 8.47842766036889956997e-32, t);
            const double v = trig_psin(t, (int)kd);
            out[r] = fabs(x) < 0x1p-26 ? x : v;
        } else if (KIND == 1) {
            const double kd = rint(fma(x, 3.18309886183790671538e-01, -0.5));
            out[r] = trig_psin(trig_reduce_m(x, fma(2.0, kd, 1.0)), (int)kd + 1);
        } else {
            double t; int q;
            trig_reduce(x, t, q);
            const double sn = trig_ksin(t), cs = trig_kcos(t);
            const double v = (q & 1) ? -cs / sn : sn / cs;
            out[r] = fabs(x) < 0x1p-26 ? x : v;
        }
    }
}

// f32 columns compute in f32 (num::Float on f32, src/functions/scalar.rs:106-452).  sin / cos: the argument is reduced in f64
// — two FMAs against pi split in two doubles, good for every |x| < 1e9 with the relative accuracy of r kept next to the zeros
// of the function — and ONE odd polynomial of degree 9 runs in f32 on [-pi/2, pi/2] (weighted least squares at Chebyshev nodes
// in 50-digit arithmetic; the error is the f32 rounding of its evaluation).  14 vector instructions against the device libm's
// ~40 with its own branches.  Checked on the host against (float)sin((double)x) over 2.4e8 points incl. every multiple of pi/2
// below 6e6 and its neighbours: <= 2.0 ulp, relative error <= 1.2e-7; sin(-0.0f) == -0.0f, cos(0) == 1.  |x| >= 1e9, inf, NaN: libm.
__device__ __forceinline__ float trigf_psin(float r, int k) {
    const float z = r * r;
    float p = 2.6052944122056942e-06f;
    p = fmaf(p, z, -0.00019809351942967623f);
    p = fmaf(p, z, 0.008333061821758747f);
    p = fmaf(p, z, -0.16666659712791443f);
    const float v = fmaf(r * z, p, r);
    return u2f(f2u(v) ^ ((uint32_t)(k & 1) << 31));
}
template <int KIND> __device__ __forceinline__ float trigf_core(float x) {   // KIND 0 sin, 1 cos; |x| < 1e9
    const double xd = (double)x;
    if (KIND == 0) {
        const double kd = rint(xd * 3.18309886183790671538e-01);
        double r = fma(-kd, 3.14159265358979311600e+00, xd);
        r = fma(-kd, 1.22464679914735317723e-16, r);
        const float v = trigf_psin((float)r, (int)kd);
        return fabsf(x) < 0x1p-13f ? x : v;     // (also keeps -0.0f)
    }
    const double kd = rint(fma(xd, 3.18309886183790671538e-01, -0.5)), md = kd + 0.5;
    double r = fma(-md, 3.14159265358979311600e+00, xd);
    r = fma(-md, 1.22464679914735317723e-16, r);
    return trigf_psin((float)r, (int)kd + 1);
}
__device__ __forceinline__ float rdf_sin(float x) { return fabsf(x) < 1.0e9f ? trigf_core<0>(x) : sinf(x); }
__device__ __forceinline__ float rdf_cos(float x) { return fabsf(x) < 1.0e9f ? trigf_core<1>(x) : cosf(x); }
__device__ __forceinline__ float rdf_tan(float x) { return tanf(x); }
template <int KIND>
__device__ __attribute__((noinline)) float rdf_trigf_any(float x) { return KIND == 0 ? rdf_sin(x) : KIND == 1 ? rdf_cos(x) : rdf_tan(x); }
template <int KIND, int R>
__device__ __forceinline__ void rdf_trig_rows(const float (&a)[R], float (&out)[R]) {
    if (KIND == 2) {
#pragma unroll
        for (int r = 0; r < R; ++r) out[r] = tanf(a[r]);
        return;
    }
    bool big = false;
#pragma unroll
    for (int r = 0; r < R; ++r) big |= !(fabsf(a[r]) < 1.0e9f);
    if (__ballot(big) != 0) {     // a lane holds a huge / non-finite argument: the per-row path, a call
#pragma unroll
        for (int r = 0; r < R; ++r) out[r] = rdf_trigf_any<KIND>(a[r]);
        return;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) out[r] = trigf_core<(KIND == 0 ? 0 : 1)>(a[r]);
}

// largest c in [0, n) with start[c] <= t (start is a non-decreasing prefix table; scalar loads)
template <class P>
__device__ __forceinline__ int64_t find_chunk(P start, int64_t n, int64_t t) {
    int64_t lo = 0, hi = n - 1;
    while (lo < hi) {
        const int64_t mid = (lo + hi + 1) >> 1;
        if (start[mid] <= t) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// The same for tile -> chunk lookups of the streaming kernels: frames usually carry equally sized RecordBatches (the
// reader's 1024-row batches, src/dataframe.rs:352), where chunk c starts at tile c * tiles_per_chunk.  Interpolating
// gives the answer with two dependent scalar loads instead of log2(n) (measured on 48 828 chunks of 1024 rows: the
// binary search alone cost ~9 us per tile); any other layout fails the check and takes the search from the guess.
template <class P>
__device__ __forceinline__ int64_t find_chunk_tile(P start, int64_t n, int64_t t) {
    if (n <= 1) return 0;
    const int64_t last = start[n - 1];
    if (t >= ((int64_t)1 << 31) || n >= ((int64_t)1 << 31)) return find_chunk(start, n, t);   // keeps t * (n - 1) in 64 bits
    int64_t g = last > 0 ? (int64_t)((uint64_t)t * (uint64_t)(n - 1) / (uint64_t)last) : n - 1;
    if (g > n - 1) g = n - 1;
    const int64_t sg = start[g];
    if (sg <= t) {
        if (g + 1 == n || start[g + 1] > t) return g;
        int64_t lo = g + 1, hi = n - 1;   // start[g + 1] <= t: the answer is at or after g + 1
        while (lo < hi) {
            const int64_t mid = (lo + hi + 1) >> 1;
            if (start[mid] <= t) lo = mid; else hi = mid - 1;
        }
        return lo;
    }
    int64_t lo = 0, hi = g - 1;
    while (lo < hi) {
        const int64_t mid = (lo + hi + 1) >> 1;
        if (start[mid] <= t) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// The same with the division replaced by a multiplication: inv = floor(((n - 1) << 32) / start[n - 1]) comes from the host.
// For equally long batches the guess is exact or one short; both neighbours are tried before any search.  (An explicit
// "uniform batches" fast path next to the search cost the specialised kernels 20 VGPRs and with them a third of their
// occupancy; this single path costs none.)
template <class P>
__device__ __forceinline__ int64_t find_chunk_tile_inv(P start, int64_t n, int64_t t, uint64_t inv) {
    if (n <= 1) return 0;
    if (inv == 0 || t >= ((int64_t)1 << 32)) return find_chunk(start, n, t);
    int64_t g = (int64_t)(((uint64_t)(uint32_t)t * inv) >> 32);
    if (g > n - 1) g = n - 1;
    const int64_t sg = start[g], s1 = start[g + 1 < n ? g + 1 : g];
    if (sg <= t) {
        if (g + 1 == n || s1 > t) return g;
        if (g + 2 == n || start[g + 2] > t) return g + 1;
        int64_t lo = g + 2, hi = n - 1;
        while (lo < hi) { const int64_t mid = (lo + hi + 1) >> 1; if (start[mid] <= t) lo = mid; else hi = mid - 1; }
        return lo;
    }
    if (g >= 1 && start[g - 1] <= t) return g - 1;
    int64_t lo = 0, hi = g >= 2 ? g - 2 : 0;
    while (lo < hi) { const int64_t mid = (lo + hi + 1) >> 1; if (start[mid] <= t) lo = mid; else hi = mid - 1; }
    return lo;
}

// Row -> chunk lookups done per LANE (take, sort keys): the same interpolation with the division replaced by a
// multiplication with `inv` = (n - 1) / start[n - 1] (chunk_lookup_scale, computed once per kernel).  The guess is
// exact or one off for equally sized chunks (three loads); other layouts finish with a search from the guess.
__device__ __forceinline__ double chunk_lookup_scale(const int64_t* start, int64_t n) {
    if (n <= 1) return 0.0;
    const int64_t last = start[n - 1];
    return last > 0 ? (double)(n - 1) / (double)last : 0.0;
}
__device__ __forceinline__ int64_t find_chunk_row(const int64_t* start, int64_t n, int64_t t, double inv) {
    if (n <= 1) return 0;
    int64_t g = (int64_t)((double)t * inv);
    g = g < 0 ? 0 : (g > n - 1 ? n - 1 : g);
    int64_t lo, hi;
    if (start[g] <= t) {
        if (g + 1 == n || start[g + 1] > t) return g;
        if (g + 2 == n || start[g + 2] > t) return g + 1;
        lo = g + 2; hi = n - 1;
    } else {
        if (g >= 1 && start[g - 1] <= t) return g - 1;
        lo = 0; hi = g >= 2 ? g - 2 : 0;
    }
    while (lo < hi) {
        const int64_t mid = (lo + hi + 1) >> 1;
        if (start[mid] <= t) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// arrow::compute::hour: floor division to seconds, floor modulo to the second of the day (constant divisors per unit)
template <int64_t U> __device__ __forceinline__ uint64_t hour_of(int64_t v) {
    int64_t q = v / U;
    if (v % U < 0) --q;
    int64_t m = q % 86400;
    if (m < 0) m += 86400;
    return (uint64_t)(m / 3600);
}

}  // namespace rdfk
