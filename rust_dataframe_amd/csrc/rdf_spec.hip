// rdf_spec.hip — ahead-of-time specialised fused kernels (expression templates).
//
// The general evaluator (rdf_eval.hip) interprets any expression tree; its price is registers.  For
// the program shapes that dominate the path — one ScalarFunctions op per call (src/functions/scalar.rs),
// a comparison against a scalar (BooleanFilter, src/expression.rs:836-859), an aggregate of a column
// or of a small fused expression (BASELINE configs C1-C3) — this file instantiates straight-line
// kernels from C++ expression templates over 8-byte columns (f64 / i64 / u64):
//
//   spec_kernel<Prog>: wave-contiguous rows, 16-byte global_load_dwordx4 (1 KiB per wave-instruction),
//   U vectors in flight per lane per column, validity as bulk scalar bitmap windows, predicate and
//   arithmetic in registers, sink = {sum,min,max,count} two-stage reduction, or 16-byte stores with
//   ballot-built validity / boolean bitmaps.
//
// A program is looked up by its canonical signature string (the host builds the same string from the
// rdf_expr_node tree); a miss falls back to the interpreter.  Everything here is compiled by build().
#include <map>
#include <string>
#include <type_traits>

#include "rdf_common.hip.h"

namespace rdfk {

typedef double dvec2 __attribute__((ext_vector_type(2)));
typedef uint64_t uvec2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------------
// expression templates

template <int DT> struct CType;
template <> struct CType<RDF_F64> { using T = double; static constexpr char tag = 'd'; };
template <> struct CType<RDF_I64> { using T = int64_t; static constexpr char tag = 'l'; };
template <> struct CType<RDF_U64> { using T = uint64_t; static constexpr char tag = 'u'; };
template <> struct CType<RDF_BOOL> { using T = bool; static constexpr char tag = 'b'; };

template <int NC, int R>
struct Ctx {
    uint64_t v[NC][R];   // raw 8-byte elements, row r of column c
    uint32_t valid[NC];  // bit r = row r of column c is valid
    uint64_t imm[4];
    uint32_t inr;        // bit r = row r exists
    uint32_t err;
};

template <class T> __device__ __forceinline__ T from_bits(uint64_t x);
template <> __device__ __forceinline__ double from_bits<double>(uint64_t x) { return u2d(x); }
template <> __device__ __forceinline__ int64_t from_bits<int64_t>(uint64_t x) { return (int64_t)x; }
template <> __device__ __forceinline__ uint64_t from_bits<uint64_t>(uint64_t x) { return x; }
template <> __device__ __forceinline__ bool from_bits<bool>(uint64_t x) { return x != 0; }
__device__ __forceinline__ uint64_t to_bits(double x) { return d2u(x); }
__device__ __forceinline__ uint64_t to_bits(int64_t x) { return (uint64_t)x; }
__device__ __forceinline__ uint64_t to_bits(uint64_t x) { return x; }
__device__ __forceinline__ uint64_t to_bits(bool x) { return (uint64_t)x; }

template <int I, int DT>
struct Col {
    static constexpr int dt = DT;
    static constexpr int ncols = I + 1;
    using T = typename CType<DT>::T;
    template <int r, class C> static __device__ __forceinline__ T eval(C& c) { return from_bits<T>(c.v[I][r]); }
    template <class C> static __device__ __forceinline__ uint32_t vmask(const C& c) { return c.valid[I]; }
    static std::string sig() { return std::string("c") + char('0' + I) + CType<DT>::tag; }
};
template <int K, int DT>
struct Imm {
    static constexpr int dt = DT;
    static constexpr int ncols = 0;
    using T = typename CType<DT>::T;
    template <int r, class C> static __device__ __forceinline__ T eval(C& c) { return from_bits<T>(c.imm[K]); }
    template <class C> static __device__ __forceinline__ uint32_t vmask(const C&) { return ~0u; }
    static std::string sig() { return std::string("k") + char('0' + K) + CType<DT>::tag; }
};

template <class T> __device__ __forceinline__ double as_f64(T x) { return (double)x; }

constexpr bool is_cmp(int op) { return op >= RDF_OP_GT && op <= RDF_OP_LE; }
constexpr bool is_logic(int op) { return op == RDF_OP_AND || op == RDF_OP_OR; }

template <int OP, class A, class B>
struct Bin {
    static_assert(is_cmp(OP) || A::dt == B::dt, "arithmetic operands share one dtype");
    static constexpr int dt = (is_cmp(OP) || is_logic(OP)) ? RDF_BOOL : A::dt;
    static constexpr int ncols = A::ncols > B::ncols ? A::ncols : B::ncols;
    using T = typename CType<dt>::T;
    template <class C> static __device__ __forceinline__ uint32_t vmask(const C& c) { return A::vmask(c) & B::vmask(c); }
    template <int r, class C> static __device__ __forceinline__ T eval(C& c) {
        const auto x = A::template eval<r>(c);
        const auto y = B::template eval<r>(c);
        if constexpr (is_cmp(OP)) {  // both sides cast to Float64 (src/expression.rs:844-845)
            const double a = as_f64(x), b = as_f64(y);
            if constexpr (OP == RDF_OP_GT) return a > b;
            else if constexpr (OP == RDF_OP_GE) return a >= b;
            else if constexpr (OP == RDF_OP_EQ) return a == b;
            else if constexpr (OP == RDF_OP_NE) return a != b;
            else if constexpr (OP == RDF_OP_LT) return a < b;
            else return a <= b;
        } else if constexpr (OP == RDF_OP_AND) return x && y;
        else if constexpr (OP == RDF_OP_OR) return x || y;
        else if constexpr (A::dt == RDF_F64) {
            if constexpr (OP == RDF_OP_ADD) return x + y;
            else if constexpr (OP == RDF_OP_SUB) return x - y;
            else if constexpr (OP == RDF_OP_MUL) return x * y;
            else if constexpr (OP == RDF_OP_DIV) {
                const bool z = y == 0.0;
                if (z && ((vmask(c) & c.inr) >> r & 1)) c.err |= 1u;
                return z ? 0.0 : x / y;
            } else if constexpr (OP == RDF_OP_ATAN2) return atan2(x, y);
            else if constexpr (OP == RDF_OP_HYPOT) return hypot(x, y);
            else return log(x) / log(y);
        } else {  // i64 / u64: wrapping
            using U = uint64_t;
            if constexpr (OP == RDF_OP_ADD) return (T)((U)x + (U)y);
            else if constexpr (OP == RDF_OP_SUB) return (T)((U)x - (U)y);
            else if constexpr (OP == RDF_OP_MUL) return (T)((U)x * (U)y);
            else {
                const bool z = y == 0;
                if (z && ((vmask(c) & c.inr) >> r & 1)) c.err |= 1u;
                if (z) return (T)0;
                if constexpr (A::dt == RDF_I64) return y == -1 ? (T)((U)0 - (U)x) : x / y;
                else return x / y;
            }
        }
    }
    static std::string sig() { return "(" + std::to_string(OP) + " " + A::sig() + " " + B::sig() + ")"; }
};

template <int OP, class A>
struct Un {
    static constexpr int dt = OP == RDF_OP_NOT ? RDF_BOOL : A::dt;
    static constexpr int ncols = A::ncols;
    using T = typename CType<dt>::T;
    template <class C> static __device__ __forceinline__ uint32_t vmask(const C& c) { return A::vmask(c); }
    template <int r, class C> static __device__ __forceinline__ T eval(C& c) {
        const auto x = A::template eval<r>(c);
        if constexpr (OP == RDF_OP_NOT) return !x;
        else if constexpr (A::dt == RDF_I64) return x < 0 ? (int64_t)((uint64_t)0 - (uint64_t)x) : x;  // abs, MIN wraps
        else if constexpr (OP == RDF_OP_ABS) return fabs(x);
        else if constexpr (OP == RDF_OP_ACOS) return acos(x);
        else if constexpr (OP == RDF_OP_ASIN) return asin(x);
        else if constexpr (OP == RDF_OP_ATAN) return atan(x);
        else if constexpr (OP == RDF_OP_CBRT) return cbrt(x);
        else if constexpr (OP == RDF_OP_CEIL) return ceil(x);
        else if constexpr (OP == RDF_OP_COS) return cos(x);
        else if constexpr (OP == RDF_OP_COSH) return cosh(x);
        else if constexpr (OP == RDF_OP_DEGREES) return x * (180.0 / 3.14159265358979323846264338327950288);
        else if constexpr (OP == RDF_OP_EXP) return exp(x);
        else if constexpr (OP == RDF_OP_EXPM1) return expm1(x);
        else if constexpr (OP == RDF_OP_FLOOR) return floor(x);
        else if constexpr (OP == RDF_OP_LOG10) return log10(x);
        else if constexpr (OP == RDF_OP_LOG2) return log2(x);
        else if constexpr (OP == RDF_OP_RADIANS) return x * (3.14159265358979323846264338327950288 / 180.0);
        else if constexpr (OP == RDF_OP_ROUND) return round(x);
        else if constexpr (OP == RDF_OP_SIN) return sin(x);
        else if constexpr (OP == RDF_OP_SINH) return sinh(x);
        else if constexpr (OP == RDF_OP_SQRT) return sqrt(x);
        else if constexpr (OP == RDF_OP_TAN) return tan(x);
        else return tanh(x);
    }
    static std::string sig() { return "[" + std::to_string(OP) + " " + A::sig() + "]"; }
};

template <int TO, class A>
struct Cast {
    static constexpr int dt = TO;
    static constexpr int ncols = A::ncols;
    using T = typename CType<TO>::T;
    template <class C> static __device__ __forceinline__ uint32_t vmask(const C& c) { return A::vmask(c); }
    template <int r, class C> static __device__ __forceinline__ T eval(C& c) {
        const auto x = A::template eval<r>(c);
        if constexpr (TO == RDF_BOOL) return x != 0;
        else if constexpr (TO == RDF_F64) return (double)x;
        else if constexpr (A::dt == RDF_F64) {  // saturating `as`
            if (x != x) return (T)0;
            if constexpr (TO == RDF_I64) {
                if (x >= 9223372036854775808.0) return INT64_MAX;
                if (x <= -9223372036854775808.0) return INT64_MIN;
                return (int64_t)x;
            } else {
                if (x <= 0.0) return (T)0;
                if (x >= 18446744073709551616.0) return ~0ull;
                return (uint64_t)x;
            }
        } else return (T)x;
    }
    static std::string sig() { return "{" + std::to_string(TO) + " " + A::sig() + "}"; }
};

struct None {
    static constexpr int ncols = 0;
    static std::string sig() { return "-"; }
};

// ------------------------------------------------------------------------------------------------
// typed running aggregates

template <int DT> struct AggT;
template <> struct AggT<RDF_F64> {
    double sum, mn, mx; int64_t cnt;
    static constexpr int cls = CLS_F64;
    __device__ __forceinline__ void init() { sum = 0.0; mn = mx = __longlong_as_double(0x7FF8000000000000ll); cnt = 0; }
    __device__ __forceinline__ void add(double v) { sum += v; mn = fmin(mn, v); mx = fmax(mx, v); ++cnt; }
    __device__ __forceinline__ uint64_t s() const { return d2u(sum); }
    __device__ __forceinline__ uint64_t a() const { return d2u(mn); }
    __device__ __forceinline__ uint64_t b() const { return d2u(mx); }
};
template <> struct AggT<RDF_I64> {
    uint64_t sum; int64_t mn, mx; int64_t cnt;
    static constexpr int cls = CLS_SIGNED;
    __device__ __forceinline__ void init() { sum = 0; mn = INT64_MAX; mx = INT64_MIN; cnt = 0; }
    __device__ __forceinline__ void add(int64_t v) { sum += (uint64_t)v; mn = v < mn ? v : mn; mx = v > mx ? v : mx; ++cnt; }
    __device__ __forceinline__ uint64_t s() const { return sum; }
    __device__ __forceinline__ uint64_t a() const { return (uint64_t)mn; }
    __device__ __forceinline__ uint64_t b() const { return (uint64_t)mx; }
};
template <> struct AggT<RDF_U64> {
    uint64_t sum, mn, mx; int64_t cnt;
    static constexpr int cls = CLS_UNSIGNED;
    __device__ __forceinline__ void init() { sum = 0; mn = ~0ull; mx = 0; cnt = 0; }
    __device__ __forceinline__ void add(uint64_t v) { sum += v; mn = v < mn ? v : mn; mx = v > mx ? v : mx; ++cnt; }
    __device__ __forceinline__ uint64_t s() const { return sum; }
    __device__ __forceinline__ uint64_t a() const { return mn; }
    __device__ __forceinline__ uint64_t b() const { return mx; }
};
template <> struct AggT<RDF_BOOL> : AggT<RDF_U64> {};

// Lane l of a 16-byte-load wave holds rows 2l and 2l+1, so the two ballots b0 (even rows) and b1 (odd
// rows) must be interleaved into Arrow's row-ordered bitmap words.  Every lane j picks the bit that
// belongs at output position j and the wave ballots again: word `half` (rows 64*half .. +63) in ~5 VALU
// instructions for the whole wave.
__device__ __forceinline__ uint64_t interleave_word(uint64_t b0, uint64_t b1, int half, int lane) {
    const uint64_t src = (lane & 1) ? b1 : b0;
    return __ballot((src >> (32 * half + (lane >> 1))) & 1);
}

// ------------------------------------------------------------------------------------------------
// the kernel

template <class PRED, class V0, class V1, int SINK_>
struct Prog {
    using Pred = PRED; using Val0 = V0; using Val1 = V1;
    static constexpr int SINK = SINK_;
    static constexpr int NC_ = (PRED::ncols > V0::ncols ? PRED::ncols : V0::ncols) > V1::ncols
                                   ? (PRED::ncols > V0::ncols ? PRED::ncols : V0::ncols) : V1::ncols;
    static constexpr int NC = NC_ < 1 ? 1 : NC_;
    static constexpr int U = NC <= 2 ? 4 : 2;  // 16-byte vectors per lane per column per iteration
    static std::string sig() { return "P:" + PRED::sig() + ";V:" + V0::sig() + ";" + V1::sig() + ";S:" + std::to_string(SINK_); }
};

template <class E, int R, int r, class C, class AGG>
__device__ __forceinline__ void agg_rows(C& c, uint32_t live, AGG& g) {
    if constexpr (r < R) {
        const auto v = E::template eval<r>(c);
        if ((live >> r) & 1) g.add(v);
        agg_rows<E, R, r + 1>(c, live, g);
    }
}
template <class E, int R, int r, class C>
__device__ __forceinline__ void pred_rows(C& c, uint32_t& keep) {
    if constexpr (r < R) {
        if (!E::template eval<r>(c)) keep &= ~(1u << r);
        pred_rows<E, R, r + 1>(c, keep);
    }
}
template <class E, int R, int r, class C>
__device__ __forceinline__ void eval_rows(C& c, uint64_t (&out)[R]) {
    if constexpr (r < R) {
        out[r] = to_bits(E::template eval<r>(c));
        eval_rows<E, R, r + 1>(c, out);
    }
}

template <class P>
__global__ __launch_bounds__(kBlock) void spec_kernel(const SpecArgs a) {
    constexpr int NC = P::NC, U = P::U, R = 2 * U;
    using Pred = typename P::Pred;
    using V0 = typename P::Val0;
    using V1 = typename P::Val1;
    constexpr bool has_pred = !std::is_same<Pred, None>::value;
    constexpr bool has_v1 = !std::is_same<V1, None>::value;
    __shared__ AggPartial red_lds[kBlock / 64];
    const int lane = threadIdx.x & 63;
    const int wave = wave_id();

    Ctx<NC, R> c;
    c.err = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) c.imm[k] = a.imm[k];
    AggT<V0::dt> g0;
    using V1e = typename std::conditional<has_v1, V1, V0>::type;
    AggT<V1e::dt> g1;
    g0.init();
    g1.init();
    uint32_t nulls = 0;
    int64_t cur_chunk = -1;

    // A "tile" is one block iteration: kBlock*U vectors = 2*kBlock*U rows of ONE chunk.
    constexpr int64_t per_iter = (int64_t)kBlock * U;
    for (int64_t tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        int64_t ch = 0, base, n;
        DevChunkCol col[NC];
        DevOutChunk out = a.out;
        if (a.nchunks == 1) {
            base = tile * per_iter;
            n = a.n;
#pragma unroll
            for (int k = 0; k < NC; ++k) col[k] = a.cols[k];
        } else {
            ch = find_chunk(a.chunk_tile_start, a.nchunks, tile);
            base = (tile - a.chunk_tile_start[ch]) * per_iter;
            n = a.chunk_len[ch];
#pragma unroll
            for (int k = 0; k < NC; ++k) col[k] = a.cols_tab[(int64_t)k * a.nchunks + ch];
            if (P::SINK == SINK_STORE) out = a.outs_tab[ch];
        }
        if (P::SINK == SINK_STORE && ch != cur_chunk) {  // one null-count atomic per (wave, chunk), not per tile
            if (cur_chunk >= 0 && lane == 0 && nulls) atomicAdd((unsigned long long*)&a.out_null_count[cur_chunk], (unsigned long long)nulls);
            nulls = 0;
            cur_chunk = ch;
        }
        const int64_t wbase = base + (int64_t)wave * (U * 64);
        const int64_t rw = 2 * wbase;  // first row of this wave
        // rows in range; `full` (wave-uniform) = every row of this wave's span exists: the common case
        // runs without per-lane bounds checks or predicated loads
        const bool full = rw + 128 * U <= n;
        if (full) {
            c.inr = (1u << R) - 1;
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const uvec2* p = (const uvec2*)((const uint64_t*)col[k].values + col[k].offset) + wbase + lane;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uvec2 t = __builtin_nontemporal_load(p + u * 64);
                    c.v[k][2 * u] = t.x;
                    c.v[k][2 * u + 1] = t.y;
                }
            }
        } else {
            c.inr = 0;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t row = 2 * (wbase + u * 64 + lane);
                c.inr |= (uint32_t)(row < n) << (2 * u) | (uint32_t)(row + 1 < n) << (2 * u + 1);
            }
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const uint64_t* p = (const uint64_t*)col[k].values + col[k].offset;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int64_t i = wbase + u * 64 + lane;
                    const uint32_t m = (c.inr >> (2 * u)) & 3u;
                    if (m == 3u) {
                        const uvec2 t = *((const uvec2*)p + i);
                        c.v[k][2 * u] = t.x;
                        c.v[k][2 * u + 1] = t.y;
                    } else {
                        c.v[k][2 * u] = m ? p[2 * i] : 0;
                        c.v[k][2 * u + 1] = 0;
                    }
                }
            }
        }
        // validity: 2U windows of 64 rows per column
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            c.valid[k] = c.inr;
            if (col[k].validity) {
                uint64_t w[2 * U];
                if (a.vec_bitmap) load_windows<2 * U>(col[k].validity, col[k].offset + rw, n - rw, w);
                else load_windows_s<2 * U>(col[k].validity, col[k].offset + rw, n - rw, w);
                uint32_t m = 0;
                const int sh = (2 * lane) & 63;
#pragma unroll
                for (int u = 0; u < U; ++u) m |= ((uint32_t)((lane < 32 ? w[2 * u] : w[2 * u + 1]) >> sh) & 3u) << (2 * u);
                c.valid[k] = m & c.inr;
            }
        }
        // (2) predicate
        uint32_t keep = c.inr;
        if constexpr (has_pred) {
            keep &= Pred::vmask(c);
            pred_rows<Pred, R, 0>(c, keep);
        }
        // (3) sink
        if constexpr (P::SINK == SINK_AGG) {
            agg_rows<V0, R, 0>(c, keep & V0::vmask(c), g0);
            if constexpr (has_v1) agg_rows<V1, R, 0>(c, keep & V1::vmask(c), g1);
        } else {
            uint64_t outv[R];
            eval_rows<V0, R, 0>(c, outv);
            const uint32_t vm = V0::vmask(c) & c.inr;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t i = wbase + u * 64 + lane;
                const uint32_t in2 = (c.inr >> (2 * u)) & 3u;
                const uint32_t v2 = (vm >> (2 * u)) & 3u;
                const uint64_t x0 = (v2 & 1u) ? outv[2 * u] : 0, x1 = (v2 & 2u) ? outv[2 * u + 1] : 0;  // null slots hold 0
                const uint64_t in0 = __ballot(in2 & 1u), in1 = __ballot(in2 & 2u);
                const bool upper = (in0 >> 32) != 0;  // the wave's second 64 rows exist
                uint64_t* const ow = (uint64_t*)out.values + ((wbase + u * 64) >> 5);
                if constexpr (V0::dt == RDF_BOOL) {
                    const uint64_t b0 = __ballot(x0 & 1), b1 = __ballot(x1 & 1);
                    const uint64_t w0 = interleave_word(b0, b1, 0, lane), w1 = interleave_word(b0, b1, 1, lane);
                    if (lane == 0 && in0) { ow[0] = w0; if (upper) ow[1] = w1; }
                } else {
                    if (in2 == 3u) { uvec2 t; t.x = x0; t.y = x1; ((uvec2*)out.values)[i] = t; }
                    else if (in2) ((uint64_t*)out.values)[2 * i] = x0;
                }
                const uint64_t vb0 = __ballot(v2 & 1u), vb1 = __ballot(v2 & 2u);
                if (out.validity) {
                    const uint64_t w0 = interleave_word(vb0, vb1, 0, lane), w1 = interleave_word(vb0, vb1, 1, lane);
                    uint64_t* const ob = (uint64_t*)out.validity + ((wbase + u * 64) >> 5);
                    if (lane == 0 && in0) { ob[0] = w0; if (upper) ob[1] = w1; }
                }
                if (lane == 0) nulls += (uint32_t)(__popcll(in0 & ~vb0) + __popcll(in1 & ~vb1));
            }
        }
    }
    if (c.err) atomicOr(a.flags, c.err);
    if constexpr (P::SINK == SINK_AGG) {
        constexpr int nv = has_v1 ? 2 : 1;
        block_reduce_agg(g0.cls, g0.s(), g0.a(), g0.b(), g0.cnt, red_lds, &a.partials[(int64_t)blockIdx.x * nv]);
        if constexpr (has_v1) block_reduce_agg(g1.cls, g1.s(), g1.a(), g1.b(), g1.cnt, red_lds, &a.partials[(int64_t)blockIdx.x * nv + 1]);
    } else {
        if (cur_chunk >= 0 && lane == 0 && nulls) atomicAdd((unsigned long long*)&a.out_null_count[cur_chunk], (unsigned long long)nulls);
    }
}

// ------------------------------------------------------------------------------------------------
// the catalog

typedef void (*SpecLaunch)(const SpecArgs&, int, hipStream_t);
struct SpecEntry { SpecLaunch launch; int rows_per_block_iter; };

template <class P>
static void launch_prog(const SpecArgs& a, int grid, hipStream_t s) {
    hipLaunchKernelGGL((spec_kernel<P>), dim3(grid), dim3(kBlock), 0, s, a);
}

static std::map<std::string, SpecEntry>& registry() {
    static std::map<std::string, SpecEntry> r;
    return r;
}
template <class P>
static void reg() { registry()[P::sig()] = SpecEntry{&launch_prog<P>, kBlock * P::U * 2}; }

using D0 = Col<0, RDF_F64>; using D1 = Col<1, RDF_F64>; using D2 = Col<2, RDF_F64>;
using L0 = Col<0, RDF_I64>; using L1 = Col<1, RDF_I64>; using L3 = Col<3, RDF_I64>;
using W0 = Col<0, RDF_U64>; using W1 = Col<1, RDF_U64>;
using KD0 = Imm<0, RDF_F64>; using KL0 = Imm<0, RDF_I64>;

template <int OP> static void reg_cmp_family() {
    // filter(x CMP c) -> aggregates of x / of another column (headline family, config C2)
    reg<Prog<Bin<OP, D0, KD0>, D0, None, SINK_AGG>>();
    reg<Prog<Bin<OP, D0, KD0>, D1, None, SINK_AGG>>();
    reg<Prog<Bin<OP, D0, KD0>, L1, None, SINK_AGG>>();
    reg<Prog<Bin<OP, L0, KD0>, L0, None, SINK_AGG>>();
    // BooleanFilter::eval_to_array of column CMP scalar / column CMP column -> mask
    reg<Prog<None, Bin<OP, D0, KD0>, None, SINK_STORE>>();
    reg<Prog<None, Bin<OP, L0, KD0>, None, SINK_STORE>>();
    reg<Prog<None, Bin<OP, D0, D1>, None, SINK_STORE>>();
    reg<Prog<None, Bin<OP, L0, L1>, None, SINK_STORE>>();
}
template <int OP> static void reg_arith_family() {
    reg<Prog<None, Bin<OP, D0, D1>, None, SINK_STORE>>();   // ScalarFunctions::add/... f64
    reg<Prog<None, Bin<OP, L0, L1>, None, SINK_STORE>>();   // i64
    reg<Prog<None, Bin<OP, W0, W1>, None, SINK_STORE>>();   // u64
    reg<Prog<None, Bin<OP, D0, KD0>, None, SINK_STORE>>();  // column OP scalar ("add_scalar", config C1)
    reg<Prog<None, Bin<OP, L0, KL0>, None, SINK_STORE>>();
}
template <int OP> static void reg_unary_f64() {
    reg<Prog<None, Un<OP, D0>, None, SINK_STORE>>();                      // ScalarFunctions::<op>
    reg<Prog<None, Un<OP, Bin<RDF_OP_ADD, D0, KD0>>, None, SINK_AGG>>();  // sum(op(x + c)) — config C1 shape
    reg<Prog<None, Un<OP, D0>, None, SINK_AGG>>();
}

static void build_registry() {
    // aggregates of a plain column (AggregateFunctions::sum/min/max/count/avg)
    reg<Prog<None, D0, None, SINK_AGG>>();
    reg<Prog<None, L0, None, SINK_AGG>>();
    reg<Prog<None, W0, None, SINK_AGG>>();
    reg<Prog<None, Cast<RDF_F64, L0>, None, SINK_AGG>>();  // avg of an i64 column
    reg<Prog<None, Cast<RDF_F64, W0>, None, SINK_AGG>>();
    reg_cmp_family<RDF_OP_GT>(); reg_cmp_family<RDF_OP_GE>(); reg_cmp_family<RDF_OP_EQ>();
    reg_cmp_family<RDF_OP_NE>(); reg_cmp_family<RDF_OP_LT>(); reg_cmp_family<RDF_OP_LE>();
    reg_arith_family<RDF_OP_ADD>(); reg_arith_family<RDF_OP_SUB>(); reg_arith_family<RDF_OP_MUL>(); reg_arith_family<RDF_OP_DIV>();
    reg<Prog<None, Bin<RDF_OP_ATAN2, D0, D1>, None, SINK_STORE>>();
    reg<Prog<None, Bin<RDF_OP_HYPOT, D0, D1>, None, SINK_STORE>>();
    reg<Prog<None, Bin<RDF_OP_LOG, D0, D1>, None, SINK_STORE>>();
    reg_unary_f64<RDF_OP_ABS>(); reg_unary_f64<RDF_OP_ACOS>(); reg_unary_f64<RDF_OP_ASIN>(); reg_unary_f64<RDF_OP_ATAN>();
    reg_unary_f64<RDF_OP_CBRT>(); reg_unary_f64<RDF_OP_CEIL>(); reg_unary_f64<RDF_OP_COS>(); reg_unary_f64<RDF_OP_COSH>();
    reg_unary_f64<RDF_OP_DEGREES>(); reg_unary_f64<RDF_OP_EXP>(); reg_unary_f64<RDF_OP_EXPM1>(); reg_unary_f64<RDF_OP_FLOOR>();
    reg_unary_f64<RDF_OP_LOG10>(); reg_unary_f64<RDF_OP_LOG2>(); reg_unary_f64<RDF_OP_RADIANS>(); reg_unary_f64<RDF_OP_ROUND>();
    reg_unary_f64<RDF_OP_SIN>(); reg_unary_f64<RDF_OP_SINH>(); reg_unary_f64<RDF_OP_SQRT>(); reg_unary_f64<RDF_OP_TAN>();
    reg_unary_f64<RDF_OP_TANH>();
    reg<Prog<None, Un<RDF_OP_ABS, L0>, None, SINK_STORE>>();
    // casts between the 8-byte types (Function::Cast, src/evaluation.rs:296-315)
    reg<Prog<None, Cast<RDF_F64, L0>, None, SINK_STORE>>();
    reg<Prog<None, Cast<RDF_I64, D0>, None, SINK_STORE>>();
    reg<Prog<None, Cast<RDF_F64, W0>, None, SINK_STORE>>();
    reg<Prog<None, Cast<RDF_U64, D0>, None, SINK_STORE>>();
    // config C3: fused a*b+c -> min/max/count, plus the i64 key column's min/max/count, one pass over 4 columns
    using FMA = Bin<RDF_OP_ADD, Bin<RDF_OP_MUL, D0, D1>, D2>;
    reg<Prog<None, FMA, L3, SINK_AGG>>();
    reg<Prog<None, FMA, None, SINK_AGG>>();
    reg<Prog<None, FMA, None, SINK_STORE>>();
}

const SpecEntry* spec_lookup(const char* sig) {
    static bool built = (build_registry(), true);
    (void)built;
    auto it = registry().find(sig);
    return it == registry().end() ? nullptr : &it->second;
}

int spec_rows_per_block_iter(const char* sig) {
    const SpecEntry* e = spec_lookup(sig);
    return e ? e->rows_per_block_iter : 0;
}
bool spec_available(const char* sig) { return spec_lookup(sig) != nullptr; }
hipError_t launch_spec(const char* sig, const SpecArgs& a, int grid, hipStream_t s) {
    const SpecEntry* e = spec_lookup(sig);
    if (!e) return hipErrorInvalidValue;
    e->launch(a, grid, s);
    return hipGetLastError();
}
int spec_catalog_size() { spec_lookup(""); return (int)registry().size(); }

}  // namespace rdfk
