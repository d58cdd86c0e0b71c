// rdf_expr.hip.h — C++ expression templates shared by the ahead-of-time specialised kernels
// (rdf_spec.hip: one element width per program; rdf_gspec.hip: grouped sink, mixed widths).
// A node exposes: dt, ncols, width (common column width, -1 when mixed), colw<k>() (element width of
// canonical column k inside this expression, 0 when unused), eval<r>(ctx), vmask(ctx), sig().
#pragma once
#include <limits>
#include <string>
#include <type_traits>

#include "rdf_common.hip.h"

namespace rdfk {

template <class S, int N> struct VecOf { typedef S type __attribute__((ext_vector_type(N))); };

// ------------------------------------------------------------------------------------------------
// expression templates

template <int DT> struct CType;
template <> struct CType<RDF_F64> { using T = double; static constexpr char tag = 'd'; static constexpr int width = 8; };
template <> struct CType<RDF_I64> { using T = int64_t; static constexpr char tag = 'l'; static constexpr int width = 8; };
template <> struct CType<RDF_U64> { using T = uint64_t; static constexpr char tag = 'u'; static constexpr int width = 8; };
template <> struct CType<RDF_F32> { using T = float; static constexpr char tag = 'f'; static constexpr int width = 4; };
template <> struct CType<RDF_I32> { using T = int32_t; static constexpr char tag = 'i'; static constexpr int width = 4; };
template <> struct CType<RDF_U32> { using T = uint32_t; static constexpr char tag = 'j'; static constexpr int width = 4; };
template <> struct CType<RDF_I8> { using T = int8_t; static constexpr char tag = 'a'; static constexpr int width = 1; };
template <> struct CType<RDF_U8> { using T = uint8_t; static constexpr char tag = 'h'; static constexpr int width = 1; };
template <> struct CType<RDF_I16> { using T = int16_t; static constexpr char tag = 's'; static constexpr int width = 2; };
template <> struct CType<RDF_U16> { using T = uint16_t; static constexpr char tag = 't'; static constexpr int width = 2; };
template <> struct CType<RDF_BOOL> { using T = bool; static constexpr char tag = 'b'; static constexpr int width = 0; };

constexpr int merge_width(int a, int b) { return a == 0 ? b : (b == 0 || b == a ? a : -1); }
constexpr bool dt_float(int dt) { return dt == RDF_F64 || dt == RDF_F32; }
constexpr bool dt_signed(int dt) { return dt == RDF_I64 || dt == RDF_I32 || dt == RDF_I16 || dt == RDF_I8; }
constexpr int cmax(int a, int b) { return a > b ? a : b; }

template <int NC, int R, class S>
struct Ctx {
    static constexpr int rows = R;
    S        v[NC][R];   // raw elements, row r of column c
    uint32_t valid[NC];  // bit r = row r of column c is valid
    uint64_t imm[kSpecImm];
    uint32_t inr;        // bit r = row r exists
    uint32_t err;
    int32_t  rt[8];      // runtime operators of the *RT nodes (wave-uniform)
};

template <class T> __device__ __forceinline__ T from_bits(uint64_t x);
template <> __device__ __forceinline__ double from_bits<double>(uint64_t x) { return u2d(x); }
template <> __device__ __forceinline__ int64_t from_bits<int64_t>(uint64_t x) { return (int64_t)x; }
template <> __device__ __forceinline__ uint64_t from_bits<uint64_t>(uint64_t x) { return x; }
template <> __device__ __forceinline__ float from_bits<float>(uint64_t x) { return __uint_as_float((uint32_t)x); }
template <> __device__ __forceinline__ int32_t from_bits<int32_t>(uint64_t x) { return (int32_t)(uint32_t)x; }
template <> __device__ __forceinline__ uint32_t from_bits<uint32_t>(uint64_t x) { return (uint32_t)x; }
template <> __device__ __forceinline__ int16_t from_bits<int16_t>(uint64_t x) { return (int16_t)(uint16_t)x; }
template <> __device__ __forceinline__ uint16_t from_bits<uint16_t>(uint64_t x) { return (uint16_t)x; }
template <> __device__ __forceinline__ int8_t from_bits<int8_t>(uint64_t x) { return (int8_t)(uint8_t)x; }
template <> __device__ __forceinline__ uint8_t from_bits<uint8_t>(uint64_t x) { return (uint8_t)x; }
template <> __device__ __forceinline__ bool from_bits<bool>(uint64_t x) { return x != 0; }
__device__ __forceinline__ uint64_t to_bits(double x) { return d2u(x); }
__device__ __forceinline__ uint64_t to_bits(int64_t x) { return (uint64_t)x; }
__device__ __forceinline__ uint64_t to_bits(uint64_t x) { return x; }
__device__ __forceinline__ uint64_t to_bits(float x) { return (uint64_t)__float_as_uint(x); }
__device__ __forceinline__ uint64_t to_bits(int32_t x) { return (uint64_t)(uint32_t)x; }
__device__ __forceinline__ uint64_t to_bits(uint32_t x) { return (uint64_t)x; }
__device__ __forceinline__ uint64_t to_bits(int16_t x) { return (uint64_t)(uint16_t)x; }
__device__ __forceinline__ uint64_t to_bits(uint16_t x) { return (uint64_t)x; }
__device__ __forceinline__ uint64_t to_bits(bool x) { return (uint64_t)x; }

// Rows at once.  Every node also answers eval_rows<R>(ctx, out[R]); the default walks eval<r>.  The runtime-operator nodes
// override it: they take their children's R values first and test the (wave-uniform) operator word ONCE per wave iteration
// instead of once per row — a 4- or 2-byte program is bound by instruction issue, not by HBM, and the per-row form spends
// as many scalar compare / branch instructions on dispatch as it spends vector instructions on data.
template <class E, int R, int r = 0, class C, class T>
__device__ __forceinline__ void eval_each(C& c, T (&out)[R]) {
    if constexpr (r < R) { out[r] = E::template eval<r>(c); eval_each<E, R, r + 1>(c, out); }
}
#define RDF_DEFAULT_EVAL_ROWS(SELF)                                                                                          \
    template <int R, class C> static __device__ __forceinline__ void eval_rows(C& c, T (&out)[R]) { eval_each<SELF, R>(c, out); }

template <int I, int DT>
struct Col {
    static constexpr int dt = DT;
    static constexpr int ncols = I + 1;
    static constexpr int width = CType<DT>::width;
    using T = typename CType<DT>::T;
    template <int k> static constexpr int colw() { return k == I ? CType<DT>::width : 0; }
    template <int r, class C> static __device__ __forceinline__ T eval(C& c) { return from_bits<T>((uint64_t)c.v[I][r]); }
    RDF_DEFAULT_EVAL_ROWS(Col)
    template <class C> static __device__ __forceinline__ uint32_t vmask(const C& c) { return c.valid[I]; }
    static std::string sig() { return std::string("c") + char('0' + I) + CType<DT>::tag; }
};
template <int K, int DT>
struct Imm {
    static constexpr int dt = DT;
    static constexpr int ncols = 0;
    static constexpr int width = 0;
    using T = typename CType<DT>::T;
    template <int k> static constexpr int colw() { return 0; }
    template <int r, class C> static __device__ __forceinline__ T eval(C& c) { return from_bits<T>(c.imm[K]); }
    RDF_DEFAULT_EVAL_ROWS(Imm)
    template <class C> static __device__ __forceinline__ uint32_t vmask(const C&) { return ~0u; }
    static std::string sig() { return std::string("k") + char('0' + K) + CType<DT>::tag; }
};

template <class T> __device__ __forceinline__ double as_f64(T x) { return (double)x; }

// which (operator, operand type) pairs the nodes below implement: the catalogs only ever named valid ones, a program compiled at
// run time (rdf_jit.cpp) is whatever the caller wrote — an unsupported pair must fail to compile (the interpreter then runs it)
constexpr bool un_float_op(int op) { return (op >= RDF_OP_ABS && op <= RDF_OP_TANH) || op == RDF_OP_COT || op == RDF_OP_SEC || op == RDF_OP_CSC; }
constexpr bool is_hour(int op) { return op >= RDF_OP_HOUR_S && op <= RDF_OP_HOUR_DAY; }
constexpr bool un_supported(int op, int dt) {
    return op == RDF_OP_NOT ? dt == RDF_BOOL
         : is_hour(op) ? (dt == RDF_I32 || dt == RDF_I64)
         : dt_float(dt) ? un_float_op(op)
         : (dt_signed(dt) && op == RDF_OP_ABS);
}
constexpr bool bin_supported(int op, int dt) {   // dt: the operands' type (comparisons convert both sides to f64: any numeric type)
    return (op >= RDF_OP_GT && op <= RDF_OP_LE) ? dt != RDF_BOOL
         : (op == RDF_OP_AND || op == RDF_OP_OR) ? dt == RDF_BOOL
         : dt_float(dt) ? (op >= RDF_OP_ADD && op <= RDF_OP_LOG)
         : (dt != RDF_BOOL && op >= RDF_OP_ADD && op <= RDF_OP_DIV);
}

constexpr bool is_cmp(int op) { return op >= RDF_OP_GT && op <= RDF_OP_LE; }
constexpr bool is_logic(int op) { return op == RDF_OP_AND || op == RDF_OP_OR; }

template <int OP, class A, class B>
struct Bin {
    static_assert(is_cmp(OP) || A::dt == B::dt, "arithmetic operands share one dtype");
    static_assert(bin_supported(OP, A::dt) && (!is_cmp(OP) || B::dt != RDF_BOOL), "binary operator / operand type not implemented by the specialised kernels");
    static constexpr int dt = (is_cmp(OP) || is_logic(OP)) ? RDF_BOOL : A::dt;
    static constexpr int ncols = A::ncols > B::ncols ? A::ncols : B::ncols;
    static constexpr int width = merge_width(A::width, B::width);   // -1: mixed (only the grouped kernels take that)
    template <int k> static constexpr int colw() { return cmax(A::template colw<k>(), B::template colw<k>()); }
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
        else if constexpr (dt_float(A::dt)) {
            if constexpr (OP == RDF_OP_ADD) return x + y;
            else if constexpr (OP == RDF_OP_SUB) return x - y;
            else if constexpr (OP == RDF_OP_MUL) return x * y;
            else if constexpr (OP == RDF_OP_DIV) {
                const bool z = y == (T)0;
                if (z && ((vmask(c) & c.inr) >> r & 1)) c.err |= 1u;
                return z ? (T)0 : x / y;
            } else if constexpr (OP == RDF_OP_ATAN2) return atan2(x, y);
            else if constexpr (OP == RDF_OP_HYPOT) return hypot(x, y);
            else return log(x) / log(y);
        } else {  // integers: wrapping
            using U = typename std::make_unsigned<T>::type;
            if constexpr (OP == RDF_OP_ADD) return (T)((U)x + (U)y);
            else if constexpr (OP == RDF_OP_SUB) return (T)((U)x - (U)y);
            else if constexpr (OP == RDF_OP_MUL) return (T)((U)x * (U)y);
            else {
                const bool z = y == 0;
                if (z && ((vmask(c) & c.inr) >> r & 1)) c.err |= 1u;
                if (z) return (T)0;
                if constexpr (dt_signed(A::dt)) return y == -1 ? (T)((U)0 - (U)x) : x / y;
                else return x / y;
            }
        }
    }
    RDF_DEFAULT_EVAL_ROWS(Bin)
    static std::string sig() { return "(" + std::to_string(OP) + " " + A::sig() + " " + B::sig() + ")"; }
};

template <int OP, class A>
struct Un {
    static_assert(un_supported(OP, A::dt), "unary operator / operand type not implemented by the specialised kernels");
    static constexpr int dt = OP == RDF_OP_NOT ? RDF_BOOL : A::dt;
    static constexpr int ncols = A::ncols;
    static constexpr int width = A::width;
    template <int k> static constexpr int colw() { return A::template colw<k>(); }
    using T = typename CType<dt>::T;
    template <class C> static __device__ __forceinline__ uint32_t vmask(const C& c) { return A::vmask(c); }
    template <int r, class C> static __device__ __forceinline__ T eval(C& c) {
        const auto x = A::template eval<r>(c);
        if constexpr (OP == RDF_OP_NOT) return !x;
        else if constexpr (OP == RDF_OP_HOUR_S) return (T)hour_of<1>((int64_t)x);
        else if constexpr (OP == RDF_OP_HOUR_MS) return (T)hour_of<1000>((int64_t)x);
        else if constexpr (OP == RDF_OP_HOUR_US) return (T)hour_of<1000000>((int64_t)x);
        else if constexpr (OP == RDF_OP_HOUR_NS) return (T)hour_of<1000000000>((int64_t)x);
        else if constexpr (OP == RDF_OP_HOUR_DAY) return (T)0;
        else if constexpr (dt_signed(A::dt)) { using U = typename std::make_unsigned<T>::type; return x < 0 ? (T)((U)0 - (U)x) : x; }  // abs, MIN wraps
        else if constexpr (A::dt == RDF_F32 && OP == RDF_OP_DEGREES) return x * 57.2957795130823208767981548141051703f;
        else if constexpr (A::dt == RDF_F32 && OP == RDF_OP_RADIANS) return x * (3.14159265358979323846264338327950288f / 180.0f);
        else if constexpr (OP == RDF_OP_ABS) return fabs(x);
        else if constexpr (OP == RDF_OP_ACOS) return acos(x);
        else if constexpr (OP == RDF_OP_ASIN) return asin(x);
        else if constexpr (OP == RDF_OP_ATAN) return atan(x);
        else if constexpr (OP == RDF_OP_CBRT) return cbrt(x);
        else if constexpr (OP == RDF_OP_CEIL) return ceil(x);
        else if constexpr (OP == RDF_OP_COS) return rdf_cos(x);
        else if constexpr (OP == RDF_OP_COSH) return cosh(x);
        else if constexpr (OP == RDF_OP_DEGREES) return x * (180.0 / 3.14159265358979323846264338327950288);
        else if constexpr (OP == RDF_OP_EXP) return exp(x);
        else if constexpr (OP == RDF_OP_EXPM1) return expm1(x);
        else if constexpr (OP == RDF_OP_FLOOR) return floor(x);
        else if constexpr (OP == RDF_OP_LOG10) return log10(x);
        else if constexpr (OP == RDF_OP_LOG2) return log2(x);
        else if constexpr (OP == RDF_OP_RADIANS) return x * (3.14159265358979323846264338327950288 / 180.0);
        else if constexpr (OP == RDF_OP_ROUND) return round(x);
        else if constexpr (OP == RDF_OP_SIN) return rdf_sin(x);
        else if constexpr (OP == RDF_OP_SINH) return sinh(x);
        else if constexpr (OP == RDF_OP_SQRT) return sqrt(x);
        else if constexpr (OP == RDF_OP_TAN) return rdf_tan(x);
        else if constexpr (OP == RDF_OP_COT) return (T)1 / rdf_tan(x);
        else if constexpr (OP == RDF_OP_SEC) return (T)1 / rdf_cos(x);
        else if constexpr (OP == RDF_OP_CSC) return (T)1 / rdf_sin(x);
        else return tanh(x);
    }
    template <int R, class C> static __device__ __forceinline__ void eval_rows(C& c, T (&out)[R]) {
        constexpr bool trig = dt_float(A::dt) && (OP == RDF_OP_SIN || OP == RDF_OP_COS || OP == RDF_OP_TAN || OP == RDF_OP_COT ||
                                                  OP == RDF_OP_SEC || OP == RDF_OP_CSC);
        if constexpr (trig) {   // one wave-wide argument test for the R rows, then R interleaved branch-free chains
            T a[R];
            A::template eval_rows<R>(c, a);
            rdf_trig_rows<(OP == RDF_OP_SIN || OP == RDF_OP_CSC) ? 0 : (OP == RDF_OP_COS || OP == RDF_OP_SEC) ? 1 : 2, R>(a, out);
            if constexpr (OP == RDF_OP_COT || OP == RDF_OP_SEC || OP == RDF_OP_CSC) {
#pragma unroll
                for (int r = 0; r < R; ++r) out[r] = (T)1 / out[r];
            }
        } else eval_each<Un, R>(c, out);
    }
    static std::string sig() { return "[" + std::to_string(OP) + " " + A::sig() + "]"; }
};

// arrow::compute::cast of the reference's era: num::cast::cast per element, NULL where the target type cannot represent
// the value.  A lossy cast therefore narrows the validity mask by looking at the values.
constexpr bool cast_lossy(int from, int to) {
    if (from == to || to == RDF_BOOL || to == RDF_F32 || to == RDF_F64 || from == RDF_BOOL) return false;
    if (dt_float(from)) return true;
    const bool fs = dt_signed(from), ts = dt_signed(to);
    const int fb = CType<RDF_I8>::width * 0 + (from == RDF_I8 || from == RDF_U8 ? 1 : from == RDF_I16 || from == RDF_U16 ? 2 : from == RDF_I32 || from == RDF_U32 ? 4 : 8);
    const int tb = to == RDF_I8 || to == RDF_U8 ? 1 : to == RDF_I16 || to == RDF_U16 ? 2 : to == RDF_I32 || to == RDF_U32 ? 4 : 8;
    if (fs == ts) return tb < fb;
    if (fs) return true;
    return tb <= fb;
}
template <int TO, class X>
__device__ __forceinline__ bool cast_fits(X x) {
    using T = typename CType<TO>::T;
    constexpr int tbits = 8 * (int)sizeof(T);
    if constexpr (std::is_floating_point<X>::value) {
        const double f = (double)x;
        if constexpr (dt_signed(TO)) {
            constexpr double lim = tbits == 64 ? 9223372036854775808.0 : (double)(1ll << (tbits - 1));
            return tbits == 64 ? (f >= -lim && f < lim) : (f > -lim - 1.0 && f < lim);
        } else {
            constexpr double lim = tbits == 64 ? 18446744073709551616.0 : (double)(1ull << (tbits % 64));
            return f > -1.0 && f < lim;
        }
    } else if constexpr (std::is_signed<X>::value) {
        if constexpr (dt_signed(TO)) return (int64_t)x >= (int64_t)std::numeric_limits<T>::min() && (int64_t)x <= (int64_t)std::numeric_limits<T>::max();
        else return x >= 0 && (uint64_t)x <= (uint64_t)std::numeric_limits<T>::max();
    } else {
        return (uint64_t)x <= (uint64_t)std::numeric_limits<T>::max();
    }
}
template <int TO, class A, int R, int r, class C>
__device__ __forceinline__ void cast_fit_rows(C& c, uint32_t& m) {
    if constexpr (r < R) {
        if (!cast_fits<TO>(A::template eval<r>(c))) m &= ~(1u << r);
        cast_fit_rows<TO, A, R, r + 1>(c, m);
    }
}

template <int TO, class A>
struct Cast {
    static constexpr int dt = TO;
    static constexpr int ncols = A::ncols;
    static constexpr int width = merge_width(A::width, CType<TO>::width);   // spec_kernel: casts keep the element width
    template <int k> static constexpr int colw() { return A::template colw<k>(); }
    using T = typename CType<TO>::T;
    template <class C> static __device__ __forceinline__ uint32_t vmask(const C& c) {
        uint32_t m = A::vmask(c);
        if constexpr (cast_lossy(A::dt, TO)) cast_fit_rows<TO, A, C::rows, 0>(const_cast<C&>(c), m);
        return m;
    }
    template <int r, class C> static __device__ __forceinline__ T eval(C& c) {
        const auto x = A::template eval<r>(c);
        if constexpr (TO == RDF_BOOL) return x != 0;
        else if constexpr (cast_lossy(A::dt, TO)) return cast_fits<TO>(x) ? (T)x : (T)0;   // (a float out of range must not reach the conversion)
        else return (T)x;
    }
    RDF_DEFAULT_EVAL_ROWS(Cast)
    static std::string sig() { return "{" + std::to_string(TO) + " " + A::sig() + "}"; }
};

// ---- runtime-operator nodes (f64): ONE compiled kernel per tree SHAPE, the operators are wave-uniform kernel
// arguments (c.rt[SLOT] = rdf_op | swap << 8), so a fused Calculate chain that is not in the exact catalog still runs
// as straight-line code: the switch below is a handful of scalar compares per row, not an interpreter.
template <int SLOT, class A, class B>
struct ArithRT {   // add / subtract / multiply / divide on f64 / f32 or (wrapping) i64 / u64 / i32 / u32
    static_assert(A::dt == B::dt && (CType<A::dt>::width == 8 || CType<A::dt>::width == 4 || CType<A::dt>::width == 2), "runtime-op arithmetic is instantiated for the 8-, 4- and 2-byte numeric types");
    static constexpr int dt = A::dt;
    static constexpr int ncols = A::ncols > B::ncols ? A::ncols : B::ncols;
    static constexpr int width = merge_width(A::width, B::width);
    template <int k> static constexpr int colw() { return cmax(A::template colw<k>(), B::template colw<k>()); }
    using T = typename CType<dt>::T;
    template <class C> static __device__ __forceinline__ uint32_t vmask(const C& c) { return A::vmask(c) & B::vmask(c); }
    template <int r, class C> static __device__ __forceinline__ T eval(C& c) {
        const T x0 = A::template eval<r>(c), y0 = B::template eval<r>(c);
        // The operator word is wave-uniform: every test below is a scalar compare + branch.  The host never sets the swap bit
        // on the commutative operators and a swapped subtraction is its own case, so the common operators cost no per-lane
        // operand select (two to four v_cndmask per node and row otherwise); only a swapped division selects.
        // (Measured: +3-6 % on the 8-byte types, where a select is two v_cndmask per operand; on the 4- and 2-byte types the
        // extra branches cost what the single-instruction selects saved, so those keep the select.)
        const int rt = c.rt[SLOT];
        const int op = rt & 0xFF;
        if constexpr (sizeof(T) == 8 && dt_float(dt)) {
            if (rt == RDF_OP_ADD) return x0 + y0;
            if (rt == RDF_OP_MUL) return x0 * y0;
            if (rt == RDF_OP_SUB) return x0 - y0;
            if (rt == (RDF_OP_SUB | 0x100)) return y0 - x0;
        } else if constexpr (sizeof(T) == 8) {
            using U0 = typename std::make_unsigned<T>::type;
            if (rt == RDF_OP_ADD) return (T)((U0)x0 + (U0)y0);
            if (rt == RDF_OP_MUL) return (T)((U0)x0 * (U0)y0);
            if (rt == RDF_OP_SUB) return (T)((U0)x0 - (U0)y0);
            if (rt == (RDF_OP_SUB | 0x100)) return (T)((U0)y0 - (U0)x0);
        }
        const bool sw = (rt >> 8) & 1;
        const T x = sw ? y0 : x0, y = sw ? x0 : y0;
        if constexpr (dt_float(dt)) {
            if (op == RDF_OP_ADD) return x + y;
            if (op == RDF_OP_SUB) return x - y;
            if (op == RDF_OP_MUL) return x * y;
            const bool z = y == (T)0;
            if (z && ((vmask(c) & c.inr) >> r & 1)) c.err |= 1u;
            return z ? (T)0 : x / y;
        } else {
            using U = typename std::make_unsigned<T>::type;
            if (op == RDF_OP_ADD) return (T)((U)x + (U)y);
            if (op == RDF_OP_SUB) return (T)((U)x - (U)y);
            if (op == RDF_OP_MUL) return (T)((U)x * (U)y);
            const bool z = y == 0;
            if (z && ((vmask(c) & c.inr) >> r & 1)) c.err |= 1u;
            if (z) return (T)0;
            if constexpr (dt_signed(dt)) return y == -1 ? (T)((U)0 - (U)x) : x / y;   // MIN / -1 wraps
            else return x / y;
        }
    }
    template <int R, class C> static __device__ __forceinline__ void eval_rows(C& c, T (&out)[R]) {
        T a[R], b[R];
        A::template eval_rows<R>(c, a);
        B::template eval_rows<R>(c, b);
        const int rt = c.rt[SLOT];
        using U0 = typename std::conditional<std::is_floating_point<T>::value, T, typename std::make_unsigned<typename std::conditional<std::is_floating_point<T>::value, int, T>::type>::type>::type;
        if (rt == RDF_OP_ADD) {
#pragma unroll
            for (int r = 0; r < R; ++r) out[r] = (T)((U0)a[r] + (U0)b[r]);
            return;
        }
        if (rt == RDF_OP_MUL) {
#pragma unroll
            for (int r = 0; r < R; ++r) out[r] = (T)((U0)a[r] * (U0)b[r]);
            return;
        }
        if (rt == RDF_OP_SUB) {
#pragma unroll
            for (int r = 0; r < R; ++r) out[r] = (T)((U0)a[r] - (U0)b[r]);
            return;
        }
        if (rt == (RDF_OP_SUB | 0x100)) {
#pragma unroll
            for (int r = 0; r < R; ++r) out[r] = (T)((U0)b[r] - (U0)a[r]);
            return;
        }
        // division (either operand order); a zero divisor at a live row raises the error flag
        const bool sw = (rt >> 8) & 1;
        const uint32_t live = vmask(c) & c.inr;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const T x = sw ? b[r] : a[r], y = sw ? a[r] : b[r];
            const bool z = y == (T)0;
            if (z && ((live >> r) & 1)) c.err |= 1u;
            if constexpr (dt_float(dt)) out[r] = z ? (T)0 : x / y;
            else if constexpr (dt_signed(dt)) { using U = typename std::make_unsigned<T>::type; out[r] = z ? (T)0 : (y == (T)-1 ? (T)((U)0 - (U)x) : (T)(x / y)); }
            else out[r] = z ? (T)0 : (T)(x / y);
        }
    }
    static std::string sig() { return "(A" + std::to_string(SLOT) + " " + A::sig() + " " + B::sig() + ")"; }
};
template <int SLOT, class A>
struct TrigRT {    // sin / cos / tan: the three the reference's Evaluate::calculate dispatches (src/evaluation.rs:250-293)
    static_assert(dt_float(A::dt), "runtime-op trig is instantiated for f64 and f32");
    static constexpr int dt = A::dt;
    static constexpr int ncols = A::ncols;
    static constexpr int width = A::width;
    template <int k> static constexpr int colw() { return A::template colw<k>(); }
    using T = typename CType<dt>::T;
    template <class C> static __device__ __forceinline__ uint32_t vmask(const C& c) { return A::vmask(c); }
    template <int r, class C> static __device__ __forceinline__ T eval(C& c) {
        const T x = A::template eval<r>(c);
        const int op = c.rt[SLOT] & 0xFF;
        if (op == RDF_OP_SIN) return rdf_sin(x);
        if (op == RDF_OP_COS) return rdf_cos(x);
        return rdf_tan(x);
    }
    template <int R, class C> static __device__ __forceinline__ void eval_rows(C& c, T (&out)[R]) {
        T a[R];
        A::template eval_rows<R>(c, a);
        const int op = c.rt[SLOT] & 0xFF;
        if (op == RDF_OP_SIN) rdf_trig_rows<0, R>(a, out);
        else if (op == RDF_OP_COS) rdf_trig_rows<1, R>(a, out);
        else rdf_trig_rows<2, R>(a, out);
    }
    static std::string sig() { return "[T" + std::to_string(SLOT) + " " + A::sig() + "]"; }
};
template <int SLOT, class A, class B>
struct CmpRT {     // gt / ge / eq / ne / lt / le, both sides as f64 (src/expression.rs:844-845)
    static constexpr int dt = RDF_BOOL;
    static constexpr int ncols = A::ncols > B::ncols ? A::ncols : B::ncols;
    static constexpr int width = merge_width(A::width, B::width);
    template <int k> static constexpr int colw() { return cmax(A::template colw<k>(), B::template colw<k>()); }
    using T = bool;
    template <class C> static __device__ __forceinline__ uint32_t vmask(const C& c) { return A::vmask(c) & B::vmask(c); }
    template <int r, class C> static __device__ __forceinline__ bool eval(C& c) {
        const double a = (double)A::template eval<r>(c), b = (double)B::template eval<r>(c);
        const int op = c.rt[SLOT] & 0xFF;   // (a swapped comparison arrives as the mirrored operator: no operand select)
        if (op == RDF_OP_GT) return a > b;
        if (op == RDF_OP_GE) return a >= b;
        if (op == RDF_OP_EQ) return a == b;
        if (op == RDF_OP_NE) return a != b;
        if (op == RDF_OP_LT) return a < b;
        return a <= b;
    }
    // bit r = the comparison holds at row r (operator tested once, not per row)
    static constexpr bool has_row_mask = true;
    template <int R, class C> static __device__ __forceinline__ uint32_t eval_mask(C& c) {
        typename A::T a[R];
        typename B::T b[R];
        A::template eval_rows<R>(c, a);
        B::template eval_rows<R>(c, b);
        const int op = c.rt[SLOT] & 0xFF;
        uint32_t m = 0;
#define RDF_CMP_ROWS(REL) _Pragma("unroll") for (int r = 0; r < R; ++r) m |= (uint32_t)((double)a[r] REL (double)b[r]) << r
        if (op == RDF_OP_GT) { RDF_CMP_ROWS(>); }
        else if (op == RDF_OP_GE) { RDF_CMP_ROWS(>=); }
        else if (op == RDF_OP_EQ) { RDF_CMP_ROWS(==); }
        else if (op == RDF_OP_NE) { RDF_CMP_ROWS(!=); }
        else if (op == RDF_OP_LT) { RDF_CMP_ROWS(<); }
        else { RDF_CMP_ROWS(<=); }
#undef RDF_CMP_ROWS
        return m;
    }
    template <int R, class C> static __device__ __forceinline__ void eval_rows(C& c, bool (&out)[R]) {
        const uint32_t m = eval_mask<R>(c);
#pragma unroll
        for (int r = 0; r < R; ++r) out[r] = (m >> r) & 1;
    }
    static std::string sig() { return "(C" + std::to_string(SLOT) + " " + A::sig() + " " + B::sig() + ")"; }
};
template <int SLOT, class A, class B>
struct LogicRT {   // and / or on booleans, NULL if either side is NULL (arrow::compute::and / or)
    static_assert(A::dt == RDF_BOOL && B::dt == RDF_BOOL, "logic over boolean operands");
    static constexpr int dt = RDF_BOOL;
    static constexpr int ncols = A::ncols > B::ncols ? A::ncols : B::ncols;
    static constexpr int width = merge_width(A::width, B::width);
    template <int k> static constexpr int colw() { return cmax(A::template colw<k>(), B::template colw<k>()); }
    using T = bool;
    template <class C> static __device__ __forceinline__ uint32_t vmask(const C& c) { return A::vmask(c) & B::vmask(c); }
    template <int r, class C> static __device__ __forceinline__ bool eval(C& c) {
        const bool x = A::template eval<r>(c), y = B::template eval<r>(c);
        return (c.rt[SLOT] & 0xFF) == RDF_OP_AND ? (x && y) : (x || y);
    }
    static constexpr bool has_row_mask = true;
    template <int R, class C> static __device__ __forceinline__ uint32_t eval_mask(C& c) {
        const uint32_t x = A::template eval_mask<R>(c), y = B::template eval_mask<R>(c);
        return (c.rt[SLOT] & 0xFF) == RDF_OP_AND ? (x & y) : (x | y);
    }
    template <int R, class C> static __device__ __forceinline__ void eval_rows(C& c, bool (&out)[R]) {
        const uint32_t m = eval_mask<R>(c);
#pragma unroll
        for (int r = 0; r < R; ++r) out[r] = (m >> r) & 1;
    }
    static std::string sig() { return "(G" + std::to_string(SLOT) + " " + A::sig() + " " + B::sig() + ")"; }
};

struct None {
    static constexpr int ncols = 0;
    static constexpr int width = 0;
    template <int k> static constexpr int colw() { return 0; }
    static std::string sig() { return "-"; }
};

}  // namespace rdfk
