/*
 * rdf_oracle.c — CPU oracle (see rdf_oracle.h).  TEST INFRASTRUCTURE ONLY, never shipped.
 *
 * Build: make -C oracle   (gcc -O3 -march=native -ffp-contract=off -shared -fPIC)
 *
 * The reference delegates its inner loops to the `arrow` crate (apache/arrow, branch
 * rust-parquet-arrow-writer, un-pinned: Cargo.toml:9; API era ~ arrow-rs 2.0.0, Aug-Oct 2020).
 * That crate is not under /root/reference, so its kernels are restated here from the Arrow
 * columnar semantics listed in SURVEY.md §8c, anchored on the reference's own call sites.
 */
#include "rdf_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static __thread char g_err[256];
const char* ora_last_error(void) { return g_err; }
#define FAIL(code, ...) do { snprintf(g_err, sizeof g_err, __VA_ARGS__); return (code); } while (0)

/* ------------------------------------------------------------------ small helpers */

static inline int bit_get(const uint8_t* b, int64_t i) { return (b[i >> 3] >> (i & 7)) & 1; }
static inline void bit_set(uint8_t* b, int64_t i) { b[i >> 3] |= (uint8_t)(1u << (i & 7)); }
static inline void bit_clr(uint8_t* b, int64_t i) { b[i >> 3] &= (uint8_t)~(1u << (i & 7)); }
static inline void bit_put(uint8_t* b, int64_t i, int v) { if (v) bit_set(b, i); else bit_clr(b, i); }

static int dtype_size(int32_t dt) {
    switch (dt) {
        case RDF_I8: case RDF_U8: return 1;
        case RDF_I16: case RDF_U16: return 2;
        case RDF_I32: case RDF_U32: case RDF_F32: return 4;
        case RDF_I64: case RDF_U64: case RDF_F64: return 8;
        default: return 0; /* BOOL is bit-packed */
    }
}
static int is_float(int32_t dt) { return dt == RDF_F32 || dt == RDF_F64; }
static int is_signed_int(int32_t dt) { return dt >= RDF_I8 && dt <= RDF_I64; }
static int is_numeric(int32_t dt) { return dt >= RDF_I8 && dt <= RDF_F64; }

/* PrimitiveArray::is_valid(i) (arrow array data: bitmap at offset + i). */
static inline int arr_valid(const rdf_array* a, int64_t i) {
    return a->validity == NULL || bit_get(a->validity, a->offset + i);
}

/* Typed element access; integers are widened to 64 bits (sign- or zero-extended). */
static inline double arr_f64(const rdf_array* a, int64_t i) {
    int64_t k = a->offset + i;
    switch (a->dtype) {
        case RDF_I8: return (double)((const int8_t*)a->values)[k];
        case RDF_I16: return (double)((const int16_t*)a->values)[k];
        case RDF_I32: return (double)((const int32_t*)a->values)[k];
        case RDF_I64: return (double)((const int64_t*)a->values)[k];
        case RDF_U8: return (double)((const uint8_t*)a->values)[k];
        case RDF_U16: return (double)((const uint16_t*)a->values)[k];
        case RDF_U32: return (double)((const uint32_t*)a->values)[k];
        case RDF_U64: return (double)((const uint64_t*)a->values)[k];
        case RDF_F32: return (double)((const float*)a->values)[k];
        case RDF_F64: return ((const double*)a->values)[k];
        case RDF_BOOL: return bit_get((const uint8_t*)a->values, k) ? 1.0 : 0.0;
        default: return 0.0;
    }
}

static void out_begin(rdf_out* o, int64_t len) {
    o->length = len;
    o->null_count = 0;
    if (o->validity) memset(o->validity, 0xFF, (size_t)((len + 7) / 8));
}
static inline void out_null(rdf_out* o, int64_t i) {
    bit_clr(o->validity, i);
    o->null_count++;
}

/* ------------------------------------------------------------------ ScalarFunctions: binary
 * src/functions/scalar.rs:16-103 -> arrow::compute::{add,subtract,multiply,divide}(chunk_a, chunk_b):
 *   length mismatch -> ComputeError; validity = AND of inputs; integers wrap; divide raises
 *   DivideByZero for a zero divisor at a valid slot (ints and floats).
 * src/functions/scalar.rs:499-523 math_op (atan2 :148, hypot :274, log :291): same shape, op(l,r)
 *   on valid slots via libm (num::Float = std f64/f32 methods = platform libm). */

#define BIN_INT_CASE(DT, T, UT)                                                                   \
    case DT: {                                                                                    \
        const T* x = (const T*)a->values + a->offset;                                             \
        const T* y = (const T*)b->values + b->offset;                                             \
        T* z = (T*)o->values;                                                                     \
        for (int64_t i = 0; i < n; i++) {                                                         \
            int v = arr_valid(a, i) && arr_valid(b, i);                                           \
            if (!v) { z[i] = 0; out_null(o, i); continue; }                                       \
            switch (op) {                                                                         \
                case RDF_OP_ADD: z[i] = (T)((UT)x[i] + (UT)y[i]); break;                          \
                case RDF_OP_SUB: z[i] = (T)((UT)x[i] - (UT)y[i]); break;                          \
                case RDF_OP_MUL: z[i] = (T)((UT)x[i] * (UT)y[i]); break;                          \
                default: /* DIV */                                                                \
                    if (y[i] == 0) FAIL(RDF_DIVIDE_BY_ZERO, "Divide by zero error");              \
                    if ((T)-1 < 0 && y[i] == (T)-1) z[i] = (T)((UT)0 - (UT)x[i]); /* MIN/-1 wraps */ \
                    else z[i] = (T)(x[i] / y[i]);                                                 \
            }                                                                                     \
        }                                                                                         \
    } break;

#define BIN_FLT_CASE(DT, T, ATAN2, HYPOT, LOG)                                                    \
    case DT: {                                                                                    \
        const T* x = (const T*)a->values + a->offset;                                             \
        const T* y = (const T*)b->values + b->offset;                                             \
        T* z = (T*)o->values;                                                                     \
        for (int64_t i = 0; i < n; i++) {                                                         \
            int v = arr_valid(a, i) && arr_valid(b, i);                                           \
            if (!v) { z[i] = 0; out_null(o, i); continue; }                                       \
            switch (op) {                                                                         \
                case RDF_OP_ADD: z[i] = x[i] + y[i]; break;                                       \
                case RDF_OP_SUB: z[i] = x[i] - y[i]; break;                                       \
                case RDF_OP_MUL: z[i] = x[i] * y[i]; break;                                       \
                case RDF_OP_DIV:                                                                  \
                    if (y[i] == 0) FAIL(RDF_DIVIDE_BY_ZERO, "Divide by zero error");              \
                    z[i] = x[i] / y[i]; break;                                                    \
                case RDF_OP_ATAN2: z[i] = ATAN2(x[i], y[i]); break;                               \
                case RDF_OP_HYPOT: z[i] = HYPOT(x[i], y[i]); break;                               \
                default: /* LOG: self.ln() / base.ln() */ z[i] = LOG(x[i]) / LOG(y[i]);           \
            }                                                                                     \
        }                                                                                         \
    } break;

static rdf_status binary_chunk(int32_t op, const rdf_array* a, const rdf_array* b, rdf_out* o) {
    if (a->length != b->length)
        FAIL(RDF_COMPUTE_ERROR, "Cannot perform math operation on arrays of different length");
    if (a->dtype != b->dtype || o->dtype != a->dtype) FAIL(RDF_INVALID_ARGUMENT, "binary op: dtype mismatch");
    if (!is_numeric(a->dtype)) FAIL(RDF_INVALID_ARGUMENT, "binary op: numeric type required");
    if (op >= RDF_OP_ATAN2 && !is_float(a->dtype)) FAIL(RDF_INVALID_ARGUMENT, "math_op: float type required");
    int64_t n = a->length;
    if (o->capacity < n) FAIL(RDF_MEMORY_ERROR, "output capacity too small");
    if ((a->validity || b->validity) && !o->validity) FAIL(RDF_INVALID_ARGUMENT, "output validity buffer required");
    out_begin(o, n);
    switch (a->dtype) {
        BIN_INT_CASE(RDF_I8, int8_t, uint8_t)
        BIN_INT_CASE(RDF_I16, int16_t, uint16_t)
        BIN_INT_CASE(RDF_I32, int32_t, uint32_t)
        BIN_INT_CASE(RDF_I64, int64_t, uint64_t)
        BIN_INT_CASE(RDF_U8, uint8_t, uint8_t)
        BIN_INT_CASE(RDF_U16, uint16_t, uint16_t)
        BIN_INT_CASE(RDF_U32, uint32_t, uint32_t)
        BIN_INT_CASE(RDF_U64, uint64_t, uint64_t)
        BIN_FLT_CASE(RDF_F32, float, atan2f, hypotf, logf)
        BIN_FLT_CASE(RDF_F64, double, atan2, hypot, log)
        default: break;
    }
    return RDF_OK;
}

rdf_status ora_binary(int32_t op, const rdf_array* a, const rdf_array* b, int64_t nchunks, rdf_out* out) {
    if (op < RDF_OP_ADD || op > RDF_OP_LOG) FAIL(RDF_INVALID_ARGUMENT, "not a binary op: %d", op);
    for (int64_t c = 0; c < nchunks; c++) { /* left.iter().zip(right.iter()).map(...) scalar.rs:28-31 */
        rdf_status s = binary_chunk(op, &a[c], &b[c], &out[c]);
        if (s != RDF_OK) return s;
    }
    return RDF_OK;
}

/* ------------------------------------------------------------------ ScalarFunctions: unary
 * src/functions/scalar.rs:525-540 scalar_op: for i in 0..len { if null -> append_null else
 * append_value(op(value(i))) } with op = num::Float::* (libm) or num::abs. */

static double deg_f64(double x) { return x * (180.0 / 3.14159265358979323846264338327950288); }
static double rad_f64(double x) { return x * (3.14159265358979323846264338327950288 / 180.0); }
/* Rust core: f32::to_degrees multiplies by the literal 57.2957795130823208767981548141051703_f32. */
static float deg_f32(float x) { return x * 57.2957795130823208767981548141051703f; }
static float rad_f32(float x) { const float pi = 3.14159265358979323846264338327950288f; return x * (pi / 180.0f); }

static double un_f64(int32_t op, double x) {
    switch (op) {
        case RDF_OP_ABS: return fabs(x);
        case RDF_OP_ACOS: return acos(x);
        case RDF_OP_ASIN: return asin(x);
        case RDF_OP_ATAN: return atan(x);
        case RDF_OP_CBRT: return cbrt(x);
        case RDF_OP_CEIL: return ceil(x);
        case RDF_OP_COS: return cos(x);
        case RDF_OP_COSH: return cosh(x);
        case RDF_OP_DEGREES: return deg_f64(x);
        case RDF_OP_EXP: return exp(x);
        case RDF_OP_EXPM1: return expm1(x);
        case RDF_OP_FLOOR: return floor(x);
        case RDF_OP_LOG10: return log10(x);
        case RDF_OP_LOG2: return log2(x);
        case RDF_OP_RADIANS: return rad_f64(x);
        case RDF_OP_ROUND: return round(x); /* half away from zero, like f64::round */
        case RDF_OP_SIN: return sin(x);
        case RDF_OP_SINH: return sinh(x);
        case RDF_OP_SQRT: return sqrt(x);
        case RDF_OP_TAN: return tan(x);
        case RDF_OP_COT: return 1.0 / tan(x);   /* ScalarFunction::{Cotangent,Secant,Cosecant} (src/expression.rs:670-672): */
        case RDF_OP_SEC: return 1.0 / cos(x);   /* named by the plan, never evaluated by the reference (:487-489 panic) --    */
        case RDF_OP_CSC: return 1.0 / sin(x);   /* the textbook reciprocals, PARITY UNPINNED BY THE REFERENCE                 */
        default: return tanh(x);
    }
}
static float un_f32(int32_t op, float x) {
    switch (op) {
        case RDF_OP_ABS: return fabsf(x);
        case RDF_OP_ACOS: return acosf(x);
        case RDF_OP_ASIN: return asinf(x);
        case RDF_OP_ATAN: return atanf(x);
        case RDF_OP_CBRT: return cbrtf(x);
        case RDF_OP_CEIL: return ceilf(x);
        case RDF_OP_COS: return cosf(x);
        case RDF_OP_COSH: return coshf(x);
        case RDF_OP_DEGREES: return deg_f32(x);
        case RDF_OP_EXP: return expf(x);
        case RDF_OP_EXPM1: return expm1f(x);
        case RDF_OP_FLOOR: return floorf(x);
        case RDF_OP_LOG10: return log10f(x);
        case RDF_OP_LOG2: return log2f(x);
        case RDF_OP_RADIANS: return rad_f32(x);
        case RDF_OP_ROUND: return roundf(x);
        case RDF_OP_SIN: return sinf(x);
        case RDF_OP_SINH: return sinhf(x);
        case RDF_OP_SQRT: return sqrtf(x);
        case RDF_OP_TAN: return tanf(x);
        case RDF_OP_COT: return 1.0f / tanf(x);
        case RDF_OP_SEC: return 1.0f / cosf(x);
        case RDF_OP_CSC: return 1.0f / sinf(x);
        default: return tanhf(x);
    }
}

#define ABS_INT_CASE(DT, T, UT)                                                                   \
    case DT: {                                                                                    \
        const T* x = (const T*)a->values + a->offset;                                             \
        T* z = (T*)o->values;                                                                     \
        for (int64_t i = 0; i < n; i++) {                                                         \
            if (!arr_valid(a, i)) { z[i] = 0; out_null(o, i); continue; }                         \
            z[i] = x[i] < 0 ? (T)((UT)0 - (UT)x[i]) : x[i]; /* num::abs; MIN wraps */             \
        }                                                                                         \
    } break;

static rdf_status unary_chunk(int32_t op, const rdf_array* a, rdf_out* o) {
    if (o->dtype != a->dtype) FAIL(RDF_INVALID_ARGUMENT, "unary op: dtype mismatch");
    if (op == RDF_OP_ABS) {
        if (!(is_float(a->dtype) || is_signed_int(a->dtype)))
            FAIL(RDF_INVALID_ARGUMENT, "abs: signed numeric type required"); /* T::Native: Signed, scalar.rs:109 */
    } else if (!is_float(a->dtype)) {
        FAIL(RDF_INVALID_ARGUMENT, "float type required"); /* T::Native: num_traits::Float */
    }
    int64_t n = a->length;
    if (o->capacity < n) FAIL(RDF_MEMORY_ERROR, "output capacity too small");
    if (a->validity && !o->validity) FAIL(RDF_INVALID_ARGUMENT, "output validity buffer required");
    out_begin(o, n);
    switch (a->dtype) {
        ABS_INT_CASE(RDF_I8, int8_t, uint8_t)
        ABS_INT_CASE(RDF_I16, int16_t, uint16_t)
        ABS_INT_CASE(RDF_I32, int32_t, uint32_t)
        ABS_INT_CASE(RDF_I64, int64_t, uint64_t)
        case RDF_F32: {
            const float* x = (const float*)a->values + a->offset;
            float* z = (float*)o->values;
            for (int64_t i = 0; i < n; i++) {
                if (!arr_valid(a, i)) { z[i] = 0; out_null(o, i); continue; }
                z[i] = un_f32(op, x[i]);
            }
        } break;
        default: {
            const double* x = (const double*)a->values + a->offset;
            double* z = (double*)o->values;
            for (int64_t i = 0; i < n; i++) {
                if (!arr_valid(a, i)) { z[i] = 0; out_null(o, i); continue; }
                z[i] = un_f64(op, x[i]);
            }
        }
    }
    return RDF_OK;
}

rdf_status ora_unary(int32_t op, const rdf_array* a, int64_t nchunks, rdf_out* out) {
    if (!((op >= RDF_OP_ABS && op <= RDF_OP_TANH) || (op >= RDF_OP_COT && op <= RDF_OP_CSC))) FAIL(RDF_INVALID_ARGUMENT, "not a unary op: %d", op);
    for (int64_t c = 0; c < nchunks; c++) { /* array.iter().map(|a| scalar_op(a, ..)) scalar.rs:111 */
        rdf_status s = unary_chunk(op, &a[c], &out[c]);
        if (s != RDF_OK) return s;
    }
    return RDF_OK;
}

/* ------------------------------------------------------------------ cast
 * src/evaluation.rs:296-315 -> arrow::compute::cast per chunk.  The arrow crate of the reference's era (branch
 * rust-parquet-arrow-writer, Aug-Oct 2020, not under /root/reference) casts numeric arrays element by element through
 * num::cast::cast (num-traits 0.2, Cargo.toml:14-15) and appends NULL where that returns None ("some casts return None,
 * such as a negative value to u{8|16|32|64}"):
 *   int -> int      Some iff the value is representable in the target type;
 *   int -> float    always Some (rounded);
 *   float -> int    None for NaN; else truncation toward zero, Some iff the truncated value fits: num-traits tests
 *                   MIN - 1 < f < MAX + 1 (float wider than the int) or MIN <= f < MAX + 1 (otherwise), unsigned: -1 < f < MAX + 1;
 *   float -> float  always Some (f64 -> f32 is `as`: rounds, overflows to infinity);
 *   numeric -> Boolean = (x != 0), Boolean -> numeric = 1 / 0 (cast_numeric_to_bool / cast_bool_to_numeric): always valid.
 * Input NULLs stay NULL.  No reference test pins a lossy cast (PARITY UNPINNED BY THE REFERENCE for those); the value
 * rules are checked against numpy / pyarrow where the value is representable (tests/test_oracle_golden.py). */

/* 1 if a cast from -> to can turn a valid slot into NULL */
static int cast_can_null(int32_t from, int32_t to) {
    if (from == to || to == RDF_BOOL || to == RDF_F32 || to == RDF_F64 || from == RDF_BOOL) return 0;
    if (from == RDF_F32 || from == RDF_F64) return 1;               /* float -> int */
    int fs = from <= RDF_I64, ts = to <= RDF_I64;
    int fb = dtype_size(from), tb = dtype_size(to);
    if (fs == ts) return tb < fb;                                     /* same signedness: narrowing only */
    if (fs) return 1;                                                 /* signed -> unsigned: negatives */
    return tb <= fb;                                                  /* unsigned -> signed of the same or a smaller width */
}
/* truncating float -> integer of `bits` bits; *ok = representable per num-traits */
static int64_t f64_to_int_checked(double f, int bits, int is_signed, int* ok) {
    *ok = 0;
    if (f != f) return 0;
    if (is_signed) {
        double lim = ldexp(1.0, bits - 1);
        if (bits == 64 ? !(f >= -lim && f < lim) : !(f > -lim - 1.0 && f < lim)) return 0;
        *ok = 1;
        return (int64_t)f;
    }
    double lim = ldexp(1.0, bits);
    if (!(f > -1.0 && f < lim)) return 0;
    *ok = 1;
    return (int64_t)(uint64_t)f;
}
/* Convert element i of `a` to dtype `to`, written at z[k] (0 where the result is NULL).  Returns 1 if representable. */
static int cast_elem(const rdf_array* a, int64_t i, int32_t to, void* zv, int64_t k) {
    int64_t s = a->offset + i;
    int from = a->dtype;
    /* fetch as (i64 | u64 | f64) by class */
    int64_t si = 0; uint64_t ui = 0; double f = 0; int cls; /* 0 signed, 1 unsigned, 2 float */
    switch (from) {
        case RDF_I8: si = ((const int8_t*)a->values)[s]; cls = 0; break;
        case RDF_I16: si = ((const int16_t*)a->values)[s]; cls = 0; break;
        case RDF_I32: si = ((const int32_t*)a->values)[s]; cls = 0; break;
        case RDF_I64: si = ((const int64_t*)a->values)[s]; cls = 0; break;
        case RDF_U8: ui = ((const uint8_t*)a->values)[s]; cls = 1; break;
        case RDF_U16: ui = ((const uint16_t*)a->values)[s]; cls = 1; break;
        case RDF_U32: ui = ((const uint32_t*)a->values)[s]; cls = 1; break;
        case RDF_U64: ui = ((const uint64_t*)a->values)[s]; cls = 1; break;
        case RDF_F32: f = ((const float*)a->values)[s]; cls = 2; break;
        case RDF_F64: f = ((const double*)a->values)[s]; cls = 2; break;
        default: ui = (uint64_t)bit_get((const uint8_t*)a->values, s); cls = 1; break; /* BOOL */
    }
    if (to == RDF_BOOL) {
        int v = cls == 0 ? (si != 0) : cls == 1 ? (ui != 0) : (f != 0.0);
        bit_put((uint8_t*)zv, k, v);
        return 1;
    }
    if (to == RDF_F64 || to == RDF_F32) {
        double d = cls == 0 ? (double)si : cls == 1 ? (double)ui : f;
        if (to == RDF_F64) ((double*)zv)[k] = d;
        else ((float*)zv)[k] = cls == 0 ? (float)si : cls == 1 ? (float)ui : (float)f;
        return 1;
    }
    int tbits = 8 * dtype_size(to), tsigned = to <= RDF_I64, ok = 1;
    uint64_t bits;
    if (cls == 2) bits = (uint64_t)f64_to_int_checked(f, tbits, tsigned, &ok);
    else {
        bits = cls == 0 ? (uint64_t)si : ui;
        if (tsigned) {
            int64_t hi = tbits == 64 ? INT64_MAX : ((int64_t)1 << (tbits - 1)) - 1, lo = -hi - 1;
            ok = cls == 0 ? (si >= lo && si <= hi) : (ui <= (uint64_t)hi);
        } else {
            uint64_t hi = tbits == 64 ? UINT64_MAX : (((uint64_t)1 << tbits) - 1);
            ok = cls == 0 ? (si >= 0 && (uint64_t)si <= hi) : (ui <= hi);
        }
    }
    if (!ok) bits = 0;
    switch (to) {
        case RDF_I8: case RDF_U8: ((uint8_t*)zv)[k] = (uint8_t)bits; break;
        case RDF_I16: case RDF_U16: ((uint16_t*)zv)[k] = (uint16_t)bits; break;
        case RDF_I32: case RDF_U32: ((uint32_t*)zv)[k] = (uint32_t)bits; break;
        default: ((uint64_t*)zv)[k] = bits; break;
    }
    return ok;
}

static rdf_status cast_chunk(const rdf_array* a, rdf_out* o) {
    int64_t n = a->length;
    if (o->capacity < n) FAIL(RDF_MEMORY_ERROR, "output capacity too small");
    if (!(is_numeric(a->dtype) || a->dtype == RDF_BOOL) || !(is_numeric(o->dtype) || o->dtype == RDF_BOOL))
        FAIL(RDF_INVALID_ARGUMENT, "cast: unsupported type");
    if ((a->validity || cast_can_null(a->dtype, o->dtype)) && !o->validity) FAIL(RDF_INVALID_ARGUMENT, "output validity buffer required");
    out_begin(o, n);
    if (o->dtype == RDF_BOOL) memset(o->values, 0, (size_t)((n + 7) / 8));
    for (int64_t i = 0; i < n; i++) {
        int ok = cast_elem(a, i, o->dtype, o->values, i);
        if (!arr_valid(a, i) || !ok) out_null(o, i);
    }
    return RDF_OK;
}

rdf_status ora_cast(const rdf_array* a, int64_t nchunks, rdf_out* out) {
    for (int64_t c = 0; c < nchunks; c++) {
        rdf_status s = cast_chunk(&a[c], &out[c]);
        if (s != RDF_OK) return s;
    }
    return RDF_OK;
}

/* ScalarFunctions::hour (src/functions/scalar.rs:267-273) -> arrow::compute::hour (arrow crate, not under /root/reference):
 * Time32 / Time64 go through PrimitiveArray::value_as_time = chrono NaiveTime::from_num_seconds_from_midnight(secs, nanos),
 * Date32 / Date64 / Timestamp through value_as_datetime = NaiveDateTime::from_timestamp(secs, nanos), whose date / time
 * split is div_mod_floor(secs, 86400); `.hour()` = second of the day / 3600.  secs = value / units_per_second for the
 * values chrono accepts (non-negative remainder); the others make the reference panic and get floor semantics here.
 * The reference has no test for hour: parity unpinned. */
static int64_t floor_div64(int64_t a, int64_t b) { int64_t q = a / b; return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q; }
static int32_t hour_value(int64_t v, int32_t unit) {
    static const int64_t per_sec[4] = {1, 1000, 1000000, 1000000000};
    if (unit == RDF_TIME_DAY) return 0;                                     /* Date32: midnight of the day */
    int64_t secs = floor_div64(v, per_sec[unit]);
    int64_t sod = secs - floor_div64(secs, 86400) * 86400;                  /* div_mod_floor */
    return (int32_t)(sod / 3600);
}
static rdf_status hour_chunk(const rdf_array* a, int32_t unit, rdf_out* o, int32_t odt) {
    if (a->dtype != RDF_I32 && a->dtype != RDF_I64) FAIL(RDF_COMPUTE_ERROR, "hour does not support type %d", a->dtype);
    if (o->dtype != odt) FAIL(RDF_INVALID_ARGUMENT, "hour: output dtype");
    int64_t n = a->length;
    if (o->capacity < n) FAIL(RDF_MEMORY_ERROR, "output capacity too small");
    if (a->validity && !o->validity) FAIL(RDF_INVALID_ARGUMENT, "output validity buffer required");
    out_begin(o, n);
    for (int64_t i = 0; i < n; i++) {
        int32_t h = 0;
        if (!arr_valid(a, i)) out_null(o, i);
        else h = hour_value(a->dtype == RDF_I32 ? ((const int32_t*)a->values)[a->offset + i] : ((const int64_t*)a->values)[a->offset + i], unit);
        if (odt == RDF_I32) ((int32_t*)o->values)[i] = h; else ((int64_t*)o->values)[i] = h;
    }
    return RDF_OK;
}
rdf_status ora_hour(const rdf_array* a, int64_t nchunks, int32_t unit, rdf_out* out) {
    if (unit < RDF_TIME_SECOND || unit > RDF_TIME_DAY) FAIL(RDF_INVALID_ARGUMENT, "hour: unknown time unit %d", unit);
    for (int64_t c = 0; c < nchunks; c++) {
        rdf_status s = hour_chunk(&a[c], unit, &out[c], RDF_I32);
        if (s != RDF_OK) return s;
    }
    return RDF_OK;
}

/* ------------------------------------------------------------------ AggregateFunctions
 * src/functions/aggregate.rs:12-93. */

/* arrow::compute::sum(chunk): None iff len == null_count, else sequential left fold over valid
 * slots.  aggregate.rs:82-93 adds `unwrap_or(default)` per chunk and returns Some(total). */
#define SUM_CASE(DT, T, UT)                                                                       \
    case DT: {                                                                                    \
        UT tot = 0;                                                                               \
        for (int64_t c = 0; c < nchunks; c++) {                                                   \
            const T* x = (const T*)a[c].values + a[c].offset;                                     \
            UT s = 0;                                                                             \
            for (int64_t i = 0; i < a[c].length; i++)                                             \
                if (arr_valid(&a[c], i)) s = (UT)(s + (UT)x[i]);                                  \
            tot = (UT)(tot + s);                                                                  \
        }                                                                                         \
        *(T*)out_scalar = (T)tot;                                                                 \
    } break;
#define SUM_FLT_CASE(DT, T)                                                                       \
    case DT: {                                                                                    \
        T tot = 0;                                                                                \
        for (int64_t c = 0; c < nchunks; c++) {                                                   \
            const T* x = (const T*)a[c].values + a[c].offset;                                     \
            T s = 0;                                                                              \
            for (int64_t i = 0; i < a[c].length; i++)                                             \
                if (arr_valid(&a[c], i)) s = s + x[i];                                            \
            tot = tot + s;                                                                        \
        }                                                                                         \
        *(T*)out_scalar = tot;                                                                    \
    } break;

rdf_status ora_sum(const rdf_array* a, int64_t nchunks, void* out_scalar, int32_t* out_is_some) {
    if (nchunks < 1) FAIL(RDF_INVALID_ARGUMENT, "sum: a column has at least one chunk"); /* table.rs:25 */
    if (!is_numeric(a[0].dtype)) FAIL(RDF_INVALID_ARGUMENT, "sum: numeric type required");
    int dt = a[0].dtype;
    switch (dt) {
        SUM_CASE(RDF_I8, int8_t, uint8_t)
        SUM_CASE(RDF_I16, int16_t, uint16_t)
        SUM_CASE(RDF_I32, int32_t, uint32_t)
        SUM_CASE(RDF_I64, int64_t, uint64_t)
        SUM_CASE(RDF_U8, uint8_t, uint8_t)
        SUM_CASE(RDF_U16, uint16_t, uint16_t)
        SUM_CASE(RDF_U32, uint32_t, uint32_t)
        SUM_CASE(RDF_U64, uint64_t, uint64_t)
        SUM_FLT_CASE(RDF_F32, float)
        SUM_FLT_CASE(RDF_F64, double)
        default: break;
    }
    *out_is_some = 1; /* Some(sum) always, aggregate.rs:92 */
    return RDF_OK;
}

/* aggregate.rs:12-31 with the evident intent: a `<` / `>` fold over
 * the valid slots of every chunk, all-null chunks skipped, None when nothing is valid.  Floats:
 * NaN never wins unless every valid value is NaN (documented divergence: the reference bounds
 * T::Native: Ord and cannot take floats at all). */
#define MINMAX_INT_CASE(DT, T)                                                                    \
    case DT: {                                                                                    \
        T best = 0;                                                                               \
        for (int64_t c = 0; c < nchunks; c++) {                                                   \
            const T* x = (const T*)a[c].values + a[c].offset;                                     \
            for (int64_t i = 0; i < a[c].length; i++) {                                           \
                if (!arr_valid(&a[c], i)) continue;                                               \
                if (!some || (want_max ? x[i] > best : x[i] < best)) { best = x[i]; some = 1; }   \
            }                                                                                     \
        }                                                                                         \
        if (some) *(T*)out_scalar = best;                                                         \
    } break;
#define MINMAX_FLT_CASE(DT, T)                                                                    \
    case DT: {                                                                                    \
        T best = (T)NAN; int any = 0;                                                             \
        for (int64_t c = 0; c < nchunks; c++) {                                                   \
            const T* x = (const T*)a[c].values + a[c].offset;                                     \
            for (int64_t i = 0; i < a[c].length; i++) {                                           \
                if (!arr_valid(&a[c], i)) continue;                                               \
                any = 1;                                                                          \
                if (x[i] != x[i]) continue;                                                       \
                if (best != best || (want_max ? x[i] > best : x[i] < best)) best = x[i];          \
            }                                                                                     \
        }                                                                                         \
        some = any;                                                                               \
        if (some) *(T*)out_scalar = best;                                                         \
    } break;

static rdf_status minmax(const rdf_array* a, int64_t nchunks, void* out_scalar, int32_t* out_is_some, int want_max) {
    int some = 0;
    if (nchunks < 1) FAIL(RDF_INVALID_ARGUMENT, "min/max: a column has at least one chunk");
    if (!is_numeric(a[0].dtype)) FAIL(RDF_INVALID_ARGUMENT, "min/max: numeric type required");
    int dt = a[0].dtype;
    switch (dt) {
        MINMAX_INT_CASE(RDF_I8, int8_t)
        MINMAX_INT_CASE(RDF_I16, int16_t)
        MINMAX_INT_CASE(RDF_I32, int32_t)
        MINMAX_INT_CASE(RDF_I64, int64_t)
        MINMAX_INT_CASE(RDF_U8, uint8_t)
        MINMAX_INT_CASE(RDF_U16, uint16_t)
        MINMAX_INT_CASE(RDF_U32, uint32_t)
        MINMAX_INT_CASE(RDF_U64, uint64_t)
        MINMAX_FLT_CASE(RDF_F32, float)
        MINMAX_FLT_CASE(RDF_F64, double)
        default: break;
    }
    *out_is_some = some;
    return RDF_OK;
}
rdf_status ora_min(const rdf_array* a, int64_t n, void* o, int32_t* s) { return minmax(a, n, o, s, 0); }
rdf_status ora_max(const rdf_array* a, int64_t n, void* o, int32_t* s) { return minmax(a, n, o, s, 1); }

/* aggregate.rs:70-80: sum += (array.len() - array.null_count()) as i64; Some(sum). */
static int64_t null_count_of(const rdf_array* a) {
    if (a->validity == NULL) return 0;
    if (a->null_count >= 0) return a->null_count;
    int64_t nn = 0;
    for (int64_t i = 0; i < a->length; i++) nn += !bit_get(a->validity, a->offset + i);
    return nn;
}
rdf_status ora_count(const rdf_array* a, int64_t nchunks, int64_t* out_count, int32_t* out_is_some) {
    int64_t sum = 0;
    for (int64_t c = 0; c < nchunks; c++) sum += a[c].length - null_count_of(&a[c]);
    *out_count = sum;
    *out_is_some = 1;
    return RDF_OK;
}

/* aggregate.rs:32-65: per-chunk running mean m += (v - m) / (i + 1 - nulls), then the
 * count-weighted merge mean += (m - mean) * len / count; None if count == 0. */
rdf_status ora_avg(const rdf_array* a, int64_t nchunks, double* out_mean, int32_t* out_is_some) {
    double mean = 0.0;
    int64_t count = 0;
    for (int64_t c = 0; c < nchunks; c++) {
        if (!is_numeric(a[c].dtype)) FAIL(RDF_INVALID_ARGUMENT, "avg: numeric type required");
        double m = 0.0;
        int64_t nulls = 0;
        for (int64_t i = 0; i < a[c].length; i++) {
            if (arr_valid(&a[c], i)) m = m + (arr_f64(&a[c], i) - m) / (double)(i + 1 - nulls);
            else nulls++;
        }
        int64_t len = a[c].length - nulls;
        count += len;
        if (count > 0) mean = mean + ((m - mean) * (double)len) / (double)count;
    }
    *out_is_some = count != 0;
    if (count != 0) *out_mean = mean;
    return RDF_OK;
}

/* ------------------------------------------------------------------ temporaries for the unfused
 * evaluators: every plan step materialises a whole array, as the reference does
 * (src/evaluation.rs:70-92, src/expression.rs:777-859). */

typedef struct {
    int32_t dtype;
    void* values;      /* offset 0 */
    uint8_t* validity; /* NULL = all valid */
    int64_t len;
    int owns;
} tmparr;

static void tmp_free(tmparr* t) {
    if (t->owns) { free(t->values); free(t->validity); }
    t->values = NULL; t->validity = NULL; t->owns = 0;
}
static rdf_array tmp_view(const tmparr* t) {
    rdf_array a = { t->values, t->validity, 0, t->len, -1, t->dtype, RDF_MEM_HOST };
    return a;
}
static size_t values_bytes(int32_t dt, int64_t n) {
    return dt == RDF_BOOL ? (size_t)((n + 63) / 64 * 8) : (size_t)(n > 0 ? n : 1) * (size_t)dtype_size(dt);
}
static int tmp_alloc(tmparr* t, int32_t dt, int64_t n, int with_validity) {
    t->dtype = dt; t->len = n; t->owns = 1;
    t->values = calloc(1, values_bytes(dt, n) + 8);
    t->validity = with_validity ? calloc(1, (size_t)((n + 63) / 64 * 8) + 8) : NULL;
    return t->values != NULL && (!with_validity || t->validity != NULL);
}
static rdf_out tmp_out(tmparr* t) {
    rdf_out o = { t->values, t->validity, t->len, 0, 0, t->dtype, RDF_MEM_HOST };
    return o;
}
/* Re-based copy of a (possibly offset) input chunk as a temporary: batch.column(i).clone(). */
static int tmp_from_array(tmparr* t, const rdf_array* a) {
    if (!tmp_alloc(t, a->dtype, a->length, a->validity != NULL)) return 0;
    if (a->dtype == RDF_BOOL) {
        for (int64_t i = 0; i < a->length; i++) bit_put((uint8_t*)t->values, i, bit_get((const uint8_t*)a->values, a->offset + i));
    } else {
        int es = dtype_size(a->dtype);
        memcpy(t->values, (const char*)a->values + a->offset * es, (size_t)a->length * (size_t)es);
    }
    if (a->validity)
        for (int64_t i = 0; i < a->length; i++) bit_put(t->validity, i, bit_get(a->validity, a->offset + i));
    return 1;
}

/* Scalar -> constant array of batch length (src/expression.rs:777-803).  Scalar::Null becomes a
 * BooleanArray of `false` (valid). */
static int tmp_from_scalar(tmparr* t, const rdf_expr_node* nd, int64_t n) {
    int32_t dt = nd->dtype == RDF_NULLTYPE ? RDF_BOOL : nd->dtype;
    if (!tmp_alloc(t, dt, n, 0)) return 0;
    for (int64_t i = 0; i < n; i++) {
        switch (dt) {
            case RDF_BOOL: bit_put((uint8_t*)t->values, i, nd->dtype == RDF_NULLTYPE ? 0 : nd->i64 != 0); break;
            case RDF_F64: ((double*)t->values)[i] = nd->f64; break;
            case RDF_F32: ((float*)t->values)[i] = (float)nd->f64; break;
            case RDF_I8: case RDF_U8: ((uint8_t*)t->values)[i] = (uint8_t)nd->i64; break;
            case RDF_I16: case RDF_U16: ((uint16_t*)t->values)[i] = (uint16_t)nd->i64; break;
            case RDF_I32: case RDF_U32: ((uint32_t*)t->values)[i] = (uint32_t)nd->i64; break;
            default: ((uint64_t*)t->values)[i] = (uint64_t)nd->i64; break;
        }
    }
    return 1;
}

static rdf_status tmp_cast(const tmparr* in, int32_t to, tmparr* out) {
    if (!tmp_alloc(out, to, in->len, in->validity != NULL || cast_can_null(in->dtype, to))) FAIL(RDF_MEMORY_ERROR, "out of memory");
    rdf_array a = tmp_view(in);
    rdf_out o = tmp_out(out);
    return cast_chunk(&a, &o);
}

/* Recursive, fully materialising evaluation of an expression node over ONE batch (chunk index c).
 *   BooleanFilter::eval_to_array, src/expression.rs:766-861:
 *     Input(Scalar) -> constant array; Input(Column) -> the batch's column;
 *     Not -> cast to Boolean, compute::not; And/Or -> (intent) cast to Boolean, compute::and/or,
 *     null if either side null; Gt..Le -> cast both sides to Float64, compare,
 *     null if either side null.
 *   Value expressions: each OP node is one Calculation step of Evaluate::calculate
 *     (src/evaluation.rs:97-323): binary arithmetic, unary float ops, cast. */
static rdf_status eval_node(const rdf_expr_node* nodes, int32_t nnodes, int32_t idx, const rdf_array* cols,
                            int32_t ncols, int64_t nchunks, int64_t c, int64_t batch_len, tmparr* res) {
    if (idx < 0 || idx >= nnodes) FAIL(RDF_INVALID_ARGUMENT, "bad node index %d", idx);
    const rdf_expr_node* nd = &nodes[idx];
    memset(res, 0, sizeof *res);
    if (nd->kind == RDF_NODE_SCALAR) {
        if (!tmp_from_scalar(res, nd, batch_len)) FAIL(RDF_MEMORY_ERROR, "out of memory");
        return RDF_OK;
    }
    if (nd->kind == RDF_NODE_COLUMN) {
        if (nd->column < 0 || nd->column >= ncols) FAIL(RDF_COMPUTE_ERROR, "Cannot find column %d", nd->column);
        if (!tmp_from_array(res, &cols[(int64_t)nd->column * nchunks + c])) FAIL(RDF_MEMORY_ERROR, "out of memory");
        return RDF_OK;
    }
    int op = nd->op;
    tmparr l = {0}, r = {0}, lc = {0}, rc = {0};
    rdf_status s = eval_node(nodes, nnodes, nd->lhs, cols, ncols, nchunks, c, batch_len, &l);
    if (s != RDF_OK) return s;
    int binary = (op >= RDF_OP_ADD && op <= RDF_OP_LOG) || (op >= RDF_OP_GT && op <= RDF_OP_LE) || op == RDF_OP_AND || op == RDF_OP_OR;
    if (binary) {
        s = eval_node(nodes, nnodes, nd->rhs, cols, ncols, nchunks, c, batch_len, &r);
        if (s != RDF_OK) { tmp_free(&l); return s; }
    }
    if (op >= RDF_OP_ADD && op <= RDF_OP_LOG) {
        if (!tmp_alloc(res, l.dtype, l.len, l.validity || r.validity)) s = RDF_MEMORY_ERROR;
        else { rdf_array a = tmp_view(&l), b = tmp_view(&r); rdf_out o = tmp_out(res); s = binary_chunk(op, &a, &b, &o); }
    } else if ((op >= RDF_OP_ABS && op <= RDF_OP_TANH) || (op >= RDF_OP_COT && op <= RDF_OP_CSC)) {
        if (!tmp_alloc(res, l.dtype, l.len, l.validity != NULL)) s = RDF_MEMORY_ERROR;
        else { rdf_array a = tmp_view(&l); rdf_out o = tmp_out(res); s = unary_chunk(op, &a, &o); }
    } else if (op >= RDF_OP_HOUR_S && op <= RDF_OP_HOUR_DAY) {                /* result keeps the operand's storage type */
        if (l.dtype != RDF_I32 && l.dtype != RDF_I64) { tmp_free(&l); FAIL(RDF_INVALID_ARGUMENT, "hour: Int32 / Int64 temporal storage required"); }
        if (!tmp_alloc(res, l.dtype, l.len, l.validity != NULL)) s = RDF_MEMORY_ERROR;
        else { rdf_array a = tmp_view(&l); rdf_out o = tmp_out(res); s = hour_chunk(&a, op - RDF_OP_HOUR_S, &o, l.dtype); }
    } else if (op == RDF_OP_CAST) {
        s = tmp_cast(&l, nd->dtype, res);
    } else if (op >= RDF_OP_GT && op <= RDF_OP_LE) {
        s = tmp_cast(&l, RDF_F64, &lc);
        if (s == RDF_OK) s = tmp_cast(&r, RDF_F64, &rc);
        if (s == RDF_OK && !tmp_alloc(res, RDF_BOOL, l.len, lc.validity || rc.validity)) s = RDF_MEMORY_ERROR;
        if (s == RDF_OK) {
            const double* x = (const double*)lc.values; const double* y = (const double*)rc.values;
            for (int64_t i = 0; i < l.len; i++) {
                int v = (!lc.validity || bit_get(lc.validity, i)) && (!rc.validity || bit_get(rc.validity, i));
                int b;
                switch (op) {
                    case RDF_OP_GT: b = x[i] > y[i]; break;
                    case RDF_OP_GE: b = x[i] >= y[i]; break;
                    case RDF_OP_EQ: b = x[i] == y[i]; break;
                    case RDF_OP_NE: b = x[i] != y[i]; break;
                    case RDF_OP_LT: b = x[i] < y[i]; break;
                    default: b = x[i] <= y[i]; break;
                }
                bit_put((uint8_t*)res->values, i, b && v); /* value bit of a null slot is 0 */
                if (res->validity) bit_put(res->validity, i, v);
            }
        }
    } else if (op == RDF_OP_NOT) {
        s = tmp_cast(&l, RDF_BOOL, &lc);
        if (s == RDF_OK && !tmp_alloc(res, RDF_BOOL, l.len, lc.validity != NULL)) s = RDF_MEMORY_ERROR;
        if (s == RDF_OK)
            for (int64_t i = 0; i < l.len; i++) {
                int v = !lc.validity || bit_get(lc.validity, i);
                bit_put((uint8_t*)res->values, i, v && !bit_get((uint8_t*)lc.values, i));
                if (res->validity) bit_put(res->validity, i, v);
            }
    } else if (op == RDF_OP_AND || op == RDF_OP_OR) {
        s = tmp_cast(&l, RDF_BOOL, &lc);
        if (s == RDF_OK) s = tmp_cast(&r, RDF_BOOL, &rc);
        if (s == RDF_OK && !tmp_alloc(res, RDF_BOOL, l.len, lc.validity || rc.validity)) s = RDF_MEMORY_ERROR;
        if (s == RDF_OK)
            for (int64_t i = 0; i < l.len; i++) {
                int v = (!lc.validity || bit_get(lc.validity, i)) && (!rc.validity || bit_get(rc.validity, i));
                int x = bit_get((uint8_t*)lc.values, i), y = bit_get((uint8_t*)rc.values, i);
                bit_put((uint8_t*)res->values, i, v && (op == RDF_OP_AND ? (x && y) : (x || y)));
                if (res->validity) bit_put(res->validity, i, v);
            }
    } else {
        snprintf(g_err, sizeof g_err, "unsupported op %d", op);
        s = RDF_INVALID_ARGUMENT;
    }
    tmp_free(&l); tmp_free(&r); tmp_free(&lc); tmp_free(&rc);
    if (s != RDF_OK) tmp_free(res);
    if (s == RDF_MEMORY_ERROR) snprintf(g_err, sizeof g_err, "out of memory");
    return s;
}

static int64_t batch_length(const rdf_array* cols, int32_t ncols, int64_t nchunks, int64_t c, rdf_status* st) {
    *st = RDF_OK;
    if (ncols <= 0) return 0;
    int64_t n = cols[c].length;
    for (int32_t k = 1; k < ncols; k++)
        if (cols[(int64_t)k * nchunks + c].length != n) { *st = RDF_COMPUTE_ERROR; snprintf(g_err, sizeof g_err, "columns of a batch differ in length"); }
    return n;
}

static rdf_status copy_tmp_to_out(const tmparr* t, rdf_out* o) {
    if (o->capacity < t->len) FAIL(RDF_MEMORY_ERROR, "output capacity too small");
    if (o->dtype != t->dtype) FAIL(RDF_INVALID_ARGUMENT, "output dtype %d != expression dtype %d", o->dtype, t->dtype);
    if (t->validity && !o->validity) FAIL(RDF_INVALID_ARGUMENT, "output validity buffer required");
    out_begin(o, t->len);
    memcpy(o->values, t->values, t->dtype == RDF_BOOL ? (size_t)((t->len + 7) / 8) : (size_t)t->len * (size_t)dtype_size(t->dtype));
    if (t->validity)
        for (int64_t i = 0; i < t->len; i++) if (!bit_get(t->validity, i)) out_null(o, i);
    return RDF_OK;
}

/* DataFrame::evaluate_boolean_filter, src/dataframe.rs:612-624: one mask array per RecordBatch. */
rdf_status ora_predicate(const rdf_expr_node* nodes, int32_t nnodes, int32_t root, const rdf_array* cols,
                         int32_t ncols, int64_t nchunks, rdf_out* mask) {
    for (int64_t c = 0; c < nchunks; c++) {
        rdf_status s;
        int64_t n = batch_length(cols, ncols, nchunks, c, &s);
        if (s != RDF_OK) return s;
        tmparr t;
        s = eval_node(nodes, nnodes, root, cols, ncols, nchunks, c, n, &t);
        if (s != RDF_OK) return s;
        if (t.dtype != RDF_BOOL) { tmp_free(&t); FAIL(RDF_INVALID_ARGUMENT, "predicate root must be boolean"); }
        s = copy_tmp_to_out(&t, &mask[c]);
        tmp_free(&t);
        if (s != RDF_OK) return s;
    }
    return RDF_OK;
}

/* ------------------------------------------------------------------ filter
 * src/table.rs:97-107: zip(chunks, mask chunks) -> arrow::compute::filter(chunk, mask): keep the
 * rows whose mask bit is 1 (and valid), in order, carrying validity. */

static inline int mask_keep(const rdf_array* m, int64_t i) {
    return bit_get((const uint8_t*)m->values, m->offset + i) && (m->validity == NULL || bit_get(m->validity, m->offset + i));
}

rdf_status ora_filter_count(const rdf_array* mask, int64_t nchunks, int64_t* counts) {
    for (int64_t c = 0; c < nchunks; c++) {
        if (mask[c].dtype != RDF_BOOL) FAIL(RDF_INVALID_ARGUMENT, "filter mask must be boolean");
        int64_t k = 0;
        for (int64_t i = 0; i < mask[c].length; i++) k += mask_keep(&mask[c], i);
        counts[c] = k;
    }
    return RDF_OK;
}

static rdf_status filter_chunk(const rdf_array* a, const rdf_array* m, rdf_out* o) {
    if (m->dtype != RDF_BOOL) FAIL(RDF_INVALID_ARGUMENT, "filter mask must be boolean");
    if (a->length != m->length) FAIL(RDF_COMPUTE_ERROR, "Filter array must have the same length as the data array");
    if (!is_numeric(a->dtype) || o->dtype != a->dtype) FAIL(RDF_INVALID_ARGUMENT, "filter: unsupported or mismatched dtype");
    if (a->validity && !o->validity) FAIL(RDF_INVALID_ARGUMENT, "output validity buffer required");
    int es = dtype_size(a->dtype);
    int64_t k = 0;
    for (int64_t i = 0; i < a->length; i++) k += mask_keep(m, i);
    if (o->capacity < k) FAIL(RDF_MEMORY_ERROR, "output capacity too small");
    out_begin(o, k);
    k = 0;
    for (int64_t i = 0; i < a->length; i++) {
        if (!mask_keep(m, i)) continue;
        memcpy((char*)o->values + k * es, (const char*)a->values + (a->offset + i) * es, (size_t)es);
        if (!arr_valid(a, i)) out_null(o, k);
        k++;
    }
    return RDF_OK;
}

rdf_status ora_filter(const rdf_array* col, const rdf_array* mask, int64_t nchunks, rdf_out* out) {
    for (int64_t c = 0; c < nchunks; c++) {
        rdf_status s = filter_chunk(&col[c], &mask[c], &out[c]);
        if (s != RDF_OK) return s;
    }
    return RDF_OK;
}

/* src/dataframe.rs:183-187: self.columns.map(|col| col.filter(&mask)). */
rdf_status ora_filter_columns(const rdf_array* cols, int32_t ncols, const rdf_array* mask, int64_t nchunks, rdf_out* outs) {
    for (int32_t k = 0; k < ncols; k++) {
        rdf_status s = ora_filter(cols + (int64_t)k * nchunks, mask, nchunks, outs + (int64_t)k * nchunks);
        if (s != RDF_OK) return s;
    }
    return RDF_OK;
}

/* ------------------------------------------------------------------ take
 * src/table.rs:218-241: values = concat(all chunks) (:221 -> :180-182); arrow::compute::take(values,
 * indices, None): out[j] = values[idx[j]]; null index -> null; values' nulls carried; the result is
 * ONE chunk whatever chunk_size says. */
rdf_status ora_take(const rdf_array* chunks, int64_t nchunks, const rdf_array* indices, rdf_out* out) {
    if (nchunks < 1) FAIL(RDF_INVALID_ARGUMENT, "take: a column has at least one chunk");
    if (indices->dtype != RDF_U32 && indices->dtype != RDF_U64) FAIL(RDF_INVALID_ARGUMENT, "take: indices must be UInt32/UInt64");
    int32_t dt = chunks[0].dtype;
    if (!is_numeric(dt) || out->dtype != dt) FAIL(RDF_INVALID_ARGUMENT, "take: unsupported or mismatched dtype");
    int es = dtype_size(dt);
    int64_t total = 0;
    int any_validity = indices->validity != NULL;
    for (int64_t c = 0; c < nchunks; c++) {
        if (chunks[c].dtype != dt) FAIL(RDF_INVALID_ARGUMENT, "take: chunks differ in dtype");
        total += chunks[c].length;
        any_validity |= chunks[c].validity != NULL;
    }
    /* Column::to_array: the concat copy */
    char* values = (char*)malloc((size_t)(total > 0 ? total : 1) * (size_t)es);
    uint8_t* valid = (uint8_t*)malloc((size_t)(total / 8 + 8));
    if (!values || !valid) { free(values); free(valid); FAIL(RDF_MEMORY_ERROR, "out of memory"); }
    memset(valid, 0xFF, (size_t)(total / 8 + 8));
    int64_t pos = 0;
    for (int64_t c = 0; c < nchunks; c++) {
        memcpy(values + pos * es, (const char*)chunks[c].values + chunks[c].offset * es, (size_t)chunks[c].length * (size_t)es);
        for (int64_t i = 0; i < chunks[c].length; i++) if (!arr_valid(&chunks[c], i)) bit_clr(valid, pos + i);
        pos += chunks[c].length;
    }
    rdf_status st = RDF_OK;
    int64_t n = indices->length;
    if (out->capacity < n) { st = RDF_MEMORY_ERROR; snprintf(g_err, sizeof g_err, "output capacity too small"); }
    else if (any_validity && !out->validity) { st = RDF_INVALID_ARGUMENT; snprintf(g_err, sizeof g_err, "output validity buffer required"); }
    else {
        out_begin(out, n);
        for (int64_t j = 0; j < n; j++) {
            if (!arr_valid(indices, j)) { memset((char*)out->values + j * es, 0, (size_t)es); out_null(out, j); continue; }
            uint64_t ix = indices->dtype == RDF_U32 ? ((const uint32_t*)indices->values)[indices->offset + j]
                                                    : ((const uint64_t*)indices->values)[indices->offset + j];
            if (ix >= (uint64_t)total) { st = RDF_COMPUTE_ERROR; snprintf(g_err, sizeof g_err, "take index %llu out of bounds (len %lld)", (unsigned long long)ix, (long long)total); break; }
            memcpy((char*)out->values + j * es, values + ix * es, (size_t)es);
            if (!bit_get(valid, (int64_t)ix)) out_null(out, j);
        }
    }
    free(values); free(valid);
    return st;
}

/* ------------------------------------------------------------------ the batch loop, unfused
 * Evaluate::evaluate, src/evaluation.rs:66-96: every step replaces the frame with freshly
 * materialised columns; Filter -> DataFrame::filter (src/dataframe.rs:178-189); the aggregates
 * are AggregateFunctions over the resulting chunk list. */

static void agg_from_chunks(const rdf_array* chunks, int64_t nchunks, rdf_agg_result* r) {
    int32_t dt = chunks[0].dtype, some = 0;
    memset(r, 0, sizeof *r);
    r->dtype = dt;
    ora_count(chunks, nchunks, &r->count, &some);
    /* count must look at the bitmap: temporaries carry null_count = -1 */
    r->is_some = r->count > 0;
    if (is_float(dt)) {
        if (dt == RDF_F64) {
            double v; int32_t s;
            ora_sum(chunks, nchunks, &v, &s); r->sum_f64 = v;
            ora_min(chunks, nchunks, &v, &s); if (s) r->min_f64 = v;
            ora_max(chunks, nchunks, &v, &s); if (s) r->max_f64 = v;
        } else {
            float v; int32_t s;
            ora_sum(chunks, nchunks, &v, &s); r->sum_f64 = v;
            ora_min(chunks, nchunks, &v, &s); if (s) r->min_f64 = v;
            ora_max(chunks, nchunks, &v, &s); if (s) r->max_f64 = v;
        }
    } else {
        uint64_t raw; int32_t s; int es = dtype_size(dt);
#define WIDEN(dst)                                                                                \
        switch (dt) {                                                                             \
            case RDF_I8: dst = (int64_t)(int8_t)raw; break;                                       \
            case RDF_I16: dst = (int64_t)(int16_t)raw; break;                                     \
            case RDF_I32: dst = (int64_t)(int32_t)raw; break;                                     \
            case RDF_U8: dst = (int64_t)(uint8_t)raw; break;                                      \
            case RDF_U16: dst = (int64_t)(uint16_t)raw; break;                                    \
            case RDF_U32: dst = (int64_t)(uint32_t)raw; break;                                    \
            default: dst = (int64_t)raw; break;                                                   \
        }
        (void)es;
        raw = 0; ora_sum(chunks, nchunks, &raw, &s); WIDEN(r->sum_i64)
        raw = 0; ora_min(chunks, nchunks, &raw, &s); if (s) { WIDEN(r->min_i64) }
        raw = 0; ora_max(chunks, nchunks, &raw, &s); if (s) { WIDEN(r->max_i64) }
#undef WIDEN
    }
}

rdf_status ora_pipeline(const rdf_program* prog, const rdf_array* cols, int32_t ncols, int64_t nchunks,
                        rdf_out* outs, rdf_agg_result* aggs) {
    if (prog->nvalues < 1 || prog->nvalues > RDF_MAX_VALUES) FAIL(RDF_INVALID_ARGUMENT, "nvalues out of range");
    if (prog->sink == RDF_SINK_STORE && prog->filter_root >= 0)
        FAIL(RDF_INVALID_ARGUMENT, "SINK_STORE with a filter: use rdf_predicate + rdf_filter_columns");
    rdf_status st = RDF_OK;
    /* materialise every value column for every batch */
    tmparr* vals = (tmparr*)calloc((size_t)(prog->nvalues * (nchunks > 0 ? nchunks : 1)), sizeof(tmparr));
    tmparr* masks = (tmparr*)calloc((size_t)(nchunks > 0 ? nchunks : 1), sizeof(tmparr));
    if (!vals || !masks) { free(vals); free(masks); FAIL(RDF_MEMORY_ERROR, "out of memory"); }
    for (int64_t c = 0; c < nchunks && st == RDF_OK; c++) {
        int64_t n = batch_length(cols, ncols, nchunks, c, &st);
        if (st != RDF_OK) break;
        for (int32_t v = 0; v < prog->nvalues && st == RDF_OK; v++)
            st = eval_node(prog->nodes, prog->nnodes, prog->value_roots[v], cols, ncols, nchunks, c, n, &vals[(int64_t)v * nchunks + c]);
        if (st == RDF_OK && prog->filter_root >= 0) {
            st = eval_node(prog->nodes, prog->nnodes, prog->filter_root, cols, ncols, nchunks, c, n, &masks[c]);
            if (st == RDF_OK && masks[c].dtype != RDF_BOOL) { st = RDF_INVALID_ARGUMENT; snprintf(g_err, sizeof g_err, "predicate root must be boolean"); }
        }
    }
    if (st == RDF_OK && prog->filter_root >= 0) {
        /* DataFrame::filter: every column filtered by the mask column */
        for (int32_t v = 0; v < prog->nvalues && st == RDF_OK; v++)
            for (int64_t c = 0; c < nchunks && st == RDF_OK; c++) {
                tmparr* t = &vals[(int64_t)v * nchunks + c];
                tmparr f;
                if (!tmp_alloc(&f, t->dtype, t->len, t->validity != NULL)) { st = RDF_MEMORY_ERROR; break; }
                rdf_array a = tmp_view(t), m = tmp_view(&masks[c]);
                rdf_out o = tmp_out(&f);
                st = filter_chunk(&a, &m, &o);
                f.len = o.length;
                tmp_free(t);
                *t = f;
            }
    }
    if (st == RDF_OK && prog->sink == RDF_SINK_STORE) {
        for (int32_t v = 0; v < prog->nvalues && st == RDF_OK; v++)
            for (int64_t c = 0; c < nchunks && st == RDF_OK; c++)
                st = copy_tmp_to_out(&vals[(int64_t)v * nchunks + c], &outs[(int64_t)v * nchunks + c]);
    } else if (st == RDF_OK) {
        rdf_array* views = (rdf_array*)calloc((size_t)(nchunks > 0 ? nchunks : 1), sizeof(rdf_array));
        for (int32_t v = 0; v < prog->nvalues; v++) {
            if (nchunks == 0) { memset(&aggs[v], 0, sizeof aggs[v]); continue; }
            for (int64_t c = 0; c < nchunks; c++) views[c] = tmp_view(&vals[(int64_t)v * nchunks + c]);
            if (!is_numeric(views[0].dtype)) { st = RDF_INVALID_ARGUMENT; snprintf(g_err, sizeof g_err, "aggregate of a non-numeric value"); break; }
            agg_from_chunks(views, nchunks, &aggs[v]);
        }
        free(views);
    }
    for (int64_t i = 0; i < (int64_t)prog->nvalues * nchunks; i++) tmp_free(&vals[i]);
    for (int64_t c = 0; c < nchunks; c++) tmp_free(&masks[c]);
    free(vals); free(masks);
    return st;
}

/* ------------------------------------------------------------------ grouped aggregation over a small dense domain
 * Transformation::GroupAggregate after Calculate/Filter steps: planned by Dataset::try_aggregate
 * (src/expression.rs:114-221), not executed by the reference (src/evaluation.rs:73 panics): PARITY UNPINNED BY
 * THE REFERENCE, SQL semantics.  Restated unfused like the batch loop above: every expression (mask, group id,
 * values) is materialised per batch, then one sequential scan folds the kept rows into per-group sums. */
static int64_t int_at(const rdf_array* a, int64_t i) {
    if (a->dtype == RDF_BOOL) return bit_get((const uint8_t*)a->values, a->offset + i);
    if (a->dtype == RDF_U64) return (int64_t)((const uint64_t*)a->values)[a->offset + i];
    int64_t k = a->offset + i;
    switch (a->dtype) {
        case RDF_I8: return ((const int8_t*)a->values)[k];
        case RDF_I16: return ((const int16_t*)a->values)[k];
        case RDF_I32: return ((const int32_t*)a->values)[k];
        case RDF_U8: return ((const uint8_t*)a->values)[k];
        case RDF_U16: return ((const uint16_t*)a->values)[k];
        case RDF_U32: return ((const uint32_t*)a->values)[k];
        default: return ((const int64_t*)a->values)[k];
    }
}
rdf_status ora_group_pipeline(const rdf_expr_node* nodes, int32_t nnodes, int32_t filter_root, int32_t group_root, int32_t ngroups,
                              const int32_t* value_roots, int32_t nvalues, const rdf_array* cols, int32_t ncols, int64_t nchunks,
                              rdf_group_result* out, int64_t* group_rows) {
    if (!nodes || nnodes <= 0) FAIL(RDF_INVALID_ARGUMENT, "empty program");
    if (!value_roots || nvalues < 1 || nvalues > RDF_MAX_GROUP_VALUES) FAIL(RDF_INVALID_ARGUMENT, "nvalues out of range");
    if (group_root < 0 || group_root >= nnodes) FAIL(RDF_INVALID_ARGUMENT, "group_root out of range");
    if (ngroups < 1 || (int64_t)(ngroups + 1) * nvalues > RDF_MAX_GROUP_SLOTS) FAIL(RDF_INVALID_ARGUMENT, "grouped aggregation: domain too large");
    if (!out) FAIL(RDF_INVALID_ARGUMENT, "null output pointer");
    const int S = ngroups + 1;
    int64_t* rows = (int64_t*)calloc((size_t)S, sizeof(int64_t));
    memset(out, 0, sizeof(rdf_group_result) * (size_t)S * (size_t)nvalues);
    rdf_status st = RDF_OK;
    int dtype_known = 0;
    for (int64_t c = 0; c < nchunks && st == RDF_OK; c++) {
        int64_t n = batch_length(cols, ncols, nchunks, c, &st);
        if (st != RDF_OK) break;
        tmparr mask, gid, val[RDF_MAX_GROUP_VALUES];
        memset(&mask, 0, sizeof mask); memset(&gid, 0, sizeof gid); memset(val, 0, sizeof val);
        if (filter_root >= 0) {
            st = eval_node(nodes, nnodes, filter_root, cols, ncols, nchunks, c, n, &mask);
            if (st == RDF_OK && mask.dtype != RDF_BOOL) { st = RDF_INVALID_ARGUMENT; snprintf(g_err, sizeof g_err, "predicate root must be boolean"); }
        }
        if (st == RDF_OK) st = eval_node(nodes, nnodes, group_root, cols, ncols, nchunks, c, n, &gid);
        if (st == RDF_OK && !(gid.dtype == RDF_BOOL || (gid.dtype >= RDF_I8 && gid.dtype <= RDF_U64))) {
            st = RDF_INVALID_ARGUMENT; snprintf(g_err, sizeof g_err, "group id expression must be integer-valued");
        }
        for (int32_t v = 0; v < nvalues && st == RDF_OK; v++) {
            st = eval_node(nodes, nnodes, value_roots[v], cols, ncols, nchunks, c, n, &val[v]);
            if (st == RDF_OK && !(is_numeric(val[v].dtype) || val[v].dtype == RDF_BOOL)) { st = RDF_INVALID_ARGUMENT; snprintf(g_err, sizeof g_err, "aggregate of a non-numeric value"); }
        }
        if (st == RDF_OK) {
            rdf_array m = tmp_view(&mask), g = tmp_view(&gid);
            if (!dtype_known) { for (int32_t v = 0; v < nvalues; v++) for (int k = 0; k < S; k++) out[(size_t)v * S + k].dtype = val[v].dtype; dtype_known = 1; }
            for (int64_t i = 0; i < n && st == RDF_OK; i++) {
                if (filter_root >= 0 && !mask_keep(&m, i)) continue;
                int64_t slot = ngroups;
                if (arr_valid(&g, i)) {
                    slot = int_at(&g, i);
                    if ((gid.dtype == RDF_U64 ? (uint64_t)slot >= (uint64_t)ngroups : (slot < 0 || slot >= ngroups))) {
                        st = RDF_COMPUTE_ERROR; snprintf(g_err, sizeof g_err, "group id outside [0, %d)", ngroups); break;
                    }
                }
                rows[slot]++;
                for (int32_t v = 0; v < nvalues; v++) {
                    rdf_array a = tmp_view(&val[v]);
                    if (!arr_valid(&a, i)) continue;
                    rdf_group_result* r = &out[(size_t)v * S + slot];
                    if (is_float(a.dtype)) r->sum_f64 += arr_f64(&a, i);
                    else r->sum_i64 = (int64_t)((uint64_t)r->sum_i64 + (uint64_t)int_at(&a, i));
                    r->count++;
                }
            }
        }
        tmp_free(&mask); tmp_free(&gid);
        for (int32_t v = 0; v < nvalues; v++) tmp_free(&val[v]);
    }
    if (st == RDF_OK) {
        for (int64_t k = 0; k < (int64_t)S * nvalues; k++) out[k].is_some = out[k].count > 0;
        if (group_rows) memcpy(group_rows, rows, sizeof(int64_t) * (size_t)S);
    }
    free(rows);
    return st;
}

/* ------------------------------------------------------------------ ArrayFunctions over List<primitive>
 * src/functions/array.rs: row i = value_slice(value_offset(i), value_length(i)) of the child values (child validity is
 * not looked at, as in the reference).  Pinned by the reference's own tests (:421-640): array_contains over
 * i32 / i64 / f64, array_position, array_remove, array_sort on the 16-value / 6-row fixture. */
typedef struct { const int32_t* off; const rdf_array* vals; const rdf_array* lst; int64_t n; } listview;
static rdf_status list_view(const rdf_list_array* l, listview* v) {
    if (!l) FAIL(RDF_INVALID_ARGUMENT, "null list");
    if (l->offsets.dtype != RDF_I32 || l->offsets.length < 1) FAIL(RDF_INVALID_ARGUMENT, "value_offsets must be Int32 with rows + 1 entries");
    if (!is_numeric(l->values.dtype)) FAIL(RDF_INVALID_ARGUMENT, "primitive numeric child values only");
    v->off = (const int32_t*)l->offsets.values + l->offsets.offset;
    v->vals = &l->values; v->lst = &l->offsets; v->n = l->offsets.length - 1;
    return RDF_OK;
}
static int list_valid(const listview* v, int64_t i) { return v->lst->validity == NULL || bit_get(v->lst->validity, v->lst->offset + i); }
/* equality of child element e with the needle, in the child's own type (IEEE for floats) */
static int elem_eq(const rdf_array* a, int64_t e, const void* needle) {
    switch (a->dtype) {
        case RDF_F64: return ((const double*)a->values)[a->offset + e] == *(const double*)needle;
        case RDF_F32: return ((const float*)a->values)[a->offset + e] == *(const float*)needle;
        default: return memcmp((const char*)a->values + (size_t)(a->offset + e) * (size_t)dtype_size(a->dtype), needle, (size_t)dtype_size(a->dtype)) == 0;
    }
}
static rdf_status list_find(const rdf_list_array* l, const void* value, rdf_out* out, int want_position) {
    listview v;
    rdf_status st = list_view(l, &v);
    if (st != RDF_OK) return st;
    if (!value || !out) FAIL(RDF_INVALID_ARGUMENT, "null argument");
    if (out->dtype != (want_position ? RDF_I32 : RDF_BOOL)) FAIL(RDF_INVALID_ARGUMENT, "output dtype");
    if (out->capacity < v.n) FAIL(RDF_MEMORY_ERROR, "output capacity too small");
    if (!want_position && v.lst->validity && !out->validity) FAIL(RDF_INVALID_ARGUMENT, "output validity buffer required");
    out_begin(out, v.n);
    for (int64_t i = 0; i < v.n; i++) {
        int32_t pos = 0;
        if (list_valid(&v, i))
            for (int32_t e = v.off[i]; e < v.off[i + 1]; e++) if (elem_eq(v.vals, e, value)) { pos = e - v.off[i] + 1; break; }
        if (want_position) ((int32_t*)out->values)[i] = pos;                 /* NULL list -> 0, array.rs:244 */
        else if (!list_valid(&v, i)) out_null(out, i);                        /* NULL list -> NULL, array.rs:24 */
        else bit_put((uint8_t*)out->values, i, pos != 0);
    }
    return RDF_OK;
}
rdf_status ora_list_contains(const rdf_list_array* l, const void* value, rdf_out* out) { return list_find(l, value, out, 0); }
rdf_status ora_list_position(const rdf_list_array* l, const void* value, rdf_out* out) { return list_find(l, value, out, 1); }

static int elem_less(const rdf_array* a, int64_t x, int64_t y) {   /* value order; NaN handled by the callers */
    int64_t i = a->offset + x, j = a->offset + y;
    switch (a->dtype) {
        case RDF_I8: return ((const int8_t*)a->values)[i] < ((const int8_t*)a->values)[j];
        case RDF_I16: return ((const int16_t*)a->values)[i] < ((const int16_t*)a->values)[j];
        case RDF_I32: return ((const int32_t*)a->values)[i] < ((const int32_t*)a->values)[j];
        case RDF_I64: return ((const int64_t*)a->values)[i] < ((const int64_t*)a->values)[j];
        case RDF_U8: return ((const uint8_t*)a->values)[i] < ((const uint8_t*)a->values)[j];
        case RDF_U16: return ((const uint16_t*)a->values)[i] < ((const uint16_t*)a->values)[j];
        case RDF_U32: return ((const uint32_t*)a->values)[i] < ((const uint32_t*)a->values)[j];
        case RDF_U64: return ((const uint64_t*)a->values)[i] < ((const uint64_t*)a->values)[j];
        case RDF_F32: return ((const float*)a->values)[i] < ((const float*)a->values)[j];
        default: return ((const double*)a->values)[i] < ((const double*)a->values)[j];
    }
}
static int elem_nan(const rdf_array* a, int64_t x) {
    if (a->dtype == RDF_F64) { double d = ((const double*)a->values)[a->offset + x]; return d != d; }
    if (a->dtype == RDF_F32) { float d = ((const float*)a->values)[a->offset + x]; return d != d; }
    return 0;
}
static rdf_status list_extreme(const rdf_list_array* l, rdf_out* out, int want_max) {
    listview v;
    rdf_status st = list_view(l, &v);
    if (st != RDF_OK) return st;
    if (!out || out->dtype != v.vals->dtype) FAIL(RDF_INVALID_ARGUMENT, "output must have the child dtype");
    if (out->capacity < v.n) FAIL(RDF_MEMORY_ERROR, "output capacity too small");
    if (!out->validity) FAIL(RDF_INVALID_ARGUMENT, "output validity buffer required");
    out_begin(out, v.n);
    size_t es = (size_t)dtype_size(v.vals->dtype);
    for (int64_t i = 0; i < v.n; i++) {
        int64_t best = -1, anynan = -1;
        if (list_valid(&v, i))
            for (int32_t e = v.off[i]; e < v.off[i + 1]; e++) {
                if (elem_nan(v.vals, e)) { anynan = e; continue; }
                if (best < 0 || (want_max ? elem_less(v.vals, best, e) : elem_less(v.vals, e, best))) best = e;
            }
        if (best < 0) best = anynan;
        if (best < 0) { out_null(out, i); memset((char*)out->values + (size_t)i * es, 0, es); }   /* NULL or EMPTY list (the reference panics on empty) */
        else memcpy((char*)out->values + (size_t)i * es, (const char*)v.vals->values + (size_t)(v.vals->offset + best) * es, es);
    }
    return RDF_OK;
}
rdf_status ora_list_max(const rdf_list_array* l, rdf_out* out) { return list_extreme(l, out, 1); }
rdf_status ora_list_min(const rdf_list_array* l, rdf_out* out) { return list_extreme(l, out, 0); }

rdf_status ora_list_remove(const rdf_list_array* l, const void* value, rdf_out* out_offsets, rdf_out* out_values) {
    listview v;
    rdf_status st = list_view(l, &v);
    if (st != RDF_OK) return st;
    if (!value || !out_offsets || !out_values) FAIL(RDF_INVALID_ARGUMENT, "null argument");
    if (out_offsets->dtype != RDF_I32 || out_values->dtype != v.vals->dtype) FAIL(RDF_INVALID_ARGUMENT, "outputs are (Int32 offsets, child dtype values)");
    if (out_offsets->capacity < v.n + 1) FAIL(RDF_MEMORY_ERROR, "output capacity too small");
    size_t es = (size_t)dtype_size(v.vals->dtype);
    int64_t o = 0;
    int32_t* oo = (int32_t*)out_offsets->values;
    for (int64_t i = 0; i < v.n; i++) {
        oo[i] = (int32_t)o;
        if (!list_valid(&v, i)) continue;                                    /* b.append(true): an empty valid list, array.rs:273 */
        for (int32_t e = v.off[i]; e < v.off[i + 1]; e++)
            if (!elem_eq(v.vals, e, value)) {
                if (o >= out_values->capacity) FAIL(RDF_MEMORY_ERROR, "values capacity too small");
                memcpy((char*)out_values->values + (size_t)o * es, (const char*)v.vals->values + (size_t)(v.vals->offset + e) * es, es);
                o++;
            }
    }
    oo[v.n] = (int32_t)o;
    out_offsets->length = v.n + 1; out_offsets->null_count = 0;
    out_values->length = o; out_values->null_count = 0;
    if (out_offsets->validity) memset(out_offsets->validity, 0xFF, (size_t)((v.n + 8) / 8));
    if (out_values->validity) memset(out_values->validity, 0xFF, (size_t)((o + 7) / 8));
    return RDF_OK;
}

static uint64_t sort_bits(const rdf_array* a, int64_t i, int* width);
static const rdf_array* g_lsort_vals;
static int lsort_cmp(const void* x, const void* y) {   /* ascending; floats in IEEE total order like the column sort */
    int32_t a = *(const int32_t*)x, b = *(const int32_t*)y;
    int w;
    uint64_t ka = sort_bits(g_lsort_vals, a, &w), kb = sort_bits(g_lsort_vals, b, &w);
    return ka < kb ? -1 : ka > kb ? 1 : (a < b ? -1 : a > b);
}
rdf_status ora_list_sort(const rdf_list_array* l, rdf_out* out_values) {
    listview v;
    rdf_status st = list_view(l, &v);
    if (st != RDF_OK) return st;
    if (!out_values || out_values->dtype != v.vals->dtype) FAIL(RDF_INVALID_ARGUMENT, "output must have the child dtype");
    int64_t first = v.off[0], total = (int64_t)v.off[v.n] - first;
    if (out_values->capacity < total) FAIL(RDF_MEMORY_ERROR, "output capacity too small");
    size_t es = (size_t)dtype_size(v.vals->dtype);
    int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)(total > 0 ? total : 1));
    for (int64_t i = 0; i < v.n; i++) {
        int32_t b = v.off[i], e = v.off[i + 1];
        for (int32_t k = b; k < e; k++) idx[k - first] = k;
        g_lsort_vals = v.vals;
        qsort(idx + (b - first), (size_t)(e - b), sizeof(int32_t), lsort_cmp);
    }
    for (int64_t k = 0; k < total; k++)
        memcpy((char*)out_values->values + (size_t)k * es, (const char*)v.vals->values + (size_t)(v.vals->offset + idx[k]) * es, es);
    free(idx);
    out_values->length = total; out_values->null_count = 0;
    if (out_values->validity) memset(out_values->validity, 0xFF, (size_t)((total + 7) / 8));
    return RDF_OK;
}

/* The set-valued ArrayFunctions (array.rs:39-153,294-326,356-399) hand each row's slice(s) to the `array_tool` crate
 * (Cargo.toml:19 `array_tool = "1"`, not under /root/reference): restated here from the crate's published
 * vec.rs as operations on small element vectors, quadratic like the original —
 *   unique():      for x in (0..len).rev() { for y in (x+1..len).rev() { if a[x] == a[y] { a.remove(y) } } }
 *   uniq(other):   out = self.unique(); for x in other.unique() { for y in (0..out.len()).rev() { if x == out[y] { out.remove(y) } } }
 *   intersect(o):  for x in self.unique() { for y in 0..o.len() { if x == o[y] { out.push(x); break } } }
 *   union(o):      (self ++ o).unique()
 *   times(q):      self.iter().cycle().take(len * q)
 * with the crate's documented examples as known answers (tests/test_list_functions.py).  The reference has no test for
 * any of the five (test_array_union is commented out, :609): parity unpinned by the reference. */
typedef struct { const rdf_array* arr; int64_t e; } eref;
static int eref_eq(eref x, eref y) {
    size_t es = (size_t)dtype_size(x.arr->dtype);
    const char* px = (const char*)x.arr->values + (size_t)(x.arr->offset + x.e) * es;
    const char* py = (const char*)y.arr->values + (size_t)(y.arr->offset + y.e) * es;
    if (x.arr->dtype == RDF_F64) return *(const double*)px == *(const double*)py;
    if (x.arr->dtype == RDF_F32) return *(const float*)px == *(const float*)py;
    return memcmp(px, py, es) == 0;
}
static void evec_remove(eref* v, int64_t* len, int64_t at) { memmove(v + at, v + at + 1, sizeof(eref) * (size_t)(*len - at - 1)); (*len)--; }
static void evec_unique(eref* v, int64_t* len) {
    for (int64_t x = *len - 1; x >= 0; x--)
        for (int64_t y = *len - 1; y > x; y--)
            if (eref_eq(v[x], v[y])) evec_remove(v, len, y);
}
enum { SET_DISTINCT, SET_EXCEPT, SET_INTERSECT, SET_UNION, SET_REPEAT };
static rdf_status list_set(const rdf_list_array* la, const rdf_list_array* lb, int op, int32_t count, rdf_out* out_offsets, rdf_out* out_values) {
    listview a, b;
    memset(&b, 0, sizeof b);
    rdf_status st = list_view(la, &a);
    if (st != RDF_OK) return st;
    if (lb) {
        st = list_view(lb, &b);
        if (st != RDF_OK) return st;
        if (a.n != b.n) FAIL(RDF_COMPUTE_ERROR, "Expected array a and b to have the same length");   /* array.rs:72-76 */
        if (a.vals->dtype != b.vals->dtype) FAIL(RDF_INVALID_ARGUMENT, "both lists must have the same child dtype");
    }
    if (!out_offsets || !out_values) FAIL(RDF_INVALID_ARGUMENT, "null argument");
    if (op == SET_REPEAT && count < 0) FAIL(RDF_INVALID_ARGUMENT, "negative count");
    if (out_offsets->dtype != RDF_I32 || out_values->dtype != a.vals->dtype) FAIL(RDF_INVALID_ARGUMENT, "outputs are (Int32 offsets, child dtype values)");
    if (out_offsets->capacity < a.n + 1) FAIL(RDF_MEMORY_ERROR, "output capacity too small");
    size_t es = (size_t)dtype_size(a.vals->dtype);
    int64_t o = 0;
    int32_t* oo = (int32_t*)out_offsets->values;
    for (int64_t i = 0; i < a.n; i++) {
        oo[i] = (int32_t)o;
        if (!list_valid(&a, i)) continue;                                   /* c.append(true): an empty valid list, array.rs:83 */
        int64_t na = a.off[i + 1] - a.off[i], nb = lb ? b.off[i + 1] - b.off[i] : 0, nu = 0, no = 0;
        eref* u = (eref*)malloc(sizeof(eref) * (size_t)(na + nb + 1));
        eref* w = (eref*)malloc(sizeof(eref) * (size_t)(nb + 1));
        eref* r = (eref*)malloc(sizeof(eref) * (size_t)(na + 1));
        if (!u || !w || !r) { free(u); free(w); free(r); FAIL(RDF_MEMORY_ERROR, "out of memory"); }
        for (int64_t k = 0; k < na; k++) u[nu++] = (eref){a.vals, a.off[i] + k};
        for (int64_t k = 0; k < nb; k++) w[k] = (eref){b.vals, b.off[i] + k};
        const eref* res = u;
        int64_t nres = 0, reps = 1;
        switch (op) {
            case SET_DISTINCT: evec_unique(u, &nu); nres = nu; break;
            case SET_EXCEPT: {
                int64_t nw = nb;
                evec_unique(u, &nu);
                evec_unique(w, &nw);
                for (int64_t x = 0; x < nw; x++)
                    for (int64_t y = nu - 1; y >= 0; y--)
                        if (eref_eq(w[x], u[y])) evec_remove(u, &nu, y);
                nres = nu;
                break;
            }
            case SET_INTERSECT:
                evec_unique(u, &nu);
                for (int64_t x = 0; x < nu; x++)
                    for (int64_t y = 0; y < nb; y++)
                        if (eref_eq(u[x], w[y])) { r[no++] = u[x]; break; }
                res = r; nres = no;
                break;
            case SET_UNION:
                for (int64_t k = 0; k < nb; k++) u[nu++] = w[k];
                evec_unique(u, &nu);
                nres = nu;
                break;
            default: nres = na; reps = count; break;                        /* times(count) */
        }
        for (int64_t t = 0; t < reps; t++)
            for (int64_t k = 0; k < nres; k++) {
                if (o >= out_values->capacity) { free(u); free(w); free(r); FAIL(RDF_MEMORY_ERROR, "values capacity too small"); }
                memcpy((char*)out_values->values + (size_t)o * es, (const char*)res[k].arr->values + (size_t)(res[k].arr->offset + res[k].e) * es, es);
                o++;
            }
        free(u); free(w); free(r);
        if (o > INT32_MAX) FAIL(RDF_COMPUTE_ERROR, "result overflows the Int32 value_offsets");
    }
    oo[a.n] = (int32_t)o;
    out_offsets->length = a.n + 1; out_offsets->null_count = 0;
    out_values->length = o; out_values->null_count = 0;
    if (out_offsets->validity) memset(out_offsets->validity, 0xFF, (size_t)((a.n + 8) / 8));
    if (out_values->validity) memset(out_values->validity, 0xFF, (size_t)((o + 7) / 8));
    return RDF_OK;
}
rdf_status ora_list_distinct(const rdf_list_array* l, rdf_out* oo, rdf_out* ov) { return list_set(l, NULL, SET_DISTINCT, 0, oo, ov); }
rdf_status ora_list_except(const rdf_list_array* a, const rdf_list_array* b, rdf_out* oo, rdf_out* ov) { if (!b) FAIL(RDF_INVALID_ARGUMENT, "null list"); return list_set(a, b, SET_EXCEPT, 0, oo, ov); }
rdf_status ora_list_intersect(const rdf_list_array* a, const rdf_list_array* b, rdf_out* oo, rdf_out* ov) { if (!b) FAIL(RDF_INVALID_ARGUMENT, "null list"); return list_set(a, b, SET_INTERSECT, 0, oo, ov); }
rdf_status ora_list_union(const rdf_list_array* a, const rdf_list_array* b, rdf_out* oo, rdf_out* ov) { if (!b) FAIL(RDF_INVALID_ARGUMENT, "null list"); return list_set(a, b, SET_UNION, 0, oo, ov); }
rdf_status ora_list_repeat(const rdf_list_array* l, int32_t count, rdf_out* oo, rdf_out* ov) { return list_set(l, NULL, SET_REPEAT, count, oo, ov); }

/* ------------------------------------------------------------------ sort
 * DataFrame::sort (src/dataframe.rs:194-214): concat every sort column (Column::to_array), then
 * arrow::compute::lexsort_to_indices with SortOptions{descending, nulls_first: false}.  Restated as a
 * stable merge sort of row indices under the lexicographic comparator (nulls last, then value order,
 * floats in IEEE total order). */
typedef struct { const rdf_array* chunks; int64_t nchunks; const int64_t* row_start; int desc; } sortcol;
static int g_sort_ncols; static const sortcol* g_sort_cols;
static uint64_t sort_bits(const rdf_array* a, int64_t i, int* width) {
    int64_t k = a->offset + i; uint64_t b;
    switch (a->dtype) {
        case RDF_I8: *width = 1; return (uint8_t)(((const uint8_t*)a->values)[k] ^ 0x80u);
        case RDF_U8: *width = 1; return ((const uint8_t*)a->values)[k];
        case RDF_I16: *width = 2; return (uint16_t)(((const uint16_t*)a->values)[k] ^ 0x8000u);
        case RDF_U16: *width = 2; return ((const uint16_t*)a->values)[k];
        case RDF_I32: *width = 4; return ((const uint32_t*)a->values)[k] ^ 0x80000000u;
        case RDF_U32: *width = 4; return ((const uint32_t*)a->values)[k];
        case RDF_F32: *width = 4; b = ((const uint32_t*)a->values)[k]; return (b & 0x80000000u) ? (uint32_t)~b : (b ^ 0x80000000u);
        case RDF_I64: *width = 8; return ((const uint64_t*)a->values)[k] ^ 0x8000000000000000ULL;
        case RDF_F64: *width = 8; b = ((const uint64_t*)a->values)[k]; return (b >> 63) ? ~b : (b ^ 0x8000000000000000ULL);
        default: *width = 8; return ((const uint64_t*)a->values)[k];
    }
}
static void sort_locate(const sortcol* sc, int64_t row, const rdf_array** a, int64_t* i) {
    int64_t c = 0;
    while (c + 1 < sc->nchunks && sc->row_start[c + 1] <= row) c++;
    *a = &sc->chunks[c]; *i = row - sc->row_start[c];
}
static int sort_cmp_rows(uint32_t x, uint32_t y) {
    for (int k = 0; k < g_sort_ncols; k++) {
        const sortcol* sc = &g_sort_cols[k];
        const rdf_array *ax, *ay; int64_t ix, iy; int w;
        sort_locate(sc, x, &ax, &ix); sort_locate(sc, y, &ay, &iy);
        int nx = !arr_valid(ax, ix), ny = !arr_valid(ay, iy);
        if (nx != ny) return nx ? 1 : -1;         /* nulls last */
        if (nx) continue;
        uint64_t bx = sort_bits(ax, ix, &w), by = sort_bits(ay, iy, &w);
        if (bx != by) { int lt = bx < by ? -1 : 1; return sc->desc ? -lt : lt; }
    }
    return 0;
}
static void sort_merge(uint32_t* v, uint32_t* tmp, int64_t n) {
    if (n < 2) return;
    int64_t h = n / 2;
    sort_merge(v, tmp, h); sort_merge(v + h, tmp, n - h);
    int64_t i = 0, j = h, o = 0;
    while (i < h && j < n) tmp[o++] = sort_cmp_rows(v[j], v[i]) < 0 ? v[j++] : v[i++];   /* stable */
    while (i < h) tmp[o++] = v[i++];
    while (j < n) tmp[o++] = v[j++];
    memcpy(v, tmp, (size_t)n * sizeof(uint32_t));
}
rdf_status ora_sort_to_indices(const rdf_array* cols, int32_t ncols, int64_t nchunks, const rdf_sort_options* opts, rdf_out* out) {
    if (ncols < 1) FAIL(RDF_COMPUTE_ERROR, "Sort criteria cannot be empty");
    int64_t* row_start = (int64_t*)calloc((size_t)nchunks + 1, sizeof(int64_t));
    sortcol* sc = (sortcol*)calloc((size_t)ncols, sizeof(sortcol));
    for (int64_t c = 0; c < nchunks; c++) row_start[c + 1] = row_start[c] + cols[c].length;
    int64_t n = row_start[nchunks];
    for (int k = 0; k < ncols; k++) { sc[k].chunks = cols + (int64_t)k * nchunks; sc[k].nchunks = nchunks; sc[k].row_start = row_start; sc[k].desc = opts ? opts[k].descending : 0; }
    rdf_status st = RDF_OK;
    if (out->capacity < n) { st = RDF_MEMORY_ERROR; snprintf(g_err, sizeof g_err, "output capacity too small"); }
    else {
        uint32_t* v = (uint32_t*)out->values;
        uint32_t* tmp = (uint32_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(uint32_t));
        for (int64_t i = 0; i < n; i++) v[i] = (uint32_t)i;
        g_sort_ncols = ncols; g_sort_cols = sc;
        sort_merge(v, tmp, n);
        free(tmp);
        out_begin(out, n);
    }
    free(row_start); free(sc);
    return st;
}

/* ------------------------------------------------------------------ join
 * calc_equijoin_indices (src/functions/join.rs:19-137): build HashMap<key bytes, Vec<row>> per side
 * (build_hash_inputs :139-215; rows with a NULL criterion go to the `nulls` lists), then per join type emit
 * (Some(l), Some(r)) for every pair of rows sharing a key, (Some(l), None) / (None, Some(r)) for the outer
 * side's rows without a partner and for its NULL-key rows.  Restated with nested loops over the rows
 * (quadratic, test sizes only) in a deterministic order: probe rows ascending, partners ascending, then —
 * FULL — the unmatched build rows ascending.  FullJoin is implemented as a true full outer join (the
 * reference's arm forgets the unmatched non-NULL rows). */
static void join_locate(const rdf_array* chunks, int64_t nchunks, int64_t row, const rdf_array** a, int64_t* i) {
    int64_t c = 0;
    while (c + 1 < nchunks && row >= chunks[c].length) { row -= chunks[c].length; c++; }
    *a = &chunks[c]; *i = row;
}
rdf_status ora_equijoin_indices_multi(const rdf_array* lk, int64_t lnc, const rdf_array* rk, int64_t rnc, int32_t nkeys, int32_t jt,
                                      rdf_out* out_left, rdf_out* out_right, int64_t* out_rows) {
    if (nkeys < 1 || nkeys > 4) FAIL(RDF_INVALID_ARGUMENT, "join: 1 to 4 key columns per side");
    int64_t nl = 0, nr = 0;
    for (int64_t c = 0; c < lnc; c++) nl += lk[c].length;
    for (int64_t c = 0; c < rnc; c++) nr += rk[c].length;
    int swap = jt == RDF_JOIN_RIGHT, outer = jt != RDF_JOIN_INNER, full = jt == RDF_JOIN_FULL;
    const rdf_array* pk = swap ? rk : lk; const rdf_array* bk = swap ? lk : rk;
    int64_t pnc = swap ? rnc : lnc, bnc = swap ? lnc : rnc, np = swap ? nr : nl, nb = swap ? nl : nr;
    /* key tuples as order-preserving bits, one row of nkeys words per table row; a NULL in any key column = NULL key */
    uint64_t* bbits = (uint64_t*)malloc((size_t)(nb + 1) * 8 * (size_t)nkeys); uint8_t* bnull = (uint8_t*)calloc((size_t)nb + 1, 1);
    uint8_t* bmatched = (uint8_t*)calloc((size_t)nb + 1, 1);
    for (int64_t j = 0; j < nb; j++)
        for (int k = 0; k < nkeys; k++) { const rdf_array* a; int64_t i; int w; join_locate(bk + (int64_t)k * bnc, bnc, j, &a, &i); bnull[j] |= !arr_valid(a, i); bbits[j * nkeys + k] = sort_bits(a, i, &w); }
    rdf_out* op = swap ? out_right : out_left; rdf_out* ob = swap ? out_left : out_right;
    int64_t rows = 0; rdf_status st = RDF_OK;
    for (int pass = 0; pass < 2 && st == RDF_OK; pass++) {   /* pass 0 counts, pass 1 writes */
        if (pass == 1) {
            *out_rows = rows;
            if (!out_left) break;
            if (out_left->capacity < rows || out_right->capacity < rows) { st = RDF_MEMORY_ERROR; snprintf(g_err, sizeof g_err, "join: output capacity too small"); break; }
            if ((outer && !ob->validity) || (full && !op->validity)) { st = RDF_INVALID_ARGUMENT; snprintf(g_err, sizeof g_err, "output validity buffer required"); break; }
            out_begin(op, rows); out_begin(ob, rows);
        }
        int64_t o = 0;
        for (int64_t x = 0; x < np; x++) {
            uint64_t bits[4]; int pn = 0; int64_t m = 0;
            for (int k = 0; k < nkeys; k++) { const rdf_array* a; int64_t i; int w; join_locate(pk + (int64_t)k * pnc, pnc, x, &a, &i); pn |= !arr_valid(a, i); bits[k] = sort_bits(a, i, &w); }
            if (!pn) for (int64_t j = 0; j < nb; j++) {
                if (bnull[j]) continue;
                int eq = 1;
                for (int k = 0; k < nkeys; k++) eq &= bbits[j * nkeys + k] == bits[k];
                if (!eq) continue;
                if (pass == 1) { ((uint32_t*)op->values)[o] = (uint32_t)x; ((uint32_t*)ob->values)[o] = (uint32_t)j; }
                bmatched[j] = 1; o++; m++;
            }
            if (m == 0 && outer) { if (pass == 1) { ((uint32_t*)op->values)[o] = (uint32_t)x; ((uint32_t*)ob->values)[o] = 0; out_null(ob, o); } o++; }
        }
        if (full) for (int64_t j = 0; j < nb; j++) if (!bmatched[j]) {
            if (pass == 1) { ((uint32_t*)op->values)[o] = 0; ((uint32_t*)ob->values)[o] = (uint32_t)j; out_null(op, o); }
            o++;
        }
        rows = o;
    }
    free(bbits); free(bnull); free(bmatched);
    return st;
}
rdf_status ora_equijoin_indices(const rdf_array* lk, int64_t lnc, const rdf_array* rk, int64_t rnc, int32_t jt,
                                rdf_out* out_left, rdf_out* out_right, int64_t* out_rows) {
    return ora_equijoin_indices_multi(lk, lnc, rk, rnc, 1, jt, out_left, out_right, out_rows);
}

/* ------------------------------------------------------------------ group-by
 * Transformation::GroupAggregate has a schema (Dataset::try_aggregate, src/expression.rs:114-221) but no
 * execution in the reference (src/evaluation.rs:73: panic!("aggregations not supported")): PARITY
 * UNPINNED BY THE REFERENCE.  Restated from SQL semantics with a sequential scan and a chained hash
 * map: NULL keys form one group, NULL values are skipped, groups are emitted in first-seen order. */
static int64_t key_at(const rdf_array* a, int64_t i) {
    int64_t k = a->offset + i;
    switch (a->dtype) {
        case RDF_I8: return ((const int8_t*)a->values)[k];
        case RDF_I16: return ((const int16_t*)a->values)[k];
        case RDF_I32: return ((const int32_t*)a->values)[k];
        case RDF_U8: return ((const uint8_t*)a->values)[k];
        case RDF_U16: return ((const uint16_t*)a->values)[k];
        case RDF_U32: return ((const uint32_t*)a->values)[k];
        default: return ((const int64_t*)a->values)[k];
    }
}
rdf_status ora_groupby_sum(const rdf_array* keys, const rdf_array* values, int64_t nchunks, int64_t max_groups,
                           rdf_out* out_keys, rdf_out* out_sums, rdf_out* out_counts) {
    if (nchunks < 1) FAIL(RDF_INVALID_ARGUMENT, "groupby: a column has at least one chunk");
    int kdt = keys[0].dtype, vdt = values ? values[0].dtype : -1;
    if (!(kdt >= RDF_I8 && kdt <= RDF_U64)) FAIL(RDF_INVALID_ARGUMENT, "groupby: integer key column required");
    if (values && !is_numeric(vdt)) FAIL(RDF_INVALID_ARGUMENT, "groupby: numeric value column required");
    int fsum = values && is_float(vdt);
    int64_t cap = 1024;
    while (cap < 4 * (max_groups + 2)) cap <<= 1;
    int64_t* slot_of = (int64_t*)malloc(sizeof(int64_t) * (size_t)cap);   /* hash slot -> group index or -1 */
    int64_t* gkey = (int64_t*)malloc(sizeof(int64_t) * (size_t)(max_groups + 2));
    int* gnull = (int*)calloc((size_t)(max_groups + 2), sizeof(int));
    double* gsumf = (double*)calloc((size_t)(max_groups + 2), sizeof(double));
    uint64_t* gsumi = (uint64_t*)calloc((size_t)(max_groups + 2), sizeof(uint64_t));
    int64_t* gcnt = (int64_t*)calloc((size_t)(max_groups + 2), sizeof(int64_t));
    if (!slot_of || !gkey || !gnull || !gsumf || !gsumi || !gcnt) FAIL(RDF_MEMORY_ERROR, "out of memory");
    for (int64_t i = 0; i < cap; i++) slot_of[i] = -1;
    int64_t ng = 0, null_group = -1, distinct = 0;
    rdf_status st = RDF_OK;
    for (int64_t c = 0; c < nchunks && st == RDF_OK; c++) {
        if (values && values[c].length != keys[c].length) { st = RDF_COMPUTE_ERROR; snprintf(g_err, sizeof g_err, "groupby: key and value chunks differ in length"); break; }
        for (int64_t i = 0; i < keys[c].length; i++) {
            int64_t g;
            if (!arr_valid(&keys[c], i)) {
                if (null_group < 0) { null_group = ng; gnull[ng] = 1; gkey[ng] = 0; ng++; }
                g = null_group;
            } else {
                int64_t k = key_at(&keys[c], i);
                uint64_t h = (uint64_t)k * 0x9E3779B97F4A7C15ULL;
                int64_t s = (int64_t)((h ^ (h >> 29)) & (uint64_t)(cap - 1));
                while (slot_of[s] >= 0 && gkey[slot_of[s]] != k) s = (s + 1) & (cap - 1);
                if (slot_of[s] < 0) {
                    if (++distinct > max_groups + 1) { st = RDF_MEMORY_ERROR; snprintf(g_err, sizeof g_err, "groupby: more than max_groups distinct keys"); break; }
                    slot_of[s] = ng; gkey[ng] = k; ng++;
                }
                g = slot_of[s];
            }
            if (!values) { gcnt[g]++; continue; }
            if (!arr_valid(&values[c], i)) continue;
            if (fsum) gsumf[g] += arr_f64(&values[c], i);
            else gsumi[g] += (uint64_t)key_at(&values[c], i);
            gcnt[g]++;
        }
    }
    if (st == RDF_OK && (out_keys->capacity < ng || out_sums->capacity < ng || out_counts->capacity < ng)) { st = RDF_MEMORY_ERROR; snprintf(g_err, sizeof g_err, "output capacity too small"); }
    if (st == RDF_OK && null_group >= 0 && !out_keys->validity) { st = RDF_INVALID_ARGUMENT; snprintf(g_err, sizeof g_err, "output validity buffer required"); }
    if (st == RDF_OK) {
        out_begin(out_keys, ng); out_begin(out_sums, ng); out_begin(out_counts, ng);
        int es = dtype_size(kdt);
        for (int64_t g = 0; g < ng; g++) {
            uint64_t kv = (uint64_t)gkey[g];
            memcpy((char*)out_keys->values + g * es, &kv, (size_t)es);
            if (gnull[g]) out_null(out_keys, g);
            if (fsum) ((double*)out_sums->values)[g] = gsumf[g];
            else ((int64_t*)out_sums->values)[g] = (int64_t)gsumi[g];
            ((int64_t*)out_counts->values)[g] = gcnt[g];
        }
    }
    free(slot_of); free(gkey); free(gnull); free(gsumf); free(gsumi); free(gcnt);
    return st;
}

/* GroupAggregate(groups, [Sum | Min | Max | Count]) over 1..4 grouping columns (AggregateFunction, src/expression.rs:696-711;
 * planned by Dataset::try_aggregate :114-221, never executed: PARITY UNPINNED BY THE REFERENCE).  SQL semantics with a
 * sequential scan: a NULL in a grouping column is a value of its own, NULL values are skipped, MIN / MAX of a group
 * without a non-NULL value is NULL, NaN loses against every number (the rule of ora_min / ora_max).  Groups in first-seen order. */
static rdf_status groupby_generic(const rdf_array* keys, int32_t nkeys, const rdf_array* values, const rdf_array* weights,
                                  int64_t nchunks, int32_t agg, int64_t max_groups,
                                  rdf_out* out_keys, rdf_out* out_values, rdf_out* out_counts) {
    if (nchunks < 1) FAIL(RDF_INVALID_ARGUMENT, "groupby: a column has at least one chunk");
    if (nkeys < 1 || nkeys > RDF_MAX_GROUP_KEYS) FAIL(RDF_INVALID_ARGUMENT, "groupby: 1..4 grouping columns");
    if (agg < RDF_AGG_SUM || agg > RDF_AGG_COUNT) FAIL(RDF_INVALID_ARGUMENT, "groupby: unknown aggregate");
    if (agg == RDF_AGG_COUNT) values = NULL; else if (!values) agg = RDF_AGG_COUNT;
    for (int k = 0; k < nkeys; k++) {
        int kdt = keys[(int64_t)k * nchunks].dtype;
        if (!(kdt >= RDF_I8 && kdt <= RDF_U64)) FAIL(RDF_INVALID_ARGUMENT, "groupby: integer key column required");
    }
    int vdt = values ? values[0].dtype : -1;
    if (values && !is_numeric(vdt)) FAIL(RDF_INVALID_ARGUMENT, "groupby: numeric value column required");
    int fval = values && is_float(vdt), uval = values && vdt == RDF_U64;
    int64_t cap = 1024;
    while (cap < 4 * (max_groups + 2)) cap <<= 1;
    int64_t G = max_groups + 2;
    int64_t* slot_of = (int64_t*)malloc(sizeof(int64_t) * (size_t)cap);
    int64_t* gkey = (int64_t*)malloc(sizeof(int64_t) * (size_t)G * (size_t)nkeys);
    uint8_t* gnull = (uint8_t*)calloc((size_t)G * (size_t)nkeys, 1);
    double* gf = (double*)calloc((size_t)G, sizeof(double));
    uint64_t* gi = (uint64_t*)calloc((size_t)G, sizeof(uint64_t));
    int64_t* gcnt = (int64_t*)calloc((size_t)G, sizeof(int64_t));
    uint8_t* ghas = (uint8_t*)calloc((size_t)G, 1);   /* MIN / MAX: a non-NaN value has been seen */
    if (!slot_of || !gkey || !gnull || !gf || !gi || !gcnt || !ghas) FAIL(RDF_MEMORY_ERROR, "out of memory");
    for (int64_t i = 0; i < cap; i++) slot_of[i] = -1;
    int64_t ng = 0;
    rdf_status st = RDF_OK;
    for (int64_t c = 0; c < nchunks && st == RDF_OK; c++) {
        int64_t len = keys[c].length;
        for (int k = 1; k < nkeys; k++) if (keys[(int64_t)k * nchunks + c].length != len) { st = RDF_COMPUTE_ERROR; snprintf(g_err, sizeof g_err, "groupby: grouping columns differ in length"); }
        if (values && values[c].length != len) { st = RDF_COMPUTE_ERROR; snprintf(g_err, sizeof g_err, "groupby: key and value chunks differ in length"); }
        if (weights && weights[c].length != len) { st = RDF_COMPUTE_ERROR; snprintf(g_err, sizeof g_err, "groupby_merge: arrays differ in length"); }
        if (st != RDF_OK) break;
        for (int64_t i = 0; i < len; i++) {
            int64_t kv[RDF_MAX_GROUP_KEYS]; uint8_t kn[RDF_MAX_GROUP_KEYS];
            uint64_t h = 0x9E3779B97F4A7C15ULL;
            for (int k = 0; k < nkeys; k++) {
                const rdf_array* ka = &keys[(int64_t)k * nchunks + c];
                kn[k] = !arr_valid(ka, i);
                kv[k] = kn[k] ? 0 : key_at(ka, i);
                h = (h ^ (uint64_t)kv[k] ^ ((uint64_t)kn[k] << 63)) * 0xBF58476D1CE4E5B9ULL;
                h ^= h >> 29;
            }
            int64_t s = (int64_t)(h & (uint64_t)(cap - 1));
            for (;;) {
                int64_t g0 = slot_of[s];
                if (g0 < 0) break;
                int same = 1;
                for (int k = 0; k < nkeys; k++) if (gkey[g0 * nkeys + k] != kv[k] || gnull[g0 * nkeys + k] != kn[k]) same = 0;
                if (same) break;
                s = (s + 1) & (cap - 1);
            }
            if (slot_of[s] < 0) {
                if (ng >= G) { st = RDF_MEMORY_ERROR; snprintf(g_err, sizeof g_err, "groupby: more than max_groups distinct keys"); break; }
                slot_of[s] = ng;
                for (int k = 0; k < nkeys; k++) { gkey[ng * nkeys + k] = kv[k]; gnull[ng * nkeys + k] = kn[k]; }
                ng++;
            }
            int64_t g = slot_of[s];
            int64_t w = weights ? ((const int64_t*)weights[c].values)[weights[c].offset + i] : 1;
            if (!values) { gcnt[g] += w; continue; }
            if (!arr_valid(&values[c], i) || w == 0) continue;
            gcnt[g] += w;
            if (agg == RDF_AGG_SUM) {
                if (fval) gf[g] += arr_f64(&values[c], i); else gi[g] += (uint64_t)key_at(&values[c], i);
            } else if (fval) {
                double v = arr_f64(&values[c], i);
                if (v != v) continue;
                if (!ghas[g] || (agg == RDF_AGG_MIN ? v < gf[g] : v > gf[g])) gf[g] = v;
                ghas[g] = 1;
            } else if (uval) {
                uint64_t v = ((const uint64_t*)values[c].values)[values[c].offset + i];
                if (!ghas[g] || (agg == RDF_AGG_MIN ? v < gi[g] : v > gi[g])) gi[g] = v;
                ghas[g] = 1;
            } else {
                int64_t v = key_at(&values[c], i);
                if (!ghas[g] || (agg == RDF_AGG_MIN ? v < (int64_t)gi[g] : v > (int64_t)gi[g])) gi[g] = (uint64_t)v;
                ghas[g] = 1;
            }
        }
    }
    if (st == RDF_OK && ng > max_groups + 2) { st = RDF_MEMORY_ERROR; snprintf(g_err, sizeof g_err, "groupby: more than max_groups distinct keys"); }
    for (int k = 0; k < nkeys && st == RDF_OK; k++) if (out_keys[k].capacity < ng) { st = RDF_MEMORY_ERROR; snprintf(g_err, sizeof g_err, "output capacity too small"); }
    if (st == RDF_OK && (out_values->capacity < ng || out_counts->capacity < ng)) { st = RDF_MEMORY_ERROR; snprintf(g_err, sizeof g_err, "output capacity too small"); }
    if (st == RDF_OK) {
        for (int k = 0; k < nkeys; k++) out_begin(&out_keys[k], ng);
        out_begin(out_values, ng); out_begin(out_counts, ng);
        for (int64_t g = 0; g < ng && st == RDF_OK; g++) {
            for (int k = 0; k < nkeys; k++) {
                int es = dtype_size(out_keys[k].dtype);
                uint64_t kvv = (uint64_t)gkey[g * nkeys + k];
                memcpy((char*)out_keys[k].values + g * es, &kvv, (size_t)es);
                if (gnull[g * nkeys + k]) {
                    if (!out_keys[k].validity) { st = RDF_INVALID_ARGUMENT; snprintf(g_err, sizeof g_err, "output validity buffer required"); break; }
                    out_null(&out_keys[k], g);
                }
            }
            if (st != RDF_OK) break;
            if (agg == RDF_AGG_MIN || agg == RDF_AGG_MAX) {
                if (gcnt[g] == 0) {
                    if (!out_values->validity) { st = RDF_INVALID_ARGUMENT; snprintf(g_err, sizeof g_err, "output validity buffer required"); break; }
                    ((uint64_t*)out_values->values)[g] = 0;
                    out_null(out_values, g);
                } else if (fval) ((double*)out_values->values)[g] = ghas[g] ? gf[g] : (double)NAN;
                else ((uint64_t*)out_values->values)[g] = gi[g];
            } else if (fval) ((double*)out_values->values)[g] = gf[g];
            else ((int64_t*)out_values->values)[g] = (int64_t)gi[g];
            ((int64_t*)out_counts->values)[g] = gcnt[g];
        }
    }
    free(slot_of); free(gkey); free(gnull); free(gf); free(gi); free(gcnt); free(ghas);
    return st;
}
rdf_status ora_groupby_agg(const rdf_array* keys, int32_t nkeys, const rdf_array* values, int64_t nchunks, int32_t agg, int64_t max_groups,
                           rdf_out* out_keys, rdf_out* out_values, rdf_out* out_counts) {
    return groupby_generic(keys, nkeys, values, NULL, nchunks, agg, max_groups, out_keys, out_values, out_counts);
}
/* merge of partial groups: the same scan with the partial's count as the row's weight (a partial with count 0 only
 * contributes its key) */
rdf_status ora_groupby_merge(const rdf_array* keys, const rdf_array* partial, const rdf_array* counts, int32_t agg, int64_t max_groups,
                             rdf_out* out_keys, rdf_out* out_values, rdf_out* out_counts) {
    if (!counts || counts->dtype != RDF_I64) FAIL(RDF_INVALID_ARGUMENT, "groupby_merge: counts are Int64");
    return groupby_generic(keys, 1, partial, counts, 1, partial ? agg : RDF_AGG_COUNT, max_groups, out_keys, out_values, out_counts);
}

/* ------------------------------------------------------------------ synthetic data
 * Counter-based generator shared (by restating the same few lines) with the device fill kernels:
 * SplitMix64 finaliser over (seed, column_id, row). */
static inline uint64_t rdf_hash64(uint64_t seed, uint64_t column_id, uint64_t row) {
    uint64_t z = (seed ^ (column_id * 0xD6E8FEB86659FD93ULL)) + (row + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
rdf_status ora_fill_uniform_f64(double* p, int64_t n, uint64_t seed, uint64_t col, int64_t first_row, double lo, double hi) {
    double span = hi - lo;
    for (int64_t i = 0; i < n; i++) {
        double u = (double)(rdf_hash64(seed, col, (uint64_t)(first_row + i)) >> 11) * (1.0 / 9007199254740992.0);
        double t = span * u;
        p[i] = lo + t;
    }
    return RDF_OK;
}
rdf_status ora_fill_uniform_i64(int64_t* p, int64_t n, uint64_t seed, uint64_t col, int64_t first_row, int64_t lo, int64_t hi) {
    if (hi <= lo) FAIL(RDF_INVALID_ARGUMENT, "fill_uniform_i64: hi must exceed lo");
    uint64_t span = (uint64_t)hi - (uint64_t)lo;
    for (int64_t i = 0; i < n; i++)
        p[i] = (int64_t)((uint64_t)lo + rdf_hash64(seed, col, (uint64_t)(first_row + i)) % span);
    return RDF_OK;
}
rdf_status ora_fill_validity(uint8_t* p, int64_t nbits, uint64_t seed, uint64_t col, int64_t first_row, double null_fraction) {
    double t = null_fraction * 4294967296.0;
    uint64_t thr = t <= 0.0 ? 0 : t >= 4294967296.0 ? 4294967296ULL : (uint64_t)t;
    memset(p, 0, (size_t)((nbits + 7) / 8));
    for (int64_t i = 0; i < nbits; i++) {
        uint64_t h = rdf_hash64(seed ^ 0xA5A5A5A5A5A5A5A5ULL, col, (uint64_t)(first_row + i)) >> 32;
        if (h >= thr) bit_set(p, i);
    }
    return RDF_OK;
}
