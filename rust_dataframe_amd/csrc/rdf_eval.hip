// rdf_eval.hip — the fused expression-tree evaluator (compiled once per feature level, -DRDF_FEAT=0|1|2).
//
//   eval_kernel<FEAT, SINK, NPRE, NVAL>
//     [Evaluate::calculate src/evaluation.rs:97-323 + ScalarFunctions src/functions/scalar.rs:16-540 +
//      BooleanFilter::eval_to_array src/expression.rs:766-861 + AggregateFunctions
//      src/functions/aggregate.rs:12-93 — a maximal run of the batch loop's steps in ONE pass over HBM]
//
// One persistent block walks 1024-row tiles (= the reference's RecordBatch size).  Wave w owns rows
// [256w, 256w+256) of the tile, lane l row 64j + l of it for j = 0..3, so every load instruction of a
// wave covers 64 consecutive elements (512 B for 8-byte types) and the wave's validity bits are four
// consecutive 64-bit windows fetched with a handful of independent scalar loads.  Per tile:
//   (1) ALL loads of the NPRE preloaded columns are issued up front (memory-level parallelism),
//   (2) the wave-uniform accumulator-machine bytecode runs over registers (scalar branches only),
//   (3) the sink consumes: coalesced stores + ballot-built bitmaps (STORE), or running
//       {sum,min,max,count} folded block-wide once at the end (AGG, two-stage reduction).
// Template parameters exist to keep VGPRs (and with them occupancy) where a streaming kernel needs
// them: FEAT 0 = arithmetic/compare/cast/boolean, 1 = + integer division, 2 = + libm functions;
// NPRE = columns held in registers; NVAL = value expressions with live aggregate state.
#include "rdf_common.hip.h"

#ifndef RDF_FEAT
#error "compile with -DRDF_FEAT=0|1|2"
#endif

namespace rdfk {

// ------------------------------------------------------------------------------------------------
// conversions (arrow::compute::cast of the reference's era = num::cast::cast per element, NULL where it returns None:
// an integer that does not fit the target, NaN or an out-of-range float on the way to an integer — num-traits 0.2 range tests)

__device__ __forceinline__ uint64_t cast_value(int from, int to, uint64_t x, bool& ok) {
    ok = true;
    if (from == to) return x;
    double f = 0.0;
    bool src_float = false;
    if (from == RDF_F64) { f = u2d(x); src_float = true; }
    else if (from == RDF_F32) { f = (double)u2f(x); src_float = true; }
    if (to == RDF_BOOL) return src_float ? (uint64_t)(f != 0.0) : (uint64_t)(x != 0);
    if (to == RDF_F64) {
        if (src_float) return d2u(f);
        return d2u(dt_is_signed(from) ? (double)(int64_t)x : (double)x);
    }
    if (to == RDF_F32) {
        if (from == RDF_F64) return f2u((float)u2d(x));
        return f2u(dt_is_signed(from) ? (float)(int64_t)x : (float)x);
    }
    const bool tsigned = dt_is_signed(to);
    int tbits = 64;
    switch (to) { case RDF_I8: case RDF_U8: tbits = 8; break; case RDF_I16: case RDF_U16: tbits = 16; break; case RDF_I32: case RDF_U32: tbits = 32; break; default: break; }
    if (src_float) {   // truncation toward zero when the truncated value fits (num-traits' range tests)
        if (tsigned) {
            const double lim = tbits == 64 ? 9223372036854775808.0 : (double)(1ll << (tbits - 1));
            ok = tbits == 64 ? (f >= -lim && f < lim) : (f > -lim - 1.0 && f < lim);
            return ok ? (uint64_t)(int64_t)f : 0;
        }
        const double lim = tbits == 64 ? 18446744073709551616.0 : (double)(1ull << tbits);
        ok = f > -1.0 && f < lim;
        return ok ? (uint64_t)f : 0;
    }
    // integer (or Boolean) -> integer: NULL unless representable
    if (from == RDF_BOOL) return x & 1;
    if (tsigned) {
        const int64_t hi = tbits == 64 ? INT64_MAX : ((int64_t)1 << (tbits - 1)) - 1, lo = -hi - 1;
        ok = dt_is_signed(from) ? ((int64_t)x >= lo && (int64_t)x <= hi) : (x <= (uint64_t)hi);
    } else {
        const uint64_t hi = tbits == 64 ? ~0ull : ((1ull << tbits) - 1);
        ok = dt_is_signed(from) ? ((int64_t)x >= 0 && x <= hi) : (x <= hi);
    }
    return ok ? x : 0;
}
__device__ __forceinline__ uint64_t cast_value(int from, int to, uint64_t x) { bool ok; return cast_value(from, to, x, ok); }

// ------------------------------------------------------------------------------------------------
// column loads.  A wave owns 256 consecutive rows of its tile; lane l holds rows 4 l .. 4 l + 3 (row(j) = rw + 4 l + j), so
// a full tile of an aligned column comes in with ONE 16-byte load per lane for 4-byte types, two for 8-byte types (8 / 4
// bytes for 2- / 1-byte types) — the interpreter used to load one element per lane and instruction (row(j) = rw + 64 j + l),
// which was most of its distance to the specialised kernels on narrow types.  Partial tiles and columns whose chunk does not
// start on a vector boundary take one load per element of the same rows.

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));
typedef uint8_t u8x4 __attribute__((ext_vector_type(4)));

// the lane's 4 bits (rows 4 l .. 4 l + 3) of a 256-bit window held as four 64-bit words
__device__ __forceinline__ uint32_t lane_nibble(const uint64_t (&w)[kVPT], int lane) {
    const int q = lane >> 4;
    const uint64_t word = q == 0 ? w[0] : q == 1 ? w[1] : q == 2 ? w[2] : w[3];
    return (uint32_t)(word >> ((lane & 15) * 4)) & 15u;
}
// the inverse: every lane's 4 bits -> the four 64-bit words of the wave's 256 rows (valid in lanes 0, 16, 32, 48: word lane / 16)
__device__ __forceinline__ uint64_t nibbles_to_word(uint32_t nib, int lane) {
    uint64_t x = (uint64_t)(nib & 15u) << ((lane & 15) * 4);
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) x |= shfl_xor64(x, m);
    return x;
}

__device__ __forceinline__ void load_col(const DevChunkCol cc, int dt, int64_t rw, int64_t clen, uint32_t inr,
                                         uint64_t (&v)[kVPT], uint32_t& valid) {
    // rw = first row of this wave's 256-row span (wave-uniform)
    const int lane = threadIdx.x & 63;
    const int64_t e0 = cc.offset + rw + (int64_t)lane * kVPT;
    const bool full = __ballot(inr != (1u << kVPT) - 1) == 0;    // every row of every lane exists
    switch (dt) {
        case RDF_I64: case RDF_U64: case RDF_F64: {
            const GlobalPtr<uint64_t> p = as_global<uint64_t>(cc.values) + e0;
            if (full && (((uintptr_t)cc.values + (uintptr_t)(cc.offset + rw) * 8) & 15) == 0) {
                typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
                const u64x2 a0 = __builtin_nontemporal_load((GlobalPtr<u64x2>)p), a1 = __builtin_nontemporal_load((GlobalPtr<u64x2>)p + 1);
                v[0] = a0[0]; v[1] = a0[1]; v[2] = a1[0]; v[3] = a1[1];
            } else {
#pragma unroll
                for (int j = 0; j < kVPT; ++j) v[j] = (inr >> j) & 1 ? __builtin_nontemporal_load(p + j) : 0;
            }
        } break;
        case RDF_I32: case RDF_U32: case RDF_F32: {
            const GlobalPtr<uint32_t> p = as_global<uint32_t>(cc.values) + e0;
            uint32_t t[kVPT];
            if (full && (((uintptr_t)cc.values + (uintptr_t)(cc.offset + rw) * 4) & 15) == 0) {
                const u32x4 a0 = __builtin_nontemporal_load((GlobalPtr<u32x4>)p);
                t[0] = a0[0]; t[1] = a0[1]; t[2] = a0[2]; t[3] = a0[3];
            } else {
#pragma unroll
                for (int j = 0; j < kVPT; ++j) t[j] = (inr >> j) & 1 ? __builtin_nontemporal_load(p + j) : 0;
            }
#pragma unroll
            for (int j = 0; j < kVPT; ++j) v[j] = dt == RDF_I32 ? (uint64_t)(int64_t)(int32_t)t[j] : (uint64_t)t[j];
        } break;
        case RDF_I16: case RDF_U16: {
            const GlobalPtr<uint16_t> p = as_global<uint16_t>(cc.values) + e0;
            uint16_t t[kVPT];
            if (full && (((uintptr_t)cc.values + (uintptr_t)(cc.offset + rw) * 2) & 7) == 0) {
                const u16x4 a0 = __builtin_nontemporal_load((GlobalPtr<u16x4>)p);
                t[0] = a0[0]; t[1] = a0[1]; t[2] = a0[2]; t[3] = a0[3];
            } else {
#pragma unroll
                for (int j = 0; j < kVPT; ++j) t[j] = (inr >> j) & 1 ? p[j] : (uint16_t)0;
            }
#pragma unroll
            for (int j = 0; j < kVPT; ++j) v[j] = dt == RDF_I16 ? (uint64_t)(int64_t)(int16_t)t[j] : (uint64_t)t[j];
        } break;
        case RDF_I8: case RDF_U8: {
            const GlobalPtr<uint8_t> p = as_global<uint8_t>(cc.values) + e0;
            uint8_t t[kVPT];
            if (full && (((uintptr_t)cc.values + (uintptr_t)(cc.offset + rw)) & 3) == 0) {
                const u8x4 a0 = __builtin_nontemporal_load((GlobalPtr<u8x4>)p);
                t[0] = a0[0]; t[1] = a0[1]; t[2] = a0[2]; t[3] = a0[3];
            } else {
#pragma unroll
                for (int j = 0; j < kVPT; ++j) t[j] = (inr >> j) & 1 ? p[j] : (uint8_t)0;
            }
#pragma unroll
            for (int j = 0; j < kVPT; ++j) v[j] = dt == RDF_I8 ? (uint64_t)(int64_t)(int8_t)t[j] : (uint64_t)t[j];
        } break;
        default: {  // RDF_BOOL: bit-packed values
            uint64_t w[kVPT];
            load_windows<kVPT>((const uint8_t*)cc.values, cc.offset + rw, clen - rw, w);
            const uint32_t nib = lane_nibble(w, lane);
#pragma unroll
            for (int j = 0; j < kVPT; ++j) v[j] = (nib >> j) & 1;
        }
    }
    valid = (1u << kVPT) - 1;
    if (cc.validity) {
        uint64_t w[kVPT];
        load_windows<kVPT>(cc.validity, cc.offset + rw, clen - rw, w);
        valid = lane_nibble(w, lane);
    }
}

// ------------------------------------------------------------------------------------------------
// arithmetic

template <int FEAT>
__device__ __forceinline__ double unary_f64(int op, double x) {
    switch (op) {
        case RDF_OP_ABS: return fabs(x);
        case RDF_OP_CEIL: return ceil(x);
        case RDF_OP_FLOOR: return floor(x);
        case RDF_OP_ROUND: return round(x);
        case RDF_OP_SQRT: return sqrt(x);
        case RDF_OP_DEGREES: return x * (180.0 / 3.14159265358979323846264338327950288);
        case RDF_OP_RADIANS: return x * (3.14159265358979323846264338327950288 / 180.0);
        default: break;
    }
    if constexpr (FEAT >= 2) {
        switch (op) {
            case RDF_OP_ACOS: return acos(x);
            case RDF_OP_ASIN: return asin(x);
            case RDF_OP_ATAN: return atan(x);
            case RDF_OP_CBRT: return cbrt(x);
            case RDF_OP_COS: return rdf_cos(x);
            case RDF_OP_COSH: return cosh(x);
            case RDF_OP_EXP: return exp(x);
            case RDF_OP_EXPM1: return expm1(x);
            case RDF_OP_LOG10: return log10(x);
            case RDF_OP_LOG2: return log2(x);
            case RDF_OP_SIN: return rdf_sin(x);
            case RDF_OP_SINH: return sinh(x);
            case RDF_OP_TAN: return rdf_tan(x);
            case RDF_OP_TANH: return tanh(x);
            case RDF_OP_COT: return 1.0 / rdf_tan(x);
            case RDF_OP_SEC: return 1.0 / rdf_cos(x);
            case RDF_OP_CSC: return 1.0 / rdf_sin(x);
            default: break;
        }
    }
    return x;
}
template <int FEAT>
__device__ __forceinline__ float unary_f32(int op, float x) {
    switch (op) {
        case RDF_OP_ABS: return fabsf(x);
        case RDF_OP_CEIL: return ceilf(x);
        case RDF_OP_FLOOR: return floorf(x);
        case RDF_OP_ROUND: return roundf(x);
        case RDF_OP_SQRT: return sqrtf(x);
        case RDF_OP_DEGREES: return x * 57.2957795130823208767981548141051703f;
        case RDF_OP_RADIANS: return x * (3.14159265358979323846264338327950288f / 180.0f);
        default: break;
    }
    if constexpr (FEAT >= 2) {
        switch (op) {
            case RDF_OP_ACOS: return acosf(x);
            case RDF_OP_ASIN: return asinf(x);
            case RDF_OP_ATAN: return atanf(x);
            case RDF_OP_CBRT: return cbrtf(x);
            case RDF_OP_COS: return rdf_cos(x);
            case RDF_OP_COSH: return coshf(x);
            case RDF_OP_EXP: return expf(x);
            case RDF_OP_EXPM1: return expm1f(x);
            case RDF_OP_LOG10: return log10f(x);
            case RDF_OP_LOG2: return log2f(x);
            case RDF_OP_SIN: return rdf_sin(x);
            case RDF_OP_SINH: return sinhf(x);
            case RDF_OP_TAN: return tanf(x);
            case RDF_OP_TANH: return tanhf(x);
            case RDF_OP_COT: return 1.0f / tanf(x);
            case RDF_OP_SEC: return 1.0f / rdf_cos(x);
            case RDF_OP_CSC: return 1.0f / rdf_sin(x);
            default: break;
        }
    }
    return x;
}

#define RDF_ROWS _Pragma("unroll") for (int j = 0; j < kVPT; ++j)

// acc = acc OP b.  `live` = rows where both sides are valid and in range (divide-by-zero is only an
// error there, like arrow's math_divide).
template <int FEAT>
__device__ __forceinline__ void apply_binary(int op, int dt, uint64_t (&acc)[kVPT], const uint64_t (&b)[kVPT],
                                             uint32_t live, uint32_t& err) {
    if (op >= RDF_OP_GT && op <= RDF_OP_LE) {  // f64 comparisons (src/expression.rs:844-852)
        switch (op) {
            case RDF_OP_GT: RDF_ROWS acc[j] = u2d(acc[j]) > u2d(b[j]); break;
            case RDF_OP_GE: RDF_ROWS acc[j] = u2d(acc[j]) >= u2d(b[j]); break;
            case RDF_OP_EQ: RDF_ROWS acc[j] = u2d(acc[j]) == u2d(b[j]); break;
            case RDF_OP_NE: RDF_ROWS acc[j] = u2d(acc[j]) != u2d(b[j]); break;
            case RDF_OP_LT: RDF_ROWS acc[j] = u2d(acc[j]) < u2d(b[j]); break;
            default: RDF_ROWS acc[j] = u2d(acc[j]) <= u2d(b[j]); break;
        }
        return;
    }
    if (op == RDF_OP_AND) { RDF_ROWS acc[j] = acc[j] & b[j]; return; }
    if (op == RDF_OP_OR) { RDF_ROWS acc[j] = acc[j] | b[j]; return; }
    if (dt == RDF_F64) {
        switch (op) {
            case RDF_OP_ADD: RDF_ROWS acc[j] = d2u(u2d(acc[j]) + u2d(b[j])); break;
            case RDF_OP_SUB: RDF_ROWS acc[j] = d2u(u2d(acc[j]) - u2d(b[j])); break;
            case RDF_OP_MUL: RDF_ROWS acc[j] = d2u(u2d(acc[j]) * u2d(b[j])); break;
            case RDF_OP_DIV:
                RDF_ROWS {
                    bool z = u2d(b[j]) == 0.0;
                    if (z && ((live >> j) & 1)) err |= 1u;
                    acc[j] = z ? 0 : d2u(u2d(acc[j]) / u2d(b[j]));
                }
                break;
            default:
                if constexpr (FEAT >= 2) {
                    if (op == RDF_OP_ATAN2) RDF_ROWS acc[j] = d2u(atan2(u2d(acc[j]), u2d(b[j])));
                    else if (op == RDF_OP_HYPOT) RDF_ROWS acc[j] = d2u(hypot(u2d(acc[j]), u2d(b[j])));
                    else RDF_ROWS acc[j] = d2u(log(u2d(acc[j])) / log(u2d(b[j])));
                }
        }
        return;
    }
    if (dt == RDF_F32) {
        switch (op) {
            case RDF_OP_ADD: RDF_ROWS acc[j] = f2u(u2f(acc[j]) + u2f(b[j])); break;
            case RDF_OP_SUB: RDF_ROWS acc[j] = f2u(u2f(acc[j]) - u2f(b[j])); break;
            case RDF_OP_MUL: RDF_ROWS acc[j] = f2u(u2f(acc[j]) * u2f(b[j])); break;
            case RDF_OP_DIV:
                RDF_ROWS {
                    bool z = u2f(b[j]) == 0.0f;
                    if (z && ((live >> j) & 1)) err |= 1u;
                    acc[j] = z ? 0 : f2u(u2f(acc[j]) / u2f(b[j]));
                }
                break;
            default:
                if constexpr (FEAT >= 2) {
                    if (op == RDF_OP_ATAN2) RDF_ROWS acc[j] = f2u(atan2f(u2f(acc[j]), u2f(b[j])));
                    else if (op == RDF_OP_HYPOT) RDF_ROWS acc[j] = f2u(hypotf(u2f(acc[j]), u2f(b[j])));
                    else RDF_ROWS acc[j] = f2u(logf(u2f(acc[j])) / logf(u2f(b[j])));
                }
        }
        return;
    }
    // integers: wrapping arithmetic in 64 bits, then re-normalised to the value's width
    switch (op) {
        case RDF_OP_ADD: RDF_ROWS acc[j] = normalize_int(dt, acc[j] + b[j]); break;
        case RDF_OP_SUB: RDF_ROWS acc[j] = normalize_int(dt, acc[j] - b[j]); break;
        case RDF_OP_MUL: RDF_ROWS acc[j] = normalize_int(dt, acc[j] * b[j]); break;
        default:  // DIV
            if constexpr (FEAT >= 1) {
                RDF_ROWS {
                    bool z = b[j] == 0;
                    if (z && ((live >> j) & 1)) err |= 1u;
                    uint64_t q;
                    if (z) q = 0;
                    else if (dt_is_signed(dt)) {
                        int64_t x = (int64_t)acc[j], y = (int64_t)b[j];
                        q = y == -1 ? (uint64_t)0 - (uint64_t)x : (uint64_t)(x / y);  // MIN / -1 wraps
                    } else q = acc[j] / b[j];
                    acc[j] = normalize_int(dt, q);
                }
            }
    }
}

template <int FEAT>
__device__ __forceinline__ void apply_unary(int op, int dt, uint64_t (&acc)[kVPT]) {
    if (op == RDF_OP_NOT) { RDF_ROWS acc[j] = acc[j] ^ 1ull; return; }
    if (FEAT >= 1 && op >= RDF_OP_HOUR_S && op <= RDF_OP_HOUR_DAY) {
        switch (op) {
            case RDF_OP_HOUR_S: RDF_ROWS acc[j] = hour_of<1>((int64_t)acc[j]); break;
            case RDF_OP_HOUR_MS: RDF_ROWS acc[j] = hour_of<1000>((int64_t)acc[j]); break;
            case RDF_OP_HOUR_US: RDF_ROWS acc[j] = hour_of<1000000>((int64_t)acc[j]); break;
            case RDF_OP_HOUR_NS: RDF_ROWS acc[j] = hour_of<1000000000>((int64_t)acc[j]); break;
            default: RDF_ROWS acc[j] = 0; break;
        }
        return;
    }
    if (dt == RDF_F64) { RDF_ROWS acc[j] = d2u(unary_f64<FEAT>(op, u2d(acc[j]))); return; }
    if (dt == RDF_F32) { RDF_ROWS acc[j] = f2u(unary_f32<FEAT>(op, u2f(acc[j]))); return; }
    // num::abs on signed integers; MIN wraps
    RDF_ROWS { int64_t x = (int64_t)acc[j]; acc[j] = normalize_int(dt, x < 0 ? (uint64_t)0 - (uint64_t)x : (uint64_t)x); }
}

// The steps most programs are made of, written without the generic operand registers: acc OP b where b(j) is a column's
// register or the step's immediate (a scalar register: no copy at all).  f64 comparisons and + - * only — what
// apply_binary does for the same (op, RDF_F64), minus the moves.
__device__ __forceinline__ bool fast_f64_op(int op) { return (op >= RDF_OP_GT && op <= RDF_OP_LE) || (op >= RDF_OP_ADD && op <= RDF_OP_MUL); }
template <class B>
__device__ __forceinline__ void fast_f64(int op, uint64_t (&acc)[kVPT], B b) {
    switch (op) {
        case RDF_OP_GT: RDF_ROWS acc[j] = u2d(acc[j]) > b(j); break;
        case RDF_OP_GE: RDF_ROWS acc[j] = u2d(acc[j]) >= b(j); break;
        case RDF_OP_EQ: RDF_ROWS acc[j] = u2d(acc[j]) == b(j); break;
        case RDF_OP_NE: RDF_ROWS acc[j] = u2d(acc[j]) != b(j); break;
        case RDF_OP_LT: RDF_ROWS acc[j] = u2d(acc[j]) < b(j); break;
        case RDF_OP_LE: RDF_ROWS acc[j] = u2d(acc[j]) <= b(j); break;
        case RDF_OP_ADD: RDF_ROWS acc[j] = d2u(u2d(acc[j]) + b(j)); break;
        case RDF_OP_SUB: RDF_ROWS acc[j] = d2u(u2d(acc[j]) - b(j)); break;
        default: RDF_ROWS acc[j] = d2u(u2d(acc[j]) * b(j)); break;
    }
}

// ------------------------------------------------------------------------------------------------

// Chunk c of column p: the one-batch copy in the kernel arguments, or the host-built table read through the constant address
// space.  (`a.nchunks == 1 ? a.inline_cols[p] : a.cols[..]` selects between a kernel-argument address and a global one: ONE flat
// load on the vector memory path, waited for with vmcnt(0) — i.e. together with every column load in flight — and its result
// counts as different in every lane, so each test on the descriptor became a per-lane branch.  Round 6, read off the ISA.)
__device__ __forceinline__ DevChunkCol chunk_col(const EvalArgs& a, int p, int64_t c) {
    if (a.nchunks == 1) return a.inline_cols[p];
    return const_col(a.cols, (int64_t)p * a.nchunks + c);
}

template <int FEAT, int SINK, int NPRE, int NVAL>
__global__ __launch_bounds__(kBlock) void eval_kernel(const EvalArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ AggPartial red_lds[kBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = wave_id();

    uint64_t g_sum[NVAL], g_mn[NVAL], g_mx[NVAL];
    int64_t g_cnt[NVAL];
    if (SINK == SINK_AGG) {
#pragma unroll
        for (int k = 0; k < NVAL; ++k) agg_init(k < a.nvalues ? a.value_cls[k] : CLS_F64, g_sum[k], g_mn[k], g_mx[k], g_cnt[k]);
    }
    uint32_t err = 0;
    // SINK_GROUP: `group_replicas` copies of the accumulator table in LDS behind the TMP spill area (a lane
    // works on copy lane % replicas, so lanes of one wave that hit the same group mostly hit different
    // addresses); row j's group id stays in a register between BC_GROUP and the BC_EMITs
    uint64_t* gtab = nullptr;
    uint32_t gid[kVPT];
    const int gS = a.ngroups + 1, gwords = group_words(a.ngroups, a.nvalues);
    if (SINK == SINK_GROUP) {
        gtab = (uint64_t*)(smem + (size_t)a.ntmp * (kVPT * kBlock * 8 + kBlock * 4));
        for (int i = tid; i < gwords * a.group_replicas; i += kBlock) gtab[i] = 0;
#pragma unroll
        for (int j = 0; j < kVPT; ++j) gid[j] = 0;
        __syncthreads();
    }
    uint64_t* const grep = SINK == SINK_GROUP ? gtab + (size_t)(lane & (a.group_replicas - 1)) * gwords : nullptr;
    // SINK_STORE: per-wave null counters, flushed with one atomic per (value, chunk) when the block
    // moves on to another chunk — never one atomic per tile
    uint32_t nullacc[NVAL];
#pragma unroll
    for (int k = 0; k < NVAL; ++k) nullacc[k] = 0;
    int64_t cur_chunk = -1;

    // Programs with ONE column in registers have little memory-level parallelism of their own (8 KB per
    // block in flight): their NEXT tile's column loads are issued before the current tile is interpreted.
    constexpr bool PF = NPRE == 1;  // (measured: with two columns the extra registers cost more occupancy than the prefetch buys)
    struct TileLoc { int64_t c, r0, clen; };
    auto locate = [&](int64_t tile) -> TileLoc {
        TileLoc t;
        if (a.nchunks == 1) { t.c = 0; t.r0 = tile * kEvalTile; t.clen = a.inline_len; }
        else {
            t.c = find_chunk_tile(as_const<int64_t>(a.chunk_tile_start), a.nchunks, tile);
            t.r0 = (tile - as_const<int64_t>(a.chunk_tile_start)[t.c]) * kEvalTile;
            t.clen = as_const<int64_t>(a.chunk_len)[t.c];
        }
        return t;
    };
    auto rows_in_range = [&](const TileLoc& t) -> uint32_t {
        const int64_t rw_ = t.r0 + (int64_t)wave * (kVPT * 64);
        uint32_t m = 0;
#pragma unroll
        for (int j = 0; j < kVPT; ++j) m |= (uint32_t)(rw_ + (int64_t)lane * kVPT + j < t.clen) << j;
        return m;
    };
    uint64_t pfv[PF ? NPRE : 1][kVPT];
    uint32_t pfvalid[PF ? NPRE : 1];
    auto preload = [&](const TileLoc& t, uint64_t (&v)[PF ? NPRE : 1][kVPT], uint32_t (&vv)[PF ? NPRE : 1]) {
        const int64_t rw_ = t.r0 + (int64_t)wave * (kVPT * 64);
        const uint32_t inr_ = rows_in_range(t);
#pragma unroll
        for (int p = 0; p < (PF ? NPRE : 0); ++p) {
            if (p < a.ncols) {
                const DevChunkCol cc = chunk_col(a, p, t.c);
                load_col(cc, a.col_dtype[p], rw_, t.clen, inr_, v[p], vv[p]);
            } else {
                vv[p] = 0;
#pragma unroll
                for (int j = 0; j < kVPT; ++j) v[p][j] = 0;
            }
        }
    };
    const uint64_t* const code_words = (const uint64_t*)a.code;
    int64_t tile = blockIdx.x;
    bool have = tile < a.ntiles;
    TileLoc tl = have ? locate(tile) : TileLoc{0, 0, 0};
    if (PF && have) preload(tl, pfv, pfvalid);

    while (have) {
        const int64_t c = tl.c, r0 = tl.r0, clen = tl.clen;
        if (SINK == SINK_STORE && c != cur_chunk) {
            if (cur_chunk >= 0 && lane == 0) {
#pragma unroll
                for (int k = 0; k < NVAL; ++k)
                    if (k < a.nvalues && nullacc[k]) {
                        atomicAdd((unsigned long long*)&a.out_null_counts[(int64_t)k * a.nchunks + cur_chunk], (unsigned long long)nullacc[k]);
                        nullacc[k] = 0;
                    }
            }
            cur_chunk = c;
        }
        const int64_t rw = r0 + (int64_t)wave * (kVPT * 64);  // this wave's first row
        const uint32_t inr = rows_in_range(tl);

        // (1) this tile's columns: taken from the prefetch registers, or loaded now (all loads up front)
        uint64_t colv[NPRE][kVPT];
        uint32_t colvalid[NPRE];
        if (PF) {
#pragma unroll
            for (int p = 0; p < NPRE; ++p) {
                colvalid[p] = pfvalid[PF ? p : 0];
#pragma unroll
                for (int j = 0; j < kVPT; ++j) colv[p][j] = pfv[PF ? p : 0][j];
            }
        } else {
#pragma unroll
            for (int p = 0; p < NPRE; ++p) {
                if (p < a.ncols) {
                    const DevChunkCol cc = chunk_col(a, p, c);
                    load_col(cc, a.col_dtype[p], rw, clen, inr, colv[p], colvalid[p]);
                } else {
                    colvalid[p] = 0;
#pragma unroll
                    for (int j = 0; j < kVPT; ++j) colv[p][j] = 0;
                }
            }
        }
        // the next tile of this block; its loads go in flight before this tile is interpreted
        const int64_t ntile = tile + gridDim.x;
        const bool nhave = ntile < a.ntiles;
        const TileLoc ntl = nhave ? locate(ntile) : tl;
        if (PF && nhave) preload(ntl, pfv, pfvalid);

        // (2) interpret
        uint64_t acc[kVPT];
        uint32_t accv = 0, keep = inr;
#pragma unroll
        for (int j = 0; j < kVPT; ++j) acc[j] = 0;

        uint64_t nw0 = code_words[0], nw1 = code_words[1];
        for (int pc = 0; pc < a.ncode; ++pc) {
            // One step = 16 bytes, fetched as two 8-byte SCALAR loads and taken apart with scalar shifts.  (Read field by field, the
            // one- and two-byte members came through the VECTOR memory path — the scalar unit has no sub-dword loads — and each
            // was followed by s_waitcnt vmcnt(0): a full round trip per step, and the first of them also waited for the NEXT tile's
            // column loads issued just above, i.e. the prefetch hid nothing.  Round 6, read off the ISA.)
            // The NEXT step's words are asked for now and used one trip later, so the scalar-cache round trip runs under this step.
            const uint64_t iw0 = nw0, iw1 = nw1;
            if (pc + 1 < a.ncode) { nw0 = code_words[2 * pc + 2]; nw1 = code_words[2 * pc + 3]; }
            Instr in;
            in.bc = (uint8_t)iw0; in.op = (uint8_t)(iw0 >> 8); in.dtype = (uint8_t)(iw0 >> 16); in.src_kind = (uint8_t)(iw0 >> 24);
            in.src_dtype = (uint8_t)(iw0 >> 32); in.swapped = (uint8_t)(iw0 >> 40) & 1; in.src = (uint16_t)(iw0 >> 48); in.imm = iw1;
            // Fast steps (round 6): a column that is already in registers is used where it lies, an immediate stays in scalar
            // registers.  Everything else falls through to the generic step below, which copies its operand into `opnd` first
            // (measured on filter -> sum: 431 vector instructions per 256-row wave tile, ~20 of every LOAD / BIN step moves).
            if (in.src_dtype == in.dtype && (int)in.src < NPRE) {
                if (in.bc == BC_LOAD && in.src_kind == SRC_COL) {
#pragma unroll
                    for (int p = 0; p < NPRE; ++p)
                        if (p == (int)in.src) {
                            accv = colvalid[p];
#pragma unroll
                            for (int j = 0; j < kVPT; ++j) acc[j] = colv[p][j];
                        }
                    continue;
                }
                if (in.bc == BC_BIN && !in.swapped && fast_f64_op(in.op) && (in.dtype == RDF_F64 || in.op >= RDF_OP_GT)) {
                    if (in.src_kind == SRC_IMM) {
                        const double bi = u2d(in.imm);
                        fast_f64(in.op, acc, [&](int) { return bi; });
                        continue;
                    }
                    if (in.src_kind == SRC_COL) {
#pragma unroll
                        for (int p = 0; p < NPRE; ++p)
                            if (p == (int)in.src) {
                                accv &= colvalid[p];
                                fast_f64(in.op, acc, [&](int j) { return u2d(colv[p][j]); });
                            }
                        continue;
                    }
                }
            }
            uint64_t opnd[kVPT];
            uint32_t opv = (1u << kVPT) - 1;
            if (in.bc == BC_LOAD || in.bc == BC_BIN) {
                if (in.src_kind == SRC_COL) {
                    const int ci = in.src;
                    bool hit = false;
#pragma unroll
                    for (int p = 0; p < NPRE; ++p)
                        if (p == ci) {
                            hit = true;
                            opv = colvalid[p];
#pragma unroll
                            for (int j = 0; j < kVPT; ++j) opnd[j] = colv[p][j];
                        }
                    if (!hit) {
                        const DevChunkCol cc = chunk_col(a, ci & (kMaxCols - 1), c);
                        load_col(cc, a.col_dtype[ci & (kMaxCols - 1)], rw, clen, inr, opnd, opv);
                        // the loaded values are waited for HERE, on the path that loaded them: left to the compiler the wait lands
                        // where the paths join, as vmcnt(0), and every step of every program then waits for the next tile's prefetch
#pragma unroll
                        for (int j = 0; j < kVPT; ++j) asm volatile("" : "+v"(opnd[j]));
                        asm volatile("" : "+v"(opv));
                    }
                } else if (in.src_kind == SRC_IMM) {
#pragma unroll
                    for (int j = 0; j < kVPT; ++j) opnd[j] = in.imm;
                } else {  // SRC_TMP
                    const uint64_t* tv = (const uint64_t*)smem + (size_t)in.src * kVPT * kBlock + tid;
#pragma unroll
                    for (int j = 0; j < kVPT; ++j) opnd[j] = tv[j * kBlock];
                    opv = ((const uint32_t*)(smem + (size_t)a.ntmp * kVPT * kBlock * 8))[in.src * kBlock + tid];
                }
                if (in.src_dtype != in.dtype) {
#pragma unroll
                    for (int j = 0; j < kVPT; ++j) opnd[j] = cast_value(in.src_dtype, in.dtype, opnd[j]);
                }
            }
            switch (in.bc) {
                case BC_LOAD:
#pragma unroll
                    for (int j = 0; j < kVPT; ++j) acc[j] = opnd[j];
                    accv = opv;
                    break;
                case BC_STORE_TMP: {
                    uint64_t* tv = (uint64_t*)smem + (size_t)in.src * kVPT * kBlock + tid;
#pragma unroll
                    for (int j = 0; j < kVPT; ++j) tv[j * kBlock] = acc[j];
                    ((uint32_t*)(smem + (size_t)a.ntmp * kVPT * kBlock * 8))[in.src * kBlock + tid] = accv;
                } break;
                case BC_BIN: {
                    if (in.swapped) {
#pragma unroll
                        for (int j = 0; j < kVPT; ++j) { uint64_t t = acc[j]; acc[j] = opnd[j]; opnd[j] = t; }
                    }
                    accv &= opv;
                    apply_binary<FEAT>(in.op, in.dtype, acc, opnd, accv & inr, err);
                } break;
                case BC_UN:
                    apply_unary<FEAT>(in.op, in.dtype, acc);
                    break;
                case BC_CAST:   // a value the target type cannot represent becomes NULL
#pragma unroll
                    for (int j = 0; j < kVPT; ++j) {
                        bool ok;
                        acc[j] = cast_value(in.src_dtype, in.dtype, acc[j], ok);
                        if (!ok) accv &= ~(1u << j);
                    }
                    break;
                case BC_FILTER:  // DataFrame::filter: rows whose predicate is false or null are dropped
                    {
                        uint32_t pass = 0;
#pragma unroll
                        for (int j = 0; j < kVPT; ++j) pass |= ((uint32_t)acc[j] & 1u) << j;
                        keep &= pass & accv;
                    }
                    break;
                case BC_GROUP:  // acc is the row's group id (an integer expression); NULL -> the extra group
                    if (SINK == SINK_GROUP) {
#pragma unroll
                        for (int j = 0; j < kVPT; ++j) {
                            uint64_t g = in.dtype == RDF_BOOL ? (acc[j] & 1) : acc[j];
                            if (!((accv >> j) & 1)) g = (uint64_t)a.ngroups;
                            else if (g >= (uint64_t)a.ngroups) {
                                if ((keep >> j) & 1) err |= 2u;
                                keep &= ~(1u << j);
                                g = 0;
                            }
                            gid[j] = (uint32_t)g;
                            if ((keep >> j) & 1) atomicAdd((unsigned long long*)&grep[2 * a.nvalues * gS + (int)g], 1ull);
                        }
                    }
                    break;
                default: {  // BC_EMIT: acc is value expression `in.src`
                    const int k = in.src;
                    if (SINK == SINK_GROUP) {
                        const int cls = a.value_cls[k & (kMaxGroupValues - 1)];
#pragma unroll
                        for (int j = 0; j < kVPT; ++j)
                            if ((keep >> j) & 1) {
                                if ((accv >> j) & 1) {
                                    uint64_t v = acc[j];
                                    if (in.dtype == RDF_F32) v = d2u((double)u2f(v));
                                    if (cls == CLS_F64) unsafeAtomicAdd((double*)&grep[k * gS + (int)gid[j]], u2d(v));
                                    else atomicAdd((unsigned long long*)&grep[k * gS + (int)gid[j]], (unsigned long long)v);
                                } else atomicAdd((unsigned long long*)&grep[(a.nvalues + k) * gS + (int)gid[j]], 1ull);  // NULL values are counted, not summed
                            }
                    } else if (SINK == SINK_AGG) {
                        const uint32_t live = keep & accv & inr;
#pragma unroll
                        for (int kk = 0; kk < NVAL; ++kk)
                            if (kk == k) {
                                const int cls = a.value_cls[kk];
#pragma unroll
                                for (int j = 0; j < kVPT; ++j)
                                    if ((live >> j) & 1) {
                                        uint64_t v = acc[j];
                                        if (in.dtype == RDF_F32) v = d2u((double)u2f(v));
                                        agg_merge(cls, g_sum[kk], g_mn[kk], g_mx[kk], g_cnt[kk], v, v, v, 1);
                                    }
                            }
                    } else {
                        DevOutChunk oc;
                        if (a.nchunks == 1) oc = a.inline_outs[k & (kMaxValues - 1)];
                        else {
                            const ConstPtr<DevOutChunk> ot = as_const<DevOutChunk>(a.outs) + ((int64_t)k * a.nchunks + c);
                            oc.values = ot->values; oc.validity = ot->validity;
                        }
                        const int dt = in.dtype;
                        uint32_t nn = 0;
                        // the lane's rows 4 l .. 4 l + 3 leave as one vector store (two for 8-byte values) when the tile is full;
                        // bitmaps: the lanes' nibbles are OR-reduced over groups of 16 lanes into the four words of the wave's rows
                        const int64_t row0 = rw + (int64_t)lane * kVPT;
                        const bool fullt = __ballot(inr != (1u << kVPT) - 1) == 0;
                        uint64_t vv[kVPT];
#pragma unroll
                        for (int j = 0; j < kVPT; ++j) vv[j] = ((accv >> j) & 1) ? acc[j] : 0;   // null slots hold 0
                        const uint32_t live = accv & inr;
                        if (dt == RDF_BOOL) {
                            uint32_t bits = 0;
#pragma unroll
                            for (int j = 0; j < kVPT; ++j) bits |= (uint32_t)(((live >> j) & 1) && (vv[j] & 1)) << j;
                            const uint64_t word = nibbles_to_word(bits, lane);
                            const uint64_t any = nibbles_to_word(inr, lane);
                            if ((lane & 15) == 0 && any) as_global_mut<uint64_t>(oc.values)[(rw >> 6) + (lane >> 4)] = word;
                        } else {
                            const int es = dt == RDF_I64 || dt == RDF_U64 || dt == RDF_F64 ? 8 : dt == RDF_I32 || dt == RDF_U32 || dt == RDF_F32 ? 4 : dt == RDF_I16 || dt == RDF_U16 ? 2 : 1;
                            const bool vec = fullt && (((uintptr_t)oc.values + (uintptr_t)rw * (uintptr_t)es) & (es >= 4 ? 15 : es == 2 ? 7 : 3)) == 0;
                            if (vec) {
                                switch (es) {
                                    case 8: {
                                        typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
                                        u64x2 s0, s1; s0[0] = vv[0]; s0[1] = vv[1]; s1[0] = vv[2]; s1[1] = vv[3];
                                        GlobalMutPtr<u64x2> q = (GlobalMutPtr<u64x2>)(as_global_mut<uint64_t>(oc.values) + row0);
                                        __builtin_nontemporal_store(s0, q); __builtin_nontemporal_store(s1, q + 1);
                                    } break;
                                    case 4: { u32x4 s0; s0[0] = (uint32_t)vv[0]; s0[1] = (uint32_t)vv[1]; s0[2] = (uint32_t)vv[2]; s0[3] = (uint32_t)vv[3];
                                              __builtin_nontemporal_store(s0, (GlobalMutPtr<u32x4>)(as_global_mut<uint32_t>(oc.values) + row0)); } break;
                                    case 2: { u16x4 s0; s0[0] = (uint16_t)vv[0]; s0[1] = (uint16_t)vv[1]; s0[2] = (uint16_t)vv[2]; s0[3] = (uint16_t)vv[3];
                                              __builtin_nontemporal_store(s0, (GlobalMutPtr<u16x4>)(as_global_mut<uint16_t>(oc.values) + row0)); } break;
                                    default: { u8x4 s0; s0[0] = (uint8_t)vv[0]; s0[1] = (uint8_t)vv[1]; s0[2] = (uint8_t)vv[2]; s0[3] = (uint8_t)vv[3];
                                               __builtin_nontemporal_store(s0, (GlobalMutPtr<u8x4>)(as_global_mut<uint8_t>(oc.values) + row0)); } break;
                                }
                            } else {
#pragma unroll
                                for (int j = 0; j < kVPT; ++j)
                                    if ((inr >> j) & 1) {
                                        switch (es) {
                                            case 8: __builtin_nontemporal_store(vv[j], as_global_mut<uint64_t>(oc.values) + row0 + j); break;
                                            case 4: __builtin_nontemporal_store((uint32_t)vv[j], as_global_mut<uint32_t>(oc.values) + row0 + j); break;
                                            case 2: as_global_mut<uint16_t>(oc.values)[row0 + j] = (uint16_t)vv[j]; break;
                                            default: as_global_mut<uint8_t>(oc.values)[row0 + j] = (uint8_t)vv[j]; break;
                                        }
                                    }
                            }
                        }
                        {
                            const uint64_t vword = nibbles_to_word(live, lane);
                            const uint64_t iword = nibbles_to_word(inr, lane);
                            if ((lane & 15) == 0 && iword) {
                                if (oc.validity) as_global_mut<uint64_t>(oc.validity)[(rw >> 6) + (lane >> 4)] = vword;
                                nn += (uint32_t)__popcll(iword & ~vword);
                            }
                        }
                        nn = (uint32_t)__shfl((int)nn, 0) + (uint32_t)__shfl((int)nn, 16) + (uint32_t)__shfl((int)nn, 32) + (uint32_t)__shfl((int)nn, 48);
#pragma unroll
                        for (int kk = 0; kk < NVAL; ++kk)
                            if (kk == k) nullacc[kk] += nn;
                    }
                }
            }
        }
        tile = ntile;
        tl = ntl;
        have = nhave;
    }

    if (SINK == SINK_STORE && cur_chunk >= 0 && lane == 0) {
#pragma unroll
        for (int k = 0; k < NVAL; ++k)
            if (k < a.nvalues && nullacc[k])
                atomicAdd((unsigned long long*)&a.out_null_counts[(int64_t)k * a.nchunks + cur_chunk], (unsigned long long)nullacc[k]);
    }
    if (err) atomicOr(a.flags, err);
    if (SINK == SINK_GROUP) {  // fold the LDS copies in copy order, one table per block
        __syncthreads();
        for (int w = tid; w < gwords; w += kBlock) {
            const bool fsum = w < a.nvalues * gS && a.value_cls[(w / gS) & (kMaxGroupValues - 1)] == CLS_F64;
            uint64_t acc0 = gtab[w];
            for (int r = 1; r < a.group_replicas; ++r) {
                const uint64_t x = gtab[(size_t)r * gwords + w];
                acc0 = fsum ? d2u(u2d(acc0) + u2d(x)) : acc0 + x;
            }
            a.group_partials[(size_t)blockIdx.x * gwords + w] = acc0;
        }
    }
    if (SINK == SINK_AGG) {
#pragma unroll
        for (int k = 0; k < NVAL; ++k)
            if (k < a.nvalues)
                block_reduce_agg(a.value_cls[k], g_sum[k], g_mn[k], g_mx[k], g_cnt[k], red_lds,
                                 &a.partials[(int64_t)blockIdx.x * a.nvalues + k]);
    }
}

// ------------------------------------------------------------------------------------------------
// launch: (NPRE, NVAL) in {(1,1),(2,1),(2,2),(4,1),(4,2),(4,4)}

template <int SINK, int NPRE, int NVAL>
static void launch_one(const EvalArgs& a, int grid, size_t lds, hipStream_t s) {
    hipLaunchKernelGGL((eval_kernel<RDF_FEAT, SINK, NPRE, NVAL>), dim3(grid), dim3(kBlock), lds, s, a);
}
template <int SINK>
static void launch_shape(const EvalArgs& a, int npre, int nval, int grid, size_t lds, hipStream_t s) {
    if (nval <= 1) {
        if (npre <= 1) launch_one<SINK, 1, 1>(a, grid, lds, s);
        else if (npre <= 2) launch_one<SINK, 2, 1>(a, grid, lds, s);
        else launch_one<SINK, 4, 1>(a, grid, lds, s);
    } else if (nval <= 2) {
        if (npre <= 2) launch_one<SINK, 2, 2>(a, grid, lds, s);
        else launch_one<SINK, 4, 2>(a, grid, lds, s);
    } else launch_one<SINK, 4, 4>(a, grid, lds, s);
}

#define RDF_CAT2(a, b) a##b
#define RDF_CAT(a, b) RDF_CAT2(a, b)
hipError_t RDF_CAT(launch_eval_feat, RDF_FEAT)(const EvalArgs& a, int sink, int grid, hipStream_t s) {
    size_t lds = (size_t)a.ntmp * (kVPT * kBlock * 8 + kBlock * 4);
    const int npre = a.ncols < kPreCols ? a.ncols : kPreCols;
    if (sink == SINK_GROUP) {
        lds += (size_t)group_words(a.ngroups, a.nvalues) * (size_t)a.group_replicas * 8;
        if (npre <= 1) launch_one<SINK_GROUP, 1, 1>(a, grid, lds, s);
        else if (npre <= 2) launch_one<SINK_GROUP, 2, 1>(a, grid, lds, s);
        else launch_one<SINK_GROUP, 4, 1>(a, grid, lds, s);
    } else if (sink == SINK_AGG) launch_shape<SINK_AGG>(a, npre, a.nvalues, grid, lds, s);
    else launch_shape<SINK_STORE>(a, npre, a.nvalues, grid, lds, s);
    return hipGetLastError();
}

}  // namespace rdfk
