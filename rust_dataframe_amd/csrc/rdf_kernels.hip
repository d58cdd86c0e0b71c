// rdf_kernels.hip — hand-written CDNA4 (gfx950) kernels for rust-dataframe's Arrow compute hot path.
//
// Everything here is HBM-bandwidth-bound integer / f64 work (no MFMA): the design rules are wide
// coalesced loads with many bytes in flight per CU, wave64 ballots / mbcnt instead of shuffles
// where a bitmap is involved, LDS only for staging (compaction) and spills, one pass over HBM per
// fused expression tree, and persistent grids sized to the 256 CUs.
//
// Kernels (reference counterpart in brackets, paths relative to the reference root):
//   eval_kernel<HEAVY,SINK>   fused expression-tree evaluator: an accumulator-machine interpreter
//                             whose opcode stream is wave-uniform
//                             [Evaluate::calculate src/evaluation.rs:97-323 + ScalarFunctions
//                              src/functions/scalar.rs:16-540 + BooleanFilter::eval_to_array
//                              src/expression.rs:766-861 + AggregateFunctions src/functions/aggregate.rs:12-93]
//   filter_agg_f64_kernel     specialised filter(x CMP c) -> {sum,min,max,count}(y) on f64, 16-B loads
//                             [DataFrame::filter src/dataframe.rs:178-189 then AggregateFunctions::sum]
//   agg_final_kernel          second stage of the two-stage reductions
//   mask_count / scan / compact   order-preserving stream compaction: per-tile popcounts of the
//                             bit-packed mask, exclusive scan, ballot-free ranks from mbcnt-style
//                             popcounts, LDS staging, coalesced stores
//                             [Column::filter -> arrow::compute::filter, src/table.rs:97-107,213-215]
//   take_kernel               gather over the virtual concatenation of a column's chunks
//                             [Column::take src/table.rs:218-241]
//   fill_*                    counter-based synthetic data (bench / tests)
#include "rdf_device.h"

namespace rdfk {

// ------------------------------------------------------------------------------------------------
// small device helpers

__device__ __forceinline__ uint64_t uniform64(uint64_t v) {
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// `nbits` (<= 64) consecutive bits of an LSB-first bitmap starting at bit `bitpos`, as the low bits
// of a u64.  base/bitpos are wave-uniform, so these are scalar loads.  Reads only aligned 8-byte
// words that contain requested bits (the ABI requires bitmaps readable to the next 8-byte boundary).
__device__ __forceinline__ uint64_t load_bits64(const uint8_t* base, int64_t bitpos, int nbits) {
    if (nbits <= 0) return 0;
    uint64_t addr = uniform64((uint64_t)(uintptr_t)base + (uint64_t)(bitpos >> 3));
    const uint64_t* w = (const uint64_t*)(uintptr_t)(addr & ~7ull);
    int sh = (int)(addr & 7) * 8 + (int)(bitpos & 7);
    uint64_t r = w[0] >> sh;
    if (sh + nbits > 64) r |= w[1] << (64 - sh);
    if (nbits < 64) r &= (1ull << nbits) - 1;
    return r;
}

__device__ __forceinline__ int clamp64(int64_t v) { return v <= 0 ? 0 : (v >= 64 ? 64 : (int)v); }

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
__device__ __forceinline__ bool dt_is_int(int dt) { return dt <= RDF_U64; }

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

// ------------------------------------------------------------------------------------------------
// conversions (arrow::compute::cast: Rust `as` semantics — float->int saturates, NaN -> 0)

__device__ __forceinline__ uint64_t cast_value(int from, int to, uint64_t x) {
    if (from == to) return x;
    // classify the source
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
    // integer targets
    if (src_float) {
        if (f != f) return 0;
        switch (to) {
            case RDF_I8: return (uint64_t)(int64_t)(f < -128.0 ? -128.0 : f > 127.0 ? 127.0 : f);
            case RDF_I16: return (uint64_t)(int64_t)(f < -32768.0 ? -32768.0 : f > 32767.0 ? 32767.0 : f);
            case RDF_I32: return (uint64_t)(int64_t)(f < -2147483648.0 ? -2147483648.0 : f > 2147483647.0 ? 2147483647.0 : f);
            case RDF_I64:
                if (f >= 9223372036854775808.0) return (uint64_t)INT64_MAX;
                if (f <= -9223372036854775808.0) return (uint64_t)INT64_MIN;
                return (uint64_t)(int64_t)f;
            case RDF_U8: return (uint64_t)(f < 0.0 ? 0.0 : f > 255.0 ? 255.0 : f);
            case RDF_U16: return (uint64_t)(f < 0.0 ? 0.0 : f > 65535.0 ? 65535.0 : f);
            case RDF_U32: return (uint64_t)(f < 0.0 ? 0.0 : f > 4294967295.0 ? 4294967295.0 : f);
            default:
                if (f <= 0.0) return 0;
                if (f >= 18446744073709551616.0) return ~0ull;
                return (uint64_t)f;
        }
    }
    return normalize_int(to, x);  // int/bool -> int: truncate
}

// ------------------------------------------------------------------------------------------------
// column loads for the evaluator: kVPT rows per thread, row(j) = r0 + j*kBlock + tid, so every
// wave-instruction touches 64 consecutive elements (512 B for 8-byte types).

__device__ __forceinline__ void load_col(const DevChunkCol cc, int dt, int64_t r0, int64_t clen, uint32_t inr,
                                         uint64_t (&v)[kVPT], uint32_t& valid) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t e0 = cc.offset + r0 + tid;
    switch (dt) {
        case RDF_I64: case RDF_U64: case RDF_F64: {
            const uint64_t* p = (const uint64_t*)cc.values + e0;
#pragma unroll
            for (int j = 0; j < kVPT; ++j) v[j] = (inr >> j) & 1 ? __builtin_nontemporal_load(p + j * kBlock) : 0;
        } break;
        case RDF_I32: case RDF_U32: case RDF_F32: {
            const uint32_t* p = (const uint32_t*)cc.values + e0;
#pragma unroll
            for (int j = 0; j < kVPT; ++j) {
                uint32_t t = (inr >> j) & 1 ? __builtin_nontemporal_load(p + j * kBlock) : 0;
                v[j] = dt == RDF_I32 ? (uint64_t)(int64_t)(int32_t)t : (uint64_t)t;
            }
        } break;
        case RDF_I16: case RDF_U16: {
            const uint16_t* p = (const uint16_t*)cc.values + e0;
#pragma unroll
            for (int j = 0; j < kVPT; ++j) {
                uint16_t t = (inr >> j) & 1 ? p[j * kBlock] : (uint16_t)0;
                v[j] = dt == RDF_I16 ? (uint64_t)(int64_t)(int16_t)t : (uint64_t)t;
            }
        } break;
        case RDF_I8: case RDF_U8: {
            const uint8_t* p = (const uint8_t*)cc.values + e0;
#pragma unroll
            for (int j = 0; j < kVPT; ++j) {
                uint8_t t = (inr >> j) & 1 ? p[j * kBlock] : (uint8_t)0;
                v[j] = dt == RDF_I8 ? (uint64_t)(int64_t)(int8_t)t : (uint64_t)t;
            }
        } break;
        default: {  // RDF_BOOL: bit-packed values
#pragma unroll
            for (int j = 0; j < kVPT; ++j) {
                int64_t rb = r0 + (int64_t)j * kBlock + wave * 64;
                uint64_t w = load_bits64((const uint8_t*)cc.values, cc.offset + rb, clamp64(clen - rb));
                v[j] = (w >> lane) & 1;
            }
        }
    }
    valid = (1u << kVPT) - 1;
    if (cc.validity) {
        valid = 0;
#pragma unroll
        for (int j = 0; j < kVPT; ++j) {
            int64_t rb = r0 + (int64_t)j * kBlock + wave * 64;
            uint64_t w = load_bits64(cc.validity, cc.offset + rb, clamp64(clen - rb));
            valid |= (uint32_t)((w >> lane) & 1) << j;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// the interpreter's arithmetic.  `key`-style dispatch keeps the (uniform) switch outside the
// per-row loop.

template <bool HEAVY>
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
    if constexpr (HEAVY) {
        switch (op) {
            case RDF_OP_ACOS: return acos(x);
            case RDF_OP_ASIN: return asin(x);
            case RDF_OP_ATAN: return atan(x);
            case RDF_OP_CBRT: return cbrt(x);
            case RDF_OP_COS: return cos(x);
            case RDF_OP_COSH: return cosh(x);
            case RDF_OP_EXP: return exp(x);
            case RDF_OP_EXPM1: return expm1(x);
            case RDF_OP_LOG10: return log10(x);
            case RDF_OP_LOG2: return log2(x);
            case RDF_OP_SIN: return sin(x);
            case RDF_OP_SINH: return sinh(x);
            case RDF_OP_TAN: return tan(x);
            case RDF_OP_TANH: return tanh(x);
            default: break;
        }
    }
    return x;
}
template <bool HEAVY>
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
    if constexpr (HEAVY) {
        switch (op) {
            case RDF_OP_ACOS: return acosf(x);
            case RDF_OP_ASIN: return asinf(x);
            case RDF_OP_ATAN: return atanf(x);
            case RDF_OP_CBRT: return cbrtf(x);
            case RDF_OP_COS: return cosf(x);
            case RDF_OP_COSH: return coshf(x);
            case RDF_OP_EXP: return expf(x);
            case RDF_OP_EXPM1: return expm1f(x);
            case RDF_OP_LOG10: return log10f(x);
            case RDF_OP_LOG2: return log2f(x);
            case RDF_OP_SIN: return sinf(x);
            case RDF_OP_SINH: return sinhf(x);
            case RDF_OP_TAN: return tanf(x);
            case RDF_OP_TANH: return tanhf(x);
            default: break;
        }
    }
    return x;
}

#define RDF_ROWS _Pragma("unroll") for (int j = 0; j < kVPT; ++j)

// acc = acc OP b.  `live` = rows where both sides are valid and in range (divide-by-zero is only an
// error there, like arrow's math_divide).
template <bool HEAVY>
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
                if constexpr (HEAVY) {
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
                if constexpr (HEAVY) {
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

template <bool HEAVY>
__device__ __forceinline__ void apply_unary(int op, int dt, uint64_t (&acc)[kVPT]) {
    if (op == RDF_OP_NOT) { RDF_ROWS acc[j] = acc[j] ^ 1ull; return; }
    if (dt == RDF_F64) { RDF_ROWS acc[j] = d2u(unary_f64<HEAVY>(op, u2d(acc[j]))); return; }
    if (dt == RDF_F32) { RDF_ROWS acc[j] = f2u(unary_f32<HEAVY>(op, u2f(acc[j]))); return; }
    // num::abs on signed integers; MIN wraps
    RDF_ROWS { int64_t x = (int64_t)acc[j]; acc[j] = normalize_int(dt, x < 0 ? (uint64_t)0 - (uint64_t)x : (uint64_t)x); }
}

// ------------------------------------------------------------------------------------------------
// eval_kernel: one persistent block walks 1024-row tiles (= the reference's RecordBatch size); per
// tile it (1) issues ALL column loads for the tile up front (memory-level parallelism), (2) runs the
// uniform bytecode over registers, (3) feeds the sink: coalesced stores + ballot-built bitmaps, or
// running {sum,min,max,count} folded block-wide once at the end (two-stage reduction).

template <bool HEAVY, int SINK>
__global__ __launch_bounds__(kBlock) void eval_kernel(const EvalArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ AggPartial red_lds[kBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    uint64_t g_sum[kMaxValues], g_mn[kMaxValues], g_mx[kMaxValues];
    int64_t g_cnt[kMaxValues];
    if (SINK == SINK_AGG) {
#pragma unroll
        for (int k = 0; k < kMaxValues; ++k) agg_init(k < a.nvalues ? a.value_cls[k] : CLS_F64, g_sum[k], g_mn[k], g_mx[k], g_cnt[k]);
    }
    uint32_t err = 0;

    for (int64_t tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        int64_t c = 0, r0, clen;
        if (a.nchunks == 1) { r0 = tile * kEvalTile; clen = a.inline_len; }
        else {
            int64_t lo = 0, hi = a.nchunks - 1;  // largest c with chunk_tile_start[c] <= tile
            while (lo < hi) { int64_t mid = (lo + hi + 1) >> 1; if (a.chunk_tile_start[mid] <= tile) lo = mid; else hi = mid - 1; }
            c = lo;
            r0 = (tile - a.chunk_tile_start[c]) * kEvalTile;
            clen = a.chunk_len[c];
        }
        uint32_t inr = 0;
#pragma unroll
        for (int j = 0; j < kVPT; ++j) inr |= (uint32_t)(r0 + (int64_t)j * kBlock + tid < clen) << j;

        // (1) preload
        uint64_t colv[kPreCols][kVPT];
        uint32_t colvalid[kPreCols];
#pragma unroll
        for (int p = 0; p < kPreCols; ++p) {
            colvalid[p] = 0;
#pragma unroll
            for (int j = 0; j < kVPT; ++j) colv[p][j] = 0;
            if (p < a.ncols) {
                const DevChunkCol cc = a.nchunks == 1 ? a.inline_cols[p] : a.cols[(int64_t)p * a.nchunks + c];
                load_col(cc, a.col_dtype[p], r0, clen, inr, colv[p], colvalid[p]);
            }
        }

        // (2) interpret
        uint64_t acc[kVPT];
        uint32_t accv = 0, keep = inr;
        uint32_t nullbits[kMaxValues];  // SINK_STORE: per-wave null counters (lane 0 only)
#pragma unroll
        for (int k = 0; k < kMaxValues; ++k) nullbits[k] = 0;
#pragma unroll
        for (int j = 0; j < kVPT; ++j) acc[j] = 0;

        for (int pc = 0; pc < a.ncode; ++pc) {
            const Instr in = a.code[pc];
            // operand fetch (LOAD and BIN)
            uint64_t opnd[kVPT];
            uint32_t opv = (1u << kVPT) - 1;
            if (in.bc == BC_LOAD || in.bc == BC_BIN) {
                if (in.src_kind == SRC_COL) {
                    const int ci = in.src;
                    bool hit = false;
#pragma unroll
                    for (int p = 0; p < kPreCols; ++p)
                        if (p == ci) {
                            hit = true;
                            opv = colvalid[p];
#pragma unroll
                            for (int j = 0; j < kVPT; ++j) opnd[j] = colv[p][j];
                        }
                    if (!hit) {
                        const DevChunkCol cc = a.nchunks == 1 ? a.inline_cols[ci & (kMaxCols - 1)] : a.cols[(int64_t)ci * a.nchunks + c];
                        load_col(cc, a.col_dtype[ci & (kMaxCols - 1)], r0, clen, inr, opnd, opv);
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
                    apply_binary<HEAVY>(in.op, in.dtype, acc, opnd, accv & inr, err);
                } break;
                case BC_UN:
                    apply_unary<HEAVY>(in.op, in.dtype, acc);
                    break;
                case BC_CAST:
#pragma unroll
                    for (int j = 0; j < kVPT; ++j) acc[j] = cast_value(in.src_dtype, in.dtype, acc[j]);
                    break;
                case BC_FILTER:  // DataFrame::filter: rows whose predicate is false or null are dropped
#pragma unroll
                    for (int j = 0; j < kVPT; ++j) keep &= ~((uint32_t)(((acc[j] & 1) == 0) || (((accv >> j) & 1) == 0)) << j);
                    break;
                default: {  // BC_EMIT: acc is value expression `in.src`
                    const int k = in.src;
                    if (SINK == SINK_AGG) {
                        const uint32_t live = keep & accv & inr;
#pragma unroll
                        for (int kk = 0; kk < kMaxValues; ++kk)
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
                        const DevOutChunk oc = a.nchunks == 1 ? a.inline_outs[k & (kMaxValues - 1)] : a.outs[(int64_t)k * a.nchunks + c];
                        const int dt = in.dtype;
#pragma unroll
                        for (int j = 0; j < kVPT; ++j) {
                            const int64_t row = r0 + (int64_t)j * kBlock + tid;
                            const bool ok = (inr >> j) & 1;
                            const bool valid = (accv >> j) & 1;
                            const uint64_t v = valid ? acc[j] : 0;  // null slots hold 0
                            if (dt == RDF_BOOL) {
                                const uint64_t bits = __ballot(ok && valid && (v & 1));
                                if (lane == 0 && r0 + (int64_t)j * kBlock + wave * 64 < clen)
                                    ((uint64_t*)oc.values)[(r0 + (int64_t)j * kBlock + wave * 64) >> 6] = bits;
                            } else if (ok) {
                                switch (dt) {
                                    case RDF_I64: case RDF_U64: case RDF_F64: ((uint64_t*)oc.values)[row] = v; break;
                                    case RDF_I32: case RDF_U32: case RDF_F32: ((uint32_t*)oc.values)[row] = (uint32_t)v; break;
                                    case RDF_I16: case RDF_U16: ((uint16_t*)oc.values)[row] = (uint16_t)v; break;
                                    default: ((uint8_t*)oc.values)[row] = (uint8_t)v; break;
                                }
                            }
                            const uint64_t vb = __ballot(ok && valid);
                            const uint64_t ib = __ballot(ok);
                            if (lane == 0 && ib) {
                                if (oc.validity) ((uint64_t*)oc.validity)[(r0 + (int64_t)j * kBlock + wave * 64) >> 6] = vb;
#pragma unroll
                                for (int kk = 0; kk < kMaxValues; ++kk)
                                    if (kk == k) nullbits[kk] += (uint32_t)__popcll(ib & ~vb);
                            }
                        }
                    }
                }
            }
        }
        if (SINK == SINK_STORE && lane == 0) {
#pragma unroll
            for (int kk = 0; kk < kMaxValues; ++kk)
                if (kk < a.nvalues && nullbits[kk])
                    atomicAdd((unsigned long long*)&a.out_null_counts[(int64_t)kk * a.nchunks + c], (unsigned long long)nullbits[kk]);
        }
        if (a.ntmp > 0) __syncthreads();  // tmp slots are thread-private, but keep tiles tidy across waves
    }

    if (err) atomicOr(a.flags, err);
    if (SINK == SINK_AGG) {
#pragma unroll
        for (int k = 0; k < kMaxValues; ++k)
            if (k < a.nvalues)
                block_reduce_agg(a.value_cls[k], g_sum[k], g_mn[k], g_mx[k], g_cnt[k], red_lds,
                                 &a.partials[(int64_t)blockIdx.x * a.nvalues + k]);
    }
}

// Second stage: fold the per-block partials (fixed order => run-to-run deterministic f64 sums).
__global__ __launch_bounds__(kBlock) void agg_final_kernel(const AggFinalArgs a) {
    __shared__ AggPartial red_lds[kBlock / 64];
    for (int k = 0; k < a.nvalues; ++k) {
        const int cls = a.value_cls[k];
        uint64_t s, mn, mx;
        int64_t cnt;
        agg_init(cls, s, mn, mx, cnt);
        for (int b = threadIdx.x; b < a.nblocks; b += kBlock) {
            const AggPartial p = a.partials[(int64_t)b * a.nvalues + k];
            agg_merge(cls, s, mn, mx, cnt, p.sum, p.mn, p.mx, p.cnt);
        }
        block_reduce_agg(cls, s, mn, mx, cnt, red_lds, &a.result[k]);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// filter_agg_f64_kernel: the headline shape, filter(x CMP c) -> sum/min/max/count(y), f64.
// 16-byte loads (global_load_dwordx4, 1 KiB per wave-instruction), 4 vectors in flight per lane
// per column, predicate and accumulation in registers; HBM traffic = 8 B/row (+ 1 bit with nulls).

template <int CMP>
__device__ __forceinline__ bool cmp_f64(double x, double c) {
    if (CMP == RDF_OP_GT) return x > c;
    if (CMP == RDF_OP_GE) return x >= c;
    if (CMP == RDF_OP_EQ) return x == c;
    if (CMP == RDF_OP_NE) return x != c;
    if (CMP == RDF_OP_LT) return x < c;
    return x <= c;
}

struct F64Agg {
    double sum, mn, mx;
    int64_t cnt;
    __device__ __forceinline__ void init() { sum = 0.0; mn = mx = __longlong_as_double(0x7FF8000000000000ll); cnt = 0; }
    __device__ __forceinline__ void add(double v) { sum += v; mn = fmin(mn, v); mx = fmax(mx, v); ++cnt; }
};

constexpr int kFU = 4;  // double2 vectors per lane per iteration
typedef double dvec2 __attribute__((ext_vector_type(2)));

template <int CMP, bool SAME, bool HASV>
__global__ __launch_bounds__(kBlock) void filter_agg_f64_kernel(const FilterAggF64Args a) {
    __shared__ AggPartial red_lds[kBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* xb = a.x + a.x_offset;
    const double* yb = SAME ? xb : a.y + a.y_offset;
    F64Agg g;
    g.init();
    // peel so the vector body is 16-byte aligned for x (and y must then share that parity)
    const int64_t head = (((uintptr_t)xb & 15) != 0 && a.n > 0) ? 1 : 0;
    const int64_t nvec = (a.n - head) >> 1;
    const bool tail = ((a.n - head) & 1) != 0;
    if (blockIdx.x == 0 && tid == 0) {
        // scalar head / tail rows
        int64_t rows[2] = {0, a.n - 1};
        bool use[2] = {head != 0, tail};
        for (int t = 0; t < 2; ++t)
            if (use[t]) {
                const int64_t r = rows[t];
                bool ok = true;
                if (HASV) {
                    if (a.x_validity) ok = ok && ((a.x_validity[(a.x_offset + r) >> 3] >> ((a.x_offset + r) & 7)) & 1);
                    if (!SAME && a.y_validity) ok = ok && ((a.y_validity[(a.y_offset + r) >> 3] >> ((a.y_offset + r) & 7)) & 1);
                }
                if (ok && cmp_f64<CMP>(xb[r], a.c)) g.add(yb[r]);
            }
    }
    const dvec2* xv = (const dvec2*)(xb + head);
    const dvec2* yv = (const dvec2*)(yb + head);  // only dereferenced when 16-byte aligned (host checks)
    const int64_t per_iter = (int64_t)kBlock * kFU;     // vectors per block iteration
    for (int64_t base = (int64_t)blockIdx.x * per_iter; base < nvec; base += (int64_t)gridDim.x * per_iter) {
        dvec2 vx[kFU], vy[kFU];
        uint32_t vb[kFU];  // 2 validity bits per vector
#pragma unroll
        for (int u = 0; u < kFU; ++u) {
            const int64_t i = base + (int64_t)u * kBlock + tid;
            if (i < nvec) {
                vx[u] = __builtin_nontemporal_load(xv + i);
                if (!SAME) vy[u] = __builtin_nontemporal_load(yv + i);
            } else {
                vx[u] = (dvec2)(0.0);
                if (!SAME) vy[u] = (dvec2)(0.0);
            }
        }
#pragma unroll
        for (int u = 0; u < kFU; ++u) {
            const int64_t i = base + (int64_t)u * kBlock + tid;
            uint32_t bits = i < nvec ? 3u : 0u;
            if (HASV) {
                // the wave's 128 rows start at row head + 2*(base + u*kBlock + wave*64)
                const int64_t rw = head + 2 * (base + (int64_t)u * kBlock + wave * 64);
                const int64_t left = a.n - (tail ? 1 : 0) - rw;  // vector-body rows remaining from rw
                if (a.x_validity) {
                    const uint64_t w0 = load_bits64(a.x_validity, a.x_offset + rw, clamp64(left));
                    const uint64_t w1 = load_bits64(a.x_validity, a.x_offset + rw + 64, clamp64(left - 64));
                    const uint64_t w = lane < 32 ? w0 : w1;
                    bits &= (uint32_t)(w >> ((2 * lane) & 63)) & 3u;
                }
                if (!SAME && a.y_validity) {
                    const uint64_t w0 = load_bits64(a.y_validity, a.y_offset + rw, clamp64(left));
                    const uint64_t w1 = load_bits64(a.y_validity, a.y_offset + rw + 64, clamp64(left - 64));
                    const uint64_t w = lane < 32 ? w0 : w1;
                    bits &= (uint32_t)(w >> ((2 * lane) & 63)) & 3u;
                }
            }
            vb[u] = bits;
        }
#pragma unroll
        for (int u = 0; u < kFU; ++u) {
            const dvec2 yy = SAME ? vx[u] : vy[u];
            if ((vb[u] & 1u) && cmp_f64<CMP>(vx[u].x, a.c)) g.add(yy.x);
            if ((vb[u] & 2u) && cmp_f64<CMP>(vx[u].y, a.c)) g.add(yy.y);
        }
    }
    block_reduce_agg(CLS_F64, d2u(g.sum), d2u(g.mn), d2u(g.mx), g.cnt, red_lds, &a.partials[blockIdx.x]);
}

// ------------------------------------------------------------------------------------------------
// stream compaction (Column::filter).  Tiles of kFilterTile = 2048 rows = 32 mask words.

__device__ __forceinline__ void locate_tile(const MaskTables& t, int64_t tile, int64_t& c, int64_t& r0, int64_t& clen) {
    int64_t lo = 0, hi = t.nchunks - 1;
    while (lo < hi) { int64_t mid = (lo + hi + 1) >> 1; if (t.chunk_tile_start[mid] <= tile) lo = mid; else hi = mid - 1; }
    c = lo;
    r0 = (tile - t.chunk_tile_start[c]) * kFilterTile;
    clen = t.chunk_len[c];
}

// keep-word q of a tile: mask value bits AND mask validity bits, rows past the chunk end cleared.
__device__ __forceinline__ uint64_t keep_word(const DevChunkCol& m, int64_t r0, int64_t clen, int q) {
    const int64_t rb = r0 + (int64_t)q * 64;
    const int nb = clamp64(clen - rb);
    uint64_t w = load_bits64((const uint8_t*)m.values, m.offset + rb, nb);
    if (m.validity) w &= load_bits64(m.validity, m.offset + rb, nb);
    return w;
}

// One thread per mask word; 32-lane groups = tiles.  Reads 1 bit/row.
__global__ __launch_bounds__(kBlock) void mask_count_kernel(const MaskTables t, int64_t* tile_counts) {
    const int64_t tile = (int64_t)blockIdx.x * (kBlock / 32) + (threadIdx.x >> 5);
    const int q = threadIdx.x & 31;
    int cnt = 0;
    if (tile < t.ntiles) {
        int64_t c, r0, clen;
        locate_tile(t, tile, c, r0, clen);
        // not wave-uniform here (two tiles per wave): plain byte-safe window load
        const DevChunkCol m = t.mask[c];
        const int64_t rb = r0 + (int64_t)q * 64;
        const int nb = clamp64(clen - rb);
        if (nb > 0) {
            const int64_t bitpos = m.offset + rb;
            uint64_t addr = (uint64_t)(uintptr_t)m.values + (uint64_t)(bitpos >> 3);
            const uint64_t* w = (const uint64_t*)(uintptr_t)(addr & ~7ull);
            int sh = (int)(addr & 7) * 8 + (int)(bitpos & 7);
            uint64_t r = w[0] >> sh;
            if (sh + nb > 64) r |= w[1] << (64 - sh);
            if (m.validity) {
                uint64_t addr2 = (uint64_t)(uintptr_t)m.validity + (uint64_t)(bitpos >> 3);
                const uint64_t* w2 = (const uint64_t*)(uintptr_t)(addr2 & ~7ull);
                int sh2 = (int)(addr2 & 7) * 8 + (int)(bitpos & 7);
                uint64_t r2 = w2[0] >> sh2;
                if (sh2 + nb > 64) r2 |= w2[1] << (64 - sh2);
                r &= r2;
            }
            if (nb < 64) r &= (1ull << nb) - 1;
            cnt = __popcll(r);
        }
    }
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) cnt += __shfl_xor(cnt, m);
    if (q == 0 && tile < t.ntiles) tile_counts[tile] = cnt;
}

// Exclusive scan of n int64 counts into scan[0..n] (scan[n] = total).  One block; each thread owns
// a contiguous segment (two cached passes over a few MB at most).
constexpr int kScanThreads = 1024;
__global__ __launch_bounds__(kScanThreads) void scan_kernel(const int64_t* counts, int64_t* scan, int64_t n) {
    __shared__ int64_t wave_tot[kScanThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t seg = (n + kScanThreads - 1) / kScanThreads;
    const int64_t b = (int64_t)tid * seg, e = b + seg < n ? b + seg : n;
    int64_t s = 0;
    for (int64_t i = b; i < e; ++i) s += counts[i];
    // inclusive scan across the wave
    int64_t inc = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)inc, d), hi = (uint32_t)__shfl_up((int)(uint32_t)((uint64_t)inc >> 32), d);
        int64_t o = (int64_t)(((uint64_t)hi << 32) | lo);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    int64_t wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += wave_tot[w];
    int64_t run = wbase + inc - s;  // exclusive prefix of this thread's segment
    for (int64_t i = b; i < e; ++i) { scan[i] = run; run += counts[i]; }
    if (tid == kScanThreads - 1) {
        int64_t tot = 0;
        for (int w = 0; w < kScanThreads / 64; ++w) tot += wave_tot[w];
        scan[n] = tot;
    }
}

// Compaction of one column of one tile: ranks come from popcounts of the tile's keep words (no
// shuffles, no atomics for the values); kept values are staged in LDS at their rank, then written
// with coalesced stores at the tile's output offset.
template <typename T>
__device__ __forceinline__ void compact_column(const DevChunkCol col, const DevOutChunk oc, int64_t* null_count_out,
                                               int64_t r0, int64_t clen, int64_t out_off, const uint64_t* keep_w,
                                               const int* wbase, int total, unsigned char* stage_raw, uint8_t* vstage) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    T* stage = (T*)stage_raw;
    const T* src = (const T*)col.values + col.offset + r0;
#pragma unroll
    for (int s = 0; s < kFilterTile / kBlock; ++s) {
        const int q = s * (kBlock / 64) + wave;
        const uint64_t m = keep_w[q];
        uint64_t vw = ~0ull;
        if (col.validity && m) vw = load_bits64(col.validity, col.offset + r0 + (int64_t)q * 64, clamp64(clen - (r0 + (int64_t)q * 64)));
        if ((m >> lane) & 1) {
            const int rank = wbase[q] + __popcll(m & ((1ull << lane) - 1));
            stage[rank] = __builtin_nontemporal_load(src + q * 64 + lane);
            if (col.validity) vstage[rank] = (uint8_t)((vw >> lane) & 1);
        }
    }
    __syncthreads();
    T* dst = (T*)oc.values + out_off;
    for (int i = tid; i < total; i += kBlock) dst[i] = stage[i];
    if (col.validity && oc.validity) {
        // out bits [out_off, out_off+total): ballot 64 aligned positions at a time, OR into the
        // (pre-zeroed) bitmap; boundary words are shared with neighbouring tiles, hence atomics.
        const int64_t first = out_off & ~63ll;
        const int64_t end = out_off + total;
        int nulls = 0;
        for (int64_t wb = first + (int64_t)wave * 64; wb < end; wb += (kBlock / 64) * 64) {
            const int64_t pos = wb + lane;
            const bool inside = pos >= out_off && pos < end;
            const bool bit = inside && vstage[pos - out_off];
            const uint64_t word = __ballot(bit);
            const uint64_t inw = __ballot(inside);
            if (lane == 0) {
                if (word) atomicOr((unsigned long long*)oc.validity + (wb >> 6), (unsigned long long)word);
                nulls += __popcll(inw & ~word);
            }
        }
        if (lane == 0 && nulls) atomicAdd((unsigned long long*)null_count_out, (unsigned long long)nulls);
    }
    __syncthreads();
}

__global__ __launch_bounds__(kBlock) void compact_kernel(const FilterArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char stage[kFilterTile * 8];
    __shared__ uint8_t vstage[kFilterTile];
    __shared__ uint64_t keep_w[kFilterTile / 64];
    __shared__ int wbase[kFilterTile / 64];
    __shared__ int total_s;
    const int tid = threadIdx.x;
    for (int64_t tile = blockIdx.x; tile < a.t.ntiles; tile += gridDim.x) {
        int64_t c, r0, clen;
        locate_tile(a.t, tile, c, r0, clen);
        if (tid < 64) {  // wave 0: the 32 keep words and their exclusive popcount scan
            const DevChunkCol m = a.t.mask[c];
            uint64_t w = 0;
            // load_bits64 wants wave-uniform addresses: loop the 32 words, lane q keeps word q
            for (int q = 0; q < kFilterTile / 64; ++q) {
                const uint64_t wq = keep_word(m, r0, clen, q);
                if (tid == q) w = wq;
            }
            int cnt = __popcll(w), inc = cnt;
#pragma unroll
            for (int d = 1; d < 32; d <<= 1) { int o = __shfl_up(inc, d); if (tid >= d) inc += o; }
            if (tid < kFilterTile / 64) { keep_w[tid] = w; wbase[tid] = inc - cnt; }
            if (tid == kFilterTile / 64 - 1) total_s = inc;
        }
        __syncthreads();
        const int total = total_s;
        const int64_t out_off = a.tile_scan[tile] - a.tile_scan[a.t.chunk_tile_start[c]];
        if (total > 0) {
            for (int k = 0; k < a.ncols; ++k) {
                const DevChunkCol col = a.cols[(int64_t)k * a.t.nchunks + c];
                const DevOutChunk oc = a.outs[(int64_t)k * a.t.nchunks + c];
                int64_t* nc = &a.out_null_counts[(int64_t)k * a.t.nchunks + c];
                switch (a.esize[k]) {
                    case 8: compact_column<uint64_t>(col, oc, nc, r0, clen, out_off, keep_w, wbase, total, stage, vstage); break;
                    case 4: compact_column<uint32_t>(col, oc, nc, r0, clen, out_off, keep_w, wbase, total, stage, vstage); break;
                    case 2: compact_column<uint16_t>(col, oc, nc, r0, clen, out_off, keep_w, wbase, total, stage, vstage); break;
                    default: compact_column<uint8_t>(col, oc, nc, r0, clen, out_off, keep_w, wbase, total, stage, vstage); break;
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// take: out[j] = concat(chunks)[idx[j]] without materialising the concat.

template <typename T, typename IDX>
__global__ __launch_bounds__(kBlock) void take_kernel(const TakeArgs a) {
    const int lane = threadIdx.x & 63;
    uint32_t err = 0;
    int nulls = 0;
    const int64_t nwaves_total = (a.n + 63) >> 6;
    for (int64_t wv = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); wv < nwaves_total; wv += (int64_t)gridDim.x * (kBlock / 64)) {
        const int64_t j = wv * 64 + lane;
        const bool inr = j < a.n;
        bool valid = inr;
        if (a.indices.validity) {
            const uint64_t w = load_bits64(a.indices.validity, a.indices.offset + wv * 64, clamp64(a.n - wv * 64));
            valid = valid && ((w >> lane) & 1);
        }
        T v = 0;
        if (valid) {
            const uint64_t ix = (uint64_t)((const IDX*)a.indices.values)[a.indices.offset + j];
            if (ix >= (uint64_t)a.total_rows) { err |= 2u; valid = false; }
            else {
                int64_t c = 0;
                if (a.nchunks > 1) {
                    int64_t lo = 0, hi = a.nchunks - 1;
                    while (lo < hi) { int64_t mid = (lo + hi + 1) >> 1; if ((uint64_t)a.chunk_row_start[mid] <= ix) lo = mid; else hi = mid - 1; }
                    c = lo;
                }
                const DevChunkCol cc = a.chunks[c];
                const int64_t e = cc.offset + (int64_t)ix - a.chunk_row_start[c];
                v = ((const T*)cc.values)[e];
                if (cc.validity) valid = (cc.validity[e >> 3] >> (e & 7)) & 1;
            }
        }
        if (inr) ((T*)a.out.values)[j] = v;
        const uint64_t vb = __ballot(valid);
        const uint64_t ib = __ballot(inr);
        if (lane == 0) {
            if (a.out.validity) ((uint64_t*)a.out.validity)[wv] = vb;
            nulls += __popcll(ib & ~vb);
        }
    }
    if (lane == 0 && nulls) atomicAdd((unsigned long long*)a.out_null_count, (unsigned long long)nulls);
    if (err) atomicOr(a.flags, err);
}

// ------------------------------------------------------------------------------------------------
// synthetic data: SplitMix64 finaliser over (seed, column, row) — restated identically in the oracle.

__device__ __forceinline__ uint64_t hash64(uint64_t seed, uint64_t col, uint64_t row) {
    uint64_t z = (seed ^ (col * 0xD6E8FEB86659FD93ull)) + (row + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ void fill_f64_kernel(double* p, int64_t n, uint64_t seed, uint64_t col, int64_t first_row, double lo, double hi) {
    const double span = hi - lo;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double u = (double)(hash64(seed, col, (uint64_t)(first_row + i)) >> 11) * (1.0 / 9007199254740992.0);
        const double t = span * u;
        p[i] = lo + t;
    }
}
__global__ void fill_i64_kernel(int64_t* p, int64_t n, uint64_t seed, uint64_t col, int64_t first_row, int64_t lo, uint64_t span) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        p[i] = (int64_t)((uint64_t)lo + hash64(seed, col, (uint64_t)(first_row + i)) % span);
}
__global__ void fill_validity_kernel(uint64_t* p, int64_t nbits, uint64_t seed, uint64_t col, int64_t first_row, uint64_t thr) {
    const int lane = threadIdx.x & 63;
    const int64_t nwords = (nbits + 63) >> 6;
    for (int64_t wv = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; wv < nwords; wv += ((int64_t)gridDim.x * blockDim.x) >> 6) {
        const int64_t i = wv * 64 + lane;
        const bool bit = i < nbits && (hash64(seed ^ 0xA5A5A5A5A5A5A5A5ull, col, (uint64_t)(first_row + i)) >> 32) >= thr;
        const uint64_t w = __ballot(bit);
        if (lane == 0) p[wv] = w;
    }
}

// ------------------------------------------------------------------------------------------------
// launch wrappers

int eval_grid_limit() {
    static int limit = 0;
    if (limit == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            limit = prop.multiProcessorCount * 8;  // 8 blocks of 4 waves = the CU's 32-wave capacity
        else
            limit = 2048;
    }
    return limit;
}

hipError_t launch_eval(const EvalArgs& a, int sink, bool heavy, int grid, hipStream_t s) {
    const size_t lds = (size_t)a.ntmp * (kVPT * kBlock * 8 + kBlock * 4);
    if (sink == SINK_AGG) {
        if (heavy) hipLaunchKernelGGL((eval_kernel<true, SINK_AGG>), dim3(grid), dim3(kBlock), lds, s, a);
        else hipLaunchKernelGGL((eval_kernel<false, SINK_AGG>), dim3(grid), dim3(kBlock), lds, s, a);
    } else {
        if (heavy) hipLaunchKernelGGL((eval_kernel<true, SINK_STORE>), dim3(grid), dim3(kBlock), lds, s, a);
        else hipLaunchKernelGGL((eval_kernel<false, SINK_STORE>), dim3(grid), dim3(kBlock), lds, s, a);
    }
    return hipGetLastError();
}

hipError_t launch_agg_final(const AggFinalArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(agg_final_kernel, dim3(1), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

template <int CMP>
static void launch_fa(const FilterAggF64Args& a, int grid, hipStream_t s) {
    const bool same = a.x == a.y && a.x_offset == a.y_offset;
    const bool hasv = a.x_validity != nullptr || (!same && a.y_validity != nullptr);
    if (same) {
        if (hasv) hipLaunchKernelGGL((filter_agg_f64_kernel<CMP, true, true>), dim3(grid), dim3(kBlock), 0, s, a);
        else hipLaunchKernelGGL((filter_agg_f64_kernel<CMP, true, false>), dim3(grid), dim3(kBlock), 0, s, a);
    } else {
        if (hasv) hipLaunchKernelGGL((filter_agg_f64_kernel<CMP, false, true>), dim3(grid), dim3(kBlock), 0, s, a);
        else hipLaunchKernelGGL((filter_agg_f64_kernel<CMP, false, false>), dim3(grid), dim3(kBlock), 0, s, a);
    }
}
hipError_t launch_filter_agg_f64(const FilterAggF64Args& a, int cmp_op, int grid, hipStream_t s) {
    switch (cmp_op) {
        case RDF_OP_GT: launch_fa<RDF_OP_GT>(a, grid, s); break;
        case RDF_OP_GE: launch_fa<RDF_OP_GE>(a, grid, s); break;
        case RDF_OP_EQ: launch_fa<RDF_OP_EQ>(a, grid, s); break;
        case RDF_OP_NE: launch_fa<RDF_OP_NE>(a, grid, s); break;
        case RDF_OP_LT: launch_fa<RDF_OP_LT>(a, grid, s); break;
        default: launch_fa<RDF_OP_LE>(a, grid, s); break;
    }
    return hipGetLastError();
}

hipError_t launch_mask_count(const MaskTables& t, int64_t* tile_counts, hipStream_t s) {
    const int64_t tiles_per_block = kBlock / 32;
    const int64_t grid = (t.ntiles + tiles_per_block - 1) / tiles_per_block;
    if (grid > 0) hipLaunchKernelGGL(mask_count_kernel, dim3((unsigned)grid), dim3(kBlock), 0, s, t, tile_counts);
    return hipGetLastError();
}
hipError_t launch_scan(const int64_t* counts, int64_t* scan, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(kScanThreads), 0, s, counts, scan, n);
    return hipGetLastError();
}
hipError_t launch_compact(const FilterArgs& a, hipStream_t s) {
    int64_t grid = a.t.ntiles < (int64_t)eval_grid_limit() ? a.t.ntiles : (int64_t)eval_grid_limit();
    if (grid > 0) hipLaunchKernelGGL(compact_kernel, dim3((unsigned)grid), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

template <typename T>
static void launch_take_t(const TakeArgs& a, int grid, hipStream_t s) {
    if (a.idx64) hipLaunchKernelGGL((take_kernel<T, uint64_t>), dim3(grid), dim3(kBlock), 0, s, a);
    else hipLaunchKernelGGL((take_kernel<T, uint32_t>), dim3(grid), dim3(kBlock), 0, s, a);
}
hipError_t launch_take(const TakeArgs& a, hipStream_t s) {
    const int64_t nwaves = (a.n + 63) >> 6;
    int64_t grid = (nwaves + (kBlock / 64) - 1) / (kBlock / 64);
    if (grid > eval_grid_limit()) grid = eval_grid_limit();
    if (grid <= 0) return hipSuccess;
    switch (a.esize) {
        case 8: launch_take_t<uint64_t>(a, (int)grid, s); break;
        case 4: launch_take_t<uint32_t>(a, (int)grid, s); break;
        case 2: launch_take_t<uint16_t>(a, (int)grid, s); break;
        default: launch_take_t<uint8_t>(a, (int)grid, s); break;
    }
    return hipGetLastError();
}

static int fill_grid(int64_t n) {
    int64_t g = (n + 255) / 256;
    if (g > 8192) g = 8192;
    return g < 1 ? 1 : (int)g;
}
hipError_t launch_fill_f64(double* p, int64_t n, uint64_t seed, uint64_t col, int64_t first_row, double lo, double hi, hipStream_t s) {
    hipLaunchKernelGGL(fill_f64_kernel, dim3(fill_grid(n)), dim3(256), 0, s, p, n, seed, col, first_row, lo, hi);
    return hipGetLastError();
}
hipError_t launch_fill_i64(int64_t* p, int64_t n, uint64_t seed, uint64_t col, int64_t first_row, int64_t lo, int64_t hi, hipStream_t s) {
    hipLaunchKernelGGL(fill_i64_kernel, dim3(fill_grid(n)), dim3(256), 0, s, p, n, seed, col, first_row, lo, (uint64_t)hi - (uint64_t)lo);
    return hipGetLastError();
}
hipError_t launch_fill_validity(uint8_t* p, int64_t nbits, uint64_t seed, uint64_t col, int64_t first_row, double null_fraction, hipStream_t s) {
    double t = null_fraction * 4294967296.0;
    uint64_t thr = t <= 0.0 ? 0 : t >= 4294967296.0 ? 4294967296ull : (uint64_t)t;
    hipLaunchKernelGGL(fill_validity_kernel, dim3(fill_grid(nbits)), dim3(256), 0, s, (uint64_t*)p, nbits, seed, col, first_row, thr);
    return hipGetLastError();
}

}  // namespace rdfk
