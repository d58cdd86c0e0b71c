// rdf_eval_lean.hip — the interpreter's kernel for the programs most programs are (round 6).
//
//   eval_lean_kernel<NPRE, NVAL>      [the same steps as rdf_eval.hip's eval_kernel<FEAT 0, SINK_AGG>: Evaluate::calculate
//     src/evaluation.rs:97-323, BooleanFilter::eval_to_array src/expression.rs:766-861, AggregateFunctions
//     src/functions/aggregate.rs:12-93 — filter -> {sum, min, max, count} of value expressions in one pass]
//
// Why a second kernel.  Read off the general kernel's ISA and counters (profiles/r06_pmc_interpreter_*.txt): a 256-row wave
// tile of filter -> sum cost 431 vector + 405 scalar instructions — 70 + 65 per bytecode step — and a SIMD retired about one
// instruction of either kind per four cycles, so the instruction COUNT was the kernel's time.  Two causes, neither a property
// of interpretation:
//   * the general kernel holds a step's operand in generic registers (`opnd`), so every LOAD / BIN step began by copying a
//     column or a literal there, and the paths of its big switch met again through register moves;
//   * some of its cases branch per LANE (guarded tail loads, divide-by-zero flags, aggregate updates under `if (live)`,
//     the vector-path bitmap loads), and one divergent branch anywhere makes the compiler structurize the WHOLE loop nest:
//     the wave-uniform dispatch is then lowered to flag registers and re-tests ("Flow" blocks) instead of plain scalar
//     branches — most of the 65 scalar instructions per step.
// This kernel has NO divergent branch — tails read clamped addresses, per-row conditions are selects, bitmaps come through
// the scalar path — and every step is dispatched on a handler index the HOST assigned (Instr::swapped bits 1..7, see
// lean_assign in rdf_capi.cpp); a binary step reads its operand where it lies: a column's registers, a scalar register
// pair (literals), or LDS (temporaries).  It takes the programs whose every step has a handler: columns of 8-byte types
// (f64 / i64 / u64) all held in registers (<= 4), f64 comparisons, f64 and 64-bit integer + - *, f64 /, AND / OR / NOT,
// i64 / u64 -> f64 casts, the filter, aggregate sinks.  Everything else runs on eval_kernel as before; results are the
// same bit for bit (the fold order of a lane's rows, of the lanes and of the blocks is eval_kernel's), except WHICH NaN a NaN
// result is (sign, payload: that follows the operand order and source modifiers of the compiled code, in either kernel).
#include "rdf_common.hip.h"

namespace rdfk {

#define RDF_ROWS _Pragma("unroll") for (int j = 0; j < kVPT; ++j)

typedef uint64_t lean_u64x2 __attribute__((ext_vector_type(2)));
typedef uint32_t lean_u32x2 __attribute__((ext_vector_type(2)));

// the lane's 4 bits (rows 4 l .. 4 l + 3) of a 256-bit window held as four 64-bit words (selects, no branch)
__device__ __forceinline__ uint32_t lean_nibble(const uint64_t (&w)[kVPT], int lane) {
    // (written as masks: left as `q == 0 ? w[0] : ...` the compiler branches per lane on q)
    const uint32_t q = (uint32_t)lane >> 4, sh = ((uint32_t)lane & 15u) * 4u;
    const uint32_t c0 = (uint32_t)(w[0] >> sh), c1 = (uint32_t)(w[1] >> sh), c2 = (uint32_t)(w[2] >> sh), c3 = (uint32_t)(w[3] >> sh);
    const uint32_t m0 = 0u - (uint32_t)(q == 0), m1 = 0u - (uint32_t)(q == 1), m2 = 0u - (uint32_t)(q == 2), m3 = 0u - (uint32_t)(q == 3);
    return ((c0 & m0) | (c1 & m1) | (c2 & m2) | (c3 & m3)) & 15u;
}

// An 8-byte column's rows rw + 4 l .. + 3 of a chunk of clen rows.  Full, 16-byte aligned spans: two vector loads per lane;
// anything else: four loads whose row index is clamped to the chunk's last row (rows past the end are never looked at: the
// step masks start from `inr`).  Both branches are wave-uniform.
__device__ __forceinline__ void lean_load8(const DevChunkCol cc, int64_t rw, int64_t clen, bool full, uint64_t (&v)[kVPT], uint32_t& valid) {
    const int lane = threadIdx.x & 63;
    const GlobalPtr<uint64_t> base = as_global<uint64_t>(cc.values) + cc.offset;
    if (full && (((uintptr_t)cc.values + (uintptr_t)(cc.offset + rw) * 8) & 15) == 0) {
        const GlobalPtr<lean_u64x2> p = (GlobalPtr<lean_u64x2>)(base + rw + (int64_t)lane * kVPT);
        const lean_u64x2 a0 = __builtin_nontemporal_load(p), a1 = __builtin_nontemporal_load(p + 1);
        v[0] = a0[0]; v[1] = a0[1]; v[2] = a1[0]; v[3] = a1[1];
    } else {
        const int64_t last = clen - 1;
        RDF_ROWS {
            const int64_t r = rw + (int64_t)lane * kVPT + j;
            v[j] = __builtin_nontemporal_load(base + (r < last ? r : last));
        }
    }
    valid = (1u << kVPT) - 1;
    if (cc.validity) {
        uint64_t w[kVPT];
        load_windows_s<kVPT>(cc.validity, cc.offset + rw, clen - rw, w);
        valid = lean_nibble(w, lane);
    }
}

// acc = acc OP b(r) for the binary handlers over the R rows a lane holds per trip; `live` = rows where both sides are valid and
// in range (a zero divisor is an error only there, like arrow's math_divide; the quotient of such a slot is 0 as in eval_kernel)
#define RDF_ROWS_R _Pragma("unroll") for (int r = 0; r < R; ++r)
template <int R, class B>
__device__ __forceinline__ void lean_bin(int h, uint64_t (&acc)[R], B b, uint32_t live, uint32_t& err) {
    switch (h) {
        case LH_F_GT: RDF_ROWS_R acc[r] = u2d(acc[r]) > u2d(b(r)); break;
        case LH_F_GE: RDF_ROWS_R acc[r] = u2d(acc[r]) >= u2d(b(r)); break;
        case LH_F_EQ: RDF_ROWS_R acc[r] = u2d(acc[r]) == u2d(b(r)); break;
        case LH_F_NE: RDF_ROWS_R acc[r] = u2d(acc[r]) != u2d(b(r)); break;
        case LH_F_LT: RDF_ROWS_R acc[r] = u2d(acc[r]) < u2d(b(r)); break;
        case LH_F_LE: RDF_ROWS_R acc[r] = u2d(acc[r]) <= u2d(b(r)); break;
        case LH_F_ADD: RDF_ROWS_R acc[r] = d2u(u2d(acc[r]) + u2d(b(r))); break;
        case LH_F_SUB: RDF_ROWS_R acc[r] = d2u(u2d(acc[r]) - u2d(b(r))); break;
        case LH_F_RSUB: RDF_ROWS_R acc[r] = d2u(u2d(b(r)) - u2d(acc[r])); break;
        case LH_F_MUL: RDF_ROWS_R acc[r] = d2u(u2d(acc[r]) * u2d(b(r))); break;
        case LH_F_DIV:
            RDF_ROWS_R {
                const bool z = u2d(b(r)) == 0.0;
                err |= (uint32_t)z & (live >> r) & 1u;
                double q = u2d(acc[r]) / u2d(b(r));
                asm volatile("" : "+v"(q));   // (keeps the division out of a per-lane branch the compiler would form around it)
                acc[r] = z ? 0 : d2u(q);
            }
            break;
        case LH_F_RDIV:
            RDF_ROWS_R {
                const bool z = u2d(acc[r]) == 0.0;
                err |= (uint32_t)z & (live >> r) & 1u;
                double q = u2d(b(r)) / u2d(acc[r]);
                asm volatile("" : "+v"(q));
                acc[r] = z ? 0 : d2u(q);
            }
            break;
        case LH_I_ADD: RDF_ROWS_R acc[r] = acc[r] + b(r); break;
        case LH_I_SUB: RDF_ROWS_R acc[r] = acc[r] - b(r); break;
        case LH_I_RSUB: RDF_ROWS_R acc[r] = b(r) - acc[r]; break;
        case LH_I_MUL: RDF_ROWS_R acc[r] = acc[r] * b(r); break;
        case LH_AND: RDF_ROWS_R acc[r] = acc[r] & b(r); break;
        default: RDF_ROWS_R acc[r] = acc[r] | b(r); break;   // LH_OR
    }
}

// block_reduce_agg (rdf_common.hip.h) without its per-lane branches — ONE divergent branch anywhere in the function and the
// compiler structurizes all of it, the step dispatch included (checked on the ISA: 162 flag-carrying "Flow" blocks in this
// kernel with the shared reduction, none with this one).  Same fold, same result: the butterfly runs in every lane, lane 0's
// totals are broadcast through scalar registers and written by all lanes, and every thread folds the four wave totals in wave
// order and stores the block's partial (256 stores of one value).
__device__ __forceinline__ void lean_block_reduce(int cls, uint64_t sum, uint64_t mn, uint64_t mx, int64_t cnt, AggPartial* lds4, AggPartial* out) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const uint64_t s2 = shfl_xor64(sum, m), mn2 = shfl_xor64(mn, m), mx2 = shfl_xor64(mx, m);
        const int64_t c2 = (int64_t)shfl_xor64((uint64_t)cnt, m);
        agg_merge(cls, sum, mn, mx, cnt, s2, mn2, mx2, c2);
    }
    sum = uniform64(sum); mn = uniform64(mn); mx = uniform64(mx); cnt = (int64_t)uniform64((uint64_t)cnt);   // lane 0's, as block_reduce_agg keeps
    const int wave = wave_id();   // (a scalar: with a per-lane index the totals read back count as per-lane values, and the fold's selects as per-lane branches)
    __syncthreads();
    lds4[wave].sum = sum; lds4[wave].mn = mn; lds4[wave].mx = mx; lds4[wave].cnt = cnt;
    __syncthreads();
    uint64_t s = lds4[0].sum, lo = lds4[0].mn, hi = lds4[0].mx;
    int64_t c = lds4[0].cnt;
    for (int w = 1; w < kBlock / 64; ++w) agg_merge(cls, s, lo, hi, c, lds4[w].sum, lds4[w].mn, lds4[w].mx, lds4[w].cnt);
    out->sum = s; out->mn = lo; out->mx = hi; out->cnt = c;
}

// SINK_STORE: the four 64-bit words of a wave's 256 rows in a bitmap (validity, or the values of a Boolean output).  The word of
// rows 64 q .. 64 q + 63 is assembled in lanes 16 q .. 16 q + 15 (nibbles OR-ed over the 16 lanes) and written by lane 16 q
// through a buffer descriptor that ends with the chunk: the other lanes' offsets lie outside it, as do words past the chunk's
// last row, and the hardware drops such stores — no per-lane branch.
__device__ __forceinline__ void lean_store_words(uint8_t* bitmap, int64_t rw, int64_t rows_left, uint32_t nibble, int lane) {
    uint64_t x = (uint64_t)(nibble & 15u) << (((uint32_t)lane & 15u) * 4u);
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) x |= shfl_xor64(x, m);
    const int64_t span = rows_left < 0 ? 0 : rows_left < (int64_t)kVPT * 64 ? rows_left : (int64_t)kVPT * 64;   // (a wave past the chunk's end: an empty descriptor)
    const int nwords = (int)((span + 63) >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(bitmap + (rw >> 6) * 8, 0, nwords * 8, 0x00020000);
    lean_u32x2 w;
    w[0] = (uint32_t)x; w[1] = (uint32_t)(x >> 32);
    const uint32_t off = ((uint32_t)lane & 15u) == 0 ? ((uint32_t)lane >> 4) * 8u : 0x7FFFFFF0u;
    __builtin_amdgcn_raw_buffer_store_b64(w, rs, (int)off, 0, 0);
}

// T = tiles per trip of the step loop.  With T = 2 a block interprets its tile AND the next one it would visit (tile + gridDim.x)
// in one pass over the bytecode — a lane holds 8 rows, a step's dispatch is paid once for both — and two tiles' columns are in
// flight per wave while it does.  A lane's rows meet its running aggregates in the order they always did (this tile's four, then
// the next tile's four), so the result does not depend on T.
template <int SINK, int NPRE, int NVAL, int T>
__global__ __launch_bounds__(kBlock) void eval_lean_kernel(const EvalArgs a) {
    constexpr int R = T * kVPT;
    constexpr uint32_t kAllRows = (1u << R) - 1u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ AggPartial red_lds[kBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = wave_id();

    uint64_t g_sum[NVAL], g_mn[NVAL], g_mx[NVAL];
    int64_t g_cnt[NVAL];
    if constexpr (SINK == SINK_AGG) {
#pragma unroll
        for (int k = 0; k < NVAL; ++k) agg_init(k < a.nvalues ? a.value_cls[k] : CLS_F64, g_sum[k], g_mn[k], g_mx[k], g_cnt[k]);
    }
    uint32_t err = 0;

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
    // rows of this lane that exist, and whether every row of the wave's span does (wave-uniform)
    auto rows_in_range = [&](const TileLoc& t, bool& full) -> uint32_t {
        const int64_t rw_ = t.r0 + (int64_t)wave * (kVPT * 64);
        const int64_t left = t.clen - rw_ - (int64_t)lane * kVPT;       // rows from this lane's first to the chunk's end
        full = t.clen - rw_ >= (int64_t)kVPT * 64;
        const int64_t n = left < 0 ? 0 : left > kVPT ? kVPT : left;
        return (1u << (uint32_t)n) - 1u;
    };
    // one tile's columns into v / vv, its row mask into inr_; a tile that does not exist (past the last one) has no rows
    auto load_tile = [&](bool exists, const TileLoc& t, uint64_t (&v)[NPRE][kVPT], uint32_t (&vv)[NPRE], uint32_t& inr_) {
        if (!exists) {
            inr_ = 0;
#pragma unroll
            for (int p = 0; p < NPRE; ++p) { vv[p] = 0; RDF_ROWS v[p][j] = 0; }
            return;
        }
        const int64_t rw_ = t.r0 + (int64_t)wave * (kVPT * 64);
        bool full;
        inr_ = rows_in_range(t, full);
#pragma unroll
        for (int p = 0; p < NPRE; ++p) {
            if (p < a.ncols) {
                // (the table through the constant address space: a scalar load.  `a.nchunks == 1 ? inline : a.cols[..]` selects
                // between a kernel-argument address and a global one, i.e. a FLAT vector load whose result the compiler must
                // treat as different in every lane — every test on the descriptor then becomes a divergent branch)
                DevChunkCol cc;
                if (a.nchunks == 1) cc = a.inline_cols[p];
                else cc = const_col(a.cols, (int64_t)p * a.nchunks + t.c);
                lean_load8(cc, rw_, t.clen, full, v[p], vv[p]);
            } else {
                vv[p] = 0;
                RDF_ROWS v[p][j] = 0;
            }
        }
    };

    const uint64_t* const code_words = (const uint64_t*)a.code;
    uint64_t* const tmp_vals = (uint64_t*)smem;                                        // [ntmp][R][kBlock]
    uint32_t* const tmp_valid = (uint32_t*)(smem + (size_t)a.ntmp * R * kBlock * 8);   // [ntmp][kBlock]: R bits each

    // the trip's tiles: tile, tile + gridDim.x, ... (T of them); the NEXT trip's columns are asked for before this one is interpreted
    int64_t tile = blockIdx.x;
    TileLoc tl[T];
    uint64_t pfv[T][NPRE][kVPT];
    uint32_t pfvalid[T][NPRE], pfinr[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int64_t tt = tile + (int64_t)t * gridDim.x;
        const bool ex = tt < a.ntiles;
        tl[t] = ex ? locate(tt) : TileLoc{0, 0, 0};
        load_tile(ex, tl[t], pfv[t], pfvalid[t], pfinr[t]);
    }
    bool have = tile < a.ntiles;

    while (have) {
        uint64_t colv[T][NPRE][kVPT];
        uint32_t colvalid[NPRE];      // R bits per column: tile t's rows at bits 4 t .. 4 t + 3
        uint32_t inr = 0;
        TileLoc cur[T];
#pragma unroll
        for (int p = 0; p < NPRE; ++p) colvalid[p] = 0;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            cur[t] = tl[t];
            inr |= pfinr[t] << (kVPT * t);
#pragma unroll
            for (int p = 0; p < NPRE; ++p) {
                colvalid[p] |= pfvalid[t][p] << (kVPT * t);
                RDF_ROWS colv[t][p][j] = pfv[t][p][j];
            }
        }
        const int64_t ntile = tile + (int64_t)T * gridDim.x;
        const bool nhave = ntile < a.ntiles;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int64_t tt = ntile + (int64_t)t * gridDim.x;
            const bool ex = tt < a.ntiles;
            tl[t] = ex ? locate(tt) : TileLoc{0, 0, 0};
            load_tile(ex, tl[t], pfv[t], pfvalid[t], pfinr[t]);
        }

        uint64_t acc[R];
        uint32_t accv = 0, keep = inr;
        RDF_ROWS_R acc[r] = 0;

        uint64_t nw0 = code_words[0], nw1 = code_words[1];
        for (int pc = 0; pc < a.ncode; ++pc) {
            // one step = 16 bytes in scalar registers; the next step's words are asked for now and used one trip later
            const uint64_t iw0 = nw0, imm = nw1;
            if (pc + 1 < a.ncode) { nw0 = code_words[2 * pc + 2]; nw1 = code_words[2 * pc + 3]; }
            const int h = (int)(iw0 >> 41) & 127, kind = (int)(iw0 >> 24) & 255, src = (int)(iw0 >> 48);
            switch (h) {
                case LH_LOAD:
                    if (kind == SRC_COL) {
#pragma unroll
                        for (int p = 0; p < NPRE; ++p)
                            if (p == src) { accv = colvalid[p]; RDF_ROWS_R acc[r] = colv[r / kVPT][p][r % kVPT]; }
                    } else if (kind == SRC_IMM) {
                        accv = kAllRows;
                        RDF_ROWS_R acc[r] = imm;
                    } else {
                        const uint64_t* tv = tmp_vals + (size_t)src * R * kBlock + tid;
                        RDF_ROWS_R acc[r] = tv[r * kBlock];
                        accv = tmp_valid[src * kBlock + tid];
                    }
                    break;
                case LH_STORE_TMP: {
                    uint64_t* tv = tmp_vals + (size_t)src * R * kBlock + tid;
                    RDF_ROWS_R tv[r * kBlock] = acc[r];
                    tmp_valid[src * kBlock + tid] = accv;
                } break;
                case LH_FILTER: {   // DataFrame::filter: rows whose predicate is false or null are dropped
                    uint32_t pass = 0;
                    RDF_ROWS_R pass |= ((uint32_t)acc[r] & 1u) << r;
                    keep &= pass & accv;
                } break;
                case LH_NOT: RDF_ROWS_R acc[r] ^= 1ull; break;
                case LH_CAST_I2F: RDF_ROWS_R acc[r] = d2u((double)(int64_t)acc[r]); break;
                case LH_CAST_U2F: RDF_ROWS_R acc[r] = d2u((double)acc[r]); break;
                case LH_EMIT: {   // acc is value expression `src`
                    if constexpr (SINK == SINK_AGG) {   // folded into the lane's running {sum, min, max, count}, rows in trip order
                        const uint32_t live = keep & accv;
#pragma unroll
                        for (int kk = 0; kk < NVAL; ++kk)
                            if (kk == src) {
                                const int cls = a.value_cls[kk];
                                g_cnt[kk] += (int64_t)__popc(live);
                                // A dead row offers each fold its identity: +0.0 / 0 to the sum (an f64 sum starts at +0.0 and can never
                                // become -0.0, so adding +0.0 changes no bit), the running minimum / maximum to min / max (fmin(x, x) = x
                                // bit for bit, NaN included).  Picked with bit masks, not `live ? a : b`: several selects on one per-lane
                                // condition are what the compiler turns into a per-lane branch.
                                if (cls == CLS_F64) {
                                    RDF_ROWS_R {
                                        const uint64_t m = 0ull - (uint64_t)((live >> r) & 1u);
                                        const uint64_t s = acc[r] & m, lo = (acc[r] & m) | (g_mn[kk] & ~m), hi = (acc[r] & m) | (g_mx[kk] & ~m);
                                        g_sum[kk] = d2u(u2d(g_sum[kk]) + u2d(s));
                                        g_mn[kk] = d2u(fmin(u2d(g_mn[kk]), u2d(lo)));
                                        g_mx[kk] = d2u(fmax(u2d(g_mx[kk]), u2d(hi)));
                                    }
                                } else if (cls == CLS_SIGNED) {
                                    RDF_ROWS_R {
                                        const uint64_t m = 0ull - (uint64_t)((live >> r) & 1u);
                                        const int64_t lo = (int64_t)((acc[r] & m) | (g_mn[kk] & ~m)), hi = (int64_t)((acc[r] & m) | (g_mx[kk] & ~m));
                                        g_sum[kk] += acc[r] & m;
                                        g_mn[kk] = (uint64_t)(lo < (int64_t)g_mn[kk] ? lo : (int64_t)g_mn[kk]);
                                        g_mx[kk] = (uint64_t)(hi > (int64_t)g_mx[kk] ? hi : (int64_t)g_mx[kk]);
                                    }
                                } else {
                                    RDF_ROWS_R {
                                        const uint64_t m = 0ull - (uint64_t)((live >> r) & 1u);
                                        const uint64_t lo = (acc[r] & m) | (g_mn[kk] & ~m), hi = (acc[r] & m) | (g_mx[kk] & ~m);
                                        g_sum[kk] += acc[r] & m;
                                        g_mn[kk] = lo < g_mn[kk] ? lo : g_mn[kk];
                                        g_mx[kk] = hi > g_mx[kk] ? hi : g_mx[kk];
                                    }
                                }
                            }
                    } else {   // SINK_STORE: each tile's rows of output column `src` (NULL slots hold 0), its validity words, its NULL count
                        const int ddt = (int)(iw0 >> 16) & 255;
#pragma unroll
                        for (int t = 0; t < T; ++t) {
                            const int64_t rw = cur[t].r0 + (int64_t)wave * (kVPT * 64), left = cur[t].clen - rw;   // (left <= 0: nothing of this wave's span exists — also a tile past the last one — and nothing is written)
                            DevOutChunk oc;
                            if (a.nchunks == 1) oc = a.inline_outs[src & (kMaxValues - 1)];
                            else {
                                const ConstPtr<DevOutChunk> ot = as_const<DevOutChunk>(a.outs) + ((int64_t)src * a.nchunks + cur[t].c);
                                oc.values = ot->values; oc.validity = ot->validity;
                            }
                            const uint32_t inr_t = (inr >> (kVPT * t)) & 15u, accv_t = (accv >> (kVPT * t)) & 15u, live = accv_t & inr_t;
                            if (ddt == RDF_BOOL) {
                                uint32_t bits = 0;
                                RDF_ROWS bits |= ((uint32_t)acc[kVPT * t + j] & 1u) << j;
                                lean_store_words((uint8_t*)oc.values, rw, left, bits & live, lane);
                            } else {
                                uint64_t vv[kVPT];
                                RDF_ROWS vv[j] = acc[kVPT * t + j] & (0ull - (uint64_t)((accv_t >> j) & 1u));
                                if (left >= (int64_t)kVPT * 64 && (((uintptr_t)oc.values + (uintptr_t)rw * 8) & 15) == 0) {
                                    lean_u64x2 s0, s1;
                                    s0[0] = vv[0]; s0[1] = vv[1]; s1[0] = vv[2]; s1[1] = vv[3];
                                    GlobalMutPtr<lean_u64x2> q = (GlobalMutPtr<lean_u64x2>)(as_global_mut<uint64_t>(oc.values) + rw + (int64_t)lane * kVPT);
                                    __builtin_nontemporal_store(s0, q); __builtin_nontemporal_store(s1, q + 1);
                                } else {   // a tail or an odd start: one bounds-checked store per row, rows past the chunk's end dropped by the hardware
                                    const int64_t span = left < 0 ? 0 : left < (int64_t)kVPT * 64 ? left : (int64_t)kVPT * 64;
                                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((uint64_t*)oc.values + rw, 0, (int)span * 8, 0x00020000);
                                    RDF_ROWS {
                                        lean_u32x2 w;
                                        w[0] = (uint32_t)vv[j]; w[1] = (uint32_t)(vv[j] >> 32);
                                        __builtin_amdgcn_raw_buffer_store_b64(w, rs, (lane * kVPT + j) * 8, 0, 0);
                                    }
                                }
                            }
                            if (oc.validity) lean_store_words(oc.validity, rw, left, live, lane);
                            // NULLs of the tile: in-range rows that are not live, counted on the scalar unit; the (rare) add goes out
                            // through a one-word descriptor that only lane 0's offset falls into
                            uint32_t nn = 0;
                            RDF_ROWS nn += (uint32_t)__popcll(__ballot(((inr_t & ~live) >> j) & 1u));
                            if (nn != 0) {
                                const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(a.out_null_counts + ((int64_t)src * a.nchunks + cur[t].c), 0, 4, 0x00020000);
                                (void)__builtin_amdgcn_raw_ptr_buffer_atomic_add_i32((int)nn, rc, lane == 0 ? 0 : 0x7FFFFFF0, 0, 0);
                            }
                        }
                    }
                } break;
                default: {   // the binary handlers: the operand is read where it lies
                    const int sdt = (int)(iw0 >> 32) & 255, ddt = (int)(iw0 >> 16) & 255;   // an integer column met in the f64 domain is converted on the way
                    if (kind == SRC_IMM) {
                        lean_bin<R>(h, acc, [&](int) { return imm; }, accv & inr, err);
                    } else if (kind == SRC_COL && sdt == ddt) {
#pragma unroll
                        for (int p = 0; p < NPRE; ++p)
                            if (p == src) {
                                accv &= colvalid[p];
                                lean_bin<R>(h, acc, [&](int r) { return colv[r / kVPT][p][r % kVPT]; }, accv & inr, err);
                            }
                    } else {
                        uint64_t opnd[R];
                        uint32_t opv = 0;
                        if (kind == SRC_COL) {
#pragma unroll
                            for (int p = 0; p < NPRE; ++p)
                                if (p == src) { opv = colvalid[p]; RDF_ROWS_R opnd[r] = colv[r / kVPT][p][r % kVPT]; }
                            if (sdt == RDF_I64) RDF_ROWS_R opnd[r] = d2u((double)(int64_t)opnd[r]);
                            else RDF_ROWS_R opnd[r] = d2u((double)opnd[r]);
                        } else {
                            const uint64_t* tv = tmp_vals + (size_t)src * R * kBlock + tid;
                            RDF_ROWS_R opnd[r] = tv[r * kBlock];
                            opv = tmp_valid[src * kBlock + tid];
                        }
                        accv &= opv;
                        lean_bin<R>(h, acc, [&](int r) { return opnd[r]; }, accv & inr, err);
                    }
                } break;
            }
        }
        tile = ntile;
        have = nhave;
    }

    // divide by zero at a live slot: bit 0 is the only flag this kernel can raise and the host cleared the word before the
    // launch, so a plain store by every lane says what atomicOr would (an atomic on one address is rewritten by the compiler
    // into "the first active lane does it": a per-lane branch)
    if (__ballot(err != 0) != 0) *(volatile uint32_t*)a.flags = 1u;
    if constexpr (SINK == SINK_AGG) {
#pragma unroll
        for (int k = 0; k < NVAL; ++k)
            if (k < a.nvalues)
                lean_block_reduce(a.value_cls[k], g_sum[k], g_mn[k], g_mx[k], g_cnt[k], red_lds, &a.partials[(int64_t)blockIdx.x * a.nvalues + k]);
    }
}

template <int SINK, int NPRE, int NVAL, int T>
static void lean_launch_one(const EvalArgs& a, int grid, hipStream_t s) {
    const size_t lds = (size_t)a.ntmp * ((size_t)T * kVPT * kBlock * 8 + kBlock * 4);
    hipLaunchKernelGGL((eval_lean_kernel<SINK, NPRE, NVAL, T>), dim3(grid), dim3(kBlock), lds, s, a);
}

// SINK_AGG: (NPRE, NVAL) in {(1,1),(2,1),(2,2),(4,1),(4,2),(4,4)} like eval_kernel; SINK_STORE keeps no per-value state: NPRE only.
// Two tiles per trip where it measured ahead (1e9 rows, tools/lean_ab.py --time): one-column programs (filter -> sum 1.83 -> 1.51 ms,
// stored 3.49 -> 3.41) and two-column aggregates (3.21 -> 3.01, 2.72 -> 2.71: 157-165 registers, three waves per SIMD instead of
// four, about pay for the saved dispatch); two-column stores lose 3 % and four columns' worth of rows for two tiles would cost
// more occupancy still.  The temporaries must fit 64 KB of LDS that way; one_tile forces one (rdf_set_option("interp_lean", 2)).
hipError_t launch_eval_lean(const EvalArgs& a, int sink, int grid, hipStream_t s, bool one_tile) {
    const int npre = a.ncols, nval = a.nvalues;
    const bool two = !one_tile && (npre <= 1 || (npre <= 2 && sink == SINK_AGG)) && (size_t)a.ntmp * (2 * kVPT * kBlock * 8 + kBlock * 4) <= 65536;
    if (sink == SINK_STORE) {
        if (npre <= 1) { if (two) lean_launch_one<SINK_STORE, 1, 1, 2>(a, grid, s); else lean_launch_one<SINK_STORE, 1, 1, 1>(a, grid, s); }
        else if (npre <= 2) lean_launch_one<SINK_STORE, 2, 1, 1>(a, grid, s);
        else lean_launch_one<SINK_STORE, 4, 1, 1>(a, grid, s);
    } else if (nval <= 1) {
        if (npre <= 1) { if (two) lean_launch_one<SINK_AGG, 1, 1, 2>(a, grid, s); else lean_launch_one<SINK_AGG, 1, 1, 1>(a, grid, s); }
        else if (npre <= 2) { if (two) lean_launch_one<SINK_AGG, 2, 1, 2>(a, grid, s); else lean_launch_one<SINK_AGG, 2, 1, 1>(a, grid, s); }
        else lean_launch_one<SINK_AGG, 4, 1, 1>(a, grid, s);
    } else if (nval <= 2) {
        if (npre <= 2) { if (two) lean_launch_one<SINK_AGG, 2, 2, 2>(a, grid, s); else lean_launch_one<SINK_AGG, 2, 2, 1>(a, grid, s); }
        else lean_launch_one<SINK_AGG, 4, 2, 1>(a, grid, s);
    } else lean_launch_one<SINK_AGG, 4, 4, 1>(a, grid, s);
    return hipGetLastError();
}

}  // namespace rdfk
