// rdf_spec.hip — ahead-of-time specialised fused kernels (expression templates).
//
// The general evaluator (rdf_eval.hip) interprets any expression tree; its price is registers.  For
// the program shapes that dominate the path — one ScalarFunctions op per call (src/functions/scalar.rs),
// a comparison against a scalar (BooleanFilter, src/expression.rs:836-859), an aggregate of a column
// or of a small fused expression (BASELINE configs C1-C3) — this file instantiates straight-line
// kernels from C++ expression templates over 8-byte (f64 / i64 / u64) or 4-byte (f32 / i32 / u32) columns
// (all columns of one program share one width, so a 16-byte vector holds the same rows of every column):
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

#include "rdf_expr.hip.h"

namespace rdfk {

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
template <> struct AggT<RDF_F32> : AggT<RDF_F64> {   // f32 values fold in f64 (rounded once at the end by the host)
    __device__ __forceinline__ void add(float v) { AggT<RDF_F64>::add((double)v); }
};
template <> struct AggT<RDF_I32> : AggT<RDF_I64> {
    __device__ __forceinline__ void add(int32_t v) { AggT<RDF_I64>::add((int64_t)v); }
};
template <> struct AggT<RDF_U32> : AggT<RDF_U64> {
    __device__ __forceinline__ void add(uint32_t v) { AggT<RDF_U64>::add((uint64_t)v); }
};

// Lane l of a 16-byte-load wave holds RV consecutive rows (RV = 2 for 8-byte, 4 for 4-byte elements), so
// the RV per-element ballots must be interleaved into Arrow's row-ordered bitmap words.  Every lane j picks
// the bit that belongs at output position j of word h (rows 64h..64h+63 of the wave-load) and the wave
// ballots again: a handful of VALU instructions per word for the whole wave.
template <int RV>
__device__ __forceinline__ uint64_t interleave_word(const uint64_t (&b)[RV], int h, int lane) {
    uint64_t src = b[0];
#pragma unroll
    for (int e = 1; e < RV; ++e) if ((lane % RV) == e) src = b[e];
    return __ballot((src >> ((64 / RV) * h + lane / RV)) & 1);
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
    static constexpr int W = merge_width(merge_width(PRED::width, V0::width), V1::width);  // element width of every column
    static_assert(W == 8 || W == 4, "a program reads columns of one width (8 or 4 bytes)");
    static constexpr int RV = 16 / W;                 // rows per 16-byte vector
    static constexpr int R = NC <= 2 ? 8 : 4;         // rows per lane per iteration
    static constexpr int U = R / RV;                  // 16-byte vectors per lane per column per iteration
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
    constexpr int NC = P::NC, U = P::U, R = P::R, RV = P::RV, W = P::W;
    using S = typename std::conditional<W == 8, uint64_t, uint32_t>::type;
    using VecS = typename VecOf<S, RV>::type;
    using Pred = typename P::Pred;
    using V0 = typename P::Val0;
    using V1 = typename P::Val1;
    constexpr bool has_pred = !std::is_same<Pred, None>::value;
    constexpr bool has_v1 = !std::is_same<V1, None>::value;
    __shared__ AggPartial red_lds[kBlock / 64];
    const int lane = threadIdx.x & 63;
    const int wave = wave_id();

    Ctx<NC, R, S> c;
    c.err = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) c.imm[k] = a.imm[k];
#pragma unroll
    for (int k = 0; k < 8; ++k) c.rt[k] = a.rt[k];
    AggT<V0::dt> g0;
    using V1e = typename std::conditional<has_v1, V1, V0>::type;
    AggT<V1e::dt> g1;
    g0.init();
    g1.init();
    uint32_t nulls = 0;
    int64_t cur_chunk = -1;

    // A "tile" is one WAVE iteration: 64*U vectors = 64*R consecutive rows of ONE chunk.  Waves walk the tile list
    // independently (wave w of block b starts at tile 4b + w), so a frame held in the reference's 1024-row batches
    // (src/dataframe.rs:352) keeps every wave busy: with block-wide tiles of 2048 rows a 1024-row chunk left two of the
    // four waves without rows.  The scalar work per wave and iteration is what it was (every wave located the block's
    // tile redundantly before).
    constexpr int64_t per_tile = (int64_t)64 * U;
    constexpr int kWaves = kBlock / 64;
    // Where a tile lives (chunk, first vector, chunk length, column descriptors): an interpolated guess into the prefix
    // table (exact or one off for equally long batches) checked against its two neighbours, all on the scalar unit.  The
    // NEXT tile is located while the current tile's vector loads are in flight, so the lookup latency is off the critical path.
    struct TileMeta { int64_t ch, base, n; DevChunkCol col[NC]; DevOutChunk out; };
    auto locate = [&](int64_t tile) -> TileMeta {
        TileMeta m;
        m.ch = 0;
        m.out = a.out;
        if (a.nchunks == 1) {
            m.base = tile * per_tile;
            m.n = a.n;
#pragma unroll
            for (int k = 0; k < NC; ++k) m.col[k] = a.cols[k];
        } else {
            m.ch = find_chunk_tile_inv(a.chunk_tile_start, a.nchunks, tile, a.tile_inv);
            m.base = (tile - a.chunk_tile_start[m.ch]) * per_tile;
            m.n = a.chunk_len[m.ch];
#pragma unroll
            for (int k = 0; k < NC; ++k) m.col[k] = a.cols_tab[(int64_t)k * a.nchunks + m.ch];
            if (P::SINK == SINK_STORE) m.out = a.outs_tab[m.ch];
        }
        return m;
    };
    const int64_t tile0 = (int64_t)blockIdx.x * kWaves + wave, tstride = (int64_t)gridDim.x * kWaves;
    TileMeta meta = locate(tile0 < a.ntiles ? tile0 : 0);
    for (int64_t tile = tile0; tile < a.ntiles; tile += tstride) {
        const int64_t ch = meta.ch, base = meta.base, n = meta.n;
        DevChunkCol col[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) col[k] = meta.col[k];
        const DevOutChunk out = meta.out;
        if (P::SINK == SINK_STORE && ch != cur_chunk) {  // one null-count atomic per (wave, chunk), not per tile
            if (cur_chunk >= 0 && lane == 0 && nulls) atomicAdd((unsigned long long*)&a.out_null_count[cur_chunk], (unsigned long long)nulls);
            nulls = 0;
            cur_chunk = ch;
        }
        const int64_t wbase = base;                             // first vector of this wave's tile
        const int64_t rw = (int64_t)RV * wbase;                 // its first row
        // `full` (wave-uniform) = every row of this wave's span exists: the common case runs without
        // per-lane bounds checks or predicated loads
        const bool full = rw + 64 * R <= n;
        if (full) {
            c.inr = (1u << R) - 1;
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                if (k > 0 && a.alias[k] >= 0) continue;   // a second use of a column already loaded for slot alias[k] (shape kernels)
                const GlobalPtr<VecS> p = (GlobalPtr<VecS>)(as_global<S>(col[k].values) + col[k].offset) + wbase + lane;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const VecS t = __builtin_nontemporal_load(p + u * 64);
#pragma unroll
                    for (int e = 0; e < RV; ++e) c.v[k][RV * u + e] = t[e];
                }
            }
#pragma unroll
            for (int k = 1; k < NC; ++k)      // aliases copy registers once every load has been issued
#pragma unroll
                for (int j = 0; j < k; ++j)
                    if (a.alias[k] == j) {
#pragma unroll
                        for (int r = 0; r < R; ++r) c.v[k][r] = c.v[j][r];
                    }
        } else {
            c.inr = 0;
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int e = 0; e < RV; ++e) c.inr |= (uint32_t)((int64_t)RV * (wbase + u * 64 + lane) + e < n) << (RV * u + e);
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const GlobalPtr<S> p = as_global<S>(col[k].values) + col[k].offset;
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int e = 0; e < RV; ++e)
                        c.v[k][RV * u + e] = ((c.inr >> (RV * u + e)) & 1) ? p[(int64_t)RV * (wbase + u * 64 + lane) + e] : (S)0;
            }
        }
        {   // the loads above are in flight: locate the next tile now
            const int64_t nt = tile + tstride;
            if (a.nchunks == 1) meta.base = nt * per_tile;
            else if (nt < a.ntiles) meta = locate(nt);
        }
        // validity: R windows of 64 rows per column for this wave; lane l's RV bits of load u sit in window
        // RV*u + (RV*l >> 6) at bit (RV*l) & 63
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            c.valid[k] = c.inr;
            if (k > 0 && a.alias[k] >= 0) {
#pragma unroll
                for (int j = 0; j < k; ++j) if (a.alias[k] == j) c.valid[k] = c.valid[j];
                continue;
            }
            if (col[k].validity) {
                uint64_t w[R];
                if (a.vec_bitmap) load_windows<R>(col[k].validity, col[k].offset + rw, n - rw, w);
                else load_windows_s<R>(col[k].validity, col[k].offset + rw, n - rw, w);
                uint32_t m = 0;
                const int sh = (RV * lane) & 63, wsel = (RV * lane) >> 6;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    uint64_t ww = w[RV * u];
#pragma unroll
                    for (int h = 1; h < RV; ++h) if (wsel == h) ww = w[RV * u + h];
                    m |= ((uint32_t)(ww >> sh) & ((1u << RV) - 1)) << (RV * u);
                }
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
            // bitmap words of this wave's 64*R rows: word w is parked in lane w, all R go out in ONE store
            uint64_t word_val = 0, word_vld = 0;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t i = wbase + u * 64 + lane;
                const uint32_t inu = (c.inr >> (RV * u)) & ((1u << RV) - 1);
                const uint32_t vu = (vm >> (RV * u)) & ((1u << RV) - 1);
                uint64_t x[RV], inb[RV], vb[RV];
#pragma unroll
                for (int e = 0; e < RV; ++e) {
                    x[e] = ((vu >> e) & 1) ? outv[RV * u + e] : 0;  // null slots hold 0
                    inb[e] = __ballot((inu >> e) & 1);
                    vb[e] = __ballot((vu >> e) & 1);
                }
                if constexpr (V0::dt == RDF_BOOL) {
                    uint64_t bb[RV];
#pragma unroll
                    for (int e = 0; e < RV; ++e) bb[e] = __ballot(x[e] & 1);
#pragma unroll
                    for (int h = 0; h < RV; ++h) {
                        const uint64_t wv = interleave_word<RV>(bb, h, lane);
                        if (lane == RV * u + h) word_val = wv;
                    }
                } else {
                    using OT = typename CType<V0::dt>::T;
                    static_assert(sizeof(OT) == W, "value width equals the column width");
                    if (inu == (1u << RV) - 1) {
                        VecS t;
#pragma unroll
                        for (int e = 0; e < RV; ++e) t[e] = (S)x[e];
                        __builtin_nontemporal_store(t, as_global_mut<VecS>(out.values) + i);   // streaming output: do not keep it in L2 / MALL
                    } else {
#pragma unroll
                        for (int e = 0; e < RV; ++e) if ((inu >> e) & 1) as_global_mut<S>(out.values)[(int64_t)RV * i + e] = (S)x[e];
                    }
                }
                if (out.validity) {
#pragma unroll
                    for (int h = 0; h < RV; ++h) {
                        const uint64_t wv = interleave_word<RV>(vb, h, lane);
                        if (lane == RV * u + h) word_vld = wv;
                    }
                }
#pragma unroll
                for (int e = 0; e < RV; ++e) nulls += (uint32_t)__popcll(inb[e] & ~vb[e]);   // wave-uniform: scalar unit
            }
            if (lane < R && rw + 64 * lane < n) {
                if constexpr (V0::dt == RDF_BOOL) as_global_mut<uint64_t>(out.values)[(rw >> 6) + lane] = word_val;
                if (out.validity) as_global_mut<uint64_t>(out.validity)[(rw >> 6) + lane] = word_vld;
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
struct SpecEntry { SpecLaunch launch; int rows_per_tile; };

template <class P>
static void launch_prog(const SpecArgs& a, int grid, hipStream_t s) {
    hipLaunchKernelGGL((spec_kernel<P>), dim3(grid), dim3(kBlock), 0, s, a);
}

static std::map<std::string, SpecEntry>& registry() {
    static std::map<std::string, SpecEntry> r;
    return r;
}
template <class P>
static void reg() { registry()[P::sig()] = SpecEntry{&launch_prog<P>, 64 * P::R}; }

using D0 = Col<0, RDF_F64>; using D1 = Col<1, RDF_F64>; using D2 = Col<2, RDF_F64>;
using L0 = Col<0, RDF_I64>; using L1 = Col<1, RDF_I64>; using L3 = Col<3, RDF_I64>;
using W0 = Col<0, RDF_U64>; using W1 = Col<1, RDF_U64>;
using KD0 = Imm<0, RDF_F64>; using KL0 = Imm<0, RDF_I64>;
using F0 = Col<0, RDF_F32>; using F1 = Col<1, RDF_F32>; using I0 = Col<0, RDF_I32>; using I1 = Col<1, RDF_I32>;
using J0 = Col<0, RDF_U32>; using J1 = Col<1, RDF_U32>;
using KF0 = Imm<0, RDF_F32>; using KI0 = Imm<0, RDF_I32>;

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
    // 4-byte columns: comparisons still happen in f64 (the scalar arrives as an f64 immediate)
    reg<Prog<Bin<OP, F0, KD0>, F0, None, SINK_AGG>>();
    reg<Prog<Bin<OP, I0, KD0>, I0, None, SINK_AGG>>();
    reg<Prog<None, Bin<OP, F0, KD0>, None, SINK_STORE>>();
    reg<Prog<None, Bin<OP, I0, KD0>, None, SINK_STORE>>();
}
template <int OP> static void reg_arith_family() {
    reg<Prog<None, Bin<OP, D0, D1>, None, SINK_STORE>>();   // ScalarFunctions::add/... f64
    reg<Prog<None, Bin<OP, L0, L1>, None, SINK_STORE>>();   // i64
    reg<Prog<None, Bin<OP, W0, W1>, None, SINK_STORE>>();   // u64
    reg<Prog<None, Bin<OP, D0, KD0>, None, SINK_STORE>>();  // column OP scalar ("add_scalar", config C1)
    reg<Prog<None, Bin<OP, L0, KL0>, None, SINK_STORE>>();
    reg<Prog<None, Bin<OP, F0, F1>, None, SINK_STORE>>();   // f32 / i32 / u32 (the reference's type matrix, src/evaluation.rs:107-238)
    reg<Prog<None, Bin<OP, I0, I1>, None, SINK_STORE>>();
    reg<Prog<None, Bin<OP, J0, J1>, None, SINK_STORE>>();
    reg<Prog<None, Bin<OP, F0, KF0>, None, SINK_STORE>>();
    reg<Prog<None, Bin<OP, I0, KI0>, None, SINK_STORE>>();
}
template <int OP> static void reg_unary_f64() {
    reg<Prog<None, Un<OP, D0>, None, SINK_STORE>>();                      // ScalarFunctions::<op>
    reg<Prog<None, Un<OP, Bin<RDF_OP_ADD, D0, KD0>>, None, SINK_AGG>>();  // sum(op(x + c)) — config C1 shape
    reg<Prog<None, Un<OP, D0>, None, SINK_AGG>>();
    reg<Prog<None, Un<OP, F0>, None, SINK_STORE>>();                      // f32
}

// ---- shape-specialised kernels with runtime operators (rdf_expr.hip.h *RT nodes).  Every leaf OCCURRENCE has its own
// canonical column / literal slot (the host maps two slots to the same column when a program reuses one: the second
// load hits in cache), operator slots are numbered in pre-order, predicate first.
template <int S, int C, int K, int DT> struct Shapes {
    template <int I> using Cd = Col<I, DT>;
    template <int I> using Kd = Imm<I, DT>;
    using c = Cd<C>;
    using cc = ArithRT<S, Cd<C>, Cd<C + 1>>;
    using ck = ArithRT<S, Cd<C>, Kd<K>>;
    using ccc = ArithRT<S, ArithRT<S + 1, Cd<C>, Cd<C + 1>>, Cd<C + 2>>;
    using cck = ArithRT<S, ArithRT<S + 1, Cd<C>, Cd<C + 1>>, Kd<K>>;
    using ckc = ArithRT<S, ArithRT<S + 1, Cd<C>, Kd<K>>, Cd<C + 1>>;
    using ckk = ArithRT<S, ArithRT<S + 1, Cd<C>, Kd<K>>, Kd<K + 1>>;
    using Tc = TrigRT<S, Cd<C>>;
    using Tcc = TrigRT<S, ArithRT<S + 1, Cd<C>, Cd<C + 1>>>;
    using Tck = TrigRT<S, ArithRT<S + 1, Cd<C>, Kd<K>>>;
};
template <class PRED, class V> static void reg_shape_agg() {
    if constexpr ((PRED::ncols > V::ncols ? PRED::ncols : V::ncols) <= 4) reg<Prog<PRED, V, None, SINK_AGG>>();
}
template <class PRED, int S, int C, int K, int DT> static void reg_shape_aggs() {
    using H = Shapes<S, C, K, DT>;
    reg_shape_agg<PRED, typename H::c>();
    reg_shape_agg<PRED, typename H::cc>(); reg_shape_agg<PRED, typename H::ck>();
    reg_shape_agg<PRED, typename H::ccc>(); reg_shape_agg<PRED, typename H::cck>();
    reg_shape_agg<PRED, typename H::ckc>(); reg_shape_agg<PRED, typename H::ckk>();
    if constexpr (DT == RDF_F64) { reg_shape_agg<PRED, typename H::Tc>(); reg_shape_agg<PRED, typename H::Tcc>(); reg_shape_agg<PRED, typename H::Tck>(); }
}
// DT = dtype of the value expression's columns and literals; PDT = dtype of the predicate's columns (compared in f64)
template <int DT, int PDT> static void reg_shape_family() {
    using H0 = Shapes<0, 0, 0, DT>;
    if constexpr (DT == PDT) {
        reg<Prog<None, typename H0::cc, None, SINK_STORE>>(); reg<Prog<None, typename H0::ck, None, SINK_STORE>>();
        reg<Prog<None, typename H0::ccc, None, SINK_STORE>>(); reg<Prog<None, typename H0::cck, None, SINK_STORE>>();
        reg<Prog<None, typename H0::ckc, None, SINK_STORE>>(); reg<Prog<None, typename H0::ckk, None, SINK_STORE>>();
        if constexpr (DT == RDF_F64) {
            reg<Prog<None, typename H0::Tc, None, SINK_STORE>>(); reg<Prog<None, typename H0::Tcc, None, SINK_STORE>>(); reg<Prog<None, typename H0::Tck, None, SINK_STORE>>();
        }
        reg_shape_aggs<None, 0, 0, 0, DT>();
    }
    using P1 = CmpRT<0, Col<0, PDT>, Imm<0, RDF_F64>>;                                   // x CMP c
    reg_shape_aggs<P1, 1, 1, 1, DT>();
    using P2 = LogicRT<0, CmpRT<1, Col<0, PDT>, Imm<0, RDF_F64>>, CmpRT<2, Col<1, PDT>, Imm<1, RDF_F64>>>;   // x CMP c AND|OR y CMP d
    reg_shape_aggs<P2, 3, 2, 2, DT>();
    if constexpr (DT == PDT) reg<Prog<None, P2, None, SINK_STORE>>();                                     // ... as a mask
}

static void build_registry() {
    reg_shape_family<RDF_F64, RDF_F64>();
    reg_shape_family<RDF_I64, RDF_I64>();
    reg_shape_family<RDF_F64, RDF_I64>();   // predicate on an i64 key, f64 measures
    reg_shape_family<RDF_I64, RDF_F64>();
    // aggregates of a plain column (AggregateFunctions::sum/min/max/count/avg)
    reg<Prog<None, D0, None, SINK_AGG>>();
    reg<Prog<None, L0, None, SINK_AGG>>();
    reg<Prog<None, W0, None, SINK_AGG>>();
    reg<Prog<None, Cast<RDF_F64, L0>, None, SINK_AGG>>();  // avg of an i64 column
    reg<Prog<None, Cast<RDF_F64, W0>, None, SINK_AGG>>();
    reg<Prog<None, F0, None, SINK_AGG>>();
    reg<Prog<None, I0, None, SINK_AGG>>();
    reg<Prog<None, J0, None, SINK_AGG>>();
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
    reg<Prog<None, Un<RDF_OP_ABS, I0>, None, SINK_STORE>>();
    // casts between the 8-byte types (Function::Cast, src/evaluation.rs:296-315)
    reg<Prog<None, Cast<RDF_F64, L0>, None, SINK_STORE>>();
    reg<Prog<None, Cast<RDF_I64, D0>, None, SINK_STORE>>();
    reg<Prog<None, Cast<RDF_F64, W0>, None, SINK_STORE>>();
    reg<Prog<None, Cast<RDF_U64, D0>, None, SINK_STORE>>();
    reg<Prog<None, Cast<RDF_F32, I0>, None, SINK_STORE>>();
    reg<Prog<None, Cast<RDF_I32, F0>, None, SINK_STORE>>();
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

int spec_rows_per_tile(const char* sig) {   // rows one wave iteration covers: tiles never straddle chunks
    const SpecEntry* e = spec_lookup(sig);
    return e ? e->rows_per_tile : 0;
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
