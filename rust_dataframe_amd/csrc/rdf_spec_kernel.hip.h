// rdf_spec_kernel.hip.h — the ahead-of-time specialised fused kernel (expression templates) and its catalog plumbing,
// shared by the translation units that instantiate it (rdf_spec.hip: exact catalog + basic shapes; rdf_spec_shapes.hip:
// the deeper / narrower shape families, one slice per TU so that make -j builds them side by side).
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
#pragma once
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
template <> struct AggT<RDF_F32> : AggT<RDF_F64> {   // f32 sums fold in f64 (rounded once at the end by the host); min / max stay f32
    float mn32, mx32;
    __device__ __forceinline__ void init() { AggT<RDF_F64>::init(); mn32 = mx32 = __uint_as_float(0x7FC00000u); }
    __device__ __forceinline__ void add(float v) { sum += (double)v; mn32 = fminf(mn32, v); mx32 = fmaxf(mx32, v); ++cnt; }
    // dead rows contribute the identities (NaN to fmin / fmax, -0.0 to the sum) through selects instead of exec masks
    template <int R, bool ALL, class T> __device__ __forceinline__ void add_rows(const T* v, uint32_t live) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const bool on = ALL || ((live >> i) & 1);
            const float xs = on ? (float)v[i] : __uint_as_float(0x7FC00000u);
            mn32 = fminf(mn32, xs);
            mx32 = fmaxf(mx32, xs);
            sum += (double)(on ? (float)v[i] : -0.0f);
        }
        cnt += ALL ? R : __popc(live & ((1u << R) - 1));
    }
    __device__ __forceinline__ uint64_t a() const { return d2u((double)mn32); }
    __device__ __forceinline__ uint64_t b() const { return d2u((double)mx32); }
};
// 4- and 2-byte integers: min / max stay 32 bits wide (one v_min / v_max per row instead of a 64-bit compare and two selects);
// 2-byte values also sum a wave iteration's rows in 32 bits (R <= 8 rows of 16 bits cannot overflow) before widening once.
template <class T32, class BASE, bool SUM32> struct AggNarrow : BASE {
    T32 mn32, mx32;
    __device__ __forceinline__ void init() {
        BASE::init();
        mn32 = std::is_signed<T32>::value ? (T32)INT32_MAX : (T32)~0u;
        mx32 = std::is_signed<T32>::value ? (T32)INT32_MIN : (T32)0;
    }
    __device__ __forceinline__ void add(T32 v) {
        this->sum += (uint64_t)(typename std::conditional<std::is_signed<T32>::value, int64_t, uint64_t>::type)v;
        mn32 = v < mn32 ? v : mn32;
        mx32 = v > mx32 ? v : mx32;
        ++this->cnt;
    }
    template <int R, bool ALL, class T> __device__ __forceinline__ void add_rows(const T* v, uint32_t live) {
        using Wide = typename std::conditional<std::is_signed<T32>::value, int64_t, uint64_t>::type;
        const T32 lo = std::is_signed<T32>::value ? (T32)INT32_MAX : (T32)~0u, hi = std::is_signed<T32>::value ? (T32)INT32_MIN : (T32)0;
        T32 s32 = 0;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            // dead rows contribute the identities through a 0 / ~0 mask (one bit-field extract, then v_bfi / v_and): no compare,
            // no selects
            const uint32_t m = ALL ? ~0u : 0u - ((live >> i) & 1u);
            const uint32_t xb = (uint32_t)(T32)v[i];
            const T32 xl = (T32)((xb & m) | ((uint32_t)lo & ~m)), xh = (T32)((xb & m) | ((uint32_t)hi & ~m)), xz = (T32)(xb & m);
            mn32 = xl < mn32 ? xl : mn32;
            mx32 = xh > mx32 ? xh : mx32;
            if (SUM32) s32 += xz;
            else this->sum += (uint64_t)(Wide)xz;
        }
        if (SUM32) this->sum += (uint64_t)(Wide)s32;
        this->cnt += ALL ? R : __popc(live & ((1u << R) - 1));
    }
    __device__ __forceinline__ uint64_t a() const { return (uint64_t)(typename std::conditional<std::is_signed<T32>::value, int64_t, uint64_t>::type)mn32; }
    __device__ __forceinline__ uint64_t b() const { return (uint64_t)(typename std::conditional<std::is_signed<T32>::value, int64_t, uint64_t>::type)mx32; }
};
template <> struct AggT<RDF_I32> : AggNarrow<int32_t, AggT<RDF_I64>, false> {};
template <> struct AggT<RDF_U32> : AggNarrow<uint32_t, AggT<RDF_U64>, false> {};
template <> struct AggT<RDF_I16> : AggNarrow<int32_t, AggT<RDF_I64>, true> {};
template <> struct AggT<RDF_U16> : AggNarrow<uint32_t, AggT<RDF_U64>, true> {};
template <> struct AggT<RDF_I8> : AggNarrow<int32_t, AggT<RDF_I64>, true> {};      // 16 rows of 8 bits per iteration: no overflow in 32 bits either
template <> struct AggT<RDF_U8> : AggNarrow<uint32_t, AggT<RDF_U64>, true> {};

// Lane l of a 16-byte-load wave holds RV consecutive rows (RV = 2 for 8-byte, 4 for 4-byte elements), so
// the RV per-element ballots must be interleaved into Arrow's row-ordered bitmap words.  Every lane j picks
// the bit that belongs at output position j of word h (rows 64h..64h+63 of the wave-load) and the wave
// ballots again: a handful of VALU instructions per word for the whole wave.
template <int RV>
__device__ __forceinline__ uint64_t interleave_word(const uint64_t (&b)[RV], int h, int lane) {
    if constexpr (RV == 1) return b[0];     // one row per lane: the ballot is already in row order
    uint64_t src = b[0];
#pragma unroll
    for (int e = 1; e < RV; ++e) if ((lane % RV) == e) src = b[e];
    return __ballot((src >> ((64 / RV) * h + lane / RV)) & 1);
}

// ------------------------------------------------------------------------------------------------
// the kernel

template <int BYTES> struct UIntOf;
template <> struct UIntOf<8> { using type = uint64_t; };
template <> struct UIntOf<4> { using type = uint32_t; };
template <> struct UIntOf<2> { using type = uint16_t; };
template <> struct UIntOf<1> { using type = uint8_t; };

template <class PRED, class V0, class V1, int SINK_>
struct Prog {
    using Pred = PRED; using Val0 = V0; using Val1 = V1;
    static constexpr int SINK = SINK_;
    static constexpr int NC_ = (PRED::ncols > V0::ncols ? PRED::ncols : V0::ncols) > V1::ncols
                                   ? (PRED::ncols > V0::ncols ? PRED::ncols : V0::ncols) : V1::ncols;
    static constexpr int NC = NC_ < 1 ? 1 : NC_;
    // element width of canonical column k (0: the program does not read slot k) and the program's WIDEST element — its columns
    // and, for a stored value, the output.  Columns narrower than W (a cast leaf: add(a: i64, cast(b: i32))) are read with
    // proportionally narrower vectors of the same rows.
    template <int k> static constexpr int colw() { return cmax(cmax(PRED::template colw<k>(), V0::template colw<k>()), V1::template colw<k>()); }
    static constexpr int out_width() {
        if constexpr (SINK_ == SINK_STORE && !std::is_same<V0, None>::value) return V0::dt == RDF_BOOL ? 0 : CType<V0::dt>::width;
        else return 0;
    }
    static constexpr int W = cmax(cmax(cmax(cmax(colw<0>(), colw<1>()), cmax(colw<2>(), colw<3>())), cmax(cmax(colw<4>(), colw<5>()), cmax(colw<6>(), colw<7>()))), out_width());
    static_assert(NC <= kSpecCols, "a program reads at most kSpecCols columns");
    static_assert(W == 8 || W == 4 || W == 2 || W == 1, "the widest element of a program is 8, 4, 2 or 1 bytes");
    // rows per vector slot: a 16-byte vector of the widest type — except for predicates stored as bit masks, which load 8 bytes
    // per lane: with one f64 per lane a compare's lane mask IS the Arrow bitmap word of those 64 rows, and every halving of the
    // rows per lane halves the ballot-and-interleave work that puts the bits in row order (25.9 vector instructions per row
    // with 16-byte loads, PMC; in time the two layouts measure the same, 1.40-1.50 ms per 1e9 f64 rows)
    static constexpr bool bool_store() {
        if constexpr (SINK_ == SINK_STORE && !std::is_same<V0, None>::value) return V0::dt == RDF_BOOL;
        else return false;
    }
    static constexpr int RV = (bool_store() ? 8 : 16) / W;
    // rows per lane per iteration.  A 16-byte vector of 2-byte elements is 8 rows.  Measured on the 4-byte types: 8 rows pay for
    // a 3-column aggregate (0.62 -> 0.75 of peak: half the per-iteration overhead) but cost a 3-column store (0.70 -> 0.65) and
    // 4-column programs (0.75 -> 0.73) more in registers than they save
    // Round 5: 8 rows per lane also for 3- / 4-column aggregates of 8-byte columns (config C3: 118 VGPRs, still four waves per
    // SIMD since the descriptors left the vector registers).  Same box, 14 launch shapes (blocks per CU x tile walk): 0.828-0.836 of
    // peak whatever the shape, against 0.767-0.833 with 4 rows (profiles/r05_tilewalk_*.jsonl); -DRDF_SPEC_R8_WIDE=0 builds the old rule.
#ifndef RDF_SPEC_R8_WIDE
#define RDF_SPEC_R8_WIDE 1
#endif
#ifndef RDF_SPEC_R4_ONECOL
#define RDF_SPEC_R4_ONECOL 0        // A/B: one-column aggregates of 8-byte values with 4 rows (two 16-byte vectors) per lane and iteration instead of 8
                                    // (measured, same box: 0.80 of peak at 3 resident blocks per CU, 0.855-0.859 at 4 + XCD swizzle, 0.62 at 2, against 0.859-0.862
                                    // for 8 rows at 2 blocks: profiles/r05_headline_4_rows_per_lane_ab_box22.jsonl)
#endif
    static constexpr int R = (W == 1 && !bool_store()) ? 16 : (RDF_SPEC_R4_ONECOL && W == 8 && NC == 1 && SINK_ == SINK_AGG) ? 4 : (NC <= 2 || W == 2 || (W == 4 && NC == 3 && SINK_ == SINK_AGG) || (RDF_SPEC_R8_WIDE && W == 8 && NC <= 4 && SINK_ == SINK_AGG)) ? 8 : 4;   // Int8 / UInt8: a 16-byte vector is 16 rows
    static constexpr int U = R / RV;                  // 16-byte vectors per lane per column per iteration
    // A/B (round 5): the NEXT tile's column loads are issued before the current tile is folded (aggregates over <= 32 VGPRs of
    // column data per tile: the bytes a wave has in flight no longer drop to zero while it computes)
#ifndef RDF_SPEC_PF
#define RDF_SPEC_PF 0
#endif
    static constexpr bool PF = RDF_SPEC_PF && SINK_ == SINK_AGG && NC * R * W <= 128;
    static std::string sig() { return "P:" + PRED::sig() + ";V:" + V0::sig() + ";" + V1::sig() + ";S:" + std::to_string(SINK_); }
};

// sinks over the R rows of a wave iteration (rows-at-once evaluation, rdf_expr.hip.h)
template <class A, class = void> struct HasAddRows : std::false_type {};
template <class A> struct HasAddRows<A, std::void_t<decltype(&A::mn32)>> : std::true_type {};
template <class E, int R, int r, class C, class AGG>
__device__ __forceinline__ void agg_rows(C& c, uint32_t live, AGG& g) {
    static_assert(r == 0, "all rows at once");
    typename E::T v[R];
    E::template eval_rows<R>(c, v);
    const bool all_live = __ballot((live & ((1u << R) - 1)) != (1u << R) - 1) == 0;
    if constexpr (HasAddRows<AGG>::value) {
        if (all_live) g.template add_rows<R, true>(v, live);
        else g.template add_rows<R, false>(v, live);
    } else if (all_live) {
        // every row of every lane counts (no filter, no nulls, a full tile — wave-uniform): the same folds in the same order
        // without the per-row bit test and exec mask (13 -> 4 VALU instructions per f64 row; sum(sin(x + c)) was VALU-bound)
#pragma unroll
        for (int i = 0; i < R; ++i) g.add(v[i]);
    } else {
#pragma unroll
        for (int i = 0; i < R; ++i) if ((live >> i) & 1) g.add(v[i]);
    }
}
template <class E, class = void> struct HasRowMask : std::false_type {};
template <class E> struct HasRowMask<E, typename std::enable_if<E::has_row_mask>::type> : std::true_type {};
template <class E, int R, int r, class C>
__device__ __forceinline__ void pred_rows(C& c, uint32_t& keep) {
    static_assert(r == 0, "all rows at once");
    if constexpr (HasRowMask<E>::value) keep &= E::template eval_mask<R>(c);
    else {
        bool v[R];
        E::template eval_rows<R>(c, v);
#pragma unroll
        for (int i = 0; i < R; ++i) if (!v[i]) keep &= ~(1u << i);
    }
}
template <class E, int R, int r, class C>
__device__ __forceinline__ void eval_rows(C& c, uint64_t (&out)[R]) {
    static_assert(r == 0, "all rows at once");
    typename E::T v[R];
    E::template eval_rows<R>(c, v);
#pragma unroll
    for (int i = 0; i < R; ++i) out[i] = to_bits(v[i]);
}

// Column loads of one wave iteration.  Lane l takes the RV rows of vector slot `vec + 64 u` (u < U) from EVERY column: a column of
// the program's widest type with one 16-byte load per slot, a narrower one (a cast leaf) with a load of RV of its own elements.
template <class P, int k, class V>
__device__ __forceinline__ void load_full_cols(const DevChunkCol (&col)[P::NC], const SpecArgs& a, int64_t vec, V& v) {
    if constexpr (k < P::NC) {
        if (!(k > 0 && a.alias[k] >= 0)) {
            constexpr int wk = P::template colw<k>() ? P::template colw<k>() : P::W;
            using Sk = typename UIntOf<wk>::type;
            using VecK = typename VecOf<Sk, P::RV>::type;
            const GlobalPtr<VecK> p = (GlobalPtr<VecK>)(as_global<Sk>(col[k].values) + col[k].offset) + vec;
#pragma unroll
            for (int u = 0; u < P::U; ++u) {
                const VecK t = __builtin_nontemporal_load(p + u * 64);
#pragma unroll
                for (int e = 0; e < P::RV; ++e) v[k][P::RV * u + e] = t[e];
            }
        }
        load_full_cols<P, k + 1>(col, a, vec, v);
    }
}
template <class P, int k, class C>
__device__ __forceinline__ void load_tail_cols(const DevChunkCol (&col)[P::NC], int64_t vec, C& c) {
    if constexpr (k < P::NC) {
        constexpr int wk = P::template colw<k>() ? P::template colw<k>() : P::W;
        using Sk = typename UIntOf<wk>::type;
        const GlobalPtr<Sk> p = as_global<Sk>(col[k].values) + col[k].offset;
#pragma unroll
        for (int u = 0; u < P::U; ++u)
#pragma unroll
            for (int e = 0; e < P::RV; ++e)
                c.v[k][P::RV * u + e] = ((c.inr >> (P::RV * u + e)) & 1) ? p[(int64_t)P::RV * (vec + u * 64) + e] : (Sk)0;
        load_tail_cols<P, k + 1>(col, vec, c);
    }
}

// (no waves-per-SIMD bound: the allocator settles at 4-5 waves, 88-120 VGPRs, for most programs; asking for 5 or 6 makes
// 140-260 of the 326 exact-catalog kernels spill 11-42 registers to scratch — measured with -Rpass-analysis=kernel-resource-usage)
template <class P>
__global__ __launch_bounds__(kBlock) void spec_kernel(const SpecArgs a) {
    constexpr int NC = P::NC, U = P::U, R = P::R, RV = P::RV, W = P::W;
    using S = typename UIntOf<W>::type;
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
    for (int k = 0; k < kSpecImm; ++k) c.imm[k] = a.imm[k];
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
    // (every table is read through the CONSTANT address space: as generic pointers out of the by-value argument struct the
    // first lookup became flat loads on the vector path, its results lived in VGPRs, and through the loop's phi so did every
    // later descriptor — the wave-uniform address and bitmap arithmetic of each tile then ran on the vector ALU)
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
            const ConstPtr<int64_t> ts = as_const<int64_t>(a.chunk_tile_start);
            m.ch = find_chunk_tile_inv(ts, a.nchunks, tile, a.tile_inv);
            m.base = (tile - ts[m.ch]) * per_tile;
            m.n = as_const<int64_t>(a.chunk_len)[m.ch];
#pragma unroll
            for (int k = 0; k < NC; ++k) m.col[k] = const_col(a.cols_tab, (int64_t)k * a.nchunks + m.ch);
            if (P::SINK == SINK_STORE) {
                const ConstPtr<DevOutChunk> ot = as_const<DevOutChunk>(a.outs_tab);
                m.out.values = ot[m.ch].values;
                m.out.validity = ot[m.ch].validity;
            }
        }
        return m;
    };
    // The tile walk.  The grid's S = gridDim.x * kWaves waves take the tile list row by row (S consecutive tiles per row);
    // wave p of row i works on tile i * S + (p + i * tile_rot) mod S.  tile_rot = 0 is the plain grid-stride walk, where a wave's
    // successive tiles lie S tiles apart — a power-of-two multiple of the CU count times the tile bytes for the usual grids,
    // i.e. always on the same memory channels; a rotation (a multiple of kWaves, so a block's waves stay on consecutive tiles)
    // makes the distance S + tile_rot whatever the grid.  xcd_swz: blocks are dealt to the XCDs round-robin (block b -> XCD
    // b mod 8); with it set, XCD x works on the x-th contiguous eighth of every row instead of on every eighth 16 KB piece.
    const int64_t nwv = (int64_t)gridDim.x * kWaves;
    int64_t vblock = blockIdx.x;
    if (a.xcd_swz && (gridDim.x & 7u) == 0) vblock = (int64_t)(blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    int64_t pos = vblock * kWaves + wave, rowb = 0;
    int64_t tile = pos;
    TileMeta meta = locate(tile < a.ntiles ? tile : 0);
    S nx[NC][R];
    bool have_next = false;
    while (tile < a.ntiles) {
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
            if (P::PF && have_next) {
#pragma unroll
                for (int k = 0; k < NC; ++k)
#pragma unroll
                    for (int r = 0; r < R; ++r) c.v[k][r] = nx[k][r];
            } else load_full_cols<P, 0>(col, a, wbase + lane, c.v);   // (a slot that repeats an earlier column — alias[k] >= 0, shape kernels — is not loaded again)
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
            load_tail_cols<P, 0>(col, wbase + lane, c);
        }
        {   // the loads above are in flight: locate the next tile now
            rowb += nwv;
            pos += a.tile_rot;
            if (pos >= nwv) pos -= nwv;
            tile = rowb + pos;
            if (a.nchunks == 1) meta.base = tile * per_tile;
            else if (tile < a.ntiles) meta = locate(tile);
            if constexpr (P::PF) {
                have_next = false;
                if (tile < a.ntiles && (int64_t)RV * meta.base + 64 * R <= meta.n) {
                    load_full_cols<P, 0>(meta.col, a, meta.base + lane, nx);
                    have_next = true;
                }
            }
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
                // (the vector-path variant of the window loads, rdf_set_option("vec_bitmap", 1), is gone from this kernel since
                // round 5: the compiler hoisted the arithmetic its two paths shared above the branch, onto every tile)
                if (full) load_windows_full_s<R>(col[k].validity, col[k].offset + rw, w);   // every row exists: no end-of-chunk masks
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
            // 2-byte elements (instruction-bound): every row of the tile exists and is valid (wave-uniform; the common case) ->
            // no per-row ballots, no null slots to zero, the validity words are all ones: 16 ballots and 8 selects per lane
            // saved.  Wider elements are memory-bound and measured slower with the second code path (0.68 -> 0.58, f32, 3 columns).
            constexpr bool kFastStore = V0::dt != RDF_BOOL && W == 2;
            const bool all_on = kFastStore && __ballot(vm != (1u << R) - 1) == 0;
            if (all_on) {
                if constexpr (kFastStore) {
                    using So = typename UIntOf<CType<V0::dt>::width>::type;
                    using VecO = typename VecOf<So, RV>::type;
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        VecO t;
#pragma unroll
                        for (int e = 0; e < RV; ++e) t[e] = (So)outv[RV * u + e];
                        __builtin_nontemporal_store(t, as_global_mut<VecO>(out.values) + (wbase + u * 64 + lane));
                    }
                    word_vld = ~0ull;
                }
            } else
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
                    using So = typename UIntOf<CType<V0::dt>::width>::type;     // the output's own element width (<= W)
                    using VecO = typename VecOf<So, RV>::type;
                    if (inu == (1u << RV) - 1) {
                        VecO t;
#pragma unroll
                        for (int e = 0; e < RV; ++e) t[e] = (So)x[e];
                        __builtin_nontemporal_store(t, as_global_mut<VecO>(out.values) + i);   // streaming output: do not keep it in L2 / MALL
                    } else {
#pragma unroll
                        for (int e = 0; e < RV; ++e) if ((inu >> e) & 1) as_global_mut<So>(out.values)[(int64_t)RV * i + e] = (So)x[e];
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
// catalog plumbing: one registry (defined in rdf_spec.hip), filled by every TU's registration function

typedef void (*SpecLaunch)(const SpecArgs&, int, hipStream_t);
struct SpecEntry { SpecLaunch launch; int rows_per_tile; };
std::map<std::string, SpecEntry>& spec_registry();

template <class P>
static void launch_prog(const SpecArgs& a, int grid, hipStream_t s) {
    hipLaunchKernelGGL((spec_kernel<P>), dim3(grid), dim3(kBlock), 0, s, a);
}
template <class P>
static void reg() { spec_registry()[P::sig()] = SpecEntry{&launch_prog<P>, 64 * P::R}; }

// ---- shape-specialised kernels with runtime operators (rdf_expr.hip.h *RT nodes).  A shape is written as a type:
// Lc / Lk = a column / literal leaf, Op<X, Y> = runtime arithmetic, Tr<X> = runtime sin / cos / tan.  Build<> numbers the
// slots the way the host's ShapeSigBuilder does: operator slots in pre-order (predicate first), every leaf OCCURRENCE
// its own canonical column / literal slot (the host maps two slots to the same column when a program reuses one: the
// second load is an alias of the first).  Canonical operand order (the host sets the operator's swap bit to get there):
// the deeper subtree first, a subtree before a leaf, a column before a literal.
struct Lc; struct Lk;
template <int FROM> struct Lx;          // a column of dtype FROM cast to the family's dtype (the plan builders' cast of the second operand)
template <class X, class Y> struct Op;
template <class X> struct Tr;
template <class Sh, int S, int C, int K, int DT> struct Build;
template <int S, int C, int K, int DT> struct Build<Lc, S, C, K, DT> {
    using type = Col<C, DT>;
    static constexpr int nS = S, nC = C + 1, nK = K;
};
template <int FROM, int S, int C, int K, int DT> struct Build<Lx<FROM>, S, C, K, DT> {
    using type = Cast<DT, Col<C, FROM>>;
    static constexpr int nS = S, nC = C + 1, nK = K;
};
template <int S, int C, int K, int DT> struct Build<Lk, S, C, K, DT> {
    using type = Imm<K, DT>;
    static constexpr int nS = S, nC = C, nK = K + 1;
};
template <class X, class Y, int S, int C, int K, int DT> struct Build<Op<X, Y>, S, C, K, DT> {
    using BX = Build<X, S + 1, C, K, DT>;
    using BY = Build<Y, BX::nS, BX::nC, BX::nK, DT>;
    using type = ArithRT<S, typename BX::type, typename BY::type>;
    static constexpr int nS = BY::nS, nC = BY::nC, nK = BY::nK;
};
template <class X, int S, int C, int K, int DT> struct Build<Tr<X>, S, C, K, DT> {
    using BX = Build<X, S + 1, C, K, DT>;
    using type = TrigRT<S, typename BX::type>;
    static constexpr int nS = BX::nS, nC = BX::nC, nK = BX::nK;
};

// the shape lists
template <class... Sh> struct ShapeList {};
using O1cc = Op<Lc, Lc>; using O1ck = Op<Lc, Lk>;
using O2ccc = Op<O1cc, Lc>; using O2cck = Op<O1cc, Lk>; using O2ckc = Op<O1ck, Lc>; using O2ckk = Op<O1ck, Lk>;
// level 1-2: l.l, (l.l).l   — level 3: ((l.l).l).l, (l.l).(l.l), ((l.l).l).(l.l)
using ShapesBasic = ShapeList<O1cc, O1ck, O2ccc, O2cck, O2ckc, O2ckk>;
using ShapesBasicTrig = ShapeList<Tr<Lc>, Tr<O1cc>, Tr<O1ck>>;
using ShapesDeep = ShapeList<
    Op<O2ccc, Lc>, Op<O2ccc, Lk>, Op<O2cck, Lc>, Op<O2cck, Lk>, Op<O2ckc, Lc>, Op<O2ckc, Lk>, Op<O2ckk, Lc>, Op<O2ckk, Lk>,
    Op<O1cc, O1cc>, Op<O1cc, O1ck>, Op<O1ck, O1cc>, Op<O1ck, O1ck>,
    Op<O2ccc, O1cc>, Op<O2ccc, O1ck>, Op<O2cck, O1cc>, Op<O2cck, O1ck>, Op<O2ckc, O1cc>, Op<O2ckc, O1ck>, Op<O2ckk, O1cc>, Op<O2ckk, O1ck>>;
using ShapesDeepTrig = ShapeList<Tr<O2ccc>, Tr<O2cck>, Tr<O2ckc>, Tr<O2ckk>>;

// predicates in front of a value shape: none, x CMP c, x CMP c AND|OR y CMP d (columns of dtype PDT, compared in f64)
template <int PDT> struct PredNone { using type = None; static constexpr int nS = 0, nC = 0, nK = 0; };
template <int PDT> struct Pred1 { using type = CmpRT<0, Col<0, PDT>, Imm<0, RDF_F64>>; static constexpr int nS = 1, nC = 1, nK = 1; };
template <int PDT> struct Pred2 {
    using type = LogicRT<0, CmpRT<1, Col<0, PDT>, Imm<0, RDF_F64>>, CmpRT<2, Col<1, PDT>, Imm<1, RDF_F64>>>;
    static constexpr int nS = 3, nC = 2, nK = 2;
};

template <class PB, int DT, int SINK, class Sh> static void reg_shape_one() {
    using B = Build<Sh, PB::nS, PB::nC, PB::nK, DT>;
    if constexpr (B::nC <= 4 && B::nK <= 4 && B::nS <= 8) reg<Prog<typename PB::type, typename B::type, None, SINK>>();
}
template <class PB, int DT, int SINK, class... Sh> static void reg_shape_list(ShapeList<Sh...>) { (reg_shape_one<PB, DT, SINK, Sh>(), ...); }

// DT = dtype of the value expression's columns and literals; PDT = dtype of the predicate's columns
template <int DT, int PDT, class Arith, class Trig, bool BARE> static void reg_shape_family_lists() {
    if constexpr (DT == PDT) {
        reg_shape_list<PredNone<PDT>, DT, SINK_STORE>(Arith{});
        reg_shape_list<PredNone<PDT>, DT, SINK_AGG>(Arith{});
        if constexpr (dt_float(DT)) { reg_shape_list<PredNone<PDT>, DT, SINK_STORE>(Trig{}); reg_shape_list<PredNone<PDT>, DT, SINK_AGG>(Trig{}); }
        if constexpr (BARE) reg_shape_list<PredNone<PDT>, DT, SINK_AGG>(ShapeList<Lc>{});
    }
    reg_shape_list<Pred1<PDT>, DT, SINK_AGG>(Arith{});
    reg_shape_list<Pred2<PDT>, DT, SINK_AGG>(Arith{});
    if constexpr (dt_float(DT)) { reg_shape_list<Pred1<PDT>, DT, SINK_AGG>(Trig{}); reg_shape_list<Pred2<PDT>, DT, SINK_AGG>(Trig{}); }
    if constexpr (BARE) { reg_shape_list<Pred1<PDT>, DT, SINK_AGG>(ShapeList<Lc>{}); reg_shape_list<Pred2<PDT>, DT, SINK_AGG>(ShapeList<Lc>{}); }
}
template <int DT, int PDT> static void reg_shape_family_basic() {
    reg_shape_family_lists<DT, PDT, ShapesBasic, ShapesBasicTrig, true>();
    if constexpr (DT == PDT) reg<Prog<None, typename Pred2<PDT>::type, None, SINK_STORE>>();   // x CMP c AND|OR y CMP d as a mask
}
template <int DT, int PDT> static void reg_shape_family_deep() { reg_shape_family_lists<DT, PDT, ShapesDeep, ShapesDeepTrig, false>(); }

// registration functions of the shape TUs (rdf_spec_shapes.hip, -DRDF_SHAPE_TU=n)
void spec_register_shapes1();
void spec_register_shapes2();
void spec_register_shapes3();
void spec_register_shapes4();
void spec_register_shapes5();
void spec_register_shapes6();
void spec_register_shapes7();
void spec_register_shapes8();

// a OP cast(b): the two-column calculation the reference plans for operands of different types (AddOperation ... DivideOperation,
// src/operation/scalar.rs:47-72: the second operand is cast to the first one's type), the cast on its own (CastOperation /
// Function::Cast, src/evaluation.rs:296-315), and sin / cos / tan of a cast column (SinOperation casts integers to Float64 first,
// src/operation/scalar.rs:256-294).  Columns keep their own widths in memory (rdf_spec_kernel: per-column load widths).
template <int DT, int FROM> static void reg_cast_pair() {
    if constexpr (DT != FROM) {
        reg_shape_list<PredNone<DT>, DT, SINK_STORE>(ShapeList<Op<Lc, Lx<FROM>>>{});
        reg_shape_list<PredNone<DT>, DT, SINK_AGG>(ShapeList<Op<Lc, Lx<FROM>>>{});
        reg<Prog<None, Cast<DT, Col<0, FROM>>, None, SINK_STORE>>();
        if constexpr (dt_float(DT)) {
            reg_shape_list<PredNone<DT>, DT, SINK_STORE>(ShapeList<Tr<Lx<FROM>>>{});
            reg_shape_list<PredNone<DT>, DT, SINK_AGG>(ShapeList<Tr<Lx<FROM>>>{});
        }
    }
}
template <int DT> static void reg_cast_family() {
    reg_cast_pair<DT, RDF_F64>(); reg_cast_pair<DT, RDF_I64>(); reg_cast_pair<DT, RDF_U64>(); reg_cast_pair<DT, RDF_F32>();
    reg_cast_pair<DT, RDF_I32>(); reg_cast_pair<DT, RDF_U32>(); reg_cast_pair<DT, RDF_I16>(); reg_cast_pair<DT, RDF_U16>();
    // Int8 / UInt8 columns reach the path through casts and filters only (Evaluate::calculate rejects their arithmetic,
    // src/evaluation.rs:239; Function::Cast accepts them, :296-315): a OP cast(b: i8), cast(b: i8), trig(cast(b: i8))
    reg_cast_pair<DT, RDF_I8>(); reg_cast_pair<DT, RDF_U8>();
}
// the 1-byte types on their own: column aggregates, x CMP c -> mask / -> aggregates of x or of a column of type V, narrowing casts
template <int B> static void reg_byte_family() {
    using X = Col<0, B>;
    using K = Imm<0, RDF_F64>;
    reg_shape_list<PredNone<B>, B, SINK_AGG>(ShapeList<Lc>{});             // aggregates of the column
    reg<Prog<None, Cast<RDF_F64, X>, None, SINK_AGG>>();                   // avg
    reg_shape_list<Pred1<B>, B, SINK_AGG>(ShapeList<Lc>{});                // filter(x CMP c) -> aggregates of a 1-byte column
    reg_shape_list<Pred2<B>, B, SINK_AGG>(ShapeList<Lc>{});
    reg_shape_list<Pred1<B>, RDF_F64, SINK_AGG>(ShapeList<Lc>{});          // ... of an f64 / i64 measure (a flag or code column filters a fact table)
    reg_shape_list<Pred1<B>, RDF_I64, SINK_AGG>(ShapeList<Lc>{});
    reg<Prog<None, typename Pred2<B>::type, None, SINK_STORE>>();          // masks
    reg<Prog<None, Bin<RDF_OP_GT, X, K>, None, SINK_STORE>>(); reg<Prog<None, Bin<RDF_OP_GE, X, K>, None, SINK_STORE>>();
    reg<Prog<None, Bin<RDF_OP_EQ, X, K>, None, SINK_STORE>>(); reg<Prog<None, Bin<RDF_OP_NE, X, K>, None, SINK_STORE>>();
    reg<Prog<None, Bin<RDF_OP_LT, X, K>, None, SINK_STORE>>(); reg<Prog<None, Bin<RDF_OP_LE, X, K>, None, SINK_STORE>>();
    reg<Prog<None, Cast<B, Col<0, RDF_I64>>, None, SINK_STORE>>();         // narrowing casts into the 1-byte types
    reg<Prog<None, Cast<B, Col<0, RDF_I32>>, None, SINK_STORE>>();
    reg<Prog<None, Cast<B, Col<0, RDF_I16>>, None, SINK_STORE>>();
    reg<Prog<None, Cast<B, Col<0, RDF_F64>>, None, SINK_STORE>>();
}

}  // namespace rdfk
