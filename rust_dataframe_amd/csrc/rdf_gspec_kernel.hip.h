// rdf_gspec_kernel.hip.h — the specialised kernel of the fused grouped sink (rdf_group_pipeline): the template shared by the
// ahead-of-time catalog (rdf_gspec.hip) and the instantiations compiled at run time for grouped programs the catalog does not
// hold (rdf_jit.cpp).  See rdf_gspec.hip for what the kernel does.
#pragma once
#include <initializer_list>
#include <map>
#include <string>
#include <tuple>
#include <utility>

#include "rdf_expr.hip.h"

namespace rdfk {

constexpr int cmaxl(std::initializer_list<int> l) {
    int m = 0;
    for (int x : l) m = x > m ? x : m;
    return m;
}

template <int NC, int R>
struct GCtx {
    static constexpr int rows = R;
    uint64_t v[NC][R];   // raw elements zero-extended (typed views re-narrow them)
    uint32_t valid[NC];
    uint64_t imm[kGSpecImm];
    uint32_t inr;
    uint32_t err;
};

template <int G_, class PRED, class GID, class... Vs>
struct GProg {
    using Pred = PRED; using Gid = GID;
    using Vals = std::tuple<Vs...>;
    static constexpr int G = G_;
    static constexpr int NV = (int)sizeof...(Vs);
    static constexpr int NC_ = cmaxl({PRED::ncols, GID::ncols, Vs::ncols...});
    static constexpr int NC = NC_ < 1 ? 1 : NC_;
    static constexpr int R = 4;   // rows per lane per tile: 2 wave-loads of 2 consecutive rows
    static_assert(NV >= 1 && NV <= kMaxGroupValues && NC <= kGSpecCols, "grouped program shape");
    template <int k> static constexpr int colw() { return cmaxl({PRED::template colw<k>(), GID::template colw<k>(), Vs::template colw<k>()...}); }
    template <int v> static constexpr bool isf() { return dt_float(std::tuple_element_t<v, Vals>::dt); }
    static constexpr uint32_t fmask() { uint32_t m = 0; int i = 0; ((m |= (uint32_t)dt_float(Vs::dt) << i, ++i), ...); return m; }
    static std::string sig() {
        std::string s = "G" + std::to_string(G) + ";P:" + PRED::sig() + ";K:" + GID::sig() + ";V:";
        ((s += Vs::sig() + ";"), ...);
        return s;
    }
};

// value of row r widened to its 64-bit accumulator class (f64 bits, or sign-/zero-extended integer)
template <class V, int r, class C>
__device__ __forceinline__ uint64_t val_bits(C& c) {
    const auto x = V::template eval<r>(c);
    if constexpr (dt_float(V::dt)) return d2u((double)x);
    else if constexpr (V::dt == RDF_BOOL) return (uint64_t)x;
    else if constexpr (dt_signed(V::dt)) return (uint64_t)(int64_t)x;
    else return (uint64_t)x;
}
template <bool F> __device__ __forceinline__ uint64_t acc_add(uint64_t a, uint64_t b) {
    if constexpr (F) return d2u(u2d(a) + u2d(b)); else return a + b;
}

template <class P, int k>
__device__ __forceinline__ void load_col_full(uint64_t (&v)[P::NC][P::R], const DevChunkCol& col, int64_t rw, int lane) {
    constexpr int w = P::template colw<k>();
    constexpr int U = P::R / 2;
    if constexpr (w == 0) {
#pragma unroll
        for (int r = 0; r < P::R; ++r) v[k][r] = 0;
    } else {
        using S = typename std::conditional<w == 8, uint64_t, typename std::conditional<w == 4, uint32_t, typename std::conditional<w == 2, uint16_t, uint8_t>::type>::type>::type;
        using V2 = typename VecOf<S, 2>::type;
        const GlobalPtr<V2> p = (GlobalPtr<V2>)(as_global<S>(col.values) + col.offset + rw) + lane;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const V2 t = __builtin_nontemporal_load(p + u * 64);
            v[k][2 * u] = (uint64_t)t[0];
            v[k][2 * u + 1] = (uint64_t)t[1];
        }
    }
}
template <class P, int k>
__device__ __forceinline__ void load_col_tail(uint64_t (&v)[P::NC][P::R], const DevChunkCol& col, int64_t rw, int lane, uint32_t inr) {
    constexpr int w = P::template colw<k>();
    constexpr int U = P::R / 2;
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int r = 2 * u + e;
            const int64_t row = col.offset + rw + 128 * u + 2 * lane + e;
            uint64_t x = 0;
            if constexpr (w != 0) {
                if ((inr >> r) & 1) {
                    if constexpr (w == 8) x = as_global<uint64_t>(col.values)[row];
                    else if constexpr (w == 4) x = as_global<uint32_t>(col.values)[row];
                    else if constexpr (w == 2) x = as_global<uint16_t>(col.values)[row];
                    else x = as_global<uint8_t>(col.values)[row];
                }
            }
            v[k][r] = x;
        }
}
template <class P, size_t... K>
__device__ __forceinline__ void load_all(uint64_t (&v)[P::NC][P::R], const DevChunkCol (&col)[P::NC], int64_t rw, int lane, uint32_t inr, bool full, std::index_sequence<K...>) {
    if (full) (load_col_full<P, (int)K>(v, col[K], rw, lane), ...);
    else (load_col_tail<P, (int)K>(v, col[K], rw, lane, inr), ...);
}

template <class P, int r, class C, size_t... I>
__device__ __forceinline__ void row_values(C& c, uint64_t (&vb)[P::NV], uint32_t (&vm)[P::NV], std::index_sequence<I...>) {
    ((vb[I] = val_bits<std::tuple_element_t<I, typename P::Vals>, r>(c)), ...);
    ((vm[I] = (std::tuple_element_t<I, typename P::Vals>::vmask(c) >> r) & 1u), ...);
}
template <class P, int g, size_t... I>
__device__ __forceinline__ void add_group(uint64_t (&sum)[P::G][P::NV], const uint64_t (&vb)[P::NV], std::index_sequence<I...>) {
    ((sum[g][I] = acc_add<P::template isf<(int)I>()>(sum[g][I], vb[I])), ...);
}
// One EXEC-masked region per group.  The empty asm with the group number keeps the G look-alike regions
// distinct: merged, they become `sum[g][v] += ...` with a dynamic index and the accumulators move to scratch.
template <class P, int GG>
__device__ __forceinline__ void add_if_group(uint64_t (&sum)[P::G][P::NV], uint32_t (&rows)[P::G], const uint64_t (&vb)[P::NV], uint32_t g) {
    if (g == (uint32_t)GG) {
        ++rows[GG];
        add_group<P, GG>(sum, vb, std::make_index_sequence<P::NV>());
        asm volatile("; group %0" ::"n"(GG));   // last in the region: common-tail sinking works backwards from here
    }
}
template <class P, size_t... GG>
__device__ __forceinline__ void add_groups(uint64_t (&sum)[P::G][P::NV], uint32_t (&rows)[P::G], const uint64_t (&vb)[P::NV], uint32_t g, std::index_sequence<GG...>) {
    (add_if_group<P, (int)GG>(sum, rows, vb, g), ...);
}

template <class P, int r, class C>
__device__ __forceinline__ void do_row(C& c, uint32_t keep, uint64_t (&sum)[P::G][P::NV], uint32_t (&rows)[P::G], uint64_t* gtab, int ngroups) {
    constexpr int NV = P::NV;
    using Gid = typename P::Gid;
    if (!((keep >> r) & 1)) return;
    uint64_t vb[NV];
    uint32_t vm[NV];
    row_values<P, r>(c, vb, vm, std::make_index_sequence<NV>());
    const bool gvalid = (Gid::vmask(c) >> r) & 1;
    const auto gi = Gid::template eval<r>(c);
    uint64_t g;
    if constexpr (Gid::dt == RDF_BOOL) g = (uint64_t)gi;
    else if constexpr (dt_signed(Gid::dt)) g = (uint64_t)(int64_t)gi;
    else g = (uint64_t)gi;
    uint32_t allv = 1;
#pragma unroll
    for (int v = 0; v < NV; ++v) allv &= vm[v];
    if (gvalid && g >= (uint64_t)ngroups) { c.err |= 2u; return; }
    if (!gvalid || !allv) {  // rare: the NULL group / a NULL value -> LDS atomics on the block's table
        const int S = ngroups + 1;
        const int slot = gvalid ? (int)g : ngroups;
        atomicAdd((unsigned long long*)&gtab[2 * NV * S + slot], 1ull);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (vm[v]) {
                if ((P::fmask() >> v) & 1) unsafeAtomicAdd((double*)&gtab[v * S + slot], u2d(vb[v]));
                else atomicAdd((unsigned long long*)&gtab[v * S + slot], (unsigned long long)vb[v]);
            } else atomicAdd((unsigned long long*)&gtab[(NV + v) * S + slot], 1ull);
        }
        return;
    }
    add_groups<P>(sum, rows, vb, (uint32_t)g, std::make_index_sequence<P::G>());
}
template <class P, class C, size_t... RR>
__device__ __forceinline__ void do_rows(C& c, uint32_t keep, uint64_t (&sum)[P::G][P::NV], uint32_t (&rows)[P::G], uint64_t* gtab, int ngroups, std::index_sequence<RR...>) {
    (do_row<P, (int)RR>(c, keep, sum, rows, gtab, ngroups), ...);
}
template <class E, int R, int r, class C>
__device__ __forceinline__ void gpred_rows(C& c, uint32_t& keep) {
    if constexpr (r < R) {
        if (!E::template eval<r>(c)) keep &= ~(1u << r);
        gpred_rows<E, R, r + 1>(c, keep);
    }
}

#ifndef RDF_GSPEC_PF
#define RDF_GSPEC_PF 1
#endif
#ifndef RDF_GSPEC_WAVES
#define RDF_GSPEC_WAVES 2
#endif
template <class P>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(RDF_GSPEC_WAVES, 8))) void gspec_kernel(const GSpecArgs a) {
    constexpr int NC = P::NC, R = P::R, U = R / 2, G = P::G, NV = P::NV;
    using Pred = typename P::Pred;
    constexpr bool has_pred = !std::is_same<Pred, None>::value;
    extern __shared__ __attribute__((aligned(16))) uint64_t gtab[];
    __shared__ uint64_t stage[kBlock / 64][G * (NV + 1)];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = wave_id();
    const int S = a.ngroups + 1, gwords = group_words(a.ngroups, NV);
    for (int i = tid; i < gwords; i += kBlock) gtab[i] = 0;
    __syncthreads();

    GCtx<NC, R> c;
    c.err = 0;
#pragma unroll
    for (int k = 0; k < kGSpecImm; ++k) c.imm[k] = a.imm[k];
    uint64_t sum[G][NV];
    uint32_t rows[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        rows[g] = 0;
#pragma unroll
        for (int v = 0; v < NV; ++v) sum[g][v] = 0;   // +0.0 and integer 0 share the bit pattern
    }

    // Where a tile lives: every table through the constant address space (scalar loads, results in SGPRs — as generic pointers
    // out of the argument struct they were flat loads whose results, and everything computed from them, lived in VGPRs).
    struct TileMeta { int64_t base, n; DevChunkCol col[NC]; };
    auto locate = [&](int64_t tile) -> TileMeta {
        TileMeta m;
        if (a.nchunks == 1) {
            m.base = tile * kEvalTile;
            m.n = a.n;
#pragma unroll
            for (int k = 0; k < NC; ++k) m.col[k] = a.cols[k];
        } else {
            const ConstPtr<int64_t> ts = as_const<int64_t>(a.chunk_tile_start);
            const int64_t ch = find_chunk_tile(ts, a.nchunks, tile);
            m.base = (tile - ts[ch]) * kEvalTile;
            m.n = as_const<int64_t>(a.chunk_len)[ch];
#pragma unroll
            for (int k = 0; k < NC; ++k) m.col[k] = const_col(a.cols_tab, (int64_t)a.col_map[k] * a.nchunks + ch);
        }
        return m;
    };
    // The kernel holds G * NV accumulators per lane, so two waves per SIMD is all that fits: while a wave folds a tile nothing of
    // its own would be in flight.  The NEXT tile's column loads are therefore issued before the current tile is folded (when that
    // tile is a full one: the common case) and are waited for at the top of the next iteration.
    uint64_t nx[NC][R];
    bool have_next = false;
    // tile walk (block-wide tiles): row i of gridDim.x tiles, block p takes tile i * gridDim.x + (p + i * tile_rot) mod gridDim.x
    int64_t pos = blockIdx.x, rowb = 0;
    if (a.xcd_swz && (gridDim.x & 7u) == 0) pos = (int64_t)(blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    int64_t tile = pos;
    TileMeta meta = locate(tile < a.ntiles ? tile : 0);
    while (tile < a.ntiles) {
        const int64_t n = meta.n;
        DevChunkCol col[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) col[k] = meta.col[k];
        const int64_t rw = meta.base + (int64_t)wave * (64 * R);   // first row of this wave
        const bool full = rw + 64 * R <= n;
        if (full) {
            c.inr = (1u << R) - 1;
            if (have_next) {
#pragma unroll
                for (int k = 0; k < NC; ++k)
#pragma unroll
                    for (int r = 0; r < R; ++r) c.v[k][r] = nx[k][r];
            } else load_all<P>(c.v, col, rw, lane, c.inr, true, std::make_index_sequence<NC>());
        } else {
            c.inr = 0;
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int e = 0; e < 2; ++e) c.inr |= (uint32_t)(rw + 128 * u + 2 * lane + e < n) << (2 * u + e);
            load_all<P>(c.v, col, rw, lane, c.inr, false, std::make_index_sequence<NC>());
        }
        rowb += gridDim.x;
        pos += a.tile_rot;
        if (pos >= (int64_t)gridDim.x) pos -= gridDim.x;
        tile = rowb + pos;
        have_next = false;
        if (tile < a.ntiles) {
            meta = locate(tile);
            const int64_t nrw = meta.base + (int64_t)wave * (64 * R);
            if (RDF_GSPEC_PF && nrw + 64 * R <= meta.n) {
                load_all<P>(nx, meta.col, nrw, lane, (1u << R) - 1, true, std::make_index_sequence<NC>());
                have_next = true;
            }
        }
        // validity: R windows of 64 rows per column; lane l's 2 bits of load u sit in window 2u + (2l >> 6) at bit (2l) & 63
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            c.valid[k] = c.inr;
            if (col[k].validity) {
                uint64_t w[R];
                if (full) load_windows_full_s<R>(col[k].validity, col[k].offset + rw, w);
                else load_windows_s<R>(col[k].validity, col[k].offset + rw, n - rw, w);
                uint32_t m = 0;
                const int sh = (2 * lane) & 63, wsel = (2 * lane) >> 6;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    // (bitwise select: a ?: / if over two array reads is folded into a dynamic index -> scratch)
                    const uint64_t ww = w[2 * u] ^ ((w[2 * u] ^ w[2 * u + 1]) & ((uint64_t)0 - (uint64_t)wsel));
                    m |= ((uint32_t)(ww >> sh) & 3u) << (2 * u);
                }
                c.valid[k] = m & c.inr;
            }
        }
        uint32_t keep = c.inr;
        if constexpr (has_pred) {
            keep &= Pred::vmask(c);
            gpred_rows<Pred, R, 0>(c, keep);
        }
        do_rows<P>(c, keep, sum, rows, gtab, a.ngroups, std::make_index_sequence<R>());
    }
    if (c.err) atomicOr(a.flags, c.err);

    // block fold: wave butterflies (fixed order), then the 4 waves in wave order, on top of the LDS table
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            uint64_t x = sum[g][v];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                const uint64_t y = shfl_xor64(x, m);
                x = ((P::fmask() >> v) & 1) ? d2u(u2d(x) + u2d(y)) : x + y;
            }
            if (lane == 0) stage[wave][g * NV + v] = x;
        }
        uint32_t rr = rows[g];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) rr += (uint32_t)__shfl_xor((int)rr, m);
        if (lane == 0) stage[wave][G * NV + g] = rr;
    }
    __syncthreads();
    if (tid < G * NV) {
        const int g = tid / NV, v = tid % NV;
        if (g < a.ngroups) {
            const bool f = (P::fmask() >> v) & 1;
            uint64_t x = gtab[v * S + g];
            for (int w = 0; w < kBlock / 64; ++w) x = f ? d2u(u2d(x) + u2d(stage[w][tid])) : x + stage[w][tid];
            gtab[v * S + g] = x;
        }
    } else if (tid < G * NV + G) {
        const int g = tid - G * NV;
        if (g < a.ngroups) {
            uint64_t x = gtab[2 * NV * S + g];
            for (int w = 0; w < kBlock / 64; ++w) x += stage[w][tid];
            gtab[2 * NV * S + g] = x;
        }
    }
    __syncthreads();
    for (int w = tid; w < gwords; w += kBlock) a.group_partials[(size_t)blockIdx.x * gwords + w] = gtab[w];
}

}  // namespace rdfk
