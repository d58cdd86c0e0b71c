// rdf_bfilter.hip — order-preserving compaction of LONG batches in one pass, block tiles held in registers
// (DataFrame::filter, src/dataframe.rs:178-189, = BooleanFilter::eval_to_array, src/expression.rs:766-861, + Column::filter on
// every column, src/table.rs:97-107,213-215; and Column::filter with a mask that is given).
//
// Why a second kernel next to rdf_filter.hip: there a tile is one WAVE's 1024 rows, which is right for the readers' 1024-row
// RecordBatches (a batch is a tile, nothing to add up) — but a batch of 1e9 rows is a million tiles whose offsets depend on
// each other, and every hop of that dependency is a memory round trip during which the wave moves no data (round 5: 3.8 ms per
// 1e9 rows against 2.5 ms for the same rows in 1024-row batches).  Here (tools/ubench_compact.hip is the probe this design was
// measured with before it was built):
//
//   * a tile is a BLOCK's rows: 8 waves x G x 64 sixteen-byte vectors per column (8192 rows of an 8-byte column at G = 8) —
//     8 times fewer participants in the prefix, one count exchange in LDS per tile;
//   * the rows stay in REGISTERS (G vectors per lane), the kept ones are staged in LDS at their rank and leave as aligned
//     16-byte stores; the predicate column of tile n + 1 is in flight while tile n is staged and stored;
//   * the prefix is not looked back for: every tile publishes its row count, ONE wave of the grid (block 0) turns counts into
//     prefixes in tile order with DPP scans, a tile polls its own 8-byte word.  Two hops instead of a walk over the window of
//     tiles in flight — and the count of tile n + 1 goes out a whole iteration BEFORE its prefix is asked for (while tile n
//     is being staged and stored), so the hops are not waited for.
//
// Measured by the probe, 1e9 f64 rows, x > 0.5, one batch: 2.02 ms (0.74 of the 8 TB/s peak on 12 GB of algorithmic traffic) with
// the prefix found this way, against 2.18 ms on the same box with the offsets handed in from a scan computed beforehand.
#include "rdf_common.hip.h"
#include "rdf_lanewin.hip.h"

namespace rdfk {

namespace {

constexpr int kBfWaves = 8, kBfBlock = kBfWaves * 64;
constexpr unsigned long long kBfCount = 1ull << 62, kBfPrefix = 2ull << 62, kBfValue = (1ull << 62) - 1;

__device__ __forceinline__ unsigned long long bf_ld(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void bf_st(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int mbcnt64(uint64_t m, int init) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, init)); }

// inclusive scan over the 64 lanes: four row shifts inside the rows of 16, two row broadcasts across them (gfx9 DPP)
__device__ __forceinline__ int wave_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return v;
}

// The scanner: tile_state[t] goes  0 -> kBfCount | rows (written by the tile's block) -> kBfPrefix | rows of the FRAME in front of
// tile t (written here).  Counts are taken as far as they have been published without a gap, up to 512 tiles per round trip.
// (every wait of this kernel gives up after kBfWaitSeconds without progress and raises abort_flag: the other waiters then leave too)
constexpr unsigned long long kBfWaitTicks = (unsigned long long)kBfWaitSeconds * 100000000ull;      // wall_clock64: 100 MHz
__device__ __forceinline__ bool bf_aborted(unsigned int* flag) { return __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0; }
__device__ __forceinline__ void bf_scanner(unsigned long long* state, int64_t ntiles, unsigned int* abort_flag) {
    const int lane = threadIdx.x & 63;
    constexpr int K = 8;
    int64_t cur = 0;
    unsigned long long running = 0;
    unsigned long long idle_since = 0;
    while (cur < ntiles) {
        unsigned long long w[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int64_t idx = cur + k * 64 + lane;
            w[k] = idx < ntiles ? bf_ld(state + idx) : 0;
        }
        int64_t base = cur;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint64_t ready = __ballot((w[k] >> 62) != 0);
            const int f = ready == ~0ull ? 64 : __builtin_ctzll(~ready);
            if (f == 0) break;
            const int val = lane < f ? (int)(w[k] & kBfValue) : 0;
            const int inc = wave_incl_scan(val);
            if (lane < f) bf_st(state + base + lane, kBfPrefix | (running + (unsigned long long)(inc - val)));
            running += (unsigned long long)(unsigned int)__builtin_amdgcn_readlane(inc, 63);
            base += f;
            if (f < 64) break;
        }
        if (base == cur) {
            __builtin_amdgcn_s_sleep(1);
            const unsigned long long now = wall_clock64();
            if (idle_since == 0) idle_since = now;
            else if (now - idle_since > kBfWaitTicks || bf_aborted(abort_flag)) {       // (wave-uniform: every lane reads the same clock and flag)
                if (lane == 0) __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
        } else idle_since = 0;
        cur = base;
    }
}

// E consecutive bits of a row-order bitmap window kept one word per lane (lane i = the wave's rows [64 i, 64 i + 64)): the bits of
// rows g * 64 E + lane * E + [0, E) — the rows the lane holds in vector g of a column
template <int E>
__device__ __forceinline__ uint32_t lane_bits(uint64_t win, int g) {
    const int lane = threadIdx.x & 63;
    uint64_t w = rl64(win, g * E);
#pragma unroll
    for (int j = 1; j < E; ++j) { const uint64_t wj = rl64(win, g * E + j); if (((lane * E) >> 6) == j) w = wj; }
    return (uint32_t)(w >> ((lane * E) & 63)) & ((1u << E) - 1);
}

template <typename T, int G, typename S, class Cmp>
__device__ __forceinline__ void bf_cmp_fields(const typename Vec16<T>::type (&x)[G], double lit, uint32_t (&kf)[G], Cmp cmp) {
    constexpr int E = 16 / (int)sizeof(T);
#pragma unroll
    for (int g = 0; g < G; ++g) {
        uint32_t f = 0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const T raw = x[g][e];
            const S v = __builtin_bit_cast(S, raw);
            f |= (cmp((double)v, lit) ? 1u : 0u) << e;
        }
        kf[g] = f;
    }
}
template <typename T, int G, typename S>
__device__ __forceinline__ void bf_op_fields(const typename Vec16<T>::type (&x)[G], int op, double lit, uint32_t (&kf)[G]) {
    switch (op) {      // wave-uniform: one scalar branch per tile, not one per element
        case RDF_OP_GT: bf_cmp_fields<T, G, S>(x, lit, kf, [](double a, double c) { return a > c; }); break;
        case RDF_OP_GE: bf_cmp_fields<T, G, S>(x, lit, kf, [](double a, double c) { return a >= c; }); break;
        case RDF_OP_EQ: bf_cmp_fields<T, G, S>(x, lit, kf, [](double a, double c) { return a == c; }); break;
        case RDF_OP_NE: bf_cmp_fields<T, G, S>(x, lit, kf, [](double a, double c) { return a != c; }); break;
        case RDF_OP_LT: bf_cmp_fields<T, G, S>(x, lit, kf, [](double a, double c) { return a < c; }); break;
        default: bf_cmp_fields<T, G, S>(x, lit, kf, [](double a, double c) { return a <= c; }); break;
    }
}
// `column CMP literal`, both sides as f64 (src/expression.rs:844-845), on the registers of one tile: E result bits per vector.
// Integer columns never convert: x -> (double)x is monotone, so `(double)x CMP literal` holds on an interval of x, which the host
// has worked out (bf_term, rdf_capi_frame.inc) — two integer compares per element instead of an emulated i64 -> f64 conversion
// (16 of them unrolled were what pushed this kernel past its 128 registers).
template <typename T, int G>
__device__ __forceinline__ void bf_term_fields(const BfTerm& tm, const typename Vec16<T>::type (&x)[G], uint32_t (&kf)[G]) {
    constexpr int E = 16 / (int)sizeof(T);
    if (tm.kind == 1) {
        const T lo = (T)tm.lo, hi = (T)tm.hi, bias = (T)tm.bias;      // keep  <=>  lo <= (x ^ bias) <= hi  (unsigned), then inverted for `!=`
        const uint32_t inv = tm.inv ? (1u << E) - 1 : 0;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            uint32_t f = 0;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const T u = x[g][e] ^ bias;
                f |= (u >= lo && u <= hi ? 1u : 0u) << e;
            }
            kf[g] = f ^ inv;
        }
    } else if constexpr (sizeof(T) == 8) bf_op_fields<T, G, double>(x, tm.op, tm.lit, kf);
    else bf_op_fields<T, G, float>(x, tm.op, tm.lit, kf);
}

// bit i of x -> bit 2 i (E = 2) / bit 4 i (E = 4): the keep bits of a lane's E consecutive rows are E ballots apart and E bits together
__device__ __forceinline__ uint64_t bf_spread2(uint64_t x) {       // 32 bits in
    x &= 0xFFFFFFFFull;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull; x = (x | (x << 8)) & 0x00FF00FF00FF00FFull; x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2)) & 0x3333333333333333ull; x = (x | (x << 1)) & 0x5555555555555555ull;
    return x;
}
__device__ __forceinline__ uint64_t bf_spread4(uint64_t x) {       // 16 bits in
    x &= 0xFFFFull;
    x = (x | (x << 24)) & 0x000000FF000000FFull; x = (x | (x << 12)) & 0x000F000F000F000Full; x = (x | (x << 6)) & 0x0303030303030303ull;
    x = (x | (x << 3)) & 0x1111111111111111ull;
    return x;
}

struct BfTile { int64_t tile, c, r0, clen, first, rw; bool valid; };      // rw: this wave's first row of the batch; valid: the batch exists (SHORT: a tile's last batches may lie past the frame's end)

}  // namespace

// T: the columns' element as raw bits (uint64_t / uint32_t: all columns of a launch are equally wide); G: 16-byte vectors per lane
// and column; MULTI: more than one column (a third register set carries the columns that follow the first); NULLS: some column,
// or the mask, has a validity bitmap; BYMASK: Column::filter with the mask given, else the predicate is evaluated here.
// SHORT: no batch is longer than a tile (the readers' 1024-row RecordBatches, chunks of a few thousand rows).  A batch then takes
// S = a.short_waves (1, 2, 4 or 8: the frame's longest batch decides) consecutive waves of ONE block, a tile is 8 / S consecutive
// batches, and a batch's kept rows start its own output: the waves of a batch add their counts up in LDS and that is the whole
// prefix — no scanner block, no tile states, nothing to wait for.  Everything else (registers, staging at the ranks, aligned
// 16-byte stores, the next tile counted under the current tile's stores) is the long-batch kernel's.
// MODE 2 (OWNED): batches longer than a tile but many of them and none a large share of the frame (65 536-row batches of a 1e9-row
// frame): a block draws whole BATCHES by ticket and walks a batch's tiles in order, so the rows in front of a tile are the block's
// own running sum — again no scanner, no tile states, nothing to wait for (the long form pays two round trips per tile for them).
template <typename T, int G, bool MULTI, bool NULLS, bool BYMASK, int MODE>
__global__ __launch_bounds__(kBfBlock, 4) void bfilter_kernel(const BFilterArgs a) {
    constexpr bool SHORT = MODE == 1, OWNED = MODE == 2;
    constexpr int E = 16 / (int)sizeof(T), WR = G * 64 * E, TR = kBfWaves * WR, NWW = G * E;
    static_assert(NWW <= 32, "one keep-word per lane, two words fetched per lane");
    using V = typename Vec16<T>::type;
    __shared__ __attribute__((aligned(16))) T stage[TR + E];
    __shared__ uint8_t vstage[NULLS ? TR : 1];
    __shared__ int wcnt[2][kBfWaves];
    __shared__ int nullacc[kMaxFilterCols];
    __shared__ int64_t sh_base, sh_tile[2];
    const FilterWArgs& fa = a.w;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if constexpr (MODE == 0) { if (blockIdx.x == 0) { if (wave == 0 && !a.stall_test) bf_scanner(a.tile_state, fa.t.ntiles, a.abort_flag); return; } }
    // the waves that share a batch: all eight of the block, or (SHORT) S of them — group wg, wave wj of its group
    const int sshift = SHORT ? a.short_shift : 3, S = 1 << sshift;
    const int wg = SHORT ? wave >> sshift : 0, wj = SHORT ? wave & (S - 1) : wave;
    const int gsz = S * 64, gt = wj * 64 + lane, gbase = wg * S * WR;         // threads of the group, this thread's index in it, the group's part of the staging buffer
    const int64_t ntiles = fa.t.ntiles;
    const bool one = fa.t.nchunks == 1;
    const int ncols = fa.ncols;
    constexpr bool by_mask = BYMASK;       // the kept rows are given by a mask (a.nterms == 0)
    const int pc0 = by_mask ? 0 : a.bterm[0].col, pc1 = a.nterms > 1 ? a.bterm[1].col : pc0;
    if (tid < kMaxFilterCols) nullacc[tid] = 0;

    // tiles are handed out in order by `nclass` ticket counters (a single one would serialise every draw of the grid: 23 ns each);
    // block b draws from counter (b - 1) mod nclass, which numbers the tiles  nclass * k + its own index
    const int nclass = a.nclass, ctr = (int)((blockIdx.x - (MODE == 0 ? 1 : 0)) % (unsigned)nclass);
    // (OWNED, thread 0 only: the tiles of the batch drawn last that have not been handed out yet)
    int64_t own_next = 0, own_end = 0;
    auto draw = [&]() __attribute__((always_inline)) -> int64_t {
        if constexpr (OWNED) {
            while (own_next >= own_end) {            // the next batch (empty ones have no tiles)
                const int64_t c = (int64_t)atomicAdd(a.ticket + ctr * 32, 1u) * nclass + ctr;
                if (c >= fa.t.nchunks) return ntiles;
                own_next = __builtin_nontemporal_load(as_global<int64_t>(fa.t.chunk_tile_start) + c);
                own_end = __builtin_nontemporal_load(as_global<int64_t>(fa.t.chunk_tile_start) + c + 1);
            }
            return own_next++;
        }
        return (int64_t)atomicAdd(a.ticket + ctr * 32, 1u) * nclass + ctr;
    };
    auto locate = [&](int64_t tile) __attribute__((always_inline)) -> BfTile {
        BfTile t;
        t.tile = tile;
        t.valid = true;
        if constexpr (SHORT) {
            const int64_t c = (tile << (3 - sshift)) + wg;
            t.valid = c < fa.t.nchunks;
            t.c = t.valid ? c : fa.t.nchunks - 1;
            t.first = tile; t.r0 = 0;
            t.clen = !t.valid ? 0 : one ? fa.len0 : as_const<int64_t>(fa.t.chunk_len)[t.c];
            t.rw = (int64_t)wj * WR;
            return t;
        }
        if (one) { t.c = 0; t.first = 0; t.r0 = tile * TR; t.clen = fa.len0; }
        else {
            const ConstPtr<int64_t> start = as_const<int64_t>(fa.t.chunk_tile_start);
            t.c = find_chunk_tile_inv(start, fa.t.nchunks, tile, fa.tile_inv);
            t.first = start[t.c];
            t.r0 = (tile - t.first) * TR;
            t.clen = as_const<int64_t>(fa.t.chunk_len)[t.c];
        }
        t.rw = t.r0 + (int64_t)wave * WR;
        return t;
    };
    auto col_of = [&](int k, int64_t c) __attribute__((always_inline)) -> DevChunkCol { return one ? fa.cols0[k] : const_col(fa.cols, (int64_t)k * fa.t.nchunks + c); };
    auto out_of = [&](int k, int64_t c) __attribute__((always_inline)) -> DevOutChunk {
        if (one) return fa.outs0[k];
        const ConstPtr<DevOutChunk> t = as_const<DevOutChunk>(fa.outs);
        DevOutChunk o;
        o.values = t[(int64_t)k * fa.t.nchunks + c].values; o.validity = t[(int64_t)k * fa.t.nchunks + c].validity;
        return o;
    };
    // this wave's rows of a tile: [rw, rw + WR) of the chunk
    auto wave_row = [&](const BfTile& t) __attribute__((always_inline)) -> int64_t { return t.rw; };
    // one column's vectors; SPARSE: only the 16-byte vectors that hold a wanted row (`need`: E-bit fields) are fetched
    auto load_col = [&](const DevChunkCol& col, const BfTile& t, V (&x)[G], bool sparse, const uint32_t (&need)[G]) {
        const int64_t rw = wave_row(t);
        const GlobalPtr<T> src = as_global<T>(col.values) + col.offset + rw;
        const bool fast = rw + WR <= t.clen && (((uintptr_t)(const void*)src) & 15) == 0;
        const V zero = {};
        if (fast) {
            const GlobalPtr<V> p = (GlobalPtr<V>)src + lane;
            if (sparse) {
#pragma unroll
                for (int g = 0; g < G; ++g) { x[g] = zero; if (need[g]) x[g] = __builtin_nontemporal_load(p + g * 64); }
            } else {
#pragma unroll
                for (int g = 0; g < G; ++g) x[g] = __builtin_nontemporal_load(p + g * 64);
            }
        } else if (SHORT && (((uintptr_t)(const void*)src) & 15) == 0 && t.clen - rw >= E) {
            // the ragged end of a batch (a frame of 1000-row batches ends EVERY wave in one): whole vectors through a clamped vector
            // index — lanes past the end re-read the last whole vector and are zeroed by a select —, the one vector across the end
            // from E uniform element loads.  No per-lane branch: the loads go out back to back like the fast path's.  (Element-wise,
            // two 8-byte loads per lane touch every line twice: 1000-row batches 3.3 ms per 1e9 rows against 2.6 this way, 2.1 for
            // 1024-row ones.  SHORT only: in the long forms, where one wave per BATCH is ragged, the same code measured slower than the
            // element path — 5000-row batches 3.12 against 2.66 ms, 7000 rows 2.63 against 2.22, same box, alternating.)
            const int64_t avail = t.clen - rw;                    // rows of the batch from this wave's first row on (< WR)
            const int nfull = (int)(avail / E), rem = (int)(avail - (int64_t)nfull * E);      // whole vectors (>= 1), rows in the one across the end
            const GlobalPtr<V> p = (GlobalPtr<V>)src;
            T sv[E];
#pragma unroll
            for (int e = 0; e < E; ++e) sv[e] = src[(int64_t)nfull * E + (e < rem ? e : 0) - (rem == 0 ? E : 0)];      // (uniform addresses inside the batch whatever rem is)
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int vi = g * 64 + lane;
                const int vc = vi < nfull ? vi : nfull - 1;
                V v = zero;
                if (!sparse || need[g]) v = __builtin_nontemporal_load(p + vc);
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    T q = v[e];
                    q = vi < nfull ? q : T(0);
                    q = (vi == nfull && e < rem) ? sv[e] : q;
                    x[g][e] = q;
                }
            }
        } else {
            // a slice that is not 16-byte aligned, a wave with fewer rows than one vector: element by element — sixteen independent guarded loads
            // per lane, issued back to back.  (A form that kept whole vectors as far as they exist and went element-wise only across
            // the end was tried for the short batches: a divergent if / else per vector, each waited for in turn — 5000-row batches
            // 4.5 ms per 1e9 rows against 2.6 with this path.)
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int q = g * 64 * E + lane * E + e;
                    x[g][e] = rw + q < t.clen ? src[q] : T(0);
                }
        }
    };
    auto win_issue = [&](const uint8_t* bitmap, int64_t offset, const BfTile& t) __attribute__((always_inline)) -> LaneWin<NWW> {
        const int64_t rw = wave_row(t);
        return lane_windows_issue<NWW>(bitmap, offset + rw, bitmap ? t.clen - rw : 0);
    };
    const uint32_t all_rows[G] = {};

    // What the count of a tile needs, requested one iteration ahead: the first predicate column's vectors and its validity
    // words (by_mask: the mask's value and validity words instead).
    struct Pre { V y[G]; LaneWin<NWW> q0, q1; };
    auto prefetch = [&](const BfTile& t, Pre& p) __attribute__((always_inline)) {
        if (by_mask) {
            const DevChunkCol m = one ? fa.mask0 : const_col(fa.t.mask, t.c);
            p.q0 = win_issue((const uint8_t*)m.values, m.offset, t);
            p.q1 = win_issue(m.validity, m.offset, t);
        } else {
            const DevChunkCol col = col_of(pc0, t.c);
            load_col(col, t, p.y, false, all_rows);
            if (NULLS) p.q0 = win_issue(col.validity, col.offset, t);
        }
    };
    V U[MULTI ? G : 1];
    // keep-words of the wave's rows in the REGISTER layout (word g * E + e, bit = lane: element e of vector g), one word per lane;
    // the block's row count is published.
    auto count_publish = [&](Pre& p, const BfTile& t, int par, uint64_t& km, int& wbase, int& bcnt) __attribute__((always_inline)) {
        uint32_t kf[G];
        if (by_mask) {
            uint64_t keep = lane_windows_finish<NWW>(p.q0);                // rows past the end of the batch are already cleared
            const DevChunkCol m = one ? fa.mask0 : const_col(fa.t.mask, t.c);
            if (m.validity) keep &= lane_windows_finish<NWW>(p.q1);
#pragma unroll
            for (int g = 0; g < G; ++g) kf[g] = lane_bits<E>(keep, g);
        } else {
            bf_term_fields<T, G>(a.bterm[0], p.y, kf);
            uint64_t vv = ~0ull;
            bool anyv = false;
            if (NULLS) {
                const DevChunkCol c0 = col_of(pc0, t.c);
                if (c0.validity) { vv &= lane_windows_finish<NWW>(p.q0); anyv = true; }
            }
            if (a.nterms > 1) {
                uint32_t k1[G];
                bool done = false;
                if constexpr (MULTI) {
                    if (pc1 != pc0) {
                        const DevChunkCol c1 = col_of(pc1, t.c);
                        load_col(c1, t, U, false, all_rows);
                        if (NULLS && c1.validity) { vv &= lane_windows_finish<NWW>(win_issue(c1.validity, c1.offset, t)); anyv = true; }
                        bf_term_fields<T, G>(a.bterm[1], U, k1);
                        done = true;
                    }
                }
                if (!done) bf_term_fields<T, G>(a.bterm[1], p.y, k1);
                // Arrow's and / or: NULL where either side is NULL -> the row is dropped (the validity words are ANDed below)
#pragma unroll
                for (int g = 0; g < G; ++g) kf[g] = a.combine == RDF_OP_AND ? (kf[g] & k1[g]) : (kf[g] | k1[g]);
            }
            if (NULLS && anyv) {
#pragma unroll
                for (int g = 0; g < G; ++g) kf[g] &= lane_bits<E>(vv, g);
            }
            const int64_t rw = wave_row(t);
            if (rw + WR > t.clen) {        // a tile at the end of its batch: rows that do not exist
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    uint32_t f = 0;
#pragma unroll
                    for (int e = 0; e < E; ++e) f |= (rw + g * 64 * E + lane * E + e < t.clen ? 1u : 0u) << e;
                    kf[g] &= f;
                }
            }
        }
        km = 0;
        int cnt = 0;
        uint64_t roww = 0;          // (keep_out) lane r: the keep bits of the wave's rows [64 r, 64 r + 64) in ROW order
#pragma unroll
        for (int g = 0; g < G; ++g) {
            uint64_t be[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const uint64_t w = __ballot((kf[g] >> e) & 1);
                if (lane == g * E + e) km = w;
                cnt += __popcll(w);
                be[e] = w;
            }
            if constexpr (!BYMASK) {
                if (a.keep_out) {       // (block-uniform) frames of two column widths: the other width's launch reads the kept rows as a mask
                    // rows 64 E g + 64 j + [0, 64) are lanes [64 j / E, 64 (j + 1) / E), E bits each: the ballots' pieces interleaved (scalar unit)
#pragma unroll
                    for (int j = 0; j < E; ++j) {
                        uint64_t word = 0;
#pragma unroll
                        for (int e = 0; e < E; ++e)
                            word |= (E == 2 ? bf_spread2(be[e] >> (32 * j)) : bf_spread4(be[e] >> (16 * j))) << e;
                        if (lane == g * E + j) roww = word;
                    }
                }
            }
        }
        if constexpr (!BYMASK) {
            if (a.keep_out && lane < NWW) {
                // into the frame's own mask: batch c's bits start at a multiple of 64 (fa.t.mask[c].values), this wave's at row rw, a
                // multiple of 64 too; words past the batch's last row belong to the next batch
                const int64_t rw = wave_row(t);
                const DevChunkCol m = one ? fa.mask0 : const_col(fa.t.mask, t.c);
                if (rw + 64 * (int64_t)lane < t.clen) ((GlobalMutPtr<uint64_t>)(void*)const_cast<void*>(m.values))[(rw >> 6) + lane] = roww;
            }
        }
        if (lane == 0) wcnt[par][wave] = cnt;
        __syncthreads();
        wbase = 0; bcnt = 0;
        if constexpr (SHORT) {
            // the batch's waves only; wbase is the wave's place in the staging buffer (its group's part + the group's waves in front of it)
#pragma unroll
            for (int w = 0; w < kBfWaves; ++w) {
                const int c = __builtin_amdgcn_readfirstlane(wcnt[par][w]);
                if ((w >> sshift) == wg) { if (w < wave) wbase += c; bcnt += c; }
            }
            wbase += gbase;
        } else {
#pragma unroll
            for (int w = 0; w < kBfWaves; ++w) { const int c = __builtin_amdgcn_readfirstlane(wcnt[par][w]); if (w < wave) wbase += c; bcnt += c; }
            if constexpr (MODE == 0) { if (tid == 0) bf_st(a.tile_state + t.tile, kBfCount | (unsigned long long)bcnt); }
        }
    };

    if (tid == 0) { sh_tile[0] = draw(); sh_tile[1] = draw(); }
    __syncthreads();
    int64_t Tc = (int64_t)uniform64((uint64_t)sh_tile[0]), Tn = (int64_t)uniform64((uint64_t)sh_tile[1]);     // (block-uniform by construction: say so, or every address derived from them is per-lane arithmetic)
    __syncthreads();
    if (Tc >= ntiles) return;
    BfTile tc = locate(Tc), tn = tc;
    Pre PA, PB;
    uint64_t km_c = 0, km_n = 0;
    int wbase_c = 0, bcnt_c = 0, wbase_n = 0, bcnt_n = 0, par = 1;
    int64_t cur_chunk = tc.c;
    prefetch(tc, PA);
    if (Tn < ntiles) { tn = locate(Tn); prefetch(tn, PB); }
    count_publish(PA, tc, 0, km_c, wbase_c, bcnt_c);

    int64_t own_rows = 0;          // (OWNED, thread 0) kept rows of the current batch up to and including the block's last tile
    auto flush_nulls = [&]() __attribute__((always_inline)) {      // (between two block barriers)
        if (NULLS && tid < ncols && nullacc[tid]) {
            atomicAdd((unsigned long long*)&fa.out_null_counts[(int64_t)tid * fa.t.nchunks + cur_chunk], (unsigned long long)nullacc[tid]);
            nullacc[tid] = 0;
        }
    };

    // One iteration.  X: the current tile's prefetched registers (counted, the count published one iteration ago), Y: the next tile's.
    auto step = [&](Pre& X, Pre& Y) __attribute__((always_inline)) -> bool {
        int64_t drawn = 0;
        if (tid == 0) drawn = draw();                    // the tile after the next: the atomic returns under everything below
        // a. the next tile: count, publish
        if (Tn < ntiles) count_publish(Y, tn, par, km_n, wbase_n, bcnt_n);
        else __syncthreads();                            // (the staging buffer, sh_base and nullacc are reused below)
        par ^= 1;
        if constexpr (!SHORT) { if (tc.c != cur_chunk) { flush_nulls(); cur_chunk = tc.c; } }
        // b. rows of the batch in front of the current tile
        if (tid == 0) {
            int64_t base = 0;               // a batch's first tile starts the batch's output
            if constexpr (OWNED) {
                // the tile in front of this one, if the batch has one, was this block's previous tile
                base = tc.first != Tc ? own_rows : 0;
                own_rows = base + bcnt_c;
            }
            if (MODE == 0 && tc.first != Tc) {
                unsigned long long w, t0 = 0;
                int spins = 0;
                auto wait_for = [&](const unsigned long long* p) __attribute__((always_inline)) {
                    for (;;) {
                        w = bf_ld(p);
                        if ((w >> 62) == 2) return;
                        __builtin_amdgcn_s_sleep(2);
                        if ((++spins & 255) == 0) {          // every ~50 us of waiting: the clock and the flag
                            const unsigned long long now = wall_clock64();
                            if (t0 == 0) t0 = now;
                            if (now - t0 > kBfWaitTicks || bf_aborted(a.abort_flag)) { __hip_atomic_store(a.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); w = 0; return; }
                        }
                    }
                };
                wait_for(a.tile_state + Tc);
                base = (int64_t)(w & kBfValue);
                // the scanner's prefixes count the rows of the FRAME: take off what lies in front of the batch (its first tile was
                // passed before this one)
                wait_for(a.tile_state + tc.first);
                base -= (int64_t)(w & kBfValue);
                if (base < 0 || base > tc.r0) base = 0;      // (only after an abort: keep the stores inside the batch's output)
            }
            sh_base = base;
            sh_tile[par] = drawn;
        }
        // c. every column: stage at the ranks, store
        uint64_t B[G][E];
        uint32_t need[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            need[g] = 0;
#pragma unroll
            for (int e = 0; e < E; ++e) { B[g][e] = rl64(km_c, g * E + e); need[g] |= (uint32_t)((B[g][e] >> lane) & 1) << e; }
        }
        const bool sparse = bcnt_c * 8 < S * WR;
        // one column's registers (and, NULLS, its validity bits) to the staging buffer, at the ranks of the kept rows
        auto stage_col = [&](const V (&x)[G], bool has_validity, const LaneWin<NWW>& q) {
            uint64_t vw = 0;
            if (NULLS && has_validity) vw = lane_windows_finish<NWW>(q);
            int base = wbase_c;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                int r = base;
#pragma unroll
                for (int e = 0; e < E; ++e) r = mbcnt64(B[g][e], r);
                uint32_t vb = (1u << E) - 1;
                if (NULLS && has_validity) vb = lane_bits<E>(vw, g);
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if ((need[g] >> e) & 1) {
                        stage[r] = x[g][e];
                        if (NULLS) vstage[r] = (uint8_t)((vb >> e) & 1);
                        ++r;
                    }
                    base += __popcll(B[g][e]);
                }
            }
        };
        // the staged run [0, bcnt_c) -> rows [tbase, tbase + bcnt_c) of the batch's output: aligned 16-byte stores
        auto store_col = [&](int k, const DevOutChunk& oc, int64_t tbase, bool has_validity) __attribute__((always_inline)) {
            const GlobalMutPtr<T> o = as_global_mut<T>(oc.values) + tbase;
            int head = (int)(((16 - (((uintptr_t)oc.values + (uintptr_t)tbase * sizeof(T)) & 15)) & 15) / sizeof(T));   // elements in front of the first aligned vector
            if (head > bcnt_c) head = bcnt_c;
            if (gt < head) o[gt] = stage[gbase + gt];
            const int nv = (bcnt_c - head) / E;
            for (int q = gt; q < nv; q += gsz) {
                const int i0 = head + q * E;
                V x;
#pragma unroll
                for (int e = 0; e < E; ++e) x[e] = stage[gbase + i0 + e];
                __builtin_nontemporal_store(x, (GlobalMutPtr<V>)(o + i0));
            }
            const int tail0 = head + nv * E;
            if (gt < bcnt_c - tail0) o[tail0 + gt] = stage[gbase + tail0 + gt];
            if (NULLS && oc.validity) {
                // out bits [tbase, tbase + bcnt_c): 64 aligned positions per wave step; the run's first and last word are shared with
                // the neighbouring tiles (the bitmap is pre-zeroed)
                const int64_t end = tbase + bcnt_c, w0 = tbase >> 6, w1 = (end + 63) >> 6;
                int nn = 0;
                for (int64_t wq = w0 + wj; wq < w1; wq += S) {
                    const int64_t pos = wq * 64 + lane;
                    const bool inside = pos >= tbase && pos < end;
                    const bool bitv = inside && (!has_validity || vstage[inside ? gbase + pos - tbase : 0]);
                    const uint64_t word = __ballot(bitv), inw = __ballot(inside);
                    if (lane == 0) {
                        if (inw == ~0ull) ((GlobalMutPtr<uint64_t>)(void*)oc.validity)[wq] = word;
                        else if (word) atomicOr((unsigned long long*)oc.validity + wq, (unsigned long long)word);
                        nn += __popcll(inw & ~word);
                    }
                }
                if constexpr (SHORT) { if (lane == 0 && nn) atomicAdd((unsigned long long*)&fa.out_null_counts[(int64_t)k * fa.t.nchunks + tc.c], (unsigned long long)nn); }
                else if (lane == 0 && nn) atomicAdd(&nullacc[k], nn);
            }
        };
        // column kk of the loop is frame column col_at(kk): the first predicate column comes first (its vectors are in X.y)
        auto col_at = [&](int kk) __attribute__((always_inline)) -> int { return by_mask ? kk : (kk == 0 ? pc0 : (kk <= pc0 ? kk - 1 : kk)); };
        DevChunkCol col = col_of(col_at(0), tc.c);
        LaneWin<NWW> qx = X.q0, qu = X.q0;
        if (by_mask) {
            load_col(col, tc, X.y, sparse, need);
            if (NULLS) qx = win_issue(col.validity, col.offset, tc);
        }
        int64_t tbase = 0, Tnn = 0;
        // outputs sized by an earlier count that does not hold any more: the batch's rows are not written (the host sees out_len > capacity)
        auto room = [&](int64_t tb) __attribute__((always_inline)) -> bool { return !fa.out_cap || tb + bcnt_c <= as_const<int64_t>(fa.out_cap)[tc.c]; };
        if constexpr (!MULTI) {
            stage_col(X.y, col.validity != nullptr, qx);
            __syncthreads();
            tbase = (int64_t)uniform64((uint64_t)sh_base); Tnn = (int64_t)uniform64((uint64_t)sh_tile[par]);
            if (room(tbase)) store_col(col_at(0), out_of(col_at(0), tc.c), tbase, col.validity != nullptr);
        } else {
            // two register sets take turns: while column kk is staged and stored, column kk + 1 is in flight
#pragma unroll 1
            for (int kk = 0; kk < ncols; kk += 2) {
                DevChunkCol col1 = col;
                if (kk + 1 < ncols) {
                    col1 = col_of(col_at(kk + 1), tc.c);
                    load_col(col1, tc, U, sparse, need);
                    if (NULLS) qu = win_issue(col1.validity, col1.offset, tc);
                }
                if (kk > 0) __syncthreads();               // the previous column has left the staging buffer
                stage_col(X.y, col.validity != nullptr, qx);
                __syncthreads();
                if (kk == 0) { tbase = (int64_t)uniform64((uint64_t)sh_base); Tnn = (int64_t)uniform64((uint64_t)sh_tile[par]); }
                if (room(tbase)) store_col(col_at(kk), out_of(col_at(kk), tc.c), tbase, col.validity != nullptr);
                if (kk + 1 < ncols) {
                    if (kk + 2 < ncols) {
                        col = col_of(col_at(kk + 2), tc.c);
                        load_col(col, tc, X.y, sparse, need);
                        if (NULLS) qx = win_issue(col.validity, col.offset, tc);
                    }
                    __syncthreads();
                    stage_col(U, col1.validity != nullptr, qu);
                    __syncthreads();
                    if (room(tbase)) store_col(col_at(kk + 1), out_of(col_at(kk + 1), tc.c), tbase, col1.validity != nullptr);
                }
            }
        }
        if constexpr (SHORT) { if (gt == 0 && a.out_len && tc.valid) a.out_len[tc.c] = bcnt_c; }
        else if (tid == 0 && a.out_len && tc.r0 + TR >= tc.clen) a.out_len[tc.c] = tbase + bcnt_c;      // the batch's last tile
        // the tile after the next: its registers are the ones the current tile has left
        BfTile tnn = tn;
        if (Tnn < ntiles) { tnn = locate(Tnn); prefetch(tnn, X); }
        tc = tn; Tc = Tn; tn = tnn; Tn = Tnn;
        km_c = km_n; wbase_c = wbase_n; bcnt_c = bcnt_n;
        return Tc < ntiles;
    };
    for (;;) {
        if (!step(PA, PB)) break;
        if (!step(PB, PA)) break;
    }
    __syncthreads();
    if constexpr (!SHORT) flush_nulls();
}

// SHORT mode: log2 of the waves a batch takes (0..3) for a frame whose longest batch has max_len rows, or -1 when a batch is
// longer than a tile
int bfilter_short_shift(int esize, int ncols, int64_t max_len) {
    const int wr = bfilter_tile_rows(esize, ncols) / kBfWaves;
    for (int sh = 0; sh <= 3; ++sh) if (max_len <= ((int64_t)wr << sh)) return sh;
    return -1;
}

int bfilter_tile_rows(int esize, int ncols) {
    const int E = 16 / esize;
    const int G = esize == 8 ? (ncols > 1 ? 4 : 8) : (ncols > 1 ? 2 : 4);      // (launch_bfilter's choice)
    return kBfWaves * G * 64 * E;
}

hipError_t launch_bfilter(const BFilterArgs& a, int esize, bool nulls, hipStream_t s) {
    if (a.w.t.ntiles <= 0) return hipSuccess;
    const bool multi = a.w.ncols > 1 || a.force_multi, by_mask = a.nterms == 0;
    const int mode = a.short_mode;          // 0: long batches (block 0 scans), 1: short batches, 2: a block owns whole batches
    const dim3 grid((unsigned)(a.nworkers + (mode == 0 ? 1 : 0))), block(kBfBlock);
#define RDF_BF_LAUNCH3(T, G, MULTI, NULLS, BYMASK) \
    do { if (mode == 1) hipLaunchKernelGGL((bfilter_kernel<T, G, MULTI, NULLS, BYMASK, 1>), grid, block, 0, s, a); \
         else if (mode == 2) hipLaunchKernelGGL((bfilter_kernel<T, G, MULTI, NULLS, BYMASK, 2>), grid, block, 0, s, a); \
         else hipLaunchKernelGGL((bfilter_kernel<T, G, MULTI, NULLS, BYMASK, 0>), grid, block, 0, s, a); } while (0)
#define RDF_BF_LAUNCH2(T, G, MULTI, NULLS) \
    do { if (by_mask) RDF_BF_LAUNCH3(T, G, MULTI, NULLS, true); else RDF_BF_LAUNCH3(T, G, MULTI, NULLS, false); } while (0)
#define RDF_BF_LAUNCH(T, G, MULTI) \
    do { if (nulls) RDF_BF_LAUNCH2(T, G, MULTI, true); else RDF_BF_LAUNCH2(T, G, MULTI, false); } while (0)
    if (esize == 8) { if (multi) RDF_BF_LAUNCH(uint64_t, 4, true); else RDF_BF_LAUNCH(uint64_t, 8, false); }
    else if (multi) RDF_BF_LAUNCH(uint32_t, 2, true);       // (four rows per vector: twice the keep-words per vector of an 8-byte column)
    else RDF_BF_LAUNCH(uint32_t, 4, false);
#undef RDF_BF_LAUNCH
#undef RDF_BF_LAUNCH2
#undef RDF_BF_LAUNCH3
    return hipGetLastError();
}

}  // namespace rdfk
