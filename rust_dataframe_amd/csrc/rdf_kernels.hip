// rdf_kernels.hip — hand-written CDNA4 (gfx950) kernels for rust-dataframe's Arrow compute hot path.
//
// Everything here is HBM-bandwidth-bound integer / f64 work (no MFMA): the design rules are wide
// coalesced loads with many bytes in flight per CU, wave64 ballots / mbcnt instead of shuffles
// where a bitmap is involved, LDS only for staging (compaction) and spills, one pass over HBM per
// fused expression tree, and persistent grids sized to the 256 CUs.
//
// Kernels (reference counterpart in brackets, paths relative to the reference root):
//   eval_kernel (rdf_eval.hip) fused expression-tree evaluator: an accumulator-machine interpreter
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
#include "rdf_common.hip.h"
#include "rdf_join_place.h"

namespace rdfk {

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
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = wave_id();
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
    // Per block iteration: kBlock*kFU vectors = 2048 rows; wave w owns the 512 consecutive rows
    // [256w, 256w+256) vectors of it, so one load instruction covers 64 consecutive 16-byte vectors
    // (1 KiB) and the wave's validity bits are eight consecutive 64-bit windows (scalar loads).
    const int64_t per_iter = (int64_t)kBlock * kFU;
    const int64_t body_rows = 2 * nvec;  // rows covered by the vector body, starting at row `head`
    for (int64_t base = (int64_t)blockIdx.x * per_iter; base < nvec; base += (int64_t)gridDim.x * per_iter) {
        const int64_t wbase = base + (int64_t)wave * (kFU * 64);  // first vector of this wave
        dvec2 vx[kFU], vy[kFU];
#pragma unroll
        for (int u = 0; u < kFU; ++u) {
            const int64_t i = wbase + u * 64 + lane;
            if (i < nvec) {
                vx[u] = __builtin_nontemporal_load(xv + i);
                if (!SAME) vy[u] = __builtin_nontemporal_load(yv + i);
            } else {
                vx[u] = (dvec2)(0.0);
                if (!SAME) vy[u] = (dvec2)(0.0);
            }
        }
        uint64_t wx[2 * kFU], wy[2 * kFU];
        if (HASV) {
            const int64_t rw = 2 * wbase;  // body-relative first row of this wave
            if (a.x_validity) load_windows<2 * kFU>(a.x_validity, a.x_offset + head + rw, body_rows - rw, wx);
            if (!SAME && a.y_validity) load_windows<2 * kFU>(a.y_validity, a.y_offset + head + rw, body_rows - rw, wy);
        }
#pragma unroll
        for (int u = 0; u < kFU; ++u) {
            const int64_t i = wbase + u * 64 + lane;
            uint32_t bits = i < nvec ? 3u : 0u;
            if (HASV) {
                // lane l holds rows 128u + 2l, 2l+1 of the wave: window 2u (l < 32) or 2u+1, bits (2l)&63, +1
                const int sh = (2 * lane) & 63;
                if (a.x_validity) bits &= (uint32_t)((lane < 32 ? wx[2 * u] : wx[2 * u + 1]) >> sh) & 3u;
                if (!SAME && a.y_validity) bits &= (uint32_t)((lane < 32 ? wy[2 * u] : wy[2 * u + 1]) >> sh) & 3u;
            }
            const dvec2 yy = SAME ? vx[u] : vy[u];
            if ((bits & 1u) && cmp_f64<CMP>(vx[u].x, a.c)) g.add(yy.x);
            if ((bits & 2u) && cmp_f64<CMP>(vx[u].y, a.c)) g.add(yy.y);
        }
    }
    block_reduce_agg(CLS_F64, d2u(g.sum), d2u(g.mn), d2u(g.mx), g.cnt, red_lds, &a.partials[blockIdx.x]);
}

// ------------------------------------------------------------------------------------------------
// stream compaction (Column::filter).  A tile never spans two chunks, and wave w of the block owns WW consecutive mask
// words of it: WW = 16 (tiles of kFilterTile = 4096 rows, 1024 per wave: the best of 2048 / 4096 / 8192 on long chunks) or
// WW = 4 (tiles of kFilterTileSmall = 1024 rows) for frames held in the reader's 1024-row RecordBatches, where a
// 4096-row tile would leave three of the four waves without rows.  The host picks per call (filter_prepare).
template <int WW> constexpr int filter_tile_rows() { return WW * 64 * (kBlock / 64); }
static_assert(filter_tile_rows<16>() == kFilterTile && filter_tile_rows<4>() == kFilterTileSmall, "tile sizes of the two instantiations");

template <int WW>
__device__ __forceinline__ void locate_tile(const MaskTables& t, int64_t tile, int64_t& c, int64_t& r0, int64_t& clen) {
    c = t.nchunks == 1 ? 0 : find_chunk_tile(t.chunk_tile_start, t.nchunks, tile);
    r0 = (tile - t.chunk_tile_start[c]) * filter_tile_rows<WW>();
    clen = t.chunk_len[c];
}

// keep-words of this wave's WW * 64 rows: mask value bits AND mask validity bits, rows past the chunk end cleared
template <int kWW>
__device__ __forceinline__ void keep_words(const DevChunkCol& m, int64_t rw, int64_t clen, uint64_t (&kw)[kWW]) {
    load_windows<kWW>((const uint8_t*)m.values, m.offset + rw, clen - rw, kw);
    if (m.validity) {
        uint64_t vw[kWW];
        load_windows<kWW>(m.validity, m.offset + rw, clen - rw, vw);
#pragma unroll
        for (int i = 0; i < kWW; ++i) kw[i] &= vw[i];
    }
}

// One wave per tile-quarter: counts the kept rows of its WW words (s_bcnt1); reads 1 bit/row.
template <int kWW>
__global__ __launch_bounds__(kBlock) void mask_count_kernel(const MaskTables t, int64_t* tile_counts) {
    __shared__ int wave_cnt[kBlock / 64];
    const int wave = wave_id();
    for (int64_t tile = blockIdx.x; tile < t.ntiles; tile += gridDim.x) {
        int64_t c, r0, clen;
        locate_tile<kWW>(t, tile, c, r0, clen);
        uint64_t kw[kWW];
        keep_words<kWW>(t.mask[c], r0 + (int64_t)wave * (kWW * 64), clen, kw);
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < kWW; ++i) cnt += __popcll(kw[i]);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) wave_cnt[wave] = cnt;
        __syncthreads();
        if (threadIdx.x == 0) tile_counts[tile] = (int64_t)wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    }
}

// The same for a mask held in one chunk: its descriptor and length travel in the kernel arguments.
__global__ __launch_bounds__(kBlock) void mask_count_one_kernel(const DevChunkCol mask, int64_t clen, int64_t ntiles, int64_t* tile_counts) {
    constexpr int kWW = 16;
    __shared__ int wave_cnt[kBlock / 64];
    const int wave = wave_id();
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint64_t kw[kWW];
        keep_words<kWW>(mask, tile * filter_tile_rows<kWW>() + (int64_t)wave * (kWW * 64), clen, kw);
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < kWW; ++i) cnt += __popcll(kw[i]);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) wave_cnt[wave] = cnt;
        __syncthreads();
        if (threadIdx.x == 0) tile_counts[tile] = (int64_t)wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    }
}

// Exclusive scan of n int64 counts into scan[0..n] (scan[n] = total), hierarchical: every block scans a
// 4096-entry segment (4 consecutive entries per thread: coalesced 32-byte reads), one block scans the
// segment totals, a third pass adds the segment offsets.  (A single-block scan of the 48 828 tile counts
// of a 1e8-row filter took 85 us — a quarter of the whole filter.)
constexpr int kScanThreads = 1024;
constexpr int kScanPer = 4;
constexpr int kScanSeg = kScanThreads * kScanPer;

__device__ __forceinline__ int64_t block_exclusive_scan(int64_t v, int64_t* wave_tot, int64_t& total) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int64_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)inc, d), hi = (uint32_t)__shfl_up((int)(uint32_t)((uint64_t)inc >> 32), d);
        int64_t o = (int64_t)(((uint64_t)hi << 32) | lo);
        if (lane >= d) inc += o;
    }
    __syncthreads();
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    int64_t wbase = 0, tot = 0;
    for (int w = 0; w < kScanThreads / 64; ++w) { if (w < wave) wbase += wave_tot[w]; tot += wave_tot[w]; }
    total = tot;
    return wbase + inc - v;
}

__global__ __launch_bounds__(kScanThreads) void scan_segments_kernel(const int64_t* counts, int64_t* scan, int64_t n, int64_t* seg_totals) {
    __shared__ int64_t wave_tot[kScanThreads / 64];
    const int64_t base = (int64_t)blockIdx.x * kScanSeg + (int64_t)threadIdx.x * kScanPer;
    int64_t v[kScanPer], s = 0;
#pragma unroll
    for (int j = 0; j < kScanPer; ++j) { v[j] = base + j < n ? counts[base + j] : 0; s += v[j]; }
    int64_t total;
    int64_t run = block_exclusive_scan(s, wave_tot, total);
#pragma unroll
    for (int j = 0; j < kScanPer; ++j) { if (base + j < n) scan[base + j] = run; run += v[j]; }
    if (threadIdx.x == 0) seg_totals[blockIdx.x] = total;
}
__global__ __launch_bounds__(kScanThreads) void scan_totals_kernel(int64_t* seg_totals, int64_t nseg, int64_t* grand_total) {
    __shared__ int64_t wave_tot[kScanThreads / 64];
    int64_t carry = 0;
    for (int64_t b = 0; b < nseg; b += kScanThreads) {
        const int64_t i = b + threadIdx.x;
        const int64_t v = i < nseg ? seg_totals[i] : 0;
        int64_t total;
        const int64_t ex = block_exclusive_scan(v, wave_tot, total);
        if (i < nseg) seg_totals[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) *grand_total = carry;
}
__global__ __launch_bounds__(kScanThreads) void scan_add_kernel(int64_t* scan, int64_t n, const int64_t* seg_offsets) {
    const int64_t off = seg_offsets[blockIdx.x];
    const int64_t base = (int64_t)blockIdx.x * kScanSeg + (int64_t)threadIdx.x * kScanPer;
#pragma unroll
    for (int j = 0; j < kScanPer; ++j) if (base + j < n) scan[base + j] += off;
}

// Compaction of one column of one wave's 512 rows.  Ranks come from popcounts of the wave's keep words
// (scalar prefix + lane prefix): no shuffles, no atomics for the values.  Kept values are staged in the
// wave's PRIVATE 4 KiB LDS region at their rank and written out as one contiguous, coalesced run at
// the wave's output offset — waves never wait for each other inside a column.
template <typename T, int kWW>
__device__ __forceinline__ void compact_wave(const DevChunkCol col, const DevOutChunk oc, int64_t rw, int64_t clen,
                                             int64_t wave_out, const uint64_t (&kw)[kWW], int wave_cnt,
                                             unsigned char* stage_raw, uint8_t* vstage, uint32_t& nulls) {
    const int lane = threadIdx.x & 63;
    T* stage = (T*)stage_raw;
    const GlobalPtr<T> src = as_global<T>(col.values) + col.offset + rw + lane;
    uint64_t vw[kWW];
    const bool hasv = col.validity != nullptr;
    if (hasv) load_windows<kWW>(col.validity, col.offset + rw, clen - rw, vw);
    T val[kWW];
#pragma unroll
    for (int i = 0; i < kWW; ++i)
        if ((kw[i] >> lane) & 1) val[i] = __builtin_nontemporal_load(src + i * 64);
    int wb = 0;
#pragma unroll
    for (int i = 0; i < kWW; ++i) {
        if ((kw[i] >> lane) & 1) {
            const int rank = wb + __popcll(kw[i] & ((1ull << lane) - 1));
            stage[rank] = val[i];
            if (hasv) vstage[rank] = (uint8_t)((vw[i] >> lane) & 1);
        }
        wb += __popcll(kw[i]);
    }
    __builtin_amdgcn_wave_barrier();  // same-wave LDS ops are executed in order; this only pins the compiler
    const GlobalMutPtr<T> dst = as_global_mut<T>(oc.values) + wave_out;
    for (int i = lane; i < wave_cnt; i += 64) __builtin_nontemporal_store(stage[i], dst + i);
    if (hasv && oc.validity) {
        // out bits [wave_out, wave_out+wave_cnt): ballot 64 aligned positions at a time, OR into the
        // (pre-zeroed) bitmap; boundary words are shared with neighbouring waves/tiles, hence atomics.
        const int64_t end = wave_out + wave_cnt;
        for (int64_t wpos = wave_out & ~63ll; wpos < end; wpos += 64) {
            const int64_t pos = wpos + lane;
            const bool inside = pos >= wave_out && pos < end;
            const bool bit = inside && vstage[pos - wave_out];
            const uint64_t word = __ballot(bit);
            const uint64_t inw = __ballot(inside);
            if (lane == 0) {
                if (word) atomicOr((unsigned long long*)oc.validity + (wpos >> 6), (unsigned long long)word);
                nulls += (uint32_t)__popcll(inw & ~word);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
}

constexpr int kCompactCols = 8;  // columns per launch (the host loops over wider frames)

// ES = common element size of all columns of the launch (8/4/2/1) or 0 for mixed sizes.
template <int ES, int kWW>
__global__ __launch_bounds__(kBlock) void compact_kernel(const FilterArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char stage[kBlock / 64][kWW * 64 * 8];  // 8 KiB (WW = 16) or 2 KiB per wave
    __shared__ uint8_t vstage[kBlock / 64][kWW * 64];
    __shared__ int wave_cnt[2][kBlock / 64];
    const int lane = threadIdx.x & 63;
    const int wave = wave_id();
    // per-wave null counters per column (in LDS: the kernel has no registers to spare), flushed once per
    // (column, chunk) — not once per tile
    __shared__ uint32_t nullacc[kBlock / 64][kCompactCols];
    if (lane < kCompactCols) nullacc[wave][lane] = 0;
    __builtin_amdgcn_wave_barrier();
    int64_t cur_chunk = -1;
    int parity = 0;
    for (int64_t tile = blockIdx.x; tile < a.t.ntiles; tile += gridDim.x, parity ^= 1) {
        int64_t c, r0, clen;
        locate_tile<kWW>(a.t, tile, c, r0, clen);
        // where this tile's kept rows start in the chunk's output: asked for before the mask words are waited on
        const int64_t out_base = a.tile_scan[tile] - a.tile_scan[a.t.chunk_tile_start[c]];
        if (c != cur_chunk) {
            if (cur_chunk >= 0 && lane < a.ncols && nullacc[wave][lane]) {
                atomicAdd((unsigned long long*)&a.out_null_counts[(int64_t)lane * a.t.nchunks + cur_chunk], (unsigned long long)nullacc[wave][lane]);
                nullacc[wave][lane] = 0;
            }
            __builtin_amdgcn_wave_barrier();
            cur_chunk = c;
        }
        const int64_t rw = r0 + (int64_t)wave * (kWW * 64);
        uint64_t kw[kWW];
        keep_words<kWW>(a.t.mask[c], rw, clen, kw);
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < kWW; ++i) cnt += __popcll(kw[i]);
        if (lane == 0) wave_cnt[parity][wave] = cnt;
        __syncthreads();  // the only block barrier per tile (wave_cnt is double-buffered by tile parity)
        int wave_base = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) if (w < wave) wave_base += wave_cnt[parity][w];
        if (cnt > 0) {
            const int64_t wave_out = out_base + wave_base;
#pragma unroll 1
            for (int k = 0; k < a.ncols; ++k) {
                const DevChunkCol col = a.cols[(int64_t)k * a.t.nchunks + c];
                const DevOutChunk oc = a.outs[(int64_t)k * a.t.nchunks + c];
                uint32_t nn = 0;
                const int es = ES ? ES : a.esize[k];
                if (es == 8) compact_wave<uint64_t, kWW>(col, oc, rw, clen, wave_out, kw, cnt, stage[wave], vstage[wave], nn);
                else if (es == 4) compact_wave<uint32_t, kWW>(col, oc, rw, clen, wave_out, kw, cnt, stage[wave], vstage[wave], nn);
                else if (es == 2) compact_wave<uint16_t, kWW>(col, oc, rw, clen, wave_out, kw, cnt, stage[wave], vstage[wave], nn);
                else compact_wave<uint8_t, kWW>(col, oc, rw, clen, wave_out, kw, cnt, stage[wave], vstage[wave], nn);
                if (lane == 0 && nn) nullacc[wave][k] += nn;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (cur_chunk >= 0 && lane < a.ncols && nullacc[wave][lane])
        atomicAdd((unsigned long long*)&a.out_null_counts[(int64_t)lane * a.t.nchunks + cur_chunk], (unsigned long long)nullacc[wave][lane]);
}

// The same for a frame in one chunk, descriptors in the kernel arguments (FilterOneArgs): the per-tile chain is
// mask words -> barrier -> values -> store, without the chunk lookup and the two descriptor-table reads in between.
template <int ES>
__global__ __launch_bounds__(kBlock) void compact_one_kernel(const FilterOneArgs a) {
    constexpr int kWW = 16;
    __shared__ __attribute__((aligned(16))) unsigned char stage[kBlock / 64][kWW * 64 * 8];
    __shared__ uint8_t vstage[kBlock / 64][kWW * 64];
    __shared__ int wave_cnt[2][kBlock / 64];
    __shared__ uint32_t nullacc[kBlock / 64][kCompactCols];
    const int lane = threadIdx.x & 63;
    const int wave = wave_id();
    if (lane < kCompactCols) nullacc[wave][lane] = 0;
    __builtin_amdgcn_wave_barrier();
    int parity = 0;
    for (int64_t tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x, parity ^= 1) {
        const int64_t out_base = a.tile_scan[tile];
        const int64_t rw = tile * filter_tile_rows<kWW>() + (int64_t)wave * (kWW * 64);
        uint64_t kw[kWW];
        keep_words<kWW>(a.mask, rw, a.clen, kw);
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < kWW; ++i) cnt += __popcll(kw[i]);
        if (lane == 0) wave_cnt[parity][wave] = cnt;
        __syncthreads();
        int wave_base = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) if (w < wave) wave_base += wave_cnt[parity][w];
        if (cnt > 0) {
            const int64_t wave_out = out_base + wave_base;
#pragma unroll 1
            for (int k = 0; k < a.ncols; ++k) {
                uint32_t nn = 0;
                const int es = ES ? ES : a.esize[k];
                if (es == 8) compact_wave<uint64_t, kWW>(a.cols[k], a.outs[k], rw, a.clen, wave_out, kw, cnt, stage[wave], vstage[wave], nn);
                else if (es == 4) compact_wave<uint32_t, kWW>(a.cols[k], a.outs[k], rw, a.clen, wave_out, kw, cnt, stage[wave], vstage[wave], nn);
                else if (es == 2) compact_wave<uint16_t, kWW>(a.cols[k], a.outs[k], rw, a.clen, wave_out, kw, cnt, stage[wave], vstage[wave], nn);
                else compact_wave<uint8_t, kWW>(a.cols[k], a.outs[k], rw, a.clen, wave_out, kw, cnt, stage[wave], vstage[wave], nn);
                if (lane == 0 && nn) nullacc[wave][k] += nn;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < a.ncols && nullacc[wave][lane]) atomicAdd((unsigned long long*)&a.out_null_counts[lane], (unsigned long long)nullacc[wave][lane]);
}

// ------------------------------------------------------------------------------------------------
// take: out[j] = concat(chunks)[idx[j]] without materialising the concat.

template <typename T, typename IDX>
__global__ __launch_bounds__(kBlock) void take_kernel(const TakeArgs a) {
    const int lane = threadIdx.x & 63;
    uint32_t err = 0;
    int nulls = 0;
    const int64_t nwaves_total = (a.n + 63) >> 6;
    const double inv = chunk_lookup_scale(a.chunk_row_start, a.nchunks);
    const DevChunkCol first = a.chunks[0];   // a one-chunk column reads its descriptor once, not per element ahead of the gather
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
            const uint64_t ix = (uint64_t)as_global<IDX>(a.indices.values)[a.indices.offset + j];
            if (ix >= (uint64_t)a.total_rows) { err |= 2u; valid = false; }
            else {
                int64_t c = 0, start = 0;
                DevChunkCol cc = first;
                if (a.nchunks > 1) {
                    c = find_chunk_row(a.chunk_row_start, a.nchunks, (int64_t)ix, inv);
                    cc = a.chunks[c];
                    start = a.chunk_row_start[c];
                }
                const int64_t e = cc.offset + (int64_t)ix - start;
                v = as_global<T>(cc.values)[e];
                if (cc.validity) valid = (cc.validity[e >> 3] >> (e & 7)) & 1;
            }
        }
        if (inr) __builtin_nontemporal_store(v, as_global_mut<T>(a.out.values) + j);
        const uint64_t vb = __ballot(valid);
        const uint64_t ib = __ballot(inr);
        if (lane == 0) {
            if (a.out.validity) as_global_mut<uint64_t>(a.out.validity)[wv] = vb;
            nulls += __popcll(ib & ~vb);
        }
    }
    if (lane == 0 && nulls) atomicAdd((unsigned long long*)a.out_null_count, (unsigned long long)nulls);
    if (err) atomicOr(a.flags, err);
}

// ------------------------------------------------------------------------------------------------
// sort to indices (DataFrame::sort -> arrow::compute::lexsort_to_indices, src/dataframe.rs:194-214): a
// stable LSD radix sort of (key, row) pairs.  Per 8-bit pass: per-tile digit histograms (digit-major),
// one exclusive scan, then a stable scatter whose in-tile ranks come from wave ballots (the lanes that
// share my digit = AND over the 8 digit bits of the ballot or its complement) + an LDS prefix over the
// (row-of-items, wave) groups.  Columns are applied last-to-first; nulls go last via one extra 1-bit pass.

__device__ __forceinline__ uint64_t sort_key_bits(const DevChunkCol& cc, int dt, int64_t e) {
    switch (dt) {
        case RDF_I8: return (uint8_t)(as_global<uint8_t>(cc.values)[e] ^ 0x80u);
        case RDF_U8: return as_global<uint8_t>(cc.values)[e];
        case RDF_I16: return (uint16_t)(as_global<uint16_t>(cc.values)[e] ^ 0x8000u);
        case RDF_U16: return as_global<uint16_t>(cc.values)[e];
        case RDF_I32: return as_global<uint32_t>(cc.values)[e] ^ 0x80000000u;
        case RDF_U32: return as_global<uint32_t>(cc.values)[e];
        case RDF_F32: { const uint32_t b = as_global<uint32_t>(cc.values)[e]; return (b & 0x80000000u) ? (uint32_t)~b : (b ^ 0x80000000u); }
        case RDF_I64: return as_global<uint64_t>(cc.values)[e] ^ 0x8000000000000000ull;
        case RDF_F64: { const uint64_t b = as_global<uint64_t>(cc.values)[e]; return (b >> 63) ? ~b : (b ^ 0x8000000000000000ull); }
        default: return as_global<uint64_t>(cc.values)[e];
    }
}

__global__ __launch_bounds__(kBlock) void sort_keys_kernel(const SortKeyArgs a, uint64_t width_mask) {
    uint64_t kmin = ~0ull, kmax = 0, rmin = ~0ull, rmax = 0;
    const double inv = chunk_lookup_scale(a.chunk_row_start, a.nchunks);
    const DevChunkCol first = a.chunks[0];
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * kBlock) {
        const int64_t row = a.idx ? (int64_t)a.idx[i] : i;
        int64_t start = 0;
        DevChunkCol cc = first;
        if (a.nchunks > 1) {
            const int64_t c = find_chunk_row(a.chunk_row_start, a.nchunks, row, inv);
            cc = a.chunks[c];
            start = a.chunk_row_start[c];
        }
        const int64_t e = cc.offset + row - start;
        uint64_t k = sort_key_bits(cc, a.dtype, e);
        if (a.descending) k = ~k & width_mask;
        bool isnull = false;
        if (cc.validity) isnull = !((cc.validity[e >> 3] >> (e & 7)) & 1);
        if (a.hash_mul && !isnull) { rmin = k < rmin ? k : rmin; rmax = k > rmax ? k : rmax; k *= a.hash_mul; }
        const uint64_t kw = isnull ? 0 : k;  // nulls are ordered by the nulls-last pass; equal keys keep them stable
        a.keys[i] = kw;
        if (!isnull) { kmin = kw < kmin ? kw : kmin; kmax = kw > kmax ? kw : kmax; }
        if (a.nullflags) { if (a.null_or) { if (isnull) a.nullflags[row] = 1; } else a.nullflags[row] = isnull; }
    }
    if (a.bit_stats) {   // key range of the column: the radix passes work on key - min, so only the bytes of (max - min) need a pass
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const uint64_t x = shfl_xor64(kmin, m), y = shfl_xor64(kmax, m);
            kmin = x < kmin ? x : kmin; kmax = y > kmax ? y : kmax;
        }
        if ((threadIdx.x & 63) == 0) { atomicMin((unsigned long long*)&a.bit_stats[0], (unsigned long long)kmin); atomicMax((unsigned long long*)&a.bit_stats[1], (unsigned long long)kmax); }
    }
    if (a.hash_mul && a.raw_stats) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const uint64_t x = shfl_xor64(rmin, m), y = shfl_xor64(rmax, m);
            rmin = x < rmin ? x : rmin; rmax = y > rmax ? y : rmax;
        }
        if ((threadIdx.x & 63) == 0 && rmin <= rmax) { atomicMin((unsigned long long*)&a.raw_stats[0], (unsigned long long)rmin); atomicMax((unsigned long long*)&a.raw_stats[1], (unsigned long long)rmax); }
    }
}

__device__ __forceinline__ int sort_digit(const SortPassArgs& a, int64_t i) {
    if (a.nullflags) return a.nullflags[a.idx_in ? (int64_t)a.idx_in[i] : i];
    return (int)(((a.keys_in[i] - a.bias) >> a.shift) & 255);
}

// Block b owns the contiguous tiles [b*tpb, (b+1)*tpb): one histogram row per BLOCK (256 x gridDim entries
// to scan, a few MB), and the scatter walks its tiles in order carrying the running digit offsets in LDS.
__global__ __launch_bounds__(kBlock) void sort_hist_kernel(const SortPassArgs a) {
    __shared__ unsigned int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t tpb = (a.ntiles + gridDim.x - 1) / gridDim.x;
    const int64_t t0 = (int64_t)blockIdx.x * tpb, t1 = t0 + tpb < a.ntiles ? t0 + tpb : a.ntiles;
    for (int64_t tile = t0; tile < t1; ++tile) {
        const int64_t base = tile * kSortTile;
#pragma unroll
        for (int j = 0; j < kSortItems; ++j) {
            const int64_t i = base + j * kBlock + threadIdx.x;
            if (i < a.n) atomicAdd(&h[sort_digit(a, i)], 1u);
        }
    }
    __syncthreads();
    a.hist[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = h[threadIdx.x];
}

template <bool PAY64>
__global__ __launch_bounds__(kBlock) void sort_scatter_kernel(const SortPassArgs a) {
    __shared__ unsigned short grp[kSortItems * (kBlock / 64)][256];  // count of each digit per (row-of-items, wave) group
    __shared__ unsigned short dbase[256];                            // tile-local exclusive prefix of the digit totals
    __shared__ uint64_t lkeys[kSortTile];                            // the tile, locally sorted by digit (stable)
    __shared__ typename std::conditional<PAY64, uint64_t, uint32_t>::type lidx[kSortTile];
    __shared__ int64_t gbase[256];                                   // running global offset of each digit for this block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    gbase[threadIdx.x] = a.hist[(int64_t)threadIdx.x * gridDim.x + blockIdx.x];
    const int64_t tpb = (a.ntiles + gridDim.x - 1) / gridDim.x;
    const int64_t t0 = (int64_t)blockIdx.x * tpb, t1 = t0 + tpb < a.ntiles ? t0 + tpb : a.ntiles;
    for (int64_t tile = t0; tile < t1; ++tile) {
        for (int i = threadIdx.x; i < kSortItems * (kBlock / 64) * 256; i += kBlock) (&grp[0][0])[i] = 0;
        __syncthreads();
        const int64_t base = tile * kSortTile;
        const int count = (int)((a.n - base) < (int64_t)kSortTile ? (a.n - base) : (int64_t)kSortTile);
        int digit[kSortItems], rank[kSortItems];
        uint64_t key[kSortItems];
        typename std::conditional<PAY64, uint64_t, uint32_t>::type idx[kSortItems];
#pragma unroll
        for (int j = 0; j < kSortItems; ++j) {
            const int64_t i = base + j * kBlock + threadIdx.x;
            const bool in = i < a.n;
            key[j] = in ? a.keys_in[i] : 0;
            if (PAY64) idx[j] = in ? a.pay_in[i] : 0;
            else idx[j] = in ? (a.idx_in ? a.idx_in[i] : (uint32_t)i) : 0;
            const int d = in ? ((!PAY64 && a.nullflags) ? (int)a.nullflags[idx[j]] : (int)(((key[j] - a.bias) >> a.shift) & 255)) : 0;
            digit[j] = d;
            uint64_t peers = __ballot(in);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const uint64_t m = __ballot((d >> b) & 1);
                peers &= ((d >> b) & 1) ? m : ~m;
            }
            rank[j] = __popcll(peers & ((1ull << lane) - 1));
            if (in && rank[j] == 0) grp[j * (kBlock / 64) + wave][d] = (unsigned short)__popcll(peers);  // the group's first holder of d
        }
        __syncthreads();
        unsigned int total_d;
        {   // thread d: exclusive prefix of digit d's counts over the groups, in item order
            unsigned int run = 0;
#pragma unroll
            for (int g = 0; g < kSortItems * (kBlock / 64); ++g) { const unsigned int c = grp[g][threadIdx.x]; grp[g][threadIdx.x] = (unsigned short)run; run += c; }
            total_d = run;
        }
        // exclusive scan of the 256 digit totals (wave scan + 4 wave sums)
        unsigned int inc = total_d;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) { const unsigned int o = __shfl_up(inc, dd); if (lane >= dd) inc += o; }
        __shared__ unsigned int wsum[kBlock / 64];
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        unsigned int wb = 0;
        for (int w = 0; w < wave; ++w) wb += wsum[w];
        dbase[threadIdx.x] = (unsigned short)(wb + inc - total_d);
        __syncthreads();
        // local stable sort by digit into LDS
#pragma unroll
        for (int j = 0; j < kSortItems; ++j) {
            const int64_t i = base + j * kBlock + threadIdx.x;
            if (i < a.n) {
                const int pos = dbase[digit[j]] + grp[j * (kBlock / 64) + wave][digit[j]] + rank[j];
                lkeys[pos] = key[j];
                lidx[pos] = idx[j];
            }
        }
        __syncthreads();
        // coalesced write-out: consecutive threads hold consecutive members of a digit run
        for (int t = threadIdx.x; t < count; t += kBlock) {
            const uint64_t kk = lkeys[t];
            const auto ii = lidx[t];
            const int d = (!PAY64 && a.nullflags) ? (int)a.nullflags[ii] : (int)(((kk - a.bias) >> a.shift) & 255);
            const int64_t dst = gbase[d] + (t - dbase[d]);
            a.keys_out[dst] = kk;
            if (PAY64) a.pay_out[dst] = ii; else a.idx_out[dst] = (uint32_t)ii;
        }
        __syncthreads();
        gbase[threadIdx.x] += total_d;  // the next tile of this block continues each digit's run
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// equi-join indices (calc_equijoin_indices, src/functions/join.rs:19-137).  The reference hashes byte-encoded
// keys into HashMap<Vec<u8>, Vec<usize>> row lists; here the build side is radix-sorted by key (the sort
// kernels above), every probe row finds its partners' range with two binary searches, and the pairs are
// written at offsets from an exclusive scan of the per-row match counts.  NULL keys never match.

// multi-column join keys: one 64-bit hash per row over the columns' order-preserving key bits (SplitMix64 finaliser per
// column, chained); rows with a NULL in any key column keep hash 0 and are excluded through nullflags
constexpr uint64_t kJoinMul = 0x9E3779B97F4A7C15ull;     // slot = (key * kJoinMul) >> tshift; odd: key -> key * kJoinMul is a bijection
__device__ __forceinline__ uint64_t join_mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ __launch_bounds__(kBlock) void join_combine_kernel(const JoinCombineArgs a) {
    uint64_t kmin = ~0ull, kmax = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * kBlock) {
        const bool isnull = a.nullflags && a.nullflags[i];
        uint64_t h = 0x9E3779B97F4A7C15ull;
        for (int k = 0; k < a.nkeys; ++k) h = join_mix(h ^ a.bits[k][i]) + 0x9E3779B97F4A7C15ull * (uint64_t)(k + 1);
        if (isnull) h = 0;
        a.out[i] = h;
        if (!isnull) { kmin = h < kmin ? h : kmin; kmax = h > kmax ? h : kmax; }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const uint64_t x = shfl_xor64(kmin, m), y = shfl_xor64(kmax, m);
        kmin = x < kmin ? x : kmin; kmax = y > kmax ? y : kmax;
    }
    if ((threadIdx.x & 63) == 0) { atomicMin((unsigned long long*)&a.bit_stats[0], (unsigned long long)kmin); atomicMax((unsigned long long*)&a.bit_stats[1], (unsigned long long)kmax); }
}
__device__ __forceinline__ bool join_verify(const JoinProbeArgs& a, int64_t i, int64_t p) {   // hash match -> compare the key tuples
    const uint32_t r = a.ridx[p];
    bool eq = true;
    for (int k = 0; k < a.nkeys; ++k) eq = eq && a.pbits[k][i] == a.bbits[k][r];
    return eq;
}
__global__ __launch_bounds__(kBlock) void join_buckets_kernel(const JoinBucketArgs a) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.nrv; i += (int64_t)gridDim.x * kBlock) {
        const uint64_t b = (a.rkeys[i] - a.kmin) >> a.bucket_shift;
        if (i == 0 || ((a.rkeys[i - 1] - a.kmin) >> a.bucket_shift) != b) a.buckets[2 * b] = (uint32_t)i;
        if (i == a.nrv - 1 || ((a.rkeys[i + 1] - a.kmin) >> a.bucket_shift) != b) a.buckets[2 * b + 1] = (uint32_t)(i + 1);
    }
}
// the table of distinct build keys: the thread at the start of a run of equal sorted keys inserts it (distinct keys: an empty slot
// is claimed with one CAS on its second word, nobody compares keys while the table is built)
__global__ __launch_bounds__(kBlock) void join_table_kernel(const JoinTableArgs a) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.nrv; i += (int64_t)gridDim.x * kBlock) {
        const uint64_t k = a.rkeys[i];
        if (i > 0 && a.rkeys[i - 1] == k) continue;
        // the end of the run of equal keys that starts here: galloping, then a binary search (one lane walking a hot key's
        // millions of duplicates one dependent load at a time would hold its wave for as long)
        int64_t j = i + 1;
        if (j < a.nrv && a.rkeys[j] == k) {
            int64_t step = 2;
            while (i + step < a.nrv && a.rkeys[i + step] == k) step <<= 1;
            int64_t l = i + (step >> 1), h = i + step < a.nrv ? i + step : a.nrv;     // rkeys[l] == k; rkeys[h] != k or h == nrv
            while (l + 1 < h) { const int64_t mid = l + ((h - l) >> 1); if (a.rkeys[mid] == k) l = mid; else h = mid; }
            j = h;
        }
        const uint32_t info = j - i == 1 ? (0x80000000u | a.ridx[i]) : (uint32_t)(j - i);
        const unsigned long long w1 = ((unsigned long long)i << 32) | info;
        uint64_t s = (k * kJoinMul) >> a.tshift;
        for (;;) {
            if (atomicCAS((unsigned long long*)&a.table[2 * s + 1], 0ull, w1) == 0ull) { a.table[2 * s] = k; break; }
            s = (s + 1) & a.tmask;
        }
    }
}
typedef uint64_t join_u64x2 __attribute__((ext_vector_type(2)));
// exclusive prefix of v over the block's threads (NT threads); total in *total
template <int NT>
__device__ __forceinline__ PlaceCM place_block_scan(PlaceCM v, PlaceCM* total) {
    __shared__ PlaceCM wtot[NT / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    PlaceCM inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        PlaceCM o; o.c = __shfl_up(inc.c, d); o.m = __shfl_up(inc.m, d);
        if (lane >= d) inc = place_join(o, inc);
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    PlaceCM ex; ex.c = __shfl_up(inc.c, 1); ex.m = __shfl_up(inc.m, 1);
    if (lane == 0) ex = place_empty();
    PlaceCM wp = place_empty();
    for (int w = 0; w < wave; ++w) wp = place_join(wp, wtot[w]);
    if (total) { PlaceCM t = wp; for (int w = wave; w < NT / 64; ++w) t = place_join(t, wtot[w]); *total = t; }
    __syncthreads();
    return place_join(wp, ex);
}
__global__ __launch_bounds__(1024) void join_place_scan_kernel(const JoinPlaceArgs a) {     // phase 1: one block
    const int64_t per = (a.ntiles + 1023) / 1024;
    const int64_t t0 = (int64_t)threadIdx.x * per, t1 = t0 + per < a.ntiles ? t0 + per : a.ntiles;
    PlaceCM acc = place_empty();
    for (int64_t t = t0; t < t1; ++t) { PlaceCM v; v.c = a.tiles[2 * t]; v.m = a.tiles[2 * t + 1]; acc = place_join(acc, v); }
    PlaceCM run = place_block_scan<1024>(acc, nullptr);
    for (int64_t t = t0; t < t1; ++t) {
        PlaceCM v; v.c = a.tiles[2 * t]; v.m = a.tiles[2 * t + 1];
        a.tiles[2 * t] = run.c; a.tiles[2 * t + 1] = run.m;
        run = place_join(run, v);
    }
}
__global__ __launch_bounds__(kBlock) void join_place_kernel(const JoinPlaceArgs a) {         // phases 0 and 2: one tile per block
    constexpr int kPer = kJoinPlaceTile / kBlock;                 // consecutive sorted rows per thread while the positions are computed
    constexpr int kPad = kJoinPlaceTile + kJoinPlaceTile / kPer + 1;     // one word of padding per thread's run: conflict-free in both access orders
    __shared__ uint64_t sk[kPad];
    __shared__ uint32_t spos[kPad], sgap[kPad];
    const int64_t base = (int64_t)blockIdx.x * kJoinPlaceTile;
    const int n = (int)(a.nrv - base < kJoinPlaceTile ? a.nrv - base : kJoinPlaceTile);
    for (int e = threadIdx.x; e < n; e += kBlock) sk[e + e / kPer] = a.rkeys[base + e];
    __syncthreads();
    const int e0 = threadIdx.x * kPer;
    const uint64_t before = e0 == 0 ? (base > 0 ? a.rkeys[base - 1] : 0) : sk[(e0 - 1) + (e0 - 1) / kPer];
    const bool no_prev = e0 == 0 && base == 0;
    uint64_t prev = before;
    PlaceCM loc = place_empty();
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int e = e0 + j;
        if (e >= n) break;
        const uint64_t h = sk[e + e / kPer];
        if ((j == 0 && no_prev) || h != prev) (void)place_push(loc, (long long)(h >> a.tshift));
        prev = h;
    }
    PlaceCM total;
    const PlaceCM ex = place_block_scan<kBlock>(loc, &total);
    if (a.phase == 0) {
        if (threadIdx.x == 0) { a.tiles[2 * blockIdx.x] = total.c; a.tiles[2 * blockIdx.x + 1] = total.m; }
        return;
    }
    PlaceCM carry; carry.c = a.tiles[2 * blockIdx.x]; carry.m = a.tiles[2 * blockIdx.x + 1];
    // this tile owns the slots from the one behind the previous tiles' last key up to its own last key (the last tile: up to the end):
    // it writes the empty ones too, so nobody clears the table beforehand and every line leaves whole
    const long long p0 = place_last(carry) + 1;
    PlaceCM run = place_join(carry, ex);        // {distinct keys before this thread's rows, max(home - index) over them}: absolute
    long long last = place_last(run);                          // slot of the distinct key before this thread's rows
    prev = before;
    bool too_far = false;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int e = e0 + j;
        if (e >= n) break;
        const uint64_t h = sk[e + e / kPer];
        uint32_t off = ~0u, gap = 0;
        if ((j == 0 && no_prev) || h != prev) {
            const long long pos = place_push(run, (long long)(h >> a.tshift));
            if (pos + 1 >= a.cap || pos - p0 >= 0xFFFFFFFFll) too_far = true;       // (the slot behind the last key stays empty: probes end there)
            else { off = (uint32_t)(pos - p0); gap = (uint32_t)(pos - last - 1); }
            last = pos;
        }
        spos[e + e / kPer] = off;
        sgap[e + e / kPer] = gap;
        prev = h;
    }
    if (too_far) atomicOr(a.flags, 1ull);
    __syncthreads();
    // consecutive lanes now take consecutive rows: their slots are neighbours, the stores of a wave fall into a few lines
    const join_u64x2 zero = {0, 0};
    for (int e = threadIdx.x; e < n; e += kBlock) {
        const uint32_t off = spos[e + e / kPer];
        if (off == ~0u) continue;
        const uint64_t h = sk[e + e / kPer];
        const long long pos = p0 + off;
        // the end of the run of equal keys that starts here (galloping, then a binary search: see join_table_kernel)
        const int64_t i = base + e;
        int64_t jn = i + 1;
        if (jn < a.nrv && (e + 1 < n ? sk[(e + 1) + (e + 1) / kPer] : a.rkeys[jn]) == h) {
            int64_t step = 2;
            while (i + step < a.nrv && a.rkeys[i + step] == h) step <<= 1;
            int64_t l = i + (step >> 1), hh = i + step < a.nrv ? i + step : a.nrv;
            while (l + 1 < hh) { const int64_t mid = l + ((hh - l) >> 1); if (a.rkeys[mid] == h) l = mid; else hh = mid; }
            jn = hh;
        }
        const uint32_t info = jn - i == 1 ? (0x80000000u | a.ridx[i]) : (uint32_t)(jn - i);
        join_u64x2 w; w[0] = h; w[1] = ((unsigned long long)i << 32) | info;
        *(join_u64x2*)(a.table + 2 * pos) = w;
        for (uint32_t g = 1, gap = sgap[e + e / kPer]; g <= gap; ++g) *(join_u64x2*)(a.table + 2 * (pos - g)) = zero;
    }
    if (blockIdx.x == gridDim.x - 1) {          // the slots behind the last key: empty up to the end of the table
        const long long end_of_keys = place_last(place_join(carry, total)) + 1;
        for (long long q = end_of_keys + threadIdx.x; q < a.cap; q += kBlock) *(join_u64x2*)(a.table + 2 * q) = zero;
    }
}
// sorted build positions [lo, hi) whose key equals probe row i's key; direct = the build row itself when the key occurs once (table path)
__device__ __forceinline__ void join_range(const JoinProbeArgs& a, int64_t i, int64_t& lo, int64_t& hi, uint32_t& direct) {
    lo = hi = 0;
    direct = 0;
    if (a.lnull && a.lnull[i]) return;
    const uint64_t k = a.lkeys[i];
    if (k < a.kmin || k > a.kmax) return;
    if (a.table) {
        const uint64_t h = k * kJoinMul;
        const uint64_t want = a.hashed ? h : k;
        uint64_t s = h >> a.tshift;
        for (;;) {
            const join_u64x2 e = *(const join_u64x2*)(a.table + 2 * s);
            if (e[1] == 0) return;
            if (a.hashed && e[0] > want) return;     // a scan-placed table holds its keys in hash order: a larger one ends the search
            if (e[0] == want) {
                const uint32_t info = (uint32_t)e[1];
                lo = (int64_t)(e[1] >> 32);
                if (info >> 31) { hi = lo + 1; direct = info; } else hi = lo + info;
                return;
            }
            s = (s + 1) & a.tmask;
        }
    }
    const uint64_t b = (k - a.kmin) >> a.bucket_shift;
    const uint2 se = *(const uint2*)(a.buckets + 2 * b);
    int64_t l = se.x, h = se.y;
    const int64_t e = h;
    while (l < h) { const int64_t m = (l + h) >> 1; if (a.rkeys[m] < k) l = m + 1; else h = m; }
    lo = l;
    h = e;
    while (l < h) { const int64_t m = (l + h) >> 1; if (a.rkeys[m] <= k) l = m + 1; else h = m; }
    hi = l;
}
__global__ __launch_bounds__(kBlock) void join_count_kernel(const JoinProbeArgs a) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.nl; i += (int64_t)gridDim.x * kBlock) {
        int64_t lo, hi;
        uint32_t direct;
        join_range(a, i, lo, hi, direct);
        int64_t c = hi - lo;
        if (a.nkeys > 1) {   // candidates share the tuple's hash: count (and mark) only the ones whose key tuples are equal
            c = 0;
            for (int64_t p = lo; p < hi; ++p)
                if (join_verify(a, i, p)) { ++c; if (a.matched) atomicOr(&a.matched[p >> 5], 1u << (p & 31)); }
        } else if (a.matched) for (int64_t p = lo; p < hi; ++p) atomicOr(&a.matched[p >> 5], 1u << (p & 31));
        a.counts[i] = (c == 0 && a.outer) ? 1 : c;
        a.first[i] = c == 0 ? ~0u : direct ? direct : (uint32_t)lo;   // the write phase does not search again (bit 31: not a position but the one build row itself)
        if (c == 0 && a.outer) atomicAdd(a.unmatched, 1ull);
    }
}
__global__ __launch_bounds__(kBlock) void join_write_kernel(const JoinProbeArgs a) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.nl; i += (int64_t)gridDim.x * kBlock) {
        int64_t o = a.offsets[i];
        const int64_t cnt = a.offsets[i + 1] - o;
        const uint32_t lo = a.first[i];
        if (lo == ~0u) {
            if (a.outer) {
                a.out_probe[o] = (uint32_t)i;
                a.out_build[o] = 0;
                atomicAnd(&a.out_build_validity[o >> 5], ~(1u << (o & 31)));
            }
            continue;
        }
        if (a.table && (lo >> 31)) {   // a key that occurs once on the build side: its row came with the table entry
            a.out_probe[o] = (uint32_t)i;
            a.out_build[o] = lo & 0x7FFFFFFFu;
            continue;
        }
        if (a.nkeys > 1) {   // walk the hash range again and emit the verified pairs (cnt of them, in sorted-position order)
            int64_t left = cnt;
            for (int64_t p = lo; left > 0; ++p)
                if (join_verify(a, i, p)) { a.out_probe[o] = (uint32_t)i; a.out_build[o] = a.ridx[p]; ++o; --left; }
            continue;
        }
        for (int64_t p = lo; p < (int64_t)lo + cnt; ++p, ++o) {
            a.out_probe[o] = (uint32_t)i;
            a.out_build[o] = a.ridx[p];
        }
    }
}
__global__ __launch_bounds__(kBlock) void join_append_kernel(const JoinAppendArgs a) {
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < a.nr; p += (int64_t)gridDim.x * kBlock) {
        const bool un = p >= a.nrv || !((a.matched[p >> 5] >> (p & 31)) & 1);
        if (!un) continue;
        const unsigned long long o = atomicAdd(a.cursor, 1ull);
        if (a.count_only) continue;
        a.out_probe[o] = 0;
        a.out_build[o] = a.ridx[p];
        atomicAnd(&a.out_probe_validity[o >> 5], ~(1u << (o & 31)));
    }
}
__global__ __launch_bounds__(kBlock) void count_bytes_kernel(const uint8_t* p, int64_t n, unsigned long long* out) {
    unsigned long long c = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) c += p[i] != 0;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) c += shfl_xor64(c, m);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

// ------------------------------------------------------------------------------------------------
// hash GROUP BY (Transformation::GroupAggregate, planned by Dataset::try_aggregate src/expression.rs:114-221,
// never executed by the reference: src/evaluation.rs:73 panics).  SQL semantics: NULL keys form one
// group, NULL values are skipped.  One global open-addressing table in HBM (it lives in L2/Infinity
// Cache for the 1e6-group configuration), 64-bit CAS to claim a slot, hardware f64 / u64 atomic adds.

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ int64_t load_key(const DevChunkCol& k, int dt, int64_t row) {
    const int64_t e = k.offset + row;
    switch (dt) {
        case RDF_I32: return as_global<int32_t>(k.values)[e];
        case RDF_U32: return (int64_t)as_global<uint32_t>(k.values)[e];
        case RDF_I16: return as_global<int16_t>(k.values)[e];
        case RDF_U16: return (int64_t)as_global<uint16_t>(k.values)[e];
        case RDF_I8: return as_global<int8_t>(k.values)[e];
        case RDF_U8: return (int64_t)as_global<uint8_t>(k.values)[e];
        default: return as_global<int64_t>(k.values)[e];
    }
}

__global__ __launch_bounds__(kBlock) void groupby_build_kernel(const GroupByArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = wave_id();
    const uint64_t mask = (uint64_t)a.t.capacity - 1;
    uint32_t err = 0;
    for (int64_t tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int64_t c = a.nchunks == 1 ? 0 : find_chunk_tile(a.chunk_tile_start, a.nchunks, tile);
        const int64_t r0 = (tile - a.chunk_tile_start[c]) * kEvalTile;
        const int64_t clen = a.chunk_len[c];
        const DevChunkCol kc = a.keys[c];
        const DevChunkCol vc = a.value_dtype >= 0 ? a.values[c] : DevChunkCol{nullptr, nullptr, 0};
        const int64_t rw = r0 + (int64_t)wave * (kVPT * 64);
        uint64_t kvw[kVPT], vvw[kVPT];
        if (kc.validity) load_windows<kVPT>(kc.validity, kc.offset + rw, clen - rw, kvw);
        if (vc.validity) load_windows<kVPT>(vc.validity, vc.offset + rw, clen - rw, vvw);
#pragma unroll
        for (int j = 0; j < kVPT; ++j) {
            const int64_t row = rw + j * 64 + lane;
            if (row >= clen) continue;
            const bool kvalid = !kc.validity || ((kvw[j] >> lane) & 1);
            const bool vvalid = a.value_dtype < 0 || !vc.validity || ((vvw[j] >> lane) & 1);
            const uint64_t key = (uint64_t)load_key(kc, a.key_dtype, row);
            int64_t slot;
            if (!kvalid) { slot = a.t.capacity + 1; a.t.special[1] = 1; }
            else if (key == kGroupEmpty) { slot = a.t.capacity; a.t.special[0] = 1; }
            else {
                uint64_t s = mix64(key) & mask;
                int64_t probes = 0;
                for (;;) {
                    const unsigned long long old = atomicCAS(&a.t.keys[s], kGroupEmpty, (unsigned long long)key);
                    if (old == kGroupEmpty) { atomicAdd(a.t.ngroups, 1u); break; }
                    if (old == key) break;
                    s = (s + 1) & mask;
                    if (++probes > a.t.capacity) { err |= 4u; break; }
                }
                slot = (int64_t)s;
            }
            if (err) break;
            if (vvalid) {
                if (a.value_dtype == RDF_F64) unsafeAtomicAdd((double*)&a.t.sums[slot], ((const double*)vc.values)[vc.offset + row]);
                else if (a.value_dtype == RDF_F32) unsafeAtomicAdd((double*)&a.t.sums[slot], (double)as_global<float>(vc.values)[vc.offset + row]);
                else if (a.value_dtype >= 0) atomicAdd(&a.t.sums[slot], (unsigned long long)load_key(vc, a.value_dtype, row));
                atomicAdd(&a.t.counts[slot], 1ull);
            }
        }
    }
    if (err) atomicOr(a.t.flags, err);
}

// Insert-or-find `key` in the global table and add (v, cnt) to its slot.  Returns false on overflow.
__device__ __forceinline__ bool global_upsert(const GroupTable& t, uint64_t mask, uint64_t key, bool is_f64, uint64_t v, uint64_t cnt) {
    uint64_t s = mix64(key) & mask;
    int64_t probes = 0;
    for (;;) {
        const unsigned long long old = atomicCAS(&t.keys[s], kGroupEmpty, (unsigned long long)key);
        if (old == kGroupEmpty) { atomicAdd(t.ngroups, 1u); break; }
        if (old == key) break;
        s = (s + 1) & mask;
        if (++probes > t.capacity) return false;
    }
    if (cnt) {
        if (is_f64) unsafeAtomicAdd((double*)&t.sums[s], u2d(v)); else atomicAdd(&t.sums[s], (unsigned long long)v);
        atomicAdd(&t.counts[s], (unsigned long long)cnt);
    }
    return true;
}

// Low-cardinality variant (max_groups <= kLdsGroups/2, e.g. TPC-H Q1's 4 groups): every block
// pre-aggregates into an LDS-resident table (ds_cmpst / ds_add, no HBM atomics per row) and merges its
// <= kLdsGroups partial groups into the global table once at the end.
constexpr int kLdsGroups = 2048;
__global__ __launch_bounds__(kBlock) void groupby_build_lds_kernel(const GroupByArgs a) {
    __shared__ unsigned long long lkeys[kLdsGroups];
    __shared__ unsigned long long lsums[kLdsGroups + 2];
    __shared__ unsigned long long lcnts[kLdsGroups + 2];
    __shared__ unsigned int lspecial[2];
    const int lane = threadIdx.x & 63;
    const int wave = wave_id();
    const uint64_t gmask = (uint64_t)a.t.capacity - 1;
    const bool is_f64 = a.value_dtype == RDF_F64 || a.value_dtype == RDF_F32;
    for (int i = threadIdx.x; i < kLdsGroups + 2; i += kBlock) {
        if (i < kLdsGroups) lkeys[i] = kGroupEmpty;
        lsums[i] = 0; lcnts[i] = 0;
    }
    if (threadIdx.x < 2) lspecial[threadIdx.x] = 0;
    __syncthreads();
    uint32_t err = 0;
    for (int64_t tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int64_t c = a.nchunks == 1 ? 0 : find_chunk_tile(a.chunk_tile_start, a.nchunks, tile);
        const int64_t r0 = (tile - a.chunk_tile_start[c]) * kEvalTile;
        const int64_t clen = a.chunk_len[c];
        const DevChunkCol kc = a.keys[c];
        const DevChunkCol vc = a.value_dtype >= 0 ? a.values[c] : DevChunkCol{nullptr, nullptr, 0};
        const int64_t rw = r0 + (int64_t)wave * (kVPT * 64);
        uint64_t kvw[kVPT], vvw[kVPT];
        if (kc.validity) load_windows<kVPT>(kc.validity, kc.offset + rw, clen - rw, kvw);
        if (vc.validity) load_windows<kVPT>(vc.validity, vc.offset + rw, clen - rw, vvw);
#pragma unroll
        for (int j = 0; j < kVPT; ++j) {
            const int64_t row = rw + j * 64 + lane;
            if (row >= clen) continue;
            const bool kvalid = !kc.validity || ((kvw[j] >> lane) & 1);
            const bool vvalid = a.value_dtype < 0 || !vc.validity || ((vvw[j] >> lane) & 1);
            const uint64_t key = (uint64_t)load_key(kc, a.key_dtype, row);
            uint64_t v = 0;
            if (a.value_dtype == RDF_F64) v = as_global<uint64_t>(vc.values)[vc.offset + row];
            else if (a.value_dtype == RDF_F32) v = d2u((double)as_global<float>(vc.values)[vc.offset + row]);
            else if (a.value_dtype >= 0) v = (uint64_t)load_key(vc, a.value_dtype, row);
            int slot = -1;
            if (!kvalid) { slot = kLdsGroups + 1; lspecial[1] = 1; }
            else if (key == kGroupEmpty) { slot = kLdsGroups; lspecial[0] = 1; }
            else {
                uint32_t s = (uint32_t)mix64(key) & (kLdsGroups - 1);
                for (int probes = 0; probes < 64; ++probes) {
                    const unsigned long long old = atomicCAS(&lkeys[s], kGroupEmpty, (unsigned long long)key);
                    if (old == kGroupEmpty || old == key) { slot = (int)s; break; }
                    s = (s + 1) & (kLdsGroups - 1);
                }
            }
            if (slot >= 0) {
                if (vvalid) {
                    if (is_f64) unsafeAtomicAdd((double*)&lsums[slot], u2d(v)); else atomicAdd(&lsums[slot], (unsigned long long)v);
                    atomicAdd(&lcnts[slot], 1ull);
                }
            } else if (!global_upsert(a.t, gmask, key, is_f64, v, vvalid ? 1 : 0)) err |= 4u;  // LDS table full: straight to HBM
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kLdsGroups + 2; i += kBlock) {
        if (i < kLdsGroups) {
            if (lkeys[i] != kGroupEmpty && !global_upsert(a.t, gmask, lkeys[i], is_f64, lsums[i], lcnts[i])) err |= 4u;
        } else if (lspecial[i - kLdsGroups]) {
            const int64_t gs = a.t.capacity + (i - kLdsGroups);
            a.t.special[i - kLdsGroups] = 1;
            if (lcnts[i]) {
                if (is_f64) unsafeAtomicAdd((double*)&a.t.sums[gs], u2d(lsums[i])); else atomicAdd(&a.t.sums[gs], lsums[i]);
                atomicAdd(&a.t.counts[gs], lcnts[i]);
            }
        }
    }
    if (err) atomicOr(a.t.flags, err);
}

__device__ __forceinline__ void store_key(void* out, int dt, unsigned idx, uint64_t key) {
    switch (dt) {
        case RDF_I32: case RDF_U32: ((uint32_t*)out)[idx] = (uint32_t)key; break;
        case RDF_I16: case RDF_U16: ((uint16_t*)out)[idx] = (uint16_t)key; break;
        case RDF_I8: case RDF_U8: ((uint8_t*)out)[idx] = (uint8_t)key; break;
        default: ((uint64_t*)out)[idx] = key; break;
    }
}

// ---- partitioned GROUP BY (high cardinality) ----
__device__ __forceinline__ uint64_t unmix64(uint64_t z) {  // inverse of mix64 (it is a bijection on 64 bits)
    z ^= z >> 31; z ^= z >> 62;
    z *= 0x319642b2d24d8ec3ull;
    z ^= z >> 27; z ^= z >> 54;
    z *= 0x96de1b173f119089ull;
    z ^= z >> 30; z ^= z >> 60;
    return z;
}
constexpr unsigned long long kHashFree = ~0ull;  // LDS free marker in hashed-key space

// (key, value) columns -> dense (mix64(key), value bits) streams; NULL keys and the one key whose hash is the
// free marker go to two dedicated accumulators.
__global__ __launch_bounds__(kBlock) void groupby_prepare_kernel(const GroupPrepArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = wave_id();
    for (int64_t tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int64_t c = a.nchunks == 1 ? 0 : find_chunk_tile(a.chunk_tile_start, a.nchunks, tile);
        const int64_t r0 = (tile - a.chunk_tile_start[c]) * kEvalTile;
        const int64_t clen = a.chunk_len[c];
        const int64_t g0 = a.chunk_row_start[c];
        const DevChunkCol kc = a.keys[c];
        const DevChunkCol vc = a.value_dtype >= 0 ? a.values[c] : DevChunkCol{nullptr, nullptr, 0};
        const int64_t rw = r0 + (int64_t)wave * (kVPT * 64);
        uint64_t kvw[kVPT];
        if (kc.validity) load_windows<kVPT>(kc.validity, kc.offset + rw, clen - rw, kvw);
#pragma unroll
        for (int j = 0; j < kVPT; ++j) {
            const int64_t row = rw + j * 64 + lane;
            if (row >= clen) continue;
            const bool kvalid = !kc.validity || ((kvw[j] >> lane) & 1);
            uint64_t v = 0;
            if (a.value_dtype == RDF_F64) v = as_global<uint64_t>(vc.values)[vc.offset + row];
            else if (a.value_dtype == RDF_F32) v = d2u((double)as_global<float>(vc.values)[vc.offset + row]);
            else if (a.value_dtype >= 0) v = (uint64_t)load_key(vc, a.value_dtype, row);
            uint64_t hk = mix64((uint64_t)load_key(kc, a.key_dtype, row));
            const bool is_f = a.value_dtype == RDF_F64 || a.value_dtype == RDF_F32;
            if (!kvalid || hk == kHashFree) {
                const int s = kvalid ? 0 : 1;
                a.special[s] = 1;
                if (is_f) unsafeAtomicAdd((double*)&a.special_sums[s], u2d(v)); else atomicAdd(&a.special_sums[s], (unsigned long long)v);
                atomicAdd(&a.special_counts[s], 1ull);
                // park the row in a partition where it is harmless: hashed key 0 with value 0 would create a bogus group,
                // so give it the hash of its own neighbour-free marker and let the aggregation skip it
                hk = kHashFree;
                v = 0;
            }
            a.hkeys[g0 + row] = hk;
            a.vals[g0 + row] = v;
        }
    }
}

// One block per partition (grid-stride): the partition's rows are a contiguous range of the sorted stream, found
// by binary search on the top hash bits; its groups live in an LDS table and are written out directly.
__global__ __launch_bounds__(kBlock) void groupby_partitions_kernel(const GroupAggArgs a) {
    __shared__ unsigned long long lkeys[kLdsGroups];
    __shared__ unsigned long long lsums[kLdsGroups];
    __shared__ unsigned long long lcnts[kLdsGroups];
    __shared__ unsigned int ngroups_s, obase_s;
    const int64_t nparts = (int64_t)1 << a.part_bits;
    const int shift = 64 - a.part_bits;
    uint32_t err = 0;
    for (int64_t p = blockIdx.x; p < nparts; p += gridDim.x) {
        // [lo, hi) = rows whose top bits equal p: lower bounds of p and p+1
        auto lower_bound = [&](uint64_t part) -> int64_t {
            if (part >= (uint64_t)nparts) return a.n;
            int64_t lo = 0, hi = a.n;
            while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((a.hkeys[mid] >> shift) < part) lo = mid + 1; else hi = mid; }
            return lo;
        };
        const int64_t lo = lower_bound((uint64_t)p), hi = lower_bound((uint64_t)p + 1);
        if (hi <= lo) continue;
        for (int i = threadIdx.x; i < kLdsGroups; i += kBlock) { lkeys[i] = kHashFree; lsums[i] = 0; lcnts[i] = 0; }
        if (threadIdx.x == 0) ngroups_s = 0;
        __syncthreads();
        for (int64_t i = lo + threadIdx.x; i < hi; i += kBlock) {
            const uint64_t hk = a.hkeys[i];
            if (hk == kHashFree) continue;  // parked special rows
            const uint64_t v = a.vals[i];
            uint32_t s = (uint32_t)(hk >> 7) & (kLdsGroups - 1);  // bits below the partition bits are still well mixed
            int slot = -1;
            for (int probes = 0; probes < kLdsGroups; ++probes) {
                const unsigned long long old = atomicCAS(&lkeys[s], kHashFree, (unsigned long long)hk);
                if (old == kHashFree) { atomicAdd(&ngroups_s, 1u); slot = (int)s; break; }
                if (old == hk) { slot = (int)s; break; }
                s = (s + 1) & (kLdsGroups - 1);
            }
            if (slot < 0) { err |= 4u; continue; }
            if (a.has_values) {
                if (a.is_f64) unsafeAtomicAdd((double*)&lsums[slot], u2d(v)); else atomicAdd(&lsums[slot], (unsigned long long)v);
            }
            atomicAdd(&lcnts[slot], 1ull);
        }
        __syncthreads();
        if (threadIdx.x == 0) obase_s = atomicAdd(a.cursor, ngroups_s);
        __syncthreads();
        if (threadIdx.x == 0) ngroups_s = 0;  // reused as the local emit cursor
        __syncthreads();
        for (int i = threadIdx.x; i < kLdsGroups; i += kBlock) {
            if (lkeys[i] == kHashFree) continue;
            const unsigned idx = obase_s + atomicAdd(&ngroups_s, 1u);
            if ((int64_t)idx >= a.max_out) { err |= 4u; continue; }
            store_key(a.out_keys, a.key_dtype, idx, unmix64(lkeys[i]));
            ((uint64_t*)a.out_sums)[idx] = lsums[i];
            a.out_counts[idx] = (int64_t)lcnts[i];
        }
        __syncthreads();
    }
    if (err) atomicOr(a.flags, err);
}

// ---- single-pass partitioned GROUP BY ----
// A block iteration covers one super-tile: 4 tiles of kEvalTile rows (each inside one chunk), thread t takes the rows
// q = j * kGbBlock + t.  All loads of the iteration are issued before anything depends on them (one row at a time
// would leave a single 8-byte load in flight per lane: latency-bound at a quarter of the bandwidth).
//
// Record format (16 bytes): word 0 = (cnt << 55) | (hashed key & (2^55 - 1)), word 1 = sum bits.  The top 9 bits of the
// hashed key are the partition number — known from where the record lies — so they carry cnt, the number of non-NULL
// values summed into word 1: 1 for an ordinary row, 0 for a row whose value is NULL (the group must still exist),
// up to kGbMaxCnt when the scatter pass has combined equal keys of one super-tile (skewed inputs).  ~0 = dead record.
// The partition hash: a bijection (the aggregation pass inverts it to emit the keys) built from ONE 64-bit multiply between
// two xor-folds.  The SplitMix64 finaliser used elsewhere needs two multiplies — 8 quarter-rate 32-bit multiplies per row,
// about a millisecond per 1e9 rows in each of the two passes that hash every row.
__device__ __forceinline__ uint64_t gb_hash(uint64_t x) { x ^= x >> 32; x *= 0x9E3779B97F4A7C15ull; return x ^ (x >> 32); }
__device__ __forceinline__ uint64_t gb_unhash(uint64_t x) { x ^= x >> 32; x *= 0xF1DE83E19937733Dull; return x ^ (x >> 32); }
constexpr uint64_t kGbKeyMask = (1ull << (64 - kGbPartBits)) - 1;
constexpr unsigned kGbMaxCnt = 510;
constexpr uint64_t kGbDead = ~0ull;
struct GbBatch {
    uint64_t key[kGbRows], val[kGbRows];   // raw key (sign-/zero-extended), value bits
    uint32_t exists, knull, vnull;         // bit j: row j exists / its key is NULL / its value is NULL
};
__device__ __forceinline__ int gb_dtype_size(int dt) {
    switch (dt) {
        case RDF_I8: case RDF_U8: return 1;
        case RDF_I16: case RDF_U16: return 2;
        case RDF_I32: case RDF_U32: case RDF_F32: return 4;
        default: return 8;
    }
}
// raw element `row` of a column of `size`-byte elements, zero-extended; `size` is block-uniform, so the branch is
// scalar and the loads of consecutive calls stay back to back
__device__ __forceinline__ uint64_t gb_load_raw(const void* base, int size, int64_t idx, bool pred) {
    uint64_t x = 0;
    if (size == 8) { if (pred) x = as_global<uint64_t>(base)[idx]; }
    else if (size == 4) { if (pred) x = as_global<uint32_t>(base)[idx]; }
    else if (size == 2) { if (pred) x = as_global<uint16_t>(base)[idx]; }
    else { if (pred) x = as_global<uint8_t>(base)[idx]; }
    return x;
}
template <bool WITH_VALUES>
__device__ __forceinline__ void gb_load_batch(const GbPartArgs& a, int64_t st, int64_t tile_end, int tid, GbBatch& b) {
    static_assert(kGbBlock * 2 == kEvalTile && kGbRows == 8, "thread t holds rows t and t + 512 of each of the 4 tiles");
    b.exists = 0; b.knull = 0; b.vnull = 0;
    const int ksz = gb_dtype_size(a.key_dtype), vsz = gb_dtype_size(a.value_dtype);
    // (1) where the 4 tiles live.  One chunk: the descriptors sit in the kernel arguments (scalar registers, no memory
    // access).  Several: every table entry is fetched BEFORE the first data load — a table read between two data loads
    // makes the wave wait for everything in flight (measured: 14 us per iteration with the lookups interleaved).
    DevChunkCol kc[kGbRows / 2], vc[kGbRows / 2];
    int64_t r0[kGbRows / 2], clen[kGbRows / 2];
#pragma unroll
    for (int tt = 0; tt < kGbRows / 2; ++tt) {
        const int64_t tile = st + tt;                 // block-uniform
        kc[tt] = DevChunkCol{nullptr, nullptr, 0}; vc[tt] = kc[tt]; r0[tt] = 0; clen[tt] = 0;
        if (tile >= tile_end) continue;
        if (a.nchunks == 1) { kc[tt] = a.key0; vc[tt] = a.val0; r0[tt] = tile * kEvalTile; clen[tt] = a.len0; }
        else {
            const int64_t c = find_chunk_tile(a.chunk_tile_start, a.nchunks, tile);
            r0[tt] = (tile - a.chunk_tile_start[c]) * kEvalTile;
            clen[tt] = a.chunk_len[c];
            kc[tt] = a.keys[c];
            if (WITH_VALUES && a.value_dtype >= 0) vc[tt] = a.values[c];
        }
    }
    // (2) every data load of the iteration, back to back
    uint32_t kb[kGbRows], vb[kGbRows];
    int kbit[kGbRows], vbit[kGbRows];
#pragma unroll
    for (int tt = 0; tt < kGbRows / 2; ++tt) {
        const int j0 = 2 * tt, j1 = 2 * tt + 1;
        kb[j0] = kb[j1] = vb[j0] = vb[j1] = 0xFFu; kbit[j0] = kbit[j1] = vbit[j0] = vbit[j1] = 0;
        const int64_t row0 = r0[tt] + tid, row1 = r0[tt] + kGbBlock + tid;
        const bool e0 = row0 < clen[tt], e1 = row1 < clen[tt];
        b.exists |= ((uint32_t)e0 << j0) | ((uint32_t)e1 << j1);
        b.key[j0] = gb_load_raw(kc[tt].values, ksz, kc[tt].offset + row0, e0);
        b.key[j1] = gb_load_raw(kc[tt].values, ksz, kc[tt].offset + row1, e1);
        if (kc[tt].validity) {
            const int64_t b0 = kc[tt].offset + row0, b1 = kc[tt].offset + row1;
            if (e0) kb[j0] = as_global<uint8_t>(kc[tt].validity)[b0 >> 3];
            if (e1) kb[j1] = as_global<uint8_t>(kc[tt].validity)[b1 >> 3];
            kbit[j0] = (int)(b0 & 7); kbit[j1] = (int)(b1 & 7);
        }
        b.val[j0] = b.val[j1] = 0;
        if (WITH_VALUES && a.value_dtype >= 0) {
            b.val[j0] = gb_load_raw(vc[tt].values, vsz, vc[tt].offset + row0, e0);
            b.val[j1] = gb_load_raw(vc[tt].values, vsz, vc[tt].offset + row1, e1);
            if (vc[tt].validity) {
                const int64_t b0 = vc[tt].offset + row0, b1 = vc[tt].offset + row1;
                if (e0) vb[j0] = as_global<uint8_t>(vc[tt].validity)[b0 >> 3];
                if (e1) vb[j1] = as_global<uint8_t>(vc[tt].validity)[b1 >> 3];
                vbit[j0] = (int)(b0 & 7); vbit[j1] = (int)(b1 & 7);
            }
        }
    }
    // (3) nothing before this point consumed a data load
#pragma unroll
    for (int j = 0; j < kGbRows; ++j) {
        b.key[j] = normalize_int(a.key_dtype, b.key[j]);
        b.knull |= (uint32_t)(((kb[j] >> kbit[j]) & 1u) == 0) << j;
        if (WITH_VALUES) {
            b.vnull |= (uint32_t)(((vb[j] >> vbit[j]) & 1u) == 0) << j;
            if (a.value_dtype == RDF_F32) b.val[j] = d2u((double)__uint_as_float((uint32_t)b.val[j]));
            else if (a.value_dtype >= 0 && a.value_dtype != RDF_F64) b.val[j] = normalize_int(a.value_dtype, b.val[j]);
            if ((b.vnull >> j) & 1) b.val[j] = 0;
        }
    }
}

__global__ __launch_bounds__(kGbBlock) void gb_hist_kernel(const GbPartArgs a) {
    constexpr int P = 1 << kGbPartBits;
    __shared__ unsigned int lc[P];
    for (int i = threadIdx.x; i < P; i += kGbBlock) lc[i] = 0;
    __syncthreads();
    // super-tiles are dealt round-robin (block b takes b, b + grid, ...); any assignment works as long as the
    // histogram and the scatter agree (measured: no faster or slower than contiguous per-block ranges)
    const int64_t t1 = a.ntiles;
    // two of the block's super-tiles per iteration: 16 key loads in flight per lane (8 measured 3.6 TB/s)
    const int64_t stride = (int64_t)gridDim.x * (kGbSuper / kEvalTile);
    for (int64_t st = (int64_t)blockIdx.x * (kGbSuper / kEvalTile); st < t1; st += 2 * stride) {
        GbBatch b0, b1;
        gb_load_batch<false>(a, st, t1, (int)threadIdx.x, b0);
        gb_load_batch<false>(a, st + stride, t1, (int)threadIdx.x, b1);
#pragma unroll
        for (int j = 0; j < kGbRows; ++j) {
            const uint64_t h0 = gb_hash(b0.key[j]), h1 = gb_hash(b1.key[j]);
            if (((b0.exists & ~b0.knull) >> j & 1) && h0 != kHashFree) atomicAdd(&lc[(unsigned)(h0 >> (64 - kGbPartBits))], 1u);
            if (((b1.exists & ~b1.knull) >> j & 1) && h1 != kHashFree) atomicAdd(&lc[(unsigned)(h1 >> (64 - kGbPartBits))], 1u);
        }
    }
    __syncthreads();
    for (int d = threadIdx.x; d < P; d += kGbBlock) a.hist[(int64_t)d * gridDim.x + blockIdx.x] = (int64_t)lc[d];
}

// Skew detector (one block): *flag = 1 when the largest partition holds more than 4x the average.  Skewed inputs
// (Zipf keys: SURVEY.md §8d C4 variant) take the scatter variant that combines equal keys inside a super-tile.
__global__ __launch_bounds__(kGbBlock) void gb_skew_kernel(const int64_t* scan, int64_t nblocks, unsigned int* flag) {
    constexpr int P = 1 << kGbPartBits;
    __shared__ unsigned long long mx;
    if (threadIdx.x == 0) mx = 0;
    __syncthreads();
    const int p = threadIdx.x;
    const unsigned long long sz = (unsigned long long)(scan[(int64_t)(p + 1) * nblocks] - scan[(int64_t)p * nblocks]);
    atomicMax(&mx, sz);
    __syncthreads();
    if (threadIdx.x == 0) { const unsigned long long total = (unsigned long long)scan[(int64_t)P * nblocks]; *flag = (total > 65536 && mx * P > 4 * total) ? 1u : 0u; }
}

constexpr int kGbBatch = 4;     // records per lane per aggregation step (loads in flight, probes interleaved); 8 measured slower (9.2 vs 5.8 ms: registers)
constexpr int kGbCache = 512;   // DEDUP: direct-mapped LDS cache of keys seen in the current super-tile
template <bool DEDUP>
__global__ __launch_bounds__(kGbBlock) void gb_scatter_kernel(const GbPartArgs a) {
    constexpr int P = 1 << kGbPartBits;
    static_assert(P == kGbBlock, "one thread per partition counter");
    extern __shared__ __attribute__((aligned(16))) uint64_t gsm[];
    uint64_t* skey = gsm;                       // [kGbSuper] staged records, grouped by partition (full hashed key)
    uint64_t* sval = gsm + kGbSuper;            // [kGbSuper]
    int64_t* gbase = (int64_t*)(gsm + 2 * kGbSuper);              // [P] next output record of (partition, this block)
    unsigned int* lcount = (unsigned int*)(gbase + P);            // [P] records of the partition in this iteration
    unsigned short* lstart = (unsigned short*)(lcount + P);       // [P] their first staging slot
    unsigned short* scnt = lstart + P;                            // [kGbSuper] cnt of the staged record
    unsigned long long* ckey = (unsigned long long*)(scnt + kGbSuper);   // DEDUP [kGbCache] hashed key owning the slot (kHashFree = empty)
    unsigned long long* cval = ckey + kGbCache;                   // DEDUP [kGbCache] sum of the absorbed rows' values
    unsigned int* ccnt = (unsigned int*)(cval + kGbCache);        // DEDUP [kGbCache] non-NULL values among them
    __shared__ unsigned int wave_tot[kGbBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    gbase[tid] = a.hist[(int64_t)tid * gridDim.x + blockIdx.x];
    lcount[tid] = 0;
    if (DEDUP) for (int i = tid; i < kGbCache; i += kGbBlock) { ckey[i] = kHashFree; cval[i] = 0; ccnt[i] = 0; }
    __syncthreads();
    const bool is_f = a.value_dtype == RDF_F64 || a.value_dtype == RDF_F32;
    const bool counts_rows = a.value_dtype < 0;   // no value column: cnt = rows
    typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
    const int64_t t1 = a.ntiles;
    for (int64_t st = (int64_t)blockIdx.x * (kGbSuper / kEvalTile); st < t1; st += (int64_t)gridDim.x * (kGbSuper / kEvalTile)) {
        // (A) load everything, then hash; specials go to global accumulators, never staged
        GbBatch b;
        gb_load_batch<true>(a, st, t1, tid, b);
        unsigned int rank[kGbRows], cnt[kGbRows];
        int cslot[kGbRows];
        uint32_t live = 0;   // rows that will be emitted as records
#pragma unroll
        for (int j = 0; j < kGbRows; ++j) {
            rank[j] = ~0u; cslot[j] = -1;
            cnt[j] = (counts_rows || !((b.vnull >> j) & 1)) ? 1u : 0u;
            if (!((b.exists >> j) & 1)) continue;
            const uint64_t hk = gb_hash(b.key[j]);
            b.key[j] = hk;
            const bool knull = (b.knull >> j) & 1;
            if (knull || hk == kHashFree) {
                const int s = knull ? 1 : 0;
                a.special[s] = 1;
                if (cnt[j]) {
                    if (a.value_dtype >= 0) { if (is_f) unsafeAtomicAdd((double*)&a.special_sums[s], u2d(b.val[j])); else atomicAdd(&a.special_sums[s], (unsigned long long)b.val[j]); }
                    atomicAdd(&a.special_counts[s], 1ull);
                }
                continue;
            }
            live |= 1u << j;
            if (DEDUP) {   // equal keys of this super-tile collapse into the record of the first row that claimed the slot
                const int s = (int)((hk >> 24) & (kGbCache - 1));
                const unsigned long long old = atomicCAS(&ckey[s], kHashFree, (unsigned long long)hk);
                if (old == kHashFree) cslot[j] = s;   // owner: emits the combined record below
                else if (old == hk) {
                    if (cnt[j]) {
                        if (a.value_dtype >= 0) { if (is_f) unsafeAtomicAdd((double*)&cval[s], u2d(b.val[j])); else atomicAdd(&cval[s], (unsigned long long)b.val[j]); }
                        atomicAdd(&ccnt[s], 1u);
                    }
                    live &= ~(1u << j);   // absorbed
                }
            }
        }
        if (DEDUP) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < kGbRows; ++j)
                if (cslot[j] >= 0) {   // fold what the slot absorbed into the owner's record, free the slot for the next super-tile
                    const int s = cslot[j];
                    b.val[j] = is_f ? d2u(u2d(b.val[j]) + u2d(cval[s])) : b.val[j] + cval[s];
                    cnt[j] += ccnt[s];
                    ckey[s] = kHashFree; cval[s] = 0; ccnt[s] = 0;
                }
        }
        // rank inside the partition (LDS atomics: 512 counters, random digits -> little contention).  A combined record
        // whose cnt does not fit the 9-bit field is continued by (0-valued) records carrying the rest of the count.
        if (a.ablate_stores == 8) { uint64_t x = 0; for (int j = 0; j < kGbRows; ++j) x ^= b.key[j] ^ b.val[j]; if (x == 0x1234567) a.special[0] = 1; continue; }   // ablation: loads + hash only
#pragma unroll
        for (int j = 0; j < kGbRows; ++j)
            if ((live >> j) & 1) {
                const unsigned int nrec = cnt[j] <= kGbMaxCnt ? 1u : (cnt[j] + kGbMaxCnt - 1) / kGbMaxCnt;
                rank[j] = atomicAdd(&lcount[(unsigned)(b.key[j] >> (64 - kGbPartBits))], nrec);
            }
        __syncthreads();
        // (B) exclusive scan of the P counters (one per thread)
        const unsigned int pc = lcount[tid];
        unsigned int inc = pc;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) { const unsigned int y = (unsigned int)__shfl_up((int)inc, m); if (lane >= m) inc += y; }
        if (lane == 63) wave_tot[wave] = inc;
        __syncthreads();
        unsigned int woff = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kGbBlock / 64; ++w) { if (w < wave) woff += wave_tot[w]; total += wave_tot[w]; }
        lstart[tid] = (unsigned short)(woff + inc - pc);
        __syncthreads();
        // (C) stage the records grouped by partition
#pragma unroll
        for (int j = 0; j < kGbRows; ++j)
            if (rank[j] != ~0u && a.ablate_stores != 7) {
                unsigned int pos = lstart[(unsigned)(b.key[j] >> (64 - kGbPartBits))] + rank[j];
                unsigned int left = cnt[j];
                skey[pos] = b.key[j];
                sval[pos] = b.val[j];
                scnt[pos] = (unsigned short)(left < kGbMaxCnt ? left : kGbMaxCnt);
                while (left > kGbMaxCnt) {   // (combining variant only)
                    left -= kGbMaxCnt;
                    ++pos;
                    skey[pos] = b.key[j];
                    sval[pos] = 0;
                    scnt[pos] = (unsigned short)(left < kGbMaxCnt ? left : kGbMaxCnt);
                }
            }
        __syncthreads();
        // (D) write them out: consecutive staging slots of one partition are consecutive output records; the partition bits
        // of the key make room for cnt
        for (unsigned int i = tid; i < total && a.ablate_stores != 7; i += kGbBlock) {
            const uint64_t k = skey[i];
            const unsigned int d = (unsigned int)(k >> (64 - kGbPartBits));
            u64x2 rec;
            rec[0] = ((uint64_t)scnt[i] << (64 - kGbPartBits)) | (k & kGbKeyMask);
            rec[1] = sval[i];
            if (a.ablate_stores != 2) ((u64x2*)a.recs)[gbase[d] + (int64_t)(i - lstart[d])] = rec;   // (nontemporal stores measured slower: 13.6 vs 10.7 ms)
        }
        __syncthreads();
        // (E) advance the block's output positions
        gbase[tid] += (int64_t)pc;
        lcount[tid] = 0;
        __syncthreads();
    }
    // absorbed rows were counted by the histogram, so a (partition, block) range may end early: the aggregation pass
    // walks the ranges by their real lengths
    if (DEDUP) a.emitted[(int64_t)tid * gridDim.x + blockIdx.x] = gbase[tid] - a.hist[(int64_t)tid * gridDim.x + blockIdx.x];
}

// One block per partition: its records are the contiguous range [scan[p * nblocks], scan[(p + 1) * nblocks]).
__global__ __launch_bounds__(kGbBlock) void gb_aggregate_kernel(const GbAggArgs a) {
    constexpr int P = 1 << kGbPartBits;
    extern __shared__ __attribute__((aligned(16))) uint64_t gsm[];
    unsigned long long* lkeys = (unsigned long long*)gsm;                 // [kGbSlots]
    unsigned long long* lsums = lkeys + kGbSlots;                         // [kGbSlots]
    unsigned int* lcnts = (unsigned int*)(lsums + kGbSlots);              // [kGbSlots]
    unsigned int* misc = lcnts + kGbSlots;                                // [0] groups, [1] output base
    typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
    uint32_t err = 0;
    for (int p = blockIdx.x; p < P; p += gridDim.x) {
        const int64_t lo = a.scan[(int64_t)p * a.nblocks], hi = a.scan[(int64_t)(p + 1) * a.nblocks];
        if (hi <= lo) continue;
        for (int i = threadIdx.x; i < kGbSlots; i += kGbBlock) { lkeys[i] = kHashFree; lsums[i] = 0; lcnts[i] = 0; }
        if (threadIdx.x == 0) misc[0] = 0;
        __syncthreads();
        uint64_t dbg_acc = 0;
        const uint64_t ptop = (uint64_t)p << (64 - kGbPartBits);
        auto upsert = [&](const u64x2 rec) {
            if (rec[0] == kGbDead) return;
            const unsigned int cnt = (unsigned int)(rec[0] >> (64 - kGbPartBits));
            const uint64_t hk = (rec[0] & kGbKeyMask) | ptop;
            if (a.ablate_lds == 1) { dbg_acc ^= hk ^ rec[1]; return; }
            // slot from the 32 bits right below the partition bits (the best-mixed bits of a multiplicative hash); multiply-shift range reduction
            uint32_t s = (uint32_t)(((uint64_t)(uint32_t)(hk >> (32 - kGbPartBits)) * (uint64_t)kGbSlots) >> 32);
            const uint32_t step = 1u + (uint32_t)(((uint64_t)(uint32_t)(hk >> 3) * (uint64_t)(kGbSlots - 1)) >> 32);
            int slot = -1;
            if (a.ablate_lds == 6) slot = (int)s;   // ablation: no key table
            else for (int probes = 0; probes < kGbSlots; ++probes) {
                unsigned long long old = lkeys[s];
                if (old != hk) {
                    if (old == kHashFree) {
                        old = atomicCAS(&lkeys[s], kHashFree, (unsigned long long)hk);
                        if (old == kHashFree) { atomicAdd(&misc[0], 1u); old = hk; }
                    }
                    if (old != hk) { s += step; if (s >= (uint32_t)kGbSlots) s -= (uint32_t)kGbSlots; continue; }
                }
                slot = (int)s;
                break;
            }
            if (slot < 0) { err |= 4u; return; }
            if (cnt == 0) return;   // a NULL value: the group exists, nothing to add
            if (a.has_values && a.ablate_lds != 5) {
                if (a.is_f64) unsafeAtomicAdd((double*)&lsums[slot], u2d(rec[1])); else atomicAdd(&lsums[slot], (unsigned long long)rec[1]);
            }
            if (a.ablate_lds != 4) atomicAdd(&lcnts[slot], cnt);
        };
        // The key probe, not the atomics, is what the table costs (ablation: without the sum or the count atomic the pass
        // takes the same 7.6 ms, without the key table 2.85 ms): every probe is a dependent LDS round trip, and a wave
        // keeps looping until its unluckiest lane has found its key.  So (1) the probes of the 4 records of a batch run
        // interleaved — one round trip serves up to 4 pending records per lane — and (2) collisions step by a second hash
        // (the table size is prime), which cuts the long clusters linear probing builds at a load of 0.5.
        auto upsert_batch = [&](const u64x2 (&rec)[kGbBatch]) {
            if (a.ablate_lds) {
#pragma unroll
                for (int u = 0; u < kGbBatch; ++u) upsert(rec[u]);
                return;
            }
            uint64_t hk[kGbBatch];
            uint32_t s[kGbBatch], step[kGbBatch];
            uint32_t pending = 0;
#pragma unroll
            for (int u = 0; u < kGbBatch; ++u) {
                hk[u] = (rec[u][0] & kGbKeyMask) | ptop;
                s[u] = (uint32_t)(((uint64_t)(uint32_t)(hk[u] >> (32 - kGbPartBits)) * (uint64_t)kGbSlots) >> 32);
                step[u] = 1u + (uint32_t)(((uint64_t)(uint32_t)(hk[u] >> 3) * (uint64_t)(kGbSlots - 1)) >> 32);
                if (rec[u][0] != kGbDead) pending |= 1u << u;
            }
            int guard = 0;
            while (__any(pending != 0)) {
                unsigned long long old[kGbBatch];
#pragma unroll
                for (int u = 0; u < kGbBatch; ++u) old[u] = ((pending >> u) & 1) ? lkeys[s[u]] : 0;
#pragma unroll
                for (int u = 0; u < kGbBatch; ++u) {
                    if (!((pending >> u) & 1)) continue;
                    unsigned long long o = old[u];
                    if (o == kHashFree) {
                        o = atomicCAS(&lkeys[s[u]], kHashFree, (unsigned long long)hk[u]);
                        if (o == kHashFree) { atomicAdd(&misc[0], 1u); o = hk[u]; }
                    }
                    if (o == hk[u]) {
                        pending &= ~(1u << u);
                        const unsigned int cnt = (unsigned int)(rec[u][0] >> (64 - kGbPartBits));
                        if (cnt) {
                            if (a.has_values) {
                                if (a.is_f64) unsafeAtomicAdd((double*)&lsums[s[u]], u2d(rec[u][1])); else atomicAdd(&lsums[s[u]], (unsigned long long)rec[u][1]);
                            }
                            atomicAdd(&lcnts[s[u]], cnt);
                        }
                    } else {
                        s[u] += step[u];
                        if (s[u] >= (uint32_t)kGbSlots) s[u] -= (uint32_t)kGbSlots;
                    }
                }
                if (++guard > kGbSlots) { if (pending) err |= 4u; break; }   // table full: more groups than promised
            }
        };
        const u64x2* recs = (const u64x2*)a.recs;
        if (a.emitted) {   // combined (skewed) inputs: per-(partition, block) sub-ranges with their real lengths, two at a time
            for (int64_t b = 0; b < a.nblocks; b += 2) {
                const int64_t s0 = a.scan[(int64_t)p * a.nblocks + b], n0 = a.emitted[(int64_t)p * a.nblocks + b];
                const bool two = b + 1 < a.nblocks;
                const int64_t s1 = two ? a.scan[(int64_t)p * a.nblocks + b + 1] : 0, n1 = two ? a.emitted[(int64_t)p * a.nblocks + b + 1] : 0;
                const int64_t nmax = n0 > n1 ? n0 : n1;
                for (int64_t i = threadIdx.x; i < nmax; i += kGbBlock) {
                    u64x2 r0, r1;
                    r0[0] = r1[0] = kGbDead; r0[1] = r1[1] = 0;
                    if (i < n0) r0 = __builtin_nontemporal_load(recs + s0 + i);
                    if (i < n1) r1 = __builtin_nontemporal_load(recs + s1 + i);
                    upsert(r0); upsert(r1);
                }
            }
        }
        int64_t i = a.emitted ? hi : lo + threadIdx.x;
        // kGbBatch independent 16-byte loads per lane, and the NEXT batch is issued before the current one is folded into LDS
        constexpr int B = kGbBatch;
        u64x2 cur[B], nxt[B];
        bool have = i + (B - 1) * kGbBlock < hi;
        if (have) {
#pragma unroll
            for (int u = 0; u < B; ++u) cur[u] = __builtin_nontemporal_load(recs + i + u * kGbBlock);
        }
        while (have) {
            const int64_t ni = i + B * kGbBlock;
            const bool nhave = ni + (B - 1) * kGbBlock < hi;
            if (nhave) {
#pragma unroll
                for (int u = 0; u < B; ++u) nxt[u] = __builtin_nontemporal_load(recs + ni + u * kGbBlock);
            }
            upsert_batch(cur);
#pragma unroll
            for (int u = 0; u < B; ++u) cur[u] = nxt[u];
            i = ni;
            have = nhave;
        }
        for (; i < hi; i += kGbBlock) upsert(__builtin_nontemporal_load(recs + i));
        if (a.ablate_lds == 1 && dbg_acc == 0x1234567) err |= 8u;   // keeps the loads alive
        __syncthreads();
        if (threadIdx.x == 0) { misc[1] = atomicAdd(a.cursor, misc[0]); misc[0] = 0; }
        __syncthreads();
        for (int k = threadIdx.x; k < kGbSlots; k += kGbBlock) {
            if (lkeys[k] == kHashFree) continue;
            const unsigned idx = misc[1] + atomicAdd(&misc[0], 1u);
            if ((int64_t)idx >= a.max_out) { err |= 4u; continue; }
            store_key(a.out_keys, a.key_dtype, idx, gb_unhash(lkeys[k]));
            ((uint64_t*)a.out_sums)[idx] = lsums[k];
            a.out_counts[idx] = (int64_t)lcnts[k];
        }
        __syncthreads();
    }
    if (err) atomicOr(a.flags, err);
}

// Occupied slots -> dense outputs (order = claim order of the output cursor, i.e. unspecified).
__global__ __launch_bounds__(kBlock) void groupby_emit_kernel(const GroupEmitArgs a) {
    const int64_t n = a.t.capacity + 2;
    for (int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x; s < n; s += (int64_t)gridDim.x * kBlock) {
        bool occ; uint64_t key = 0; bool knull = false;
        if (s < a.t.capacity) { key = a.t.keys[s]; occ = key != kGroupEmpty; }
        else if (s == a.t.capacity) { key = kGroupEmpty; occ = a.t.special[0] != 0; }
        else { knull = true; occ = a.t.special[1] != 0; }
        if (!occ) continue;
        const unsigned idx = atomicAdd(a.cursor, 1u);
        store_key(a.out_keys, a.key_dtype, idx, key);
        if (a.out_keys_validity) {
            if (!knull) atomicOr((unsigned int*)a.out_keys_validity + (idx >> 5), 1u << (idx & 31));
        }
        ((uint64_t*)a.out_sums)[idx] = a.t.sums[s];
        a.out_counts[idx] = (int64_t)a.t.counts[s];
    }
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
        int cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        // 8 blocks of 4 waves = the CU's 32-wave capacity: 2048 on the 256-CU MI355X in SPX mode.  The blocks-per-CU choices of
        // run_program (and the XCD-contiguous tile walk: blocks are dealt to 8 XCDs round-robin) were measured on that part; a
        // partitioned (CPX / DPX) or down-binned device gets a grid in proportion to ITS CUs, the same ratios, and a walk that is
        // merely a permutation there — nothing breaks, the fractions of DESIGN.md are not claimed for it.
        limit = cus > 0 ? cus * 8 : 2048;
        // (A/B of persistent grid sizes on every kernel that sizes its grid from this limit: RDF_GRID_BLOCKS_PER_CU=7 / 5 / ...)
        if (const char* e = getenv("RDF_GRID_BLOCKS_PER_CU")) { const int m = atoi(e); if (m >= 1 && m <= 8 && cus > 0) limit = cus * m; }
    }
    return limit;
}

hipError_t launch_eval_feat0(const EvalArgs& a, int sink, int grid, hipStream_t s);
hipError_t launch_eval_feat1(const EvalArgs& a, int sink, int grid, hipStream_t s);
hipError_t launch_eval_feat2(const EvalArgs& a, int sink, int grid, hipStream_t s);
hipError_t launch_eval(const EvalArgs& a, int sink, int feat, int grid, hipStream_t s) {
    if (feat <= 0) return launch_eval_feat0(a, sink, grid, s);
    if (feat == 1) return launch_eval_feat1(a, sink, grid, s);
    return launch_eval_feat2(a, sink, grid, s);
}

// SINK_GROUP: per-block tables folded by one wave per table word (lane l takes blocks l, l+64, ...; then a
// butterfly): a fixed order, so the result is deterministic given the per-block tables
__global__ __launch_bounds__(64) void group_final_kernel(const GroupFinalArgs a) {
    const int w = blockIdx.x, lane = threadIdx.x;
    const int S = a.ngroups + 1;
    const bool fsum = w < a.nvalues * S && a.value_cls[(w / S) & (kMaxGroupValues - 1)] == CLS_F64;
    uint64_t acc = 0;
    if (fsum) {
        double d = 0.0;
        for (int b = lane; b < a.nblocks; b += 64) d += u2d(a.partials[(size_t)b * a.words + w]);
        acc = d2u(d);
    } else {
        for (int b = lane; b < a.nblocks; b += 64) acc += a.partials[(size_t)b * a.words + w];
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const uint64_t y = shfl_xor64(acc, m);
        acc = fsum ? d2u(u2d(acc) + u2d(y)) : acc + y;
    }
    if (lane == 0) a.result[w] = acc;
}
hipError_t launch_group_final(const GroupFinalArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(group_final_kernel, dim3(a.words), dim3(64), 0, s, a);
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

hipError_t launch_mask_count_one(const DevChunkCol& mask, int64_t clen, int64_t ntiles, int64_t* tile_counts, hipStream_t s) {
    const int64_t grid = ntiles < (int64_t)eval_grid_limit() ? ntiles : (int64_t)eval_grid_limit();
    if (grid > 0) hipLaunchKernelGGL(mask_count_one_kernel, dim3((unsigned)grid), dim3(kBlock), 0, s, mask, clen, ntiles, tile_counts);
    return hipGetLastError();
}
hipError_t launch_mask_count(const MaskTables& t, int tile_rows, int64_t* tile_counts, hipStream_t s) {
    int64_t grid = t.ntiles < (int64_t)eval_grid_limit() ? t.ntiles : (int64_t)eval_grid_limit();
    if (grid > 0) {
        if (tile_rows == kFilterTileSmall) hipLaunchKernelGGL(mask_count_kernel<4>, dim3((unsigned)grid), dim3(kBlock), 0, s, t, tile_counts);
        else hipLaunchKernelGGL(mask_count_kernel<16>, dim3((unsigned)grid), dim3(kBlock), 0, s, t, tile_counts);
    }
    return hipGetLastError();
}
// `scratch` holds ceil(n / 4096) + 1 int64 words.
hipError_t launch_scan(const int64_t* counts, int64_t* scan, int64_t n, int64_t* scratch, hipStream_t s) {
    const int64_t nseg = (n + kScanSeg - 1) / kScanSeg;
    if (nseg > 0) hipLaunchKernelGGL(scan_segments_kernel, dim3((unsigned)nseg), dim3(kScanThreads), 0, s, counts, scan, n, scratch);
    hipLaunchKernelGGL(scan_totals_kernel, dim3(1), dim3(kScanThreads), 0, s, scratch, nseg, scan + n);
    if (nseg > 1) hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)nseg), dim3(kScanThreads), 0, s, scan, n, scratch);
    return hipGetLastError();
}
int64_t scan_scratch_words(int64_t n) { return (n + kScanSeg - 1) / kScanSeg + 2; }
hipError_t launch_compact(const FilterArgs& a, int tile_rows, hipStream_t s) {
    int64_t grid = a.t.ntiles < (int64_t)eval_grid_limit() ? a.t.ntiles : (int64_t)eval_grid_limit();
    if (grid <= 0) return hipSuccess;
    int es = a.esize[0];
    for (int k = 1; k < a.ncols; ++k) if (a.esize[k] != es) es = 0;
#define RDF_COMPACT_LAUNCH(WW)                                                                                          \
    switch (es) {                                                                                                       \
        case 8: hipLaunchKernelGGL((compact_kernel<8, WW>), dim3((unsigned)grid), dim3(kBlock), 0, s, a); break;        \
        case 4: hipLaunchKernelGGL((compact_kernel<4, WW>), dim3((unsigned)grid), dim3(kBlock), 0, s, a); break;        \
        case 2: hipLaunchKernelGGL((compact_kernel<2, WW>), dim3((unsigned)grid), dim3(kBlock), 0, s, a); break;        \
        case 1: hipLaunchKernelGGL((compact_kernel<1, WW>), dim3((unsigned)grid), dim3(kBlock), 0, s, a); break;        \
        default: hipLaunchKernelGGL((compact_kernel<0, WW>), dim3((unsigned)grid), dim3(kBlock), 0, s, a); break;       \
    }
    if (tile_rows == kFilterTileSmall) { RDF_COMPACT_LAUNCH(4) } else { RDF_COMPACT_LAUNCH(16) }
#undef RDF_COMPACT_LAUNCH
    return hipGetLastError();
}

hipError_t launch_compact_one(const FilterOneArgs& a, hipStream_t s) {
    int64_t grid = a.ntiles < (int64_t)eval_grid_limit() ? a.ntiles : (int64_t)eval_grid_limit();
    if (grid <= 0) return hipSuccess;
    int es = a.esize[0];
    for (int k = 1; k < a.ncols; ++k) if (a.esize[k] != es) es = 0;
    switch (es) {
        case 8: hipLaunchKernelGGL((compact_one_kernel<8>), dim3((unsigned)grid), dim3(kBlock), 0, s, a); break;
        case 4: hipLaunchKernelGGL((compact_one_kernel<4>), dim3((unsigned)grid), dim3(kBlock), 0, s, a); break;
        case 2: hipLaunchKernelGGL((compact_one_kernel<2>), dim3((unsigned)grid), dim3(kBlock), 0, s, a); break;
        case 1: hipLaunchKernelGGL((compact_one_kernel<1>), dim3((unsigned)grid), dim3(kBlock), 0, s, a); break;
        default: hipLaunchKernelGGL((compact_one_kernel<0>), dim3((unsigned)grid), dim3(kBlock), 0, s, a); break;
    }
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

hipError_t launch_sort_keys(const SortKeyArgs& a, hipStream_t s) {
    int64_t grid = (a.n + kBlock - 1) / kBlock;
    if (grid > eval_grid_limit()) grid = eval_grid_limit();
    int w = 8;
    switch (a.dtype) { case RDF_I8: case RDF_U8: w = 1; break; case RDF_I16: case RDF_U16: w = 2; break;
                       case RDF_I32: case RDF_U32: case RDF_F32: w = 4; break; default: break; }
    const uint64_t mask = w == 8 ? ~0ull : ((1ull << (8 * w)) - 1);
    if (grid > 0) hipLaunchKernelGGL(sort_keys_kernel, dim3((unsigned)grid), dim3(kBlock), 0, s, a, mask);
    return hipGetLastError();
}
int sort_grid(int64_t ntiles) {  // both sort kernels must agree on it: it fixes the tile -> block ownership
    const int64_t lim = (int64_t)eval_grid_limit() / 2;  // 24.5 KiB of LDS per block: 4 blocks per CU
    const int64_t g = ntiles < lim ? ntiles : lim;
    return g < 1 ? 1 : (int)g;
}
hipError_t launch_sort_hist(const SortPassArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(sort_hist_kernel, dim3(sort_grid(a.ntiles)), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_sort_scatter(const SortPassArgs& a, hipStream_t s) {
    hipLaunchKernelGGL((sort_scatter_kernel<false>), dim3(sort_grid(a.ntiles)), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_sort_hist64(const SortPassArgs& a, hipStream_t s) { return launch_sort_hist(a, s); }
hipError_t launch_sort_scatter64(const SortPassArgs& a, hipStream_t s) {
    hipLaunchKernelGGL((sort_scatter_kernel<true>), dim3(sort_grid(a.ntiles)), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_groupby_build(const GroupByArgs& a, hipStream_t s) {
    int64_t grid = a.ntiles < (int64_t)eval_grid_limit() ? a.ntiles : (int64_t)eval_grid_limit();
    if (grid <= 0) return hipSuccess;
    if (a.max_groups <= kLdsGroups / 2) hipLaunchKernelGGL(groupby_build_lds_kernel, dim3((unsigned)grid), dim3(kBlock), 0, s, a);
    else hipLaunchKernelGGL(groupby_build_kernel, dim3((unsigned)grid), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}
static int rows_grid(int64_t n) {
    int64_t g = (n + kBlock - 1) / kBlock;
    if (g > eval_grid_limit()) g = eval_grid_limit();
    return g < 1 ? 1 : (int)g;
}
// A few KB ... MB of tables out of page-locked host memory, fetched by a kernel over the link instead of queued on a copy engine: while
// the streamed batch loop has a slab's upload in flight, a host-to-device copy — hipMemcpy most of all — waits behind it in the
// engine's queue (measured: rdf_frame_pin took 3 ms per slab, most of a slab's upload time, for 1.5 KB of tables).
__global__ __launch_bounds__(kBlock) void copy_small_kernel(const uint8_t* src, uint8_t* dst, size_t bytes) {
    const size_t tid = (size_t)blockIdx.x * kBlock + threadIdx.x, nt = (size_t)gridDim.x * kBlock;
    if ((((uintptr_t)src | (uintptr_t)dst) & 7) == 0) {
        const size_t nw = bytes >> 3;
        for (size_t i = tid; i < nw; i += nt) ((uint64_t*)dst)[i] = ((const uint64_t*)src)[i];
        for (size_t i = (nw << 3) + tid; i < bytes; i += nt) dst[i] = src[i];
    } else for (size_t i = tid; i < bytes; i += nt) dst[i] = src[i];
}
hipError_t launch_copy_small(const void* src_pinned, void* dst_dev, size_t bytes, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    size_t g = (bytes / 8 + kBlock - 1) / kBlock;
    if (g > 256) g = 256;
    hipLaunchKernelGGL(copy_small_kernel, dim3((unsigned)(g < 1 ? 1 : g)), dim3(kBlock), 0, s, (const uint8_t*)src_pinned, (uint8_t*)dst_dev, bytes);
    return hipGetLastError();
}
hipError_t launch_join_buckets(const JoinBucketArgs& a, hipStream_t s) {
    if (a.nrv > 0) hipLaunchKernelGGL(join_buckets_kernel, dim3(rows_grid(a.nrv)), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_join_table(const JoinTableArgs& a, hipStream_t s) {
    if (a.nrv > 0) hipLaunchKernelGGL(join_table_kernel, dim3(rows_grid(a.nrv)), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}
// distinct values of a SORTED key array (equal keys are neighbours): rows whose key differs from the row in front of them
__global__ __launch_bounds__(kBlock) void join_distinct_kernel(const uint64_t* keys, int64_t n, unsigned long long* out) {
    __shared__ unsigned int part[kBlock / 64];
    unsigned int mine = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const uint64_t k = __builtin_nontemporal_load(as_global<uint64_t>(keys) + i);
        const uint64_t p = i > 0 ? as_global<uint64_t>(keys)[i - 1] : ~k;
        mine += k != p ? 1u : 0u;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mine += (unsigned int)__shfl_xor((int)mine, d);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < kBlock / 64; ++w) t += part[w];
        if (t) atomicAdd(out, t);
    }
}
hipError_t launch_join_distinct(const uint64_t* keys, int64_t n, unsigned long long* out, hipStream_t s) {     // *out zeroed by the caller
    if (n > 0) hipLaunchKernelGGL(join_distinct_kernel, dim3(rows_grid(n)), dim3(kBlock), 0, s, keys, n, out);
    return hipGetLastError();
}
hipError_t launch_join_place(JoinPlaceArgs a, hipStream_t s) {
    if (a.nrv <= 0) return hipSuccess;
    a.phase = 0;
    hipLaunchKernelGGL(join_place_kernel, dim3((unsigned)a.ntiles), dim3(kBlock), 0, s, a);
    hipLaunchKernelGGL(join_place_scan_kernel, dim3(1), dim3(1024), 0, s, a);
    a.phase = 2;
    hipLaunchKernelGGL(join_place_kernel, dim3((unsigned)a.ntiles), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_join_combine(const JoinCombineArgs& a, hipStream_t s) {
    if (a.n > 0) hipLaunchKernelGGL(join_combine_kernel, dim3(rows_grid(a.n)), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_join_count(const JoinProbeArgs& a, hipStream_t s) {
    if (a.nl > 0) hipLaunchKernelGGL(join_count_kernel, dim3(rows_grid(a.nl)), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_join_write(const JoinProbeArgs& a, hipStream_t s) {
    if (a.nl > 0) hipLaunchKernelGGL(join_write_kernel, dim3(rows_grid(a.nl)), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_join_append(const JoinAppendArgs& a, hipStream_t s) {
    if (a.nr > 0) hipLaunchKernelGGL(join_append_kernel, dim3(rows_grid(a.nr)), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_count_bytes(const uint8_t* p, int64_t n, unsigned long long* out, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(count_bytes_kernel, dim3(rows_grid(n)), dim3(kBlock), 0, s, p, n, out);
    return hipGetLastError();
}
hipError_t launch_gb_hist(const GbPartArgs& a, int grid, hipStream_t s) {
    hipLaunchKernelGGL(gb_hist_kernel, dim3(grid), dim3(kGbBlock), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_gb_skew(const int64_t* scan, int64_t nblocks, unsigned int* flag, hipStream_t s) {
    hipLaunchKernelGGL(gb_skew_kernel, dim3(1), dim3(kGbBlock), 0, s, scan, nblocks, flag);
    return hipGetLastError();
}
hipError_t launch_gb_scatter(const GbPartArgs& a, int grid, bool dedup, hipStream_t s) {
    constexpr int P = 1 << kGbPartBits;
    size_t lds = (size_t)2 * kGbSuper * 8 + (size_t)P * (8 + 4 + 2) + (size_t)kGbSuper * 2;   // 80 896 B: two blocks per CU
    if (dedup) {
        lds += (size_t)kGbCache * (8 + 8 + 4);
        (void)hipFuncSetAttribute((const void*)gb_scatter_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((gb_scatter_kernel<true>), dim3(grid), dim3(kGbBlock), lds, s, a);
    } else {
        (void)hipFuncSetAttribute((const void*)gb_scatter_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   // > 64 KB
        hipLaunchKernelGGL((gb_scatter_kernel<false>), dim3(grid), dim3(kGbBlock), lds, s, a);
    }
    return hipGetLastError();
}
hipError_t launch_gb_aggregate(const GbAggArgs& a, hipStream_t s) {
    const size_t lds = (size_t)kGbSlots * 20 + 16;
    (void)hipFuncSetAttribute((const void*)gb_aggregate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   // > 64 KB
    hipLaunchKernelGGL(gb_aggregate_kernel, dim3(1 << kGbPartBits), dim3(kGbBlock), lds, s, a);
    return hipGetLastError();
}
hipError_t launch_groupby_prepare(const GroupPrepArgs& a, hipStream_t s) {
    int64_t grid = a.ntiles < (int64_t)eval_grid_limit() ? a.ntiles : (int64_t)eval_grid_limit();
    if (grid > 0) hipLaunchKernelGGL(groupby_prepare_kernel, dim3((unsigned)grid), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_groupby_partitions(const GroupAggArgs& a, hipStream_t s) {
    int64_t grid = (int64_t)1 << a.part_bits;
    const int64_t lim = 3 * (int64_t)eval_grid_limit() / 8;  // 48 KiB of LDS per block: 3 blocks per CU
    if (grid > lim) grid = lim;
    hipLaunchKernelGGL(groupby_partitions_kernel, dim3((unsigned)grid), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_groupby_emit(const GroupEmitArgs& a, hipStream_t s) {
    int64_t grid = (a.t.capacity + 2 + kBlock - 1) / kBlock;
    if (grid > eval_grid_limit()) grid = eval_grid_limit();
    hipLaunchKernelGGL(groupby_emit_kernel, dim3((unsigned)grid), dim3(kBlock), 0, s, a);
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
