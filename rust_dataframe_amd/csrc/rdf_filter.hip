// rdf_filter.hip — order-preserving stream compaction, wave-granular (Column::filter -> ChunkedArray::filter ->
// arrow::compute::filter per chunk pair, src/table.rs:97-107,213-215; DataFrame::filter's per-column loop,
// src/dataframe.rs:178-189, as ONE pass over the mask).
//
//   fcount_kernel -> scan -> fcompact_kernel
//
// A tile is ONE WAVE's work: WW mask words = WW * 64 rows of ONE chunk (WW = 8: 512 rows, half of one of the reference
// readers' RecordBatches, src/dataframe.rs:352 — 16 words per wave cost 192 VGPRs —; WW = 4 for frames of even shorter chunks).  The scan runs over wave
// tiles, so a wave knows where its kept rows go without talking to its neighbours: no block barrier, no block-wide tile
// that a 1024-row chunk would fill to a quarter (first generation: 0.28 of the HBM peak on 1024-row batches), and a
// block's four waves work on four different chunks at once.  Per tile and column:
//   * dense tiles (>= 1/4 of the rows kept, all rows present, 16-byte aligned) load whole 16-byte vectors — half the
//     load instructions of the per-row form; sparse tiles load only the kept rows (a 64-byte sector with no kept row is
//     never fetched);
//   * ranks = scalar prefix over the wave's keep-words + popcount of the lane's word below its bit: no shuffles;
//   * kept values are staged in the wave's private LDS region, shifted by (output position mod vector) so that aligned
//     16-byte vectors of the staging area are aligned 16-byte vectors of the output: the run leaves as dwordx4 stores.
#include <type_traits>

#include "rdf_common.hip.h"
#include "rdf_lanewin.hip.h"

namespace rdfk {

template <int WW> constexpr int wtile_rows() { return WW * 64; }

template <int WW>
__device__ __forceinline__ void wkeep_words(const DevChunkCol& m, int64_t rw, int64_t clen, uint64_t (&kw)[WW]) {
    load_windows<WW>((const uint8_t*)m.values, m.offset + rw, clen - rw, kw);
    if (m.validity) {
        uint64_t vw[WW];
        load_windows<WW>(m.validity, m.offset + rw, clen - rw, vw);
#pragma unroll
        for (int i = 0; i < WW; ++i) kw[i] &= vw[i];
    }
}

struct WTile { int64_t c, r0, clen; };
template <int WW>
__device__ __forceinline__ WTile wlocate(const FilterWArgs& a, int64_t tile) {
    WTile t;
    if (a.t.nchunks == 1) { t.c = 0; t.r0 = tile * wtile_rows<WW>(); t.clen = a.len0; }
    else {
        t.c = find_chunk_tile_inv(a.t.chunk_tile_start, a.t.nchunks, tile, a.tile_inv);
        t.r0 = (tile - a.t.chunk_tile_start[t.c]) * wtile_rows<WW>();
        t.clen = a.t.chunk_len[t.c];
    }
    return t;
}

template <int WW>
__global__ __launch_bounds__(kBlock) void fcount_kernel(const FilterWArgs a, int64_t* tile_counts) {
    const int lane = threadIdx.x & 63;
    const int wave = wave_id();
    constexpr int kWaves = kBlock / 64;
    for (int64_t tile = (int64_t)blockIdx.x * kWaves + wave; tile < a.t.ntiles; tile += (int64_t)gridDim.x * kWaves) {
        const WTile t = wlocate<WW>(a, tile);
        const DevChunkCol m = a.t.nchunks == 1 ? a.mask0 : a.t.mask[t.c];
        uint64_t kw[WW];
        wkeep_words<WW>(m, t.r0, t.clen, kw);
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < WW; ++i) cnt += __popcll(kw[i]);
        if (lane == 0) tile_counts[tile] = cnt;
    }
}


// Vector loads of one column's tile.
template <typename T, int WW>
__device__ __forceinline__ bool wvec_ok(const DevChunkCol& col, int64_t rw, int64_t clen) {
    constexpr int E = 16 / (int)sizeof(T);
    return E > 1 && WW % E == 0 && rw + WW * 64 <= clen && (((uintptr_t)(const void*)(as_global<T>(col.values) + col.offset + rw)) & 15) == 0;
}
template <typename T, int WW>
__device__ __forceinline__ void wvec_load(const DevChunkCol& col, int64_t rw, typename Vec16<T>::type (&v)[WW * (int)sizeof(T) / 16 > 0 ? WW * (int)sizeof(T) / 16 : 1]) {
    constexpr int E = 16 / (int)sizeof(T);
    using V = typename Vec16<T>::type;
    const int lane = threadIdx.x & 63;
    const GlobalPtr<V> p = (GlobalPtr<V>)(as_global<T>(col.values) + col.offset + rw) + lane;
#pragma unroll
    for (int g = 0; g < WW / E; ++g) v[g] = __builtin_nontemporal_load(p + g * 64);
}

// One column of one wave tile.  `stage`: (WW * 64 + 16 / sizeof(T)) elements of the wave's private LDS region.
template <typename T, int WW>
__device__ __forceinline__ void wcompact(const DevChunkCol col, const DevOutChunk oc, int64_t rw, int64_t clen, int64_t wave_out,
                                         const uint64_t (&kw)[WW], int cnt, unsigned char* stage_raw, uint8_t* vstage, uint32_t& nulls) {
    constexpr int E = 16 / (int)sizeof(T);        // elements per 16-byte vector
    using V = typename Vec16<T>::type;
    const int lane = threadIdx.x & 63;
    T* stage = (T*)stage_raw;
    const bool hasv = col.validity != nullptr;
    uint64_t vw[WW];
    if (hasv) load_windows<WW>(col.validity, col.offset + rw, clen - rw, vw);
    const GlobalPtr<T> src0 = as_global<T>(col.values) + col.offset + rw;
    const bool vec_out = (((uintptr_t)oc.values) & 15) == 0;
    const int shift = vec_out ? (int)(wave_out & (E - 1)) : 0;
    const bool dense = cnt * 4 >= WW * 64 && wvec_ok<T, WW>(col, rw, clen);
    if (dense) {
        if constexpr (E > 1 && WW % E == 0) {
            constexpr int NG = WW / E;                // groups of 64 vectors = 64 * E rows = E mask words
            V v[NG];
            wvec_load<T, WW>(col, rw, v);
            const int widx = (E * lane) >> 6, bit = (E * lane) & 63;    // this lane's E rows: bits [bit, bit + E) of word E * g + widx
            int wb = shift;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                uint64_t w = kw[E * g], wv = hasv ? vw[E * g] : 0;
                int before = wb;
#pragma unroll
                for (int h = 1; h < E; ++h) {
                    if (widx >= h) before += __popcll(kw[E * g + h - 1]);
                    if (widx == h) { w = kw[E * g + h]; if (hasv) wv = vw[E * g + h]; }
                }
                before += __popcll(w & ((1ull << bit) - 1));
#pragma unroll
                for (int e = 0; e < E; ++e)
                    if ((w >> (bit + e)) & 1) {
                        stage[before] = v[g][e];
                        if (hasv) vstage[before] = (uint8_t)((wv >> (bit + e)) & 1);
                        ++before;
                    }
#pragma unroll
                for (int h = 0; h < E; ++h) wb += __popcll(kw[E * g + h]);
            }
        }
    } else {
        const GlobalPtr<T> src = src0 + lane;
        T val[WW];
#pragma unroll
        for (int i = 0; i < WW; ++i)
            if ((kw[i] >> lane) & 1) val[i] = __builtin_nontemporal_load(src + i * 64);
        int wb = shift;
#pragma unroll
        for (int i = 0; i < WW; ++i) {
            if ((kw[i] >> lane) & 1) {
                const int rank = wb + __popcll(kw[i] & ((1ull << lane) - 1));
                stage[rank] = val[i];
                if (hasv) vstage[rank] = (uint8_t)((vw[i] >> lane) & 1);
            }
            wb += __popcll(kw[i]);
        }
    }
    __builtin_amdgcn_wave_barrier();  // same-wave LDS ops are executed in order; this only pins the compiler
    // slots [shift, shift + cnt) of the staging area are output elements [wave_out, wave_out + cnt)
    const int total = shift + cnt;
    if (vec_out && E > 1) {
        const GlobalMutPtr<T> dst0 = as_global_mut<T>(oc.values) + (wave_out - shift);
        for (int j = lane; j * E < total; j += 64) {
            const int s0 = j * E;
            if (s0 >= shift && s0 + E <= total) __builtin_nontemporal_store(*(const V*)&stage[s0], (GlobalMutPtr<V>)dst0 + j);
            else {
#pragma unroll
                for (int e = 0; e < E; ++e) if (s0 + e >= shift && s0 + e < total) dst0[s0 + e] = stage[s0 + e];
            }
        }
    } else {
        const GlobalMutPtr<T> dst = as_global_mut<T>(oc.values) + wave_out;
        for (int i = lane; i < cnt; i += 64) __builtin_nontemporal_store(stage[shift + i], dst + i);
    }
    if (hasv && oc.validity) {
        // out bits [wave_out, wave_out + cnt): ballot 64 aligned positions at a time, OR into the (pre-zeroed) bitmap;
        // boundary words are shared with neighbouring tiles, hence atomics
        const int64_t end = wave_out + cnt;
        for (int64_t wpos = wave_out & ~63ll; wpos < end; wpos += 64) {
            const int64_t pos = wpos + lane;
            const bool inside = pos >= wave_out && pos < end;
            const bool bitv = inside && vstage[shift + (pos - wave_out)];
            const uint64_t word = __ballot(bitv);
            const uint64_t inw = __ballot(inside);
            if (lane == 0) {
                if (word) atomicOr((unsigned long long*)oc.validity + (wpos >> 6), (unsigned long long)word);
                nulls += (uint32_t)__popcll(inw & ~word);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// ES = common element size of all columns of the launch (8 / 4 / 2 / 1) or 0 for mixed sizes.
// PF: the NEXT tile's place (chunk search, chunk length, output offset, mask descriptor — a chain of dependent scalar
// loads for chunked frames) is looked up while the current tile's values are in flight.
template <int ES, int WW, bool PF>
__global__ __launch_bounds__(kBlock, ES ? 4 : 2) void fcompact_kernel(const FilterWArgs a) {
    constexpr int kWaves = kBlock / 64;
    __shared__ __attribute__((aligned(16))) unsigned char stage[kWaves][WW * 64 * 8 + 16];
    __shared__ uint8_t vstage[kWaves][WW * 64 + 16];
    __shared__ uint32_t nullacc[kWaves][kMaxFilterCols];   // per-wave null counters per column, flushed once per (column, chunk)
    const int lane = threadIdx.x & 63;
    const int wave = wave_id();
    if (lane < kMaxFilterCols) nullacc[wave][lane] = 0;
    __builtin_amdgcn_wave_barrier();
    int64_t cur_chunk = -1;
    const bool one = a.t.nchunks == 1;
    struct Meta { WTile t; int64_t wave_out; DevChunkCol m; };
    auto locate_all = [&](int64_t tile) -> Meta {
        Meta mt;
        mt.t = wlocate<WW>(a, tile);
        // where this tile's kept rows start in the chunk's output
        mt.wave_out = one ? a.tile_scan[tile] : a.tile_scan[tile] - a.tile_scan[a.t.chunk_tile_start[mt.t.c]];
        mt.m = one ? a.mask0 : a.t.mask[mt.t.c];
        return mt;
    };
    const int64_t tile0 = (int64_t)blockIdx.x * kWaves + wave, tstride = (int64_t)gridDim.x * kWaves;
    Meta meta;
    if (PF && tile0 < a.t.ntiles) meta = locate_all(tile0);
    for (int64_t tile = tile0; tile < a.t.ntiles; tile += tstride) {
        if (!PF) meta = locate_all(tile);
        const WTile t = meta.t;
        const int64_t wave_out = meta.wave_out;
        if (t.c != cur_chunk) {
            if (cur_chunk >= 0 && lane < a.ncols && nullacc[wave][lane]) {
                atomicAdd((unsigned long long*)&a.out_null_counts[(int64_t)lane * a.t.nchunks + cur_chunk], (unsigned long long)nullacc[wave][lane]);
                nullacc[wave][lane] = 0;
            }
            __builtin_amdgcn_wave_barrier();
            cur_chunk = t.c;
        }
        // (Requesting column 0's vectors before the mask words are known, and column k + 1's while column k is staged,
        // was measured and dropped: the double-buffered vectors push the kernel past 128 VGPRs — one column 2.8 -> 3.1 ms,
        // two 4.4 -> 4.9 ms per 1e9 rows.)
        uint64_t kw[WW];
        wkeep_words<WW>(meta.m, t.r0, t.clen, kw);
        if (PF && tile + tstride < a.t.ntiles) meta = locate_all(tile + tstride);
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < WW; ++i) cnt += __popcll(kw[i]);
        if (cnt == 0) continue;
#pragma unroll 1
        for (int k = 0; k < a.ncols; ++k) {
            const DevChunkCol col = one ? a.cols0[k] : a.cols[(int64_t)k * a.t.nchunks + t.c];
            const DevOutChunk oc = one ? a.outs0[k] : a.outs[(int64_t)k * a.t.nchunks + t.c];
            uint32_t nn = 0;
            const int es = ES ? ES : a.esize[k];
            if (es == 8) wcompact<uint64_t, WW>(col, oc, t.r0, t.clen, wave_out, kw, cnt, stage[wave], vstage[wave], nn);
            else if (es == 4) wcompact<uint32_t, WW>(col, oc, t.r0, t.clen, wave_out, kw, cnt, stage[wave], vstage[wave], nn);
            else if (es == 2) wcompact<uint16_t, WW>(col, oc, t.r0, t.clen, wave_out, kw, cnt, stage[wave], vstage[wave], nn);
            else wcompact<uint8_t, WW>(col, oc, t.r0, t.clen, wave_out, kw, cnt, stage[wave], vstage[wave], nn);
            if (lane == 0 && nn) nullacc[wave][k] += nn;
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (cur_chunk >= 0 && lane < a.ncols && nullacc[wave][lane])
        atomicAdd((unsigned long long*)&a.out_null_counts[(int64_t)lane * a.t.nchunks + cur_chunk], (unsigned long long)nullacc[wave][lane]);
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA compaction: a wave's 1024-row tile of an 8- or 4-byte column goes global -> LDS with buffer-less
// `global_load_lds_dwordx4` (no VGPRs hold data in flight: 8 KB per wave, 16 waves per CU = 128 KB per CU under way,
// twice what the register-staged kernels manage at their 120-190 VGPRs), is compacted IN PLACE in LDS (a kept element
// moves to a lower address than any element not yet read), and leaves as aligned 16-byte stores.  The DMA is issued
// before the mask words are waited for: one memory round trip per tile and column instead of two.  Tiles that are not
// full or not 16-byte aligned, and 1- / 2-byte columns, take the register path (two 512-row halves).
// Chosen by the host for filters that keep at least 1/8 of the rows (a selective filter should not fetch every sector).

template <typename T>
__device__ __forceinline__ void dma_tile(const DevChunkCol& col, int64_t rw, unsigned char* raw_bytes) {
    constexpr int E = 16 / (int)sizeof(T);
    const int lane = threadIdx.x & 63;
    const GlobalPtr<T> src = as_global<T>(col.values) + col.offset + rw;
    constexpr int NI = kWDmaTile * (int)sizeof(T) / 1024;    // 1 KiB per wave instruction
#pragma unroll
    for (int g = 0; g < NI; ++g)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (g * 64 + lane) * E),
                                         (LdsPtr)(raw_bytes + 16 + g * 1024), 16, 0, 0);
}
// The same for a tile at the END of its batch.  nvec: whole 16-byte vectors the batch holds from row rw on; lanes past them re-read the
// last whole vector — no address leaves the batch — and dma_tail brings the rows of the vector across the end.  (A function of its
// own: with the clamp in dma_tile the full tiles of 1024-row batches ran 25 % slower.)
template <typename T>
__device__ __forceinline__ void dma_tile_end(const DevChunkCol& col, int64_t rw, unsigned char* raw_bytes, int nvec) {
    constexpr int E = 16 / (int)sizeof(T);
    const int lane = threadIdx.x & 63;
    const GlobalPtr<T> src = as_global<T>(col.values) + col.offset + rw;
    constexpr int NI = kWDmaTile * (int)sizeof(T) / 1024;    // 1 KiB per wave instruction
#pragma unroll
    for (int g = 0; g < NI; ++g) {
        const int vi = g * 64 + lane;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (vi < nvec ? vi : nvec - 1) * E),
                                         (LdsPtr)(raw_bytes + 16 + g * 1024), 16, 0, 0);
    }
}
// a partial tile in LDS (dma_tile with nvec): the rows of the vector that lies across the batch's end, element by element
// (call after the tile has landed; rows past the end hold copies of the last whole vector: their keep bits are cleared)
template <typename T>
__device__ __forceinline__ void dma_tail(const DevChunkCol& col, int64_t rw, int64_t avail, unsigned char* raw_bytes) {
    constexpr int E = 16 / (int)sizeof(T);
    const int lane = threadIdx.x & 63;
    const int nfull = (int)(avail / E), rem = (int)(avail - (int64_t)nfull * E);
    if (lane < rem) ((T*)raw_bytes)[E + nfull * E + lane] = (as_global<T>(col.values) + col.offset + rw)[nfull * E + lane];
}
// keep-words of a partial tile: lane i's word without the rows past the batch's end
__device__ __forceinline__ uint64_t tail_words(int64_t avail) {
    const int64_t left = avail - 64 * (int64_t)(threadIdx.x & 63);
    return left >= 64 ? ~0ull : left <= 0 ? 0ull : (1ull << left) - 1;
}

// raw tile at element index E.. of `raw`; kept elements end up at [shift, shift + cnt).  kwv: lane i = keep-word i.
template <typename T, bool ENDS = false>
__device__ __forceinline__ void dma_compact(const DevChunkCol col, const DevOutChunk oc, int64_t rw, int64_t clen, int64_t wave_out,
                                            uint64_t kwv, int cnt, unsigned char* raw_bytes, uint8_t* vstage, uint32_t& nulls) {
    constexpr int E = 16 / (int)sizeof(T);
    using V = typename Vec16<T>::type;
    const int lane = threadIdx.x & 63;
    T* raw = (T*)raw_bytes;
    const bool hasv = col.validity != nullptr;
    uint64_t vwv = 0;
    if (hasv) vwv = lane_windows<16>(col.validity, col.offset + rw, clen - rw);
    const int shift = (int)(wave_out & (E - 1));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the tile has landed in LDS (nothing else orders a ds_read behind an LDS-DMA)
    if constexpr (ENDS) { if (clen - rw < kWDmaTile) dma_tail<T>(col, rw, clen - rw, raw_bytes); }      // (a tile at the end of its batch, brought in through clamped addresses)
    int wb = shift;
#pragma unroll
    for (int b = 0; b < 16; b += 4) {   // four words per step: the reads of a step complete before its writes start, and every
        T x[4];                         // write lands below the step's read frontier
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = raw[E + (b + i) * 64 + lane];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint64_t w = rl64(kwv, b + i);
            if ((w >> lane) & 1) {
                const int pos = wb + __popcll(w & ((1ull << lane) - 1));
                raw[pos] = x[i];
                if (hasv) vstage[pos] = (uint8_t)((rl64(vwv, b + i) >> lane) & 1);
            }
            wb += __popcll(w);
        }
    }
    __builtin_amdgcn_wave_barrier();
    const int total = shift + cnt;
    const GlobalMutPtr<T> dst0 = as_global_mut<T>(oc.values) + (wave_out - shift);
    for (int j = lane; j * E < total; j += 64) {
        const int s0 = j * E;
        if (s0 >= shift && s0 + E <= total) __builtin_nontemporal_store(*(const V*)&raw[s0], (GlobalMutPtr<V>)dst0 + j);
        else {
#pragma unroll
            for (int e = 0; e < E; ++e) if (s0 + e >= shift && s0 + e < total) dst0[s0 + e] = raw[s0 + e];
        }
    }
    if (hasv && oc.validity) {
        const int64_t end = wave_out + cnt;
        for (int64_t wpos = wave_out & ~63ll; wpos < end; wpos += 64) {
            const int64_t pos = wpos + lane;
            const bool inside = pos >= wave_out && pos < end;
            const bool bitv = inside && vstage[shift + (pos - wave_out)];
            const uint64_t word = __ballot(bitv);
            const uint64_t inw = __ballot(inside);
            if (lane == 0) {
                if (word) atomicOr((unsigned long long*)oc.validity + (wpos >> 6), (unsigned long long)word);
                nulls += (uint32_t)__popcll(inw & ~word);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// A tile the DMA path cannot take (launches hold 8- and 4-byte columns only: the host sends anything else to fcompact_kernel).
template <typename T>
__device__ __forceinline__ void slow_compact(const DevChunkCol col, const DevOutChunk oc, int64_t rw, int64_t wave_out, uint64_t kwv, int cnt,
                                          unsigned char* stage_raw, uint8_t* vstage, uint32_t& nulls) {
    const int lane = threadIdx.x & 63;
    T* stage = (T*)stage_raw;
    const bool hasv = col.validity != nullptr;
    const GlobalPtr<T> src = as_global<T>(col.values) + col.offset + rw + lane;
    int wb = 0;
#pragma unroll 1
    for (int i = 0; i < 16; ++i) {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)kwv, i), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(kwv >> 32), i);
        const uint64_t w = ((uint64_t)hi << 32) | lo;
        if (w == 0) continue;
        if ((w >> lane) & 1) {
            const int pos = wb + __popcll(w & ((1ull << lane) - 1));
            stage[pos] = src[i * 64];
            if (hasv) {
                const int64_t e = col.offset + rw + i * 64 + lane;
                vstage[pos] = (uint8_t)((as_global<uint8_t>(col.validity)[e >> 3] >> (e & 7)) & 1);
            }
        }
        wb += __popcll(w);
    }
    __builtin_amdgcn_wave_barrier();
    const GlobalMutPtr<T> dst = as_global_mut<T>(oc.values) + wave_out;
    for (int i = lane; i < cnt; i += 64) dst[i] = stage[i];
    if (hasv && oc.validity) {
        const int64_t end = wave_out + cnt;
        for (int64_t wpos = wave_out & ~63ll; wpos < end; wpos += 64) {
            const int64_t pos = wpos + lane;
            const bool inside = pos >= wave_out && pos < end;
            const bool bitv = inside && vstage[pos - wave_out];
            const uint64_t word = __ballot(bitv);
            const uint64_t inw = __ballot(inside);
            if (lane == 0) {
                if (word) atomicOr((unsigned long long*)oc.validity + (wpos >> 6), (unsigned long long)word);
                nulls += (uint32_t)__popcll(inw & ~word);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// (ENDS: as in ffilter_dma_kernel below — tiles at the end of their chunk through the DMA path too; FilterWArgs::ends)
template <bool ENDS>
__global__ __launch_bounds__(kBlock, 4) void fcompact_dma_kernel(const FilterWArgs a) {
    constexpr int WW = kWDmaTile / 64, kWaves = kBlock / 64;
    __shared__ __attribute__((aligned(16))) unsigned char stage[kWaves][WW * 64 * 8 + 32];
    __shared__ uint8_t vstage[kWaves][WW * 64 + 16];
    __shared__ uint32_t nullacc[kWaves][kMaxFilterCols];
    const int lane = threadIdx.x & 63;
    const int wave = wave_id();
    if (lane < kMaxFilterCols) nullacc[wave][lane] = 0;
    __builtin_amdgcn_wave_barrier();
    int64_t cur_chunk = -1;
    const bool one = a.t.nchunks == 1;
    // A tile's place — chunk search, chunk length, output offset, mask and column-0 descriptors: three rounds of dependent
    // scalar loads that miss every cache for a frame of ~1e6 batches — is looked up one iteration ahead, BETWEEN the issue
    // of the current tile's loads (LDS-DMA + mask words) and their first use: scalar loads wait on their own counter, so
    // the lookup runs under the vector loads' latency (measured: the lookups cost 1024-row batches 1.3 ms per 1e9 rows).
    struct Meta { WTile t; int64_t wave_out; DevChunkCol m, c0; };
    auto locate_all = [&](int64_t tile) -> Meta {
        Meta mt;
        mt.t = wlocate<WW>(a, tile);
        mt.wave_out = !a.tile_scan ? 0 : one ? a.tile_scan[tile] : a.tile_scan[tile] - a.tile_scan[a.t.chunk_tile_start[mt.t.c]];      // (no scan: every chunk is one tile)
        mt.m = one ? a.mask0 : a.t.mask[mt.t.c];
        mt.c0 = one ? a.cols0[0] : a.cols[mt.t.c];
        return mt;
    };
    const int64_t tile0 = (int64_t)blockIdx.x * kWaves + wave, tstride = (int64_t)gridDim.x * kWaves;
    Meta meta;
    if (tile0 < a.t.ntiles) meta = locate_all(tile0);
    for (int64_t tile = tile0; tile < a.t.ntiles; tile += tstride) {
        const WTile t = meta.t;
        const int64_t wave_out = meta.wave_out;
        const DevChunkCol m = meta.m;
        if (t.c != cur_chunk) {
            if (cur_chunk >= 0 && lane < a.ncols && nullacc[wave][lane]) {
                atomicAdd((unsigned long long*)&a.out_null_counts[(int64_t)lane * a.t.nchunks + cur_chunk], (unsigned long long)nullacc[wave][lane]);
                nullacc[wave][lane] = 0;
            }
            __builtin_amdgcn_wave_barrier();
            cur_chunk = t.c;
        }
        const bool full = t.r0 + WW * 64 <= t.clen;
        // column 0's tile is requested before the mask words are waited for
        DevChunkCol col = meta.c0;
        int es = a.esize[0];
        const int64_t avail = t.clen - t.r0;
        auto dma_ok = [&](const DevChunkCol& c, int e) {
            return (full || (ENDS && avail * e >= 16)) && (e == 8 || e == 4) && (((uintptr_t)((const char*)c.values + (c.offset + t.r0) * e)) & 15) == 0;
        };
        auto dma_in = [&](const DevChunkCol& c, int e) __attribute__((always_inline)) {
            if (!ENDS || full) { if (e == 8) dma_tile<uint64_t>(c, t.r0, stage[wave]); else dma_tile<uint32_t>(c, t.r0, stage[wave]); }
            else if (e == 8) dma_tile_end<uint64_t>(c, t.r0, stage[wave], (int)(avail / 2));
            else dma_tile_end<uint32_t>(c, t.r0, stage[wave], (int)(avail / 4));
        };
        bool dma = dma_ok(col, es);
        if (dma) dma_in(col, es);
        const LaneWin<WW> q0 = lane_windows_issue<WW>((const uint8_t*)m.values, m.offset + t.r0, t.clen - t.r0);
        LaneWin<WW> q1 = q0;
        if (m.validity) q1 = lane_windows_issue<WW>(m.validity, m.offset + t.r0, t.clen - t.r0);
        if (tile + tstride < a.t.ntiles) meta = locate_all(tile + tstride);      // under the loads just issued
        uint64_t kwv = lane_windows_finish<WW>(q0);                              // lane i = keep-word i
        if (m.validity) kwv &= lane_windows_finish<WW>(q1);
        int cnt = __popcll(kwv);
#pragma unroll
        for (int d = 1; d < WW; d <<= 1) cnt += __shfl_xor(cnt, d);
        cnt = __builtin_amdgcn_readfirstlane(cnt);
        if (a.out_len && lane == 0 && cnt) a.out_len[t.c] = cnt;      // (one-pass Column::filter over one-tile chunks: the count never existed before)
        const bool no_room = a.out_cap && (int64_t)cnt > a.out_cap[t.c];      // outputs sized by an earlier count that does not hold any more: nothing is written
        if (cnt == 0 || no_room) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); continue; }   // the DMA must not land in the next tile's buffer
#pragma unroll 1
        for (int k = 0; k < a.ncols; ++k) {
            if (k > 0) {
                col = one ? a.cols0[k] : a.cols[(int64_t)k * a.t.nchunks + t.c];
                es = a.esize[k];
                dma = dma_ok(col, es);
                if (dma) dma_in(col, es);
            }
            const DevOutChunk oc = one ? a.outs0[k] : a.outs[(int64_t)k * a.t.nchunks + t.c];
            const bool vec_out = (((uintptr_t)oc.values) & 15) == 0;
            uint32_t nn = 0;
            if (dma && vec_out) {
                if (es == 8) dma_compact<uint64_t, ENDS>(col, oc, t.r0, t.clen, wave_out, kwv, cnt, stage[wave], vstage[wave], nn);
                else dma_compact<uint32_t, ENDS>(col, oc, t.r0, t.clen, wave_out, kwv, cnt, stage[wave], vstage[wave], nn);
            } else {
                if (dma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                // chunk tails and slices that are not 16-byte aligned: one mask word at a time
                if (es == 8) slow_compact<uint64_t>(col, oc, t.r0, wave_out, kwv, cnt, stage[wave], vstage[wave], nn);
                else slow_compact<uint32_t>(col, oc, t.r0, wave_out, kwv, cnt, stage[wave], vstage[wave], nn);
            }
            if (lane == 0 && nn) nullacc[wave][k] += nn;
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (cur_chunk >= 0 && lane < a.ncols && nullacc[wave][lane])
        atomicAdd((unsigned long long*)&a.out_null_counts[(int64_t)lane * a.t.nchunks + cur_chunk], (unsigned long long)nullacc[wave][lane]);
}


// ------------------------------------------------------------------------------------------------
// One-pass DataFrame::filter (src/dataframe.rs:178-189): BooleanFilter::eval_to_array (src/expression.rs:766-861) of a
// `column CMP literal [AND | OR column CMP literal]` predicate evaluated on the tile the LDS-DMA compaction has just fetched,
// then every column compacted with the ranks — 8 B read + 8 s B written per row and column, nothing else: the three-pass form
// (predicate -> mask, count, compact) moves the predicate's columns twice and a mask three times.
//
// Where do a tile's kept rows go?  Batch c of the OUTPUT starts at the 64-row rounded position batch c's mask has in the
// frame (the descriptors come from frame_tables_kernel before this kernel runs), so positions are relative to the batch:
// a batch that is one tile (the readers' 1024-row batches, src/dataframe.rs:352) needs nothing from its neighbours; the
// tiles of a longer batch are taken by ticket and find their offsets by decoupled look-back over the tiles of the SAME
// batch (8-byte {status, rows} granules, agent scope: the XCDs' L2s are not coherent).  The price is an output buffer
// as long as the input (288 GB of HBM: the kept rows alone are written).

// keep-words of a full tile that sits in LDS (dma_tile's layout): lane i < 16 ends up holding word i
template <typename T>
__device__ __forceinline__ uint64_t pred_words_lds(const unsigned char* raw_bytes, int op, double lit) {
    constexpr int E = 16 / (int)sizeof(T);
    const int lane = threadIdx.x & 63;
    const T* raw = (const T*)raw_bytes;
    uint64_t kw = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint64_t w = __ballot(fcmp(op, (double)raw[E + i * 64 + lane], lit));
        if (lane == i) kw = w;
    }
    return kw;
}
// the same straight from memory: chunk tails and slices the DMA cannot take
template <typename T>
__device__ __forceinline__ uint64_t pred_words_global(const DevChunkCol& col, int64_t rw, int64_t clen, int op, double lit) {
    const int lane = threadIdx.x & 63;
    const GlobalPtr<T> src = as_global<T>(col.values) + col.offset + rw + lane;
    uint64_t kw = 0;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const bool in = rw + i * 64 + lane < clen;
        const T x = in ? src[i * 64] : T(0);
        const uint64_t w = __ballot(in && fcmp(op, (double)x, lit));
        if (lane == i) kw = w;
    }
    return kw;
}
__device__ __forceinline__ uint64_t pred_words(const FusedTerm& tm, const DevChunkCol& col, int64_t rw, int64_t clen, bool in_lds, const unsigned char* raw) {
    switch (tm.dtype) {
        case RDF_F64: return in_lds ? pred_words_lds<double>(raw, tm.op, tm.lit) : pred_words_global<double>(col, rw, clen, tm.op, tm.lit);
        case RDF_I64: return in_lds ? pred_words_lds<int64_t>(raw, tm.op, tm.lit) : pred_words_global<int64_t>(col, rw, clen, tm.op, tm.lit);
        case RDF_U64: return in_lds ? pred_words_lds<uint64_t>(raw, tm.op, tm.lit) : pred_words_global<uint64_t>(col, rw, clen, tm.op, tm.lit);
        case RDF_F32: return in_lds ? pred_words_lds<float>(raw, tm.op, tm.lit) : pred_words_global<float>(col, rw, clen, tm.op, tm.lit);
        case RDF_I32: return in_lds ? pred_words_lds<int32_t>(raw, tm.op, tm.lit) : pred_words_global<int32_t>(col, rw, clen, tm.op, tm.lit);
        default: return in_lds ? pred_words_lds<uint32_t>(raw, tm.op, tm.lit) : pred_words_global<uint32_t>(col, rw, clen, tm.op, tm.lit);
    }
}

constexpr unsigned long long kFfAggregate = 1ull << 62, kFfPrefix = 2ull << 62, kFfValue = (1ull << 62) - 1;

// ENDS: tiles at the END of their batch take the DMA path as well (frames whose batch lengths are not multiples of the tile: host's
// choice, FusedFilterArgs::ends).  A second instantiation, not a branch: with the end-of-batch code compiled in, the full tiles of
// 1024-row batches ran 15 - 20 % slower (2.47 -> 2.94 ms per 1e9 rows) although none of it executed for them.
template <bool ENDS>
__global__ __launch_bounds__(kBlock, 4) void ffilter_dma_kernel(const FusedFilterArgs fa) {
    const FilterWArgs& a = fa.w;
    constexpr int WW = kWDmaTile / 64, kWaves = kBlock / 64;
    __shared__ __attribute__((aligned(16))) unsigned char stage[kWaves][WW * 64 * 8 + 32];
    __shared__ uint8_t vstage[kWaves][WW * 64 + 16];
    __shared__ uint32_t nullacc[kWaves][kMaxFilterCols];
    const int lane = threadIdx.x & 63;
    const int wave = wave_id();
    if (lane < kMaxFilterCols) nullacc[wave][lane] = 0;
    __builtin_amdgcn_wave_barrier();
    int64_t cur_chunk = -1;
    const bool one = a.t.nchunks == 1;
    const int pc0 = fa.term[0].col, pc1 = fa.nterms > 1 ? fa.term[1].col : fa.term[0].col;
    struct Meta { WTile t; int64_t first; DevChunkCol p0; };
    auto locate_all = [&](int64_t tile) -> Meta {
        Meta mt;
        mt.t = wlocate<WW>(a, tile);
        mt.first = tile - mt.t.r0 / (WW * 64);            // the batch's first tile
        mt.p0 = one ? a.cols0[pc0] : a.cols[(int64_t)pc0 * a.t.nchunks + mt.t.c];
        return mt;
    };
    const int64_t tstride = (int64_t)gridDim.x * kWaves;
    // Look-back mode hands tiles out in order — a tile's predecessors are then finished or held by a running wave, whatever
    // part of the grid is resident.  ONE ticket counter serialises a million same-address atomics (23 ns each, measured:
    // 23 ms per 1e9 rows); 64 counters a cache line apart, wave (block, wave) drawing tile 64 v + its counter's number, take
    // 1/64 of that each — every counter is served by many waves, so the counters advance together.
    constexpr int kCounters = 64;
    const int my_counter = (int)((blockIdx.x * kWaves + wave) % kCounters);
    auto take = [&](int64_t prev) -> int64_t {           // the next tile of this wave
        if (!fa.lookback) return prev < 0 ? (int64_t)blockIdx.x * kWaves + wave : prev + tstride;
        unsigned int tk = 0;
        if (lane == 0) tk = atomicAdd(fa.ticket + my_counter * 32, 1u);
        return (int64_t)(unsigned int)__builtin_amdgcn_readfirstlane((int)tk) * kCounters + my_counter;
    };
    int64_t tile = take(-1);
    Meta meta;
    if (!fa.lookback && tile < a.t.ntiles) meta = locate_all(tile);
    while (tile < a.t.ntiles) {
        // look-back mode: a ticket is drawn only when the wave is ready to work on it — a tile held while its wave still compacts
        // the previous one keeps every later tile of the batch waiting for its row count
        if (fa.lookback) meta = locate_all(tile);
        const WTile t = meta.t;
        const int64_t first = meta.first;
        if (t.c != cur_chunk) {
            if (cur_chunk >= 0 && lane < a.ncols && nullacc[wave][lane]) {
                atomicAdd((unsigned long long*)&a.out_null_counts[(int64_t)lane * a.t.nchunks + cur_chunk], (unsigned long long)nullacc[wave][lane]);
                nullacc[wave][lane] = 0;
            }
            __builtin_amdgcn_wave_barrier();
            cur_chunk = t.c;
        }
        const bool full = t.r0 + WW * 64 <= t.clen;
        const int64_t avail = t.clen - t.r0;                  // rows of the batch from the tile's first row on
        // Round 6: a tile at the END of its batch takes the DMA path too (clamped vector addresses, the vector across the end patched
        // in, the keep bits of rows past the end cleared) — batches of 1000 rows, whose every tile is one, ran at 8.3 ms per 1e9 rows
        // of an f64 + i32 frame against 3.5 for 1024-row batches.
        auto dma_ok = [&](const DevChunkCol& c, int e) {
            return (full || (ENDS && avail * e >= 16)) && (e == 8 || e == 4) && (((uintptr_t)((const char*)c.values + (c.offset + t.r0) * e)) & 15) == 0;
        };
        auto nvec_of = [&](int e) { return (int)(avail * e / 16); };
        // ---- the predicate, on the tile(s) it reads
        DevChunkCol col = meta.p0;
        int es = a.esize[pc0];
        bool dma = dma_ok(col, es);
        auto dma_in = [&](const DevChunkCol& c, int e) __attribute__((always_inline)) {
            if (!ENDS || full) { if (e == 8) dma_tile<uint64_t>(c, t.r0, stage[wave]); else dma_tile<uint32_t>(c, t.r0, stage[wave]); }
            else if (e == 8) dma_tile_end<uint64_t>(c, t.r0, stage[wave], nvec_of(8));
            else dma_tile_end<uint32_t>(c, t.r0, stage[wave], nvec_of(4));
        };
        if (dma) dma_in(col, es);
        LaneWin<WW> qv = lane_windows_issue<WW>(col.validity, col.offset + t.r0, col.validity ? t.clen - t.r0 : 0);
        int64_t next = 0;
        if (!fa.lookback) {
            next = take(tile);
            if (next < a.t.ntiles) meta = locate_all(next);           // under the loads just issued
        }
        if (dma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (ENDS) { if (dma && !full) { if (es == 8) dma_tail<uint64_t>(col, t.r0, avail, stage[wave]); else dma_tail<uint32_t>(col, t.r0, avail, stage[wave]); } }
        uint64_t kwv = pred_words(fa.term[0], col, t.r0, t.clen, dma, stage[wave]);
        if constexpr (ENDS) { if (dma && !full) kwv &= tail_words(avail); }
        if (col.validity) kwv &= lane_windows_finish<WW>(qv);
        int in_lds = dma ? pc0 : -1;
        if (fa.nterms > 1) {
            if (pc1 != pc0) {
                col = one ? a.cols0[pc1] : a.cols[(int64_t)pc1 * a.t.nchunks + t.c];
                es = a.esize[pc1];
                dma = dma_ok(col, es);
                if (dma) {
                    dma_in(col, es);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if constexpr (ENDS) { if (!full) { if (es == 8) dma_tail<uint64_t>(col, t.r0, avail, stage[wave]); else dma_tail<uint32_t>(col, t.r0, avail, stage[wave]); } }
                }
                in_lds = dma ? pc1 : -1;
            }
            uint64_t k1 = pred_words(fa.term[1], col, t.r0, t.clen, in_lds == pc1, stage[wave]);
            if constexpr (ENDS) { if (in_lds == pc1 && !full) k1 &= tail_words(avail); }
            uint64_t v0 = ~0ull, v1 = ~0ull;                          // Arrow's and / or: NULL if either side is NULL -> the row is dropped
            if (col.validity) v1 = lane_windows<WW>(col.validity, col.offset + t.r0, t.clen - t.r0);
            // (term 0's validity is already folded into kwv: a NULL there cleared the bit; for OR the other side must not set it again)
            const DevChunkCol c0 = one ? a.cols0[pc0] : a.cols[(int64_t)pc0 * a.t.nchunks + t.c];
            if (c0.validity) v0 = lane_windows<WW>(c0.validity, c0.offset + t.r0, t.clen - t.r0);
            kwv = (fa.combine == RDF_OP_AND ? (kwv & k1) : (kwv | k1)) & v0 & v1;
        }
        int cnt = __popcll(kwv);
#pragma unroll
        for (int d = 1; d < WW; d <<= 1) cnt += __shfl_xor(cnt, d);
        cnt = __builtin_amdgcn_readfirstlane(cnt);
        // ---- where the kept rows go inside the batch's output
        int64_t wave_out = 0;
        const bool last = t.r0 + WW * 64 >= t.clen;
        if (fa.lookback) {
            // Two levels, so that a tile never walks the whole window of tiles in flight (4096 waves: a flat look-back spent
            // ~60 dependent memory round trips per tile, 23 ms per 1e9 rows of ONE batch): inside a super-tile of 64 tiles the
            // predecessors' row counts come with ONE 64-wide read; the super-tiles' totals form a second chain, 64 times shorter,
            // walked the same way.  A tile publishes its count before it waits for anything, a super-tile's last tile publishes
            // the total as soon as its own local prefix is known: nobody waits for a tile that waits.
            auto ld = [](const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
            auto st = [&](unsigned long long* p, unsigned long long v) { if (lane == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
            unsigned long long* super_state = fa.tile_state + a.t.ntiles + 8;
            st(fa.tile_state + tile, kFfAggregate | (unsigned long long)cnt);
            const int64_t j = tile - first;                 // tile of its batch
            const int64_t sfirst = first + (j & ~63ll);     // first tile of its super-tile
            const int nb = (int)(tile - sfirst);
            long long local = 0, before = 0;
            const bool closes = (j & 63) == 63 || last;    // the super-tile's last tile
            int64_t q = (j >> 6) - 1;                       // super-tiles of this batch before mine
            // Round 5 (lookback == 2): the walk over the super-tiles' totals — one word per super-tile, a cache line apart: up to 64
            // line fetches per step — is done by ONE tile per super-tile, its first, which leaves what it found (the rows of the batch
            // in front of the super-tile) in a word of its own; the other 63 tiles read that one word — with lane 63 of the SAME
            // 64-wide read that brings their neighbours' counts (a tile has at most 63 neighbours before it), so a tile of a batch of
            // any length pays one memory round trip for its position, as a tile of a 64-tile batch does.  Every wait is for a tile
            // with an earlier ticket, as before.
            unsigned long long* super_excl = super_state + a.t.ntiles + 8;
            const bool follower = fa.lookback >= 2 && q >= 0 && (j & 63) != 0;
            bool need_local = nb > 0, need_before = follower;
            if (!need_local && closes) st(super_state + sfirst, kFfAggregate | (unsigned long long)cnt);
            while (need_local || need_before) {
                const unsigned long long w = need_local && lane < nb ? ld(fa.tile_state + sfirst + lane) : kFfAggregate;
                const unsigned long long lw = need_before && lane == 63 ? ld(super_excl + sfirst) : kFfPrefix;
                bool progress = false;
                if (need_local && __ballot((w >> 62) != 0) == ~0ull) {
                    local = lane < nb ? (long long)(w & kFfValue) : 0;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) local += __shfl_xor(local, d);
                    need_local = false;
                    progress = true;
                    // the super-tile's total leaves as soon as it is known: the tile that walks the totals never waits for a tile that waits
                    if (closes) st(super_state + sfirst, kFfAggregate | (unsigned long long)(local + cnt));
                }
                if (need_before && __ballot((lw >> 62) != 0) == ~0ull) {
                    before = (long long)(__shfl(lw, 63) & kFfValue);
                    need_before = false;
                    progress = true;
                }
                if (!progress) __builtin_amdgcn_s_sleep(2);
            }
            if (follower) q = -1;
            const bool leads = fa.lookback >= 2 && q >= 0;
            // lookback == 3: the leader adds up the nearest super-tiles from their tiles' OWN counts (one 64-wide read each, all in
            // flight together with the walk over the older totals) instead of waiting for each of them to be added up by its last
            // tile first — a total is one more memory round trip away than the counts it is made of, and every tile of the leader's
            // super-tile waits that long.  Super-tiles further back than kRawSupers have their totals out by the time anybody asks.
            constexpr int kRawSupers = 8;
            const int nraw = leads && fa.lookback == 3 ? (int)(q + 1 < kRawSupers ? q + 1 : kRawSupers) : 0;
            bool need_raw = nraw > 0;
            q -= nraw;
            while (q >= 0 || need_raw) {
                unsigned long long r[kRawSupers];
#pragma unroll
                for (int k = 0; k < kRawSupers; ++k) r[k] = need_raw && k < nraw ? ld(fa.tile_state + sfirst - 64 * (k + 1) + lane) : kFfAggregate;
                const int64_t idx = q - lane;
                const unsigned long long w = q >= 0 && idx >= 0 ? ld(super_state + first + idx * 64) : kFfPrefix;
                bool progress = false;
                if (need_raw) {
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < kRawSupers; ++k) ok = ok && (r[k] >> 62) != 0;
                    if (__ballot(ok) == ~0ull) {
                        long long v = 0;
#pragma unroll
                        for (int k = 0; k < kRawSupers; ++k) v += (long long)(r[k] & kFfValue);
#pragma unroll
                        for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);
                        before += v;
                        need_raw = false;
                        progress = true;
                    }
                }
                if (q >= 0) {
                    const uint64_t ready = __ballot((w >> 62) != 0), pref = __ballot((w >> 62) == 2);
                    const int pl = pref ? __builtin_ctzll(pref) : 63;                    // the nearest super-tile that knows its prefix
                    const uint64_t need = pl == 63 ? ~0ull : ((2ull << pl) - 1);
                    if ((ready & need) == need) {
                        long long v = lane <= pl ? (long long)(w & kFfValue) : 0;
#pragma unroll
                        for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);
                        before += v;
                        q = pref ? -1 : q - 64;
                        progress = true;
                    }
                }
                if (!progress) __builtin_amdgcn_s_sleep(2);
            }
            if (leads) st(super_excl + sfirst, kFfPrefix | (unsigned long long)before);
            if (closes) st(super_state + sfirst, kFfPrefix | (unsigned long long)(before + local + cnt));
            wave_out = before + local;
        }
        if (last && lane == 0) fa.out_len[t.c] = wave_out + cnt;
        if (cnt > 0) {
            const bool sparse = cnt * 8 < WW * 64;        // few rows kept: the other columns fetch only the sectors that hold one
#pragma unroll 1
            for (int kk = 0; kk < a.ncols; ++kk) {
                // the column whose tile is already in LDS goes first
                const int k = in_lds < 0 ? kk : (kk == 0 ? in_lds : (kk <= in_lds ? kk - 1 : kk));
                const bool have = k == in_lds;
                col = one ? a.cols0[k] : a.cols[(int64_t)k * a.t.nchunks + t.c];
                es = a.esize[k];
                dma = have || (!sparse && dma_ok(col, es));
                if (dma && !have) dma_in(col, es);
                const DevOutChunk oc = one ? a.outs0[k] : a.outs[(int64_t)k * a.t.nchunks + t.c];
                const bool vec_out = (((uintptr_t)oc.values) & 15) == 0;
                uint32_t nn = 0;
                if (dma && vec_out) {
                    if (es == 8) dma_compact<uint64_t, ENDS>(col, oc, t.r0, t.clen, wave_out, kwv, cnt, stage[wave], vstage[wave], nn);
                    else dma_compact<uint32_t, ENDS>(col, oc, t.r0, t.clen, wave_out, kwv, cnt, stage[wave], vstage[wave], nn);
                } else {
                    if (dma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (es == 8) slow_compact<uint64_t>(col, oc, t.r0, wave_out, kwv, cnt, stage[wave], vstage[wave], nn);
                    else slow_compact<uint32_t>(col, oc, t.r0, wave_out, kwv, cnt, stage[wave], vstage[wave], nn);
                }
                if (lane == 0 && nn) nullacc[wave][k] += nn;
            }
        }
        tile = fa.lookback ? take(tile) : next;
    }
    __builtin_amdgcn_wave_barrier();
    if (cur_chunk >= 0 && lane < a.ncols && nullacc[wave][lane])
        atomicAdd((unsigned long long*)&a.out_null_counts[(int64_t)lane * a.t.nchunks + cur_chunk], (unsigned long long)nullacc[wave][lane]);
}

static int wgrid(int64_t ntiles, int per_cu) {
    const int64_t blocks = (ntiles + kBlock / 64 - 1) / (kBlock / 64);
    const int64_t lim = (int64_t)eval_grid_limit() / 8 * per_cu;
    return (int)(blocks < lim ? (blocks < 1 ? 1 : blocks) : lim);
}
hipError_t launch_fcount(const FilterWArgs& a, int tile_rows, int64_t* tile_counts, hipStream_t s) {
    if (a.t.ntiles <= 0) return hipSuccess;
    const int grid = wgrid(a.t.ntiles, 8);
    if (tile_rows == kWDmaTile) hipLaunchKernelGGL(fcount_kernel<16>, dim3(grid), dim3(kBlock), 0, s, a, tile_counts);
    else if (tile_rows == kWTileSmall) hipLaunchKernelGGL(fcount_kernel<4>, dim3(grid), dim3(kBlock), 0, s, a, tile_counts);
    else hipLaunchKernelGGL(fcount_kernel<8>, dim3(grid), dim3(kBlock), 0, s, a, tile_counts);
    return hipGetLastError();
}
hipError_t launch_ffilter(const FusedFilterArgs& a, hipStream_t s) {
    if (a.w.t.ntiles <= 0) return hipSuccess;
    if (a.ends) hipLaunchKernelGGL(ffilter_dma_kernel<true>, dim3(wgrid(a.w.t.ntiles, 4)), dim3(kBlock), 0, s, a);
    else hipLaunchKernelGGL(ffilter_dma_kernel<false>, dim3(wgrid(a.w.t.ntiles, 4)), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_fcompact(const FilterWArgs& a, int tile_rows, hipStream_t s) {
    if (a.t.ntiles <= 0) return hipSuccess;
    if (tile_rows == kWDmaTile) {
        if (a.ends) hipLaunchKernelGGL(fcompact_dma_kernel<true>, dim3(wgrid(a.t.ntiles, 4)), dim3(kBlock), 0, s, a);
        else hipLaunchKernelGGL(fcompact_dma_kernel<false>, dim3(wgrid(a.t.ntiles, 4)), dim3(kBlock), 0, s, a);
        return hipGetLastError();
    }
    int es = a.esize[0];
    for (int k = 1; k < a.ncols; ++k) if (a.esize[k] != es) es = 0;
    const int grid = wgrid(a.t.ntiles, tile_rows == kWTileSmall ? 8 : 4);   // 19 KB of LDS per block at WW = 8
    const bool pf = a.t.nchunks > 1 && a.prefetch;   // one chunk: nothing to look up
#define RDF_FCOMPACT_LAUNCH(WW, PF)                                                                                    \
    switch (es) {                                                                                                      \
        case 8: hipLaunchKernelGGL((fcompact_kernel<8, WW, PF>), dim3(grid), dim3(kBlock), 0, s, a); break;            \
        case 4: hipLaunchKernelGGL((fcompact_kernel<4, WW, PF>), dim3(grid), dim3(kBlock), 0, s, a); break;            \
        case 2: hipLaunchKernelGGL((fcompact_kernel<2, WW, PF>), dim3(grid), dim3(kBlock), 0, s, a); break;            \
        case 1: hipLaunchKernelGGL((fcompact_kernel<1, WW, PF>), dim3(grid), dim3(kBlock), 0, s, a); break;            \
        default: hipLaunchKernelGGL((fcompact_kernel<0, WW, PF>), dim3(grid), dim3(kBlock), 0, s, a); break;           \
    }
    if (tile_rows == kWTileSmall) { if (pf) { RDF_FCOMPACT_LAUNCH(4, true) } else { RDF_FCOMPACT_LAUNCH(4, false) } }
    else { if (pf) { RDF_FCOMPACT_LAUNCH(8, true) } else { RDF_FCOMPACT_LAUNCH(8, false) } }
#undef RDF_FCOMPACT_LAUNCH
    return hipGetLastError();
}

}  // namespace rdfk
