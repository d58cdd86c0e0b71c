// rdf_frameops.hip — device side of the frame-level operators (rdf_filter_frame / rdf_take_columns / rdf_take_frame /
// rdf_sort_frame / rdf_groupby_agg_frame): what turns "a DataFrame holds its RecordBatches for its lifetime"
// (src/dataframe.rs:30-48) into device-resident descriptor tables, so that an operator over a frame of a million
// 1024-row batches (src/dataframe.rs:352) launches kernels and never walks the batch list on the host.
//
//   frame_totals_kernel    per-chunk kept rows of a compaction from the per-tile scan; 64-row padded lengths
//   frame_tables_kernel    positions (scan of the padded lengths) -> output descriptors of every column and the
//                          descriptor table of the frame the operator returns
//   take_cols_kernel       DataFrame::take's per-column loop (src/dataframe.rs:216-222, 705-711) as ONE gather pass: the
//                          index list is read once, the row -> chunk lookup is done once, and the M gathers of a row are
//                          in flight together
#include "rdf_common.hip.h"

namespace rdfk {

typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));

// kept rows of chunk c = scan[tile_start[c + 1]] - scan[tile_start[c]]; the returned frame keeps the batch boundaries
// (ChunkedArray::filter, src/table.rs:97-107) and starts every batch on a 64-row boundary of the column's buffer, so
// values stay 16-byte aligned and bitmaps 8-byte aligned whatever the keep counts are.
__global__ __launch_bounds__(kBlock) void frame_totals_kernel(const int64_t* __restrict__ tile_scan, const int64_t* __restrict__ chunk_tile_start,
                                                              int64_t nchunks, int64_t* __restrict__ out_len, int64_t* __restrict__ padded) {
    for (int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x; c < nchunks; c += (int64_t)gridDim.x * kBlock) {
        const int64_t n = tile_scan[chunk_tile_start[c + 1]] - tile_scan[chunk_tile_start[c]];
        out_len[c] = n;
        padded[c] = (n + 63) & ~(int64_t)63;
    }
}

__global__ __launch_bounds__(kBlock) void frame_tables_kernel(const FrameTabArgs a) {
    for (int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x; c < a.nchunks; c += (int64_t)gridDim.x * kBlock) {
        const int64_t pos = a.pos[c];
#pragma unroll 1
        for (int k = 0; k < a.ncols; ++k) {
            char* v = a.values[k] + pos * a.esize[k];
            uint8_t* b = a.validity[k] ? a.validity[k] + (pos >> 3) : nullptr;
            const int64_t i = (int64_t)k * a.nchunks + c;
            if (a.outs) a.outs[i] = DevOutChunk{v, b};
            if (a.cols) a.cols[i] = DevChunkCol{v, b, 0};
        }
    }
}

// Layout of a frame's OWN buffers when the batch lengths are known on the device only: one kernel writes the descriptors of
// a column whose chunk c starts `pos[c]` elements into the buffer.  BOOL columns (bit-packed masks) use pos / 8 bytes.
__global__ __launch_bounds__(kBlock) void frame_mask_tables_kernel(const int64_t* __restrict__ pos, int64_t nchunks, uint8_t* values, uint8_t* validity,
                                                                  DevOutChunk* __restrict__ outs, DevChunkCol* __restrict__ cols) {
    for (int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x; c < nchunks; c += (int64_t)gridDim.x * kBlock) {
        uint8_t* v = values + (pos[c] >> 3);
        uint8_t* b = validity ? validity + (pos[c] >> 3) : nullptr;
        outs[c] = DevOutChunk{v, b};
        cols[c] = DevChunkCol{v, nullptr, 0};     // the value bit of a NULL predicate row is 0: compaction needs the values only
    }
}

// Keep counts per compaction tile straight from the frame's own mask: batch c's bits start at bit pos[c] (a multiple of 64) of
// one buffer, so a tile's count is the popcount of at most tile_rows / 64 consecutive words — one thread per tile, no wave-wide
// window loads, no descriptor per tile (fcount_kernel needs 0.69 ms per 1e9 rows in 1024-row batches for the same numbers).
__global__ __launch_bounds__(kBlock) void frame_mask_count_kernel(const uint64_t* __restrict__ mask, const int64_t* __restrict__ pos,
                                                                  const int64_t* __restrict__ chunk_tile_start, const int64_t* __restrict__ chunk_len,
                                                                  int64_t nchunks, int64_t ntiles, int tile_rows, uint64_t tile_inv, int64_t* __restrict__ counts) {
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < ntiles; t += (int64_t)gridDim.x * kBlock) {
        const int64_t c = find_chunk_tile_inv(chunk_tile_start, nchunks, t, tile_inv);
        const int64_t row0 = (t - chunk_tile_start[c]) * tile_rows;
        int64_t n = chunk_len[c] - row0;
        n = n > tile_rows ? tile_rows : n;
        const uint64_t* w = mask + ((pos[c] + row0) >> 6);
        int64_t cnt = 0;
        for (int64_t i = 0; i * 64 < n; ++i) {
            uint64_t x = __builtin_nontemporal_load(as_global<uint64_t>(w) + i);
            if (n - i * 64 < 64) x &= (1ull << (n - i * 64)) - 1;
            cnt += __popcll(x);
        }
        counts[t] = cnt;
    }
}
hipError_t launch_frame_mask_count(const uint64_t* mask, const int64_t* pos, const int64_t* chunk_tile_start, const int64_t* chunk_len, int64_t nchunks,
                                   int64_t ntiles, int tile_rows, uint64_t tile_inv, int64_t* counts, hipStream_t s) {
    if (ntiles <= 0) return hipSuccess;
    const int64_t grid = std::min<int64_t>((ntiles + kBlock - 1) / kBlock, (int64_t)eval_grid_limit() * 4);
    hipLaunchKernelGGL(frame_mask_count_kernel, dim3((unsigned)grid), dim3(kBlock), 0, s, mask, pos, chunk_tile_start, chunk_len, nchunks, ntiles, tile_rows, tile_inv, counts);
    return hipGetLastError();
}

// padded[c] = round_up(len[c], 64)
__global__ __launch_bounds__(kBlock) void frame_pad_kernel(const int64_t* __restrict__ len, int64_t n, int64_t* __restrict__ padded) {
    for (int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x; c < n; c += (int64_t)gridDim.x * kBlock) padded[c] = (len[c] + 63) & ~(int64_t)63;
}

// ------------------------------------------------------------------------------------------------
// take over every column of a frame at once.  A wave owns kTakeU x 64 consecutive output rows per iteration: their indices
// come in with kTakeU coalesced loads, each lane resolves its kTakeU rows to (chunk, element) ONCE, and then the
// kTakeU x ncols gathers are issued back to back — a random 8-byte read costs a whole memory transaction whichever column
// it is for, so what the per-column loop of the reference pays M times over (index read, lookup, latency) is paid once,
// and the M transactions of a row overlap.

template <typename T>
__device__ __forceinline__ void take_store(const DevOutChunk& out, int64_t j, bool inr, T v) {
    if (inr) __builtin_nontemporal_store(v, as_global_mut<T>(out.values) + j);
}

template <typename IDX, int kTakeU>
__global__ __launch_bounds__(kBlock) void take_cols_kernel(const TakeColsArgs a) {
    const int lane = threadIdx.x & 63;
    uint32_t err = 0;
    uint32_t nulls[kMaxFilterCols];
#pragma unroll
    for (int k = 0; k < kMaxFilterCols; ++k) nulls[k] = 0;
    const int64_t per_wave = (int64_t)kTakeU * 64;
    const int64_t nw = (a.n + per_wave - 1) / per_wave;
    const double inv = a.uniform_len > 0 ? 1.0 / (double)a.uniform_len : chunk_lookup_scale(a.chunk_row_start, a.nchunks);
    for (int64_t wv = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); wv < nw; wv += (int64_t)gridDim.x * (kBlock / 64)) {
        const int64_t j0 = wv * per_wave;
        uint64_t ix[kTakeU];
        bool valid[kTakeU], inr[kTakeU];
        // indices (+ their validity words)
#pragma unroll
        for (int u = 0; u < kTakeU; ++u) {
            const int64_t j = j0 + u * 64 + lane;
            inr[u] = j < a.n;
            ix[u] = inr[u] ? (uint64_t)__builtin_nontemporal_load(as_global<IDX>(a.indices.values) + a.indices.offset + j) : 0;
        }
#pragma unroll
        for (int u = 0; u < kTakeU; ++u) {
            valid[u] = inr[u];
            if (a.indices.validity) {
                const uint64_t w = load_bits64(a.indices.validity, a.indices.offset + j0 + u * 64, clamp64(a.n - (j0 + u * 64)));
                valid[u] = valid[u] && ((w >> lane) & 1);
            }
            if (valid[u] && ix[u] >= (uint64_t)a.total_rows) { err |= 2u; valid[u] = false; }
        }
        // row -> (chunk, element within the chunk): once for all columns
        int64_t ch[kTakeU], el[kTakeU];
#pragma unroll
        for (int u = 0; u < kTakeU; ++u) {
            ch[u] = 0; el[u] = (int64_t)ix[u];
            if (a.need_lookup && valid[u]) {
                if (a.uniform_len > 0) {
                    int64_t g = (int64_t)((double)ix[u] * inv);
                    if (g * a.uniform_len > (int64_t)ix[u]) --g;
                    else if ((g + 1) * a.uniform_len <= (int64_t)ix[u]) ++g;
                    ch[u] = g; el[u] = (int64_t)ix[u] - g * a.uniform_len;
                } else {
                    ch[u] = find_chunk_row(a.chunk_row_start, a.nchunks, (int64_t)ix[u], inv);
                    el[u] = (int64_t)ix[u] - a.chunk_row_start[ch[u]];
                }
            }
        }
#pragma unroll 1
        for (int k = 0; k < a.ncols; ++k) {
            const int es = a.esize[k];
            uint64_t v[kTakeU];
            bool vv[kTakeU];
            int64_t e[kTakeU];
            const uint8_t* vb[kTakeU];
            // A column whose batches are consecutive slices of one buffer is ONE descriptor, read from the kernel arguments on the
            // scalar unit; only a column of separately allocated batches fetches a descriptor per row.  (As one `cc = contig ?
            // cols0[k] : table[..]` the compiler selected between two POINTERS and loaded the descriptor per lane through the flat
            // path for both — a dependent memory latency in front of every gather: a sequential take ran at 0.34 of peak where
            // take_kernel reaches 0.63.)
            auto fetch = [&](int u, const DevChunkCol& cc, int64_t within) {
                e[u] = cc.offset + within;
                vb[u] = cc.validity;
                switch (es) {
                    case 8: v[u] = as_global<uint64_t>(cc.values)[e[u]]; break;
                    case 4: v[u] = as_global<uint32_t>(cc.values)[e[u]]; break;
                    case 2: v[u] = as_global<uint16_t>(cc.values)[e[u]]; break;
                    default: v[u] = as_global<uint8_t>(cc.values)[e[u]]; break;
                }
            };
#pragma unroll
            for (int u = 0; u < kTakeU; ++u) { v[u] = 0; vv[u] = valid[u]; vb[u] = nullptr; e[u] = 0; }
            if (a.contig[k]) {
                DevChunkCol c0;
                c0.values = (const void*)uniform64((uint64_t)(uintptr_t)a.cols0[k].values);
                c0.validity = (const uint8_t*)uniform64((uint64_t)(uintptr_t)a.cols0[k].validity);
                c0.offset = (int64_t)uniform64((uint64_t)a.cols0[k].offset);
#pragma unroll
                for (int u = 0; u < kTakeU; ++u) if (valid[u]) fetch(u, c0, (int64_t)ix[u]);
            } else {
#pragma unroll
                for (int u = 0; u < kTakeU; ++u) {
                    if (!valid[u]) continue;
                    const DevChunkCol* t = a.cols_tab + ((int64_t)k * a.nchunks + ch[u]);
                    DevChunkCol cc;
                    cc.values = t->values; cc.validity = t->validity; cc.offset = t->offset;
                    fetch(u, cc, el[u]);
                }
            }
            if (a.col_nullable[k]) {
#pragma unroll
                for (int u = 0; u < kTakeU; ++u)
                    if (vv[u] && vb[u]) vv[u] = (as_global<uint8_t>(vb[u])[e[u] >> 3] >> (e[u] & 7)) & 1;
            }
            const DevOutChunk out = a.outs[k];
#pragma unroll
            for (int u = 0; u < kTakeU; ++u) {
                const int64_t j = j0 + u * 64 + lane;
                switch (es) {
                    case 8: take_store<uint64_t>(out, j, inr[u], v[u]); break;
                    case 4: take_store<uint32_t>(out, j, inr[u], (uint32_t)v[u]); break;
                    case 2: take_store<uint16_t>(out, j, inr[u], (uint16_t)v[u]); break;
                    default: take_store<uint8_t>(out, j, inr[u], (uint8_t)v[u]); break;
                }
                if (out.validity) {
                    const uint64_t bal = __ballot(vv[u]);
                    const uint64_t ib = __ballot(inr[u]);
                    if (lane == 0 && ib) {
                        as_global_mut<uint64_t>(out.validity)[(j0 >> 6) + u] = bal;
                        nulls[k] += (uint32_t)__popcll(ib & ~bal);
                    }
                }
            }
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < kMaxFilterCols; ++k)
            if (k < a.ncols && nulls[k]) atomicAdd((unsigned long long*)&a.out_null_counts[k], (unsigned long long)nulls[k]);
    }
    if (err) atomicOr(a.flags, err);
}

// ------------------------------------------------------------------------------------------------
// take through row records.  A random 8-byte read costs a whole 128-byte memory transaction (PMC: 140 B per gathered element,
// profiles/r02_pmc_hbm_traffic_by_kernel.json), so gathering M columns by the same index list costs M transactions per row
// however the loads are scheduled.  When the index list is long compared with the frame (DataFrame::sort and join take EVERY
// row, src/dataframe.rs:216-222, 705-711) it is cheaper to first interleave the columns into row records of NSLOT 8-byte
// slots (one streaming pass: M x 8 bytes read, one record written per row) and then gather RECORDS: one transaction per row
// brings all M values, which leave as M coalesced column stores.  Validity bits of the row ride in one more slot.
template <int NSLOT>
__global__ __launch_bounds__(kBlock) void rows_pack_kernel(const TakeRowsArgs a) {
    const double inv = a.uniform_len > 0 ? 1.0 / (double)a.uniform_len : chunk_lookup_scale(a.chunk_row_start, a.nchunks);
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < a.total_rows; r += (int64_t)gridDim.x * kBlock) {
        int64_t ch = 0, el = r;
        if (a.need_lookup) {
            if (a.uniform_len > 0) {
                int64_t g = (int64_t)((double)r * inv);
                if (g * a.uniform_len > r) --g;
                else if ((g + 1) * a.uniform_len <= r) ++g;
                ch = g; el = r - g * a.uniform_len;
            } else {
                ch = find_chunk_row(a.chunk_row_start, a.nchunks, r, inv);
                el = r - a.chunk_row_start[ch];
            }
        }
        uint64_t rec[NSLOT];
        uint64_t flags = 0;
#pragma unroll
        for (int k = 0; k < NSLOT; ++k) {
            rec[k] = 0;
            if (k >= a.ncols) continue;
            DevChunkCol cc = a.cols0[k];
            int64_t within = r;
            if (!a.contig[k]) {
                const DevChunkCol* t = a.cols_tab + ((int64_t)k * a.nchunks + ch);
                cc.values = t->values; cc.validity = t->validity; cc.offset = t->offset;
                within = el;
            }
            const int64_t e = cc.offset + within;
            switch (a.esize[k]) {
                case 8: rec[k] = as_global<uint64_t>(cc.values)[e]; break;
                case 4: rec[k] = as_global<uint32_t>(cc.values)[e]; break;
                case 2: rec[k] = as_global<uint16_t>(cc.values)[e]; break;
                default: rec[k] = as_global<uint8_t>(cc.values)[e]; break;
            }
            bool ok = true;
            if (cc.validity) ok = (as_global<uint8_t>(cc.validity)[e >> 3] >> (e & 7)) & 1;
            flags |= (uint64_t)ok << k;
        }
        if (a.flag_slot >= 0) {
#pragma unroll
            for (int k = 0; k < NSLOT; ++k) if (k == a.flag_slot) rec[k] = flags;
        }
        GlobalMutPtr<u64x2> dst = (GlobalMutPtr<u64x2>)(a.recs + r * NSLOT);
#pragma unroll
        for (int k = 0; k < NSLOT; k += 2) { u64x2 w; w[0] = rec[k]; w[1] = rec[k + 1]; __builtin_nontemporal_store(w, dst + k / 2); }
    }
}

template <int NSLOT, typename IDX>
__global__ __launch_bounds__(kBlock) void rows_gather_kernel(const TakeRowsArgs a) {
    const int lane = threadIdx.x & 63;
    uint32_t err = 0;
    uint32_t nulls[NSLOT];
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) nulls[k] = 0;
    const int64_t nw = (a.n + 63) >> 6;
    for (int64_t wv = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); wv < nw; wv += (int64_t)gridDim.x * (kBlock / 64)) {
        const int64_t j = wv * 64 + lane;
        const bool inr = j < a.n;
        bool valid = inr;
        if (a.indices.validity) {
            const uint64_t w = load_bits64(a.indices.validity, a.indices.offset + wv * 64, clamp64(a.n - wv * 64));
            valid = valid && ((w >> lane) & 1);
        }
        uint64_t ix = 0;
        if (valid) {
            ix = (uint64_t)__builtin_nontemporal_load(as_global<IDX>(a.indices.values) + a.indices.offset + j);
            if (ix >= (uint64_t)a.total_rows) { err |= 2u; valid = false; }
        }
        uint64_t rec[NSLOT];
#pragma unroll
        for (int k = 0; k < NSLOT; ++k) rec[k] = 0;
        if (valid) {
            GlobalPtr<u64x2> src = (GlobalPtr<u64x2>)(a.recs + ix * NSLOT);
#pragma unroll
            for (int k = 0; k < NSLOT; k += 2) { const u64x2 w = src[k / 2]; rec[k] = w[0]; rec[k + 1] = w[1]; }
        }
        uint64_t flags = ~0ull;
        if (a.flag_slot >= 0) {
#pragma unroll
            for (int k = 0; k < NSLOT; ++k) if (k == a.flag_slot) flags = rec[k];
        }
        const uint64_t ib = __ballot(inr);
#pragma unroll
        for (int k = 0; k < NSLOT; ++k) {
            if (k >= a.ncols) continue;
            const DevOutChunk out = a.outs[k];
            if (inr) {
                switch (a.esize[k]) {
                    case 8: __builtin_nontemporal_store(rec[k], as_global_mut<uint64_t>(out.values) + j); break;
                    case 4: __builtin_nontemporal_store((uint32_t)rec[k], as_global_mut<uint32_t>(out.values) + j); break;
                    case 2: __builtin_nontemporal_store((uint16_t)rec[k], as_global_mut<uint16_t>(out.values) + j); break;
                    default: __builtin_nontemporal_store((uint8_t)rec[k], as_global_mut<uint8_t>(out.values) + j); break;
                }
            }
            if (out.validity) {
                const uint64_t bal = __ballot(valid && ((flags >> k) & 1));
                if (lane == 0) { as_global_mut<uint64_t>(out.validity)[wv] = bal; nulls[k] += (uint32_t)__popcll(ib & ~bal); }
            }
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NSLOT; ++k)
            if (k < a.ncols && nulls[k]) atomicAdd((unsigned long long*)&a.out_null_counts[k], (unsigned long long)nulls[k]);
    }
    if (err) atomicOr(a.flags, err);
}

template <int NSLOT>
static void launch_rows_t(const TakeRowsArgs& a, hipStream_t s) {
    int64_t g1 = (a.total_rows + kBlock - 1) / kBlock;
    if (g1 > (int64_t)eval_grid_limit() * 4) g1 = (int64_t)eval_grid_limit() * 4;
    if (g1 > 0) hipLaunchKernelGGL((rows_pack_kernel<NSLOT>), dim3((unsigned)g1), dim3(kBlock), 0, s, a);
    const int64_t nw = (a.n + 63) >> 6;
    int64_t g2 = (nw + (kBlock / 64) - 1) / (kBlock / 64);
    if (g2 > (int64_t)eval_grid_limit() * 4) g2 = (int64_t)eval_grid_limit() * 4;
    if (g2 <= 0) return;
    if (a.idx64) hipLaunchKernelGGL((rows_gather_kernel<NSLOT, uint64_t>), dim3((unsigned)g2), dim3(kBlock), 0, s, a);
    else hipLaunchKernelGGL((rows_gather_kernel<NSLOT, uint32_t>), dim3((unsigned)g2), dim3(kBlock), 0, s, a);
}
hipError_t launch_take_rows(const TakeRowsArgs& a, hipStream_t s) {
    switch (a.nslot) {
        case 2: launch_rows_t<2>(a, s); break;
        case 4: launch_rows_t<4>(a, s); break;
        case 8: launch_rows_t<8>(a, s); break;
        default: launch_rows_t<16>(a, s); break;
    }
    return hipGetLastError();
}

// How local is an index list?  4096 sampled neighbours: |idx[i + 1] - idx[i]| < 64 rows counts as "near".  A sorted or
// sequential list gathers whole lines column by column at streaming speed and must not take the row-record detour.
template <typename IDX>
__global__ __launch_bounds__(kBlock) void idx_locality_kernel(const DevChunkCol indices, int64_t n, unsigned int* near) {
    const int64_t samples = 4096;
    unsigned int c = 0;
    for (int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x; s < samples; s += (int64_t)gridDim.x * kBlock) {
        const int64_t i = (n - 1) / samples * s;
        if (i + 1 < n) {
            const int64_t a0 = (int64_t)as_global<IDX>(indices.values)[indices.offset + i], a1 = (int64_t)as_global<IDX>(indices.values)[indices.offset + i + 1];
            const int64_t d = a1 - a0;
            c += (d < 64 && d > -64) ? 1u : 0u;
        }
    }
    if (c) atomicAdd(near, c);
}
hipError_t launch_idx_locality(const DevChunkCol& indices, int64_t n, bool idx64, unsigned int* near, hipStream_t s) {
    if (idx64) hipLaunchKernelGGL((idx_locality_kernel<uint64_t>), dim3(16), dim3(kBlock), 0, s, indices, n, near);
    else hipLaunchKernelGGL((idx_locality_kernel<uint32_t>), dim3(16), dim3(kBlock), 0, s, indices, n, near);
    return hipGetLastError();
}

hipError_t launch_frame_totals(const int64_t* tile_scan, const int64_t* chunk_tile_start, int64_t nchunks, int64_t* out_len, int64_t* padded, hipStream_t s) {
    if (nchunks <= 0) return hipSuccess;
    const int64_t grid = std::min<int64_t>((nchunks + kBlock - 1) / kBlock, eval_grid_limit());
    hipLaunchKernelGGL(frame_totals_kernel, dim3((unsigned)grid), dim3(kBlock), 0, s, tile_scan, chunk_tile_start, nchunks, out_len, padded);
    return hipGetLastError();
}
hipError_t launch_frame_tables(const FrameTabArgs& a, hipStream_t s) {
    if (a.nchunks <= 0) return hipSuccess;
    const int64_t grid = std::min<int64_t>((a.nchunks + kBlock - 1) / kBlock, eval_grid_limit());
    hipLaunchKernelGGL(frame_tables_kernel, dim3((unsigned)grid), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_frame_mask_tables(const int64_t* pos, int64_t nchunks, uint8_t* values, uint8_t* validity, DevOutChunk* outs, DevChunkCol* cols, hipStream_t s) {
    if (nchunks <= 0) return hipSuccess;
    const int64_t grid = std::min<int64_t>((nchunks + kBlock - 1) / kBlock, eval_grid_limit());
    hipLaunchKernelGGL(frame_mask_tables_kernel, dim3((unsigned)grid), dim3(kBlock), 0, s, pos, nchunks, values, validity, outs, cols);
    return hipGetLastError();
}
hipError_t launch_frame_pad(const int64_t* len, int64_t n, int64_t* padded, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const int64_t grid = std::min<int64_t>((n + kBlock - 1) / kBlock, eval_grid_limit());
    hipLaunchKernelGGL(frame_pad_kernel, dim3((unsigned)grid), dim3(kBlock), 0, s, len, n, padded);
    return hipGetLastError();
}
template <int U>
static void launch_take_cols_u(const TakeColsArgs& a, hipStream_t s) {
    const int64_t nw = (a.n + (int64_t)U * 64 - 1) / ((int64_t)U * 64);
    int64_t grid = (nw + (kBlock / 64) - 1) / (kBlock / 64);
    if (grid > eval_grid_limit()) grid = eval_grid_limit();
    if (grid <= 0) return;
    if (a.idx64) hipLaunchKernelGGL((take_cols_kernel<uint64_t, U>), dim3((unsigned)grid), dim3(kBlock), 0, s, a);
    else hipLaunchKernelGGL((take_cols_kernel<uint32_t, U>), dim3((unsigned)grid), dim3(kBlock), 0, s, a);
}
hipError_t launch_take_cols(const TakeColsArgs& a, hipStream_t s) {
    // rows per lane and iteration: 4 puts the gathers of several columns of a row in flight together (random lists: the
    // transactions overlap); a single column gathers best one row at a time, like take_kernel (RDF_TAKE_U overrides for A/B)
    static const int forced = [] { const char* e = getenv("RDF_TAKE_U"); return e ? atoi(e) : 0; }();
    const int u = forced == 1 || forced == 2 || forced == 4 ? forced : 4;
    if (u == 1) launch_take_cols_u<1>(a, s);
    else if (u == 2) launch_take_cols_u<2>(a, s);
    else launch_take_cols_u<4>(a, s);
    return hipGetLastError();
}

}  // namespace rdfk
