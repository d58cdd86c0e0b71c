// rdf_list.hip — ArrayFunctions over List<primitive> columns (src/functions/array.rs:15-399): the per-row kernels of the
// reference (one ListArray row = value_slice(value_offset(i), value_length(i)) of the child values) as segmented device
// kernels.  Child validity is ignored exactly like the reference's value_slice() does.
//
//   list_find_kernel<T>  : contains / position on short lists, 64 rows per wave: the rows' contiguous span is compared
//                          element-parallel, a row's answer is the first set bit of its range of the match mask
//   list_extreme_kernel<T>: max / min on short lists, one LIST ROW per lane walking the LDS-staged span; results
//                          leave as coalesced stores / ballot-built bitmap words
//   list_wave_kernel<T>  : one list row per WAVE (long lists): lanes stride the slice, butterfly reduction
//   list_row_ids_kernel  : element -> row number (for the sort / remove compositions in rdf_capi.cpp)
//   list_remove_kernel<T>: array_remove in two passes (count per row -> scan -> write at the scanned offsets)
//   list_set_kernel<T>   : distinct / except / intersect / union / repeat, the same two passes with a membership rule
#include "rdf_common.hip.h"
#include <type_traits>

namespace rdfk {

template <class T> struct ListKey;   // order-preserving unsigned key (max / min) and equality
template <> struct ListKey<double> {
    static __device__ __forceinline__ uint64_t key(double v) { const uint64_t b = d2u(v); return (b >> 63) ? ~b : (b ^ 0x8000000000000000ull); }
    static __device__ __forceinline__ double unkey(uint64_t k) { return u2d((k >> 63) ? (k ^ 0x8000000000000000ull) : ~k); }
};
template <> struct ListKey<float> {
    static __device__ __forceinline__ uint64_t key(float v) { const uint32_t b = __float_as_uint(v); return (b >> 31) ? (uint32_t)~b : (b ^ 0x80000000u); }
    static __device__ __forceinline__ float unkey(uint64_t k) { const uint32_t x = (uint32_t)k; return __uint_as_float((x >> 31) ? (x ^ 0x80000000u) : ~x); }
};
#define RDF_LIST_INT_KEY(T, U, BIAS)                                                                              \
    template <> struct ListKey<T> {                                                                               \
        static __device__ __forceinline__ uint64_t key(T v) { return (uint64_t)((U)v ^ (U)BIAS); }                \
        static __device__ __forceinline__ T unkey(uint64_t k) { return (T)((U)k ^ (U)BIAS); }                     \
    };
RDF_LIST_INT_KEY(int8_t, uint8_t, 0x80u) RDF_LIST_INT_KEY(int16_t, uint16_t, 0x8000u) RDF_LIST_INT_KEY(int32_t, uint32_t, 0x80000000u)
RDF_LIST_INT_KEY(int64_t, uint64_t, 0x8000000000000000ull)
RDF_LIST_INT_KEY(uint8_t, uint8_t, 0) RDF_LIST_INT_KEY(uint16_t, uint16_t, 0) RDF_LIST_INT_KEY(uint32_t, uint32_t, 0) RDF_LIST_INT_KEY(uint64_t, uint64_t, 0)

template <class T> __device__ __forceinline__ T list_needle(uint64_t bits);
template <> __device__ __forceinline__ double list_needle<double>(uint64_t b) { return u2d(b); }
template <> __device__ __forceinline__ float list_needle<float>(uint64_t b) { return __uint_as_float((uint32_t)b); }
#define RDF_LIST_NEEDLE(T) template <> __device__ __forceinline__ T list_needle<T>(uint64_t b) { return (T)b; }
RDF_LIST_NEEDLE(int8_t) RDF_LIST_NEEDLE(int16_t) RDF_LIST_NEEDLE(int32_t) RDF_LIST_NEEDLE(int64_t)
RDF_LIST_NEEDLE(uint8_t) RDF_LIST_NEEDLE(uint16_t) RDF_LIST_NEEDLE(uint32_t) RDF_LIST_NEEDLE(uint64_t)

// NaN never wins a max / min unless every element is NaN (the column aggregates' rule)
template <class T> __device__ __forceinline__ bool list_is_nan(T v) { return v != v; }

constexpr int kListStage = 1024;   // child elements per wave staged in LDS (8 KiB for 8-byte children)
// LDS of the two instantiations: the match mask of contains / position (128 B per wave — occupancy is then set by
// registers alone) or the staged span of max / min
template <class T, bool FIND> struct ListRowsLds;
template <class T> struct ListRowsLds<T, true> { uint64_t match[kBlock / 64][kListStage / 64]; };
template <class T> struct ListRowsLds<T, false> { T stage[kBlock / 64][kListStage]; };

// What a wave needs to know about its 64 rows before it can touch the child values: the list validity word, the lane's
// own slice and the span covered by all 64 (value_offsets are monotone, also across NULL rows).  None of the loads
// depends on another, and the NEXT group's are issued before the current group's values are consumed: the
// offsets -> values dependency costs one memory round trip per wave iteration instead of two or three.
struct ListGroup { uint64_t lv; int32_t rb, re, span0, span1; };
__device__ __forceinline__ ListGroup list_group_load(const ListArgs& a, GlobalPtr<int32_t> off, int64_t wv, int lane) {
    ListGroup g;
    const int64_t row = wv * 64 + lane;
    const bool inr = row < a.n;
    g.lv = ~0ull;
    if (a.offsets.validity) g.lv = load_bits64(a.offsets.validity, a.offsets.offset + wv * 64, clamp64(a.n - wv * 64));
    g.rb = off[inr ? row : a.n];
    g.re = off[inr ? row + 1 : a.n];
    const int64_t rlast = wv * 64 + 64 < a.n ? wv * 64 + 64 : a.n;
    g.span0 = off[wv * 64];
    g.span1 = off[rlast];
    return g;
}

template <class T, bool FIND>
__device__ __forceinline__ void list_rows_impl(const ListArgs& a) {
    __shared__ ListRowsLds<T, FIND> lds;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const GlobalPtr<int32_t> off = as_global<int32_t>(a.offsets.values) + a.offsets.offset;
    const GlobalPtr<T> vals = as_global<T>(a.values.values) + a.values.offset;
    const T needle = list_needle<T>(a.needle);
    int nulls = 0;
    const int64_t nwaves = (a.n + 63) >> 6, stride = (int64_t)gridDim.x * (kBlock / 64);
    int64_t wv = (int64_t)blockIdx.x * (kBlock / 64) + w;
    ListGroup next = list_group_load(a, off, wv < nwaves ? wv : 0, lane);
    for (; wv < nwaves; wv += stride) {
        const ListGroup g = next;
        if (wv + stride < nwaves) next = list_group_load(a, off, wv + stride, lane);
        const int64_t row = wv * 64 + lane;
        const bool inr = row < a.n;
        const bool lvalid = inr && ((g.lv >> lane) & 1);
        const int32_t b = lvalid ? g.rb : 0, e = lvalid ? g.re : 0;
        const int32_t span0 = g.span0, span = g.span1 - g.span0;
        const bool staged = span <= kListStage;
        bool found = false;
        int32_t pos = 0;
        uint64_t best = a.op == LIST_MAX ? 0 : ~0ull;
        bool any = false, nan_seen = false;
        if constexpr (FIND) {
            if (staged) {
                // contains / position need no walk at all: the span is compared element-parallel (coalesced loads, one
                // ballot per 64 elements = one word of the match mask), and a row's answer is the first set bit of its range
                const int nw = (span + 63) >> 6;
                for (int t0 = 0; t0 < nw; t0 += 4) {   // four loads in flight per lane
                    T v[4];
                    bool ok[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { const int32_t i = (t0 + u) * 64 + lane; ok[u] = i < span; v[u] = ok[u] ? vals[span0 + i] : (T)0; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint64_t m = __ballot(ok[u] && v[u] == needle);
                        if (lane == u && t0 + u < nw) lds.match[w][t0 + u] = m;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                const int32_t lo = b - span0, hi = e - span0;
                for (int32_t wi = lo >> 6; b < e && !found && wi * 64 < hi; ++wi) {
                    uint64_t m = lds.match[w][wi];
                    if (wi == lo >> 6) m &= ~0ull << (lo & 63);
                    if (hi - wi * 64 < 64) m &= (1ull << (hi - wi * 64)) - 1;
                    if (m) { found = true; pos = wi * 64 + __builtin_ctzll(m) - lo + 1; }
                }
                __builtin_amdgcn_wave_barrier();   // the next iteration rewrites the mask
            } else {
                for (int32_t i = b; i < e && !found; ++i)
                    if (vals[i] == needle) { found = true; pos = i - b + 1; }
            }
        } else {
            // max / min: short spans are copied into LDS with coalesced loads and the lanes walk their slices there; a
            // lane reading its slice straight from HBM touches a different cache line than its neighbours in every
            // iteration (measured 0.20 of peak on rows of 10 f64)
            if (staged) {
                for (int32_t i = lane; i < span; i += 64) lds.stage[w][i] = vals[span0 + i];
                __builtin_amdgcn_wave_barrier();   // same-wave LDS operations execute in order; this pins the compiler
            }
            for (int32_t i = b; i < e; ++i) {
                const T v = staged ? lds.stage[w][i - span0] : vals[i];
                if (list_is_nan(v)) { nan_seen = true; continue; }
                const uint64_t k = ListKey<T>::key(v);
                best = a.op == LIST_MAX ? (k > best ? k : best) : (k < best ? k : best);
                any = true;
            }
            __builtin_amdgcn_wave_barrier();   // the next iteration refills the staging area
        }
        if (a.op == LIST_CONTAINS) {   // NULL list -> NULL, else true / false (array.rs:15-37)
            const uint64_t vb = __ballot(lvalid), bits = __ballot(lvalid && found), ib = __ballot(inr);
            if (lane == 0 && ib) {
                as_global_mut<uint64_t>(a.out.values)[wv] = bits;
                if (a.out.validity) as_global_mut<uint64_t>(a.out.validity)[wv] = vb;
                nulls += __popcll(ib & ~vb);
            }
        } else if (a.op == LIST_POSITION) {   // NULL list or not found -> 0, never NULL (array.rs:233-260)
            if (inr) as_global_mut<int32_t>(a.out.values)[row] = pos;
            const uint64_t ib = __ballot(inr);
            if (lane == 0 && ib && a.out.validity) as_global_mut<uint64_t>(a.out.validity)[wv] = ib;
        } else {   // max / min: NULL for a NULL list; an EMPTY list is NULL too (the reference unwraps None and panics, array.rs:201)
            const bool ok = lvalid && (any || nan_seen);
            T r = (T)0;
            if (any) r = ListKey<T>::unkey(best);
            else if (nan_seen) r = (T)__builtin_nanf("");
            if (inr) as_global_mut<T>(a.out.values)[row] = ok ? r : (T)0;
            const uint64_t vb = __ballot(ok), ib = __ballot(inr);
            if (lane == 0 && ib) {
                if (a.out.validity) as_global_mut<uint64_t>(a.out.validity)[wv] = vb;
                nulls += __popcll(ib & ~vb);
            }
        }
    }
    if (lane == 0 && nulls) atomicAdd((unsigned long long*)a.out_null_count, (unsigned long long)nulls);
}
template <class T> __global__ __launch_bounds__(kBlock) void list_find_kernel(const ListArgs a) { list_rows_impl<T, true>(a); }
template <class T> __global__ __launch_bounds__(kBlock) void list_extreme_kernel(const ListArgs a) { list_rows_impl<T, false>(a); }

// One row per wave.  Bitmap outputs are pre-zeroed by the host and OR-ed in (rows of one word belong to different waves).
template <class T>
__global__ __launch_bounds__(kBlock) void list_wave_kernel(const ListArgs a) {
    const int lane = threadIdx.x & 63;
    const GlobalPtr<int32_t> off = as_global<int32_t>(a.offsets.values) + a.offsets.offset;
    const GlobalPtr<T> vals = as_global<T>(a.values.values) + a.values.offset;
    const T needle = list_needle<T>(a.needle);
    int nulls = 0;
    for (int64_t row = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); row < a.n; row += (int64_t)gridDim.x * (kBlock / 64)) {
        bool lvalid = true;
        if (a.offsets.validity) { const int64_t bi = a.offsets.offset + row; lvalid = (as_global<uint8_t>(a.offsets.validity)[bi >> 3] >> (bi & 7)) & 1; }
        const int32_t b = lvalid ? off[row] : 0, e = lvalid ? off[row + 1] : 0;
        uint32_t first = ~0u;
        uint64_t best = a.op == LIST_MAX ? 0 : ~0ull;
        bool any = false, nan_seen = false;
        for (int32_t i = b + lane; i < e; i += 64) {
            const T v = vals[i];
            if (a.op == LIST_CONTAINS || a.op == LIST_POSITION) {
                if (first == ~0u && v == needle) first = (uint32_t)(i - b + 1);
            } else {
                if (list_is_nan(v)) { nan_seen = true; continue; }
                const uint64_t k = ListKey<T>::key(v);
                best = a.op == LIST_MAX ? (k > best ? k : best) : (k < best ? k : best);
                any = true;
            }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const uint32_t f2 = (uint32_t)__shfl_xor((int)first, m);
            first = f2 < first ? f2 : first;
            const uint64_t b2 = shfl_xor64(best, m);
            best = a.op == LIST_MAX ? (b2 > best ? b2 : best) : (b2 < best ? b2 : best);
        }
        any = __ballot(any) != 0;
        nan_seen = __ballot(nan_seen) != 0;
        if (lane != 0) continue;
        const uint32_t bit = 1u << (row & 31);
        if (a.op == LIST_CONTAINS) {
            if (lvalid && first != ~0u) atomicOr((unsigned int*)a.out.values + (row >> 5), bit);
            if (a.out.validity && lvalid) atomicOr((unsigned int*)a.out.validity + (row >> 5), bit);
            nulls += !lvalid;
        } else if (a.op == LIST_POSITION) {
            as_global_mut<int32_t>(a.out.values)[row] = first == ~0u ? 0 : (int32_t)first;
            if (a.out.validity) atomicOr((unsigned int*)a.out.validity + (row >> 5), bit);
        } else {
            const bool ok = lvalid && (any || nan_seen);
            T r = (T)0;
            if (any) r = ListKey<T>::unkey(best);
            else if (nan_seen) r = (T)__builtin_nanf("");
            as_global_mut<T>(a.out.values)[row] = ok ? r : (T)0;
            if (a.out.validity && ok) atomicOr((unsigned int*)a.out.validity + (row >> 5), bit);
            nulls += !ok;
        }
    }
    if (lane == 0 && nulls) atomicAdd((unsigned long long*)a.out_null_count, (unsigned long long)nulls);
}

// row_ids[e - first] = row of child element e, for e in [first, last): every row writes its own slice (NULL lists have
// no elements to write: their slice is skipped by the consumers through the offsets)
__global__ __launch_bounds__(kBlock) void list_row_ids_kernel(const ListArgs a, uint32_t* row_ids, int32_t first) {
    const GlobalPtr<int32_t> off = as_global<int32_t>(a.offsets.values) + a.offsets.offset;
    const int lane = threadIdx.x & 63;
    for (int64_t row = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); row < a.n; row += (int64_t)gridDim.x * (kBlock / 64)) {
        const int32_t b = off[row], e = off[row + 1];
        for (int32_t i = b + lane; i < e; i += 64) row_ids[i - first] = (uint32_t)row;
    }
}

// array_remove, pass 1: kept[row] = elements of the row that differ from the needle (a NULL list keeps none);
// pass 2 (scan != nullptr): the kept elements are written at the scanned offsets, order preserved.
// Short rows (the 64 rows of a wave span <= kListStage child elements) are handled element-parallel: the span is
// compared with coalesced loads, one ballot per 64 elements gives a word of the DROP mask (matches + the elements of NULL
// rows), a row's count is a popcount over its bit range, and in pass 2 element i lands at
// scan[first row] + i - (dropped elements before i): both passes stream.  Longer spans fall back to a walk per lane.
template <class T>
__global__ __launch_bounds__(kBlock) void list_remove_kernel(const ListArgs a) {
    __shared__ T stage_in[kBlock / 64][kListStage];
    __shared__ unsigned long long drop[kBlock / 64][kListStage / 64];
    __shared__ uint32_t pre[kBlock / 64][kListStage / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const GlobalPtr<int32_t> off = as_global<int32_t>(a.offsets.values) + a.offsets.offset;
    const GlobalPtr<T> vals = as_global<T>(a.values.values) + a.values.offset;
    const T needle = list_needle<T>(a.needle);
    const int64_t nwaves = (a.n + 63) >> 6, stride = (int64_t)gridDim.x * (kBlock / 64);
    int64_t wv = (int64_t)blockIdx.x * (kBlock / 64) + w;
    ListGroup next = list_group_load(a, off, wv < nwaves ? wv : 0, lane);
    for (; wv < nwaves; wv += stride) {
        const ListGroup g = next;
        if (wv + stride < nwaves) next = list_group_load(a, off, wv + stride, lane);   // see list_rows_impl
        const int64_t row = wv * 64 + lane;
        const bool inr = row < a.n;
        const bool lvalid = inr && ((g.lv >> lane) & 1);
        const int32_t rb = inr ? g.rb : 0, re = inr ? g.re : 0;   // the row's slice, NULL or not
        const int32_t b = lvalid ? rb : 0, e = lvalid ? re : 0;
        const int32_t span0 = g.span0, span = g.span1 - g.span0;
        if (span <= kListStage) {
            const int nw = (span + 63) >> 6;
            for (int t0 = 0; t0 < nw; t0 += 4) {   // four loads in flight per lane
                T v[4];
                bool ok[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int32_t i = (t0 + u) * 64 + lane; ok[u] = i < span; v[u] = ok[u] ? vals[span0 + i] : (T)0; }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint64_t m = __ballot(!ok[u] || v[u] == needle);   // positions past the span count as dropped
                    if (lane == u && t0 + u < nw) drop[w][t0 + u] = m;
                    if (a.scan && ok[u]) stage_in[w][(t0 + u) * 64 + lane] = v[u];
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (inr && !lvalid && re > rb) {   // a NULL row that owns child elements: they are dropped
                const int32_t lo = rb - span0, hi = re - span0;
                for (int32_t wi = lo >> 6; wi * 64 < hi; ++wi) {
                    unsigned long long m = ~0ull;
                    if (wi == lo >> 6) m &= ~0ull << (lo & 63);
                    if (hi - wi * 64 < 64) m &= (1ull << (hi - wi * 64)) - 1;
                    atomicOr(&drop[w][wi], m);
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (!a.scan) {
                int64_t c = e - b;
                const int32_t lo = b - span0, hi = e - span0;
                for (int32_t wi = lo >> 6; b < e && wi * 64 < hi; ++wi) {
                    unsigned long long m = drop[w][wi];
                    if (wi == lo >> 6) m &= ~0ull << (lo & 63);
                    if (hi - wi * 64 < 64) m &= (1ull << (hi - wi * 64)) - 1;
                    c -= __popcll(m);
                }
                if (inr) a.kept[row] = c;
            } else {
                if (lane < nw) { uint32_t p = 0; for (int k = 0; k < lane; ++k) p += (uint32_t)__popcll(drop[w][k]); pre[w][lane] = p; }
                __builtin_amdgcn_wave_barrier();
                const int64_t out0 = a.scan[wv * 64];
                for (int t = 0; t < nw; ++t) {
                    const unsigned long long m = drop[w][t];
                    if (!((m >> lane) & 1)) {
                        const int32_t i = t * 64 + lane;
                        as_global_mut<T>(a.out.values)[out0 + i - (int64_t)(pre[w][t] + (uint32_t)__popcll(m & ((1ull << lane) - 1)))] = stage_in[w][i];
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();   // the next iteration refills the staging areas
            continue;
        }
        if (!a.scan) {
            int64_t c = 0;
            for (int32_t i = b; i < e; ++i) c += vals[i] != needle;
            if (inr) a.kept[row] = c;
        } else {
            int64_t o = inr ? a.scan[row] : 0;
            for (int32_t i = b; i < e; ++i) {
                const T v = vals[i];
                if (v != needle) as_global_mut<T>(a.out.values)[o++] = v;
            }
        }
    }
}

// value_offsets of the result: int32 from the int64 exclusive scan
__global__ void list_offsets_kernel(const int64_t* scan, int64_t n1, int32_t* out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n1; i += (int64_t)gridDim.x * blockDim.x) out[i] = (int32_t)scan[i];
}

#define RDF_LIST_DISPATCH(KERNEL)                                                                                           \
    switch (a.dtype) {                                                                                                      \
        case RDF_I8: hipLaunchKernelGGL((KERNEL<int8_t>), dim3(grid), dim3(kBlock), 0, s, a); break;                        \
        case RDF_I16: hipLaunchKernelGGL((KERNEL<int16_t>), dim3(grid), dim3(kBlock), 0, s, a); break;                      \
        case RDF_I32: hipLaunchKernelGGL((KERNEL<int32_t>), dim3(grid), dim3(kBlock), 0, s, a); break;                      \
        case RDF_I64: hipLaunchKernelGGL((KERNEL<int64_t>), dim3(grid), dim3(kBlock), 0, s, a); break;                      \
        case RDF_U8: hipLaunchKernelGGL((KERNEL<uint8_t>), dim3(grid), dim3(kBlock), 0, s, a); break;                       \
        case RDF_U16: hipLaunchKernelGGL((KERNEL<uint16_t>), dim3(grid), dim3(kBlock), 0, s, a); break;                     \
        case RDF_U32: hipLaunchKernelGGL((KERNEL<uint32_t>), dim3(grid), dim3(kBlock), 0, s, a); break;                     \
        case RDF_U64: hipLaunchKernelGGL((KERNEL<uint64_t>), dim3(grid), dim3(kBlock), 0, s, a); break;                     \
        case RDF_F32: hipLaunchKernelGGL((KERNEL<float>), dim3(grid), dim3(kBlock), 0, s, a); break;                        \
        default: hipLaunchKernelGGL((KERNEL<double>), dim3(grid), dim3(kBlock), 0, s, a); break;                            \
    }

hipError_t launch_list_op(const ListArgs& a, bool wave_per_row, hipStream_t s) {
    if (a.n <= 0) return hipSuccess;
    const int64_t units = wave_per_row ? a.n : (a.n + 63) / 64;   // waves of work
    int64_t grid64 = (units + (kBlock / 64) - 1) / (kBlock / 64);
    if (grid64 > eval_grid_limit()) grid64 = eval_grid_limit();
    const int grid = (int)(grid64 < 1 ? 1 : grid64);
    if (wave_per_row) { RDF_LIST_DISPATCH(list_wave_kernel) }
    else if (a.op == LIST_CONTAINS || a.op == LIST_POSITION) { RDF_LIST_DISPATCH(list_find_kernel) }
    else { RDF_LIST_DISPATCH(list_extreme_kernel) }
    return hipGetLastError();
}
hipError_t launch_list_row_ids(const ListArgs& a, uint32_t* row_ids, int32_t first, hipStream_t s) {
    if (a.n <= 0) return hipSuccess;
    int64_t grid = (a.n + (kBlock / 64) - 1) / (kBlock / 64);
    if (grid > eval_grid_limit()) grid = eval_grid_limit();
    hipLaunchKernelGGL(list_row_ids_kernel, dim3((unsigned)grid), dim3(kBlock), 0, s, a, row_ids, first);
    return hipGetLastError();
}
// array_remove with one row per WAVE (long rows): 64 elements at a time, the kept ones ranked by a ballot prefix
template <class T>
__global__ __launch_bounds__(kBlock) void list_remove_wave_kernel(const ListArgs a) {
    const int lane = threadIdx.x & 63;
    const GlobalPtr<int32_t> off = as_global<int32_t>(a.offsets.values) + a.offsets.offset;
    const GlobalPtr<T> vals = as_global<T>(a.values.values) + a.values.offset;
    const T needle = list_needle<T>(a.needle);
    for (int64_t row = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); row < a.n; row += (int64_t)gridDim.x * (kBlock / 64)) {
        bool lvalid = true;
        if (a.offsets.validity) { const int64_t bi = a.offsets.offset + row; lvalid = (as_global<uint8_t>(a.offsets.validity)[bi >> 3] >> (bi & 7)) & 1; }
        const int32_t b = lvalid ? off[row] : 0, e = lvalid ? off[row + 1] : 0;
        int64_t o = a.scan ? a.scan[row] : 0, c = 0;
        for (int32_t i0 = b; i0 < e; i0 += 64) {
            const int32_t i = i0 + lane;
            T v = (T)0;
            bool keep = false;
            if (i < e) { v = vals[i]; keep = v != needle; }
            const uint64_t m = __ballot(keep);
            if (a.scan) {
                if (keep) as_global_mut<T>(a.out.values)[o + __popcll(m & ((1ull << lane) - 1))] = v;
                o += __popcll(m);
            } else c += __popcll(m);
        }
        if (!a.scan && lane == 0) a.kept[row] = c;
    }
}
hipError_t launch_list_remove(const ListArgs& a, bool wave_per_row, hipStream_t s) {
    if (a.n <= 0) return hipSuccess;
    int64_t grid64 = wave_per_row ? (a.n + (kBlock / 64) - 1) / (kBlock / 64) : ((a.n + 63) / 64 + (kBlock / 64) - 1) / (kBlock / 64);
    if (grid64 > eval_grid_limit()) grid64 = eval_grid_limit();
    const int grid = (int)grid64;
    if (wave_per_row) { RDF_LIST_DISPATCH(list_remove_wave_kernel) } else { RDF_LIST_DISPATCH(list_remove_kernel) }
    return hipGetLastError();
}
// array_distinct / array_except / array_intersect / array_union / array_repeat (array.rs:39-153,294-399; the per-row
// set algebra is the array_tool crate's: `unique` keeps first occurrences in order, `uniq(b)` = unique(a) without the
// members of b, `intersect(b)` = unique(a) restricted to the members of b, `union(b)` = unique(a ++ b), `times(c)` =
// the slice c times over).  Equality is the element type's `==` (NaN equals nothing, -0.0 == 0.0).  Every output
// element is an input element that passes a membership test against the elements before it, so the kernels are the
// array_remove ones with a different keep rule: pass 1 counts per row, pass 2 writes at the scanned offsets.
//
//   list_set_kernel      : one row per lane over runs of consecutive rows whose slices fit the LDS staging area
//                          (<= 768 elements per input): the crate's quadratic compare loops, from LDS.  A row that
//                          does not fit on its own goes to a worklist instead.
//   list_set_wave_kernel : one row per wave, any length, linear work: an open-addressing table per row (2 slots per
//                          element, in device scratch) maps a value to the SMALLEST element index holding it
//                          (CAS to claim a slot, atomicMin on the index); "first occurrence" and "member of b" are then
//                          table lookups.  Pass 1 builds the tables and counts, pass 2 only looks up.
constexpr int kSetBlockMax = 1024;   // elements per input of a row the block-per-row kernel takes (tables in LDS)
constexpr int kSetStage = 768;   // child elements of each input (and of the output) staged per wave: 72 KiB per block for 8-byte children
constexpr uint32_t kSetEmpty = 0xFFFFFFFFu;

template <class T> __device__ __forceinline__ uint32_t set_hash(T v) {
    uint64_t bits;
    if constexpr (sizeof(T) == 8 && !std::is_integral<T>::value) bits = d2u((double)v + 0.0);          // -0.0 hashes like 0.0
    else if constexpr (!std::is_integral<T>::value) bits = __float_as_uint((float)v + 0.0f);
    else bits = (uint64_t)(int64_t)v;
    bits ^= bits >> 32;
    bits *= 0x9E3779B97F4A7C15ull;
    return (uint32_t)(bits >> 32);
}
__device__ __forceinline__ uint32_t set_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// One row's table: `size` slots (>= 2 * elements) holding row-relative element indices
template <class T> struct SetTable {
    uint32_t* tab;
    uint32_t size;
    GlobalPtr<T> vals;   // the row's first element
    __device__ __forceinline__ uint32_t home(T v) const { return (uint32_t)(((uint64_t)set_hash<T>(v) * size) >> 32); }
    __device__ void insert(uint32_t i, T v) const {
        uint32_t s = home(v);
        for (;;) {
            uint32_t cur = set_load(tab + s);
            if (cur == kSetEmpty) {
                cur = atomicCAS(tab + s, kSetEmpty, i);
                if (cur == kSetEmpty) return;
            }
            if (vals[cur] == v) { atomicMin(tab + s, i); return; }
            s = s + 1 == size ? 0 : s + 1;
        }
    }
    // the smallest index holding v, kSetEmpty when v is not a member
    __device__ uint32_t find(T v) const {
        if (size == 0) return kSetEmpty;
        uint32_t s = home(v);
        for (;;) {
            const uint32_t cur = set_load(tab + s);
            if (cur == kSetEmpty || vals[cur] == v) return cur;
            s = s + 1 == size ? 0 : s + 1;
        }
    }
};

template <class T>
__global__ __launch_bounds__(kBlock) void list_set_kernel(const ListArgs a) {
    __shared__ T sa[kBlock / 64][kSetStage];
    __shared__ T sb[kBlock / 64][kSetStage];
    __shared__ T so[kBlock / 64][kSetStage];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const GlobalPtr<int32_t> offa = as_global<int32_t>(a.offsets.values) + a.offsets.offset;
    const GlobalPtr<T> va = as_global<T>(a.values.values) + a.values.offset;
    const bool two = a.op == LIST_EXCEPT || a.op == LIST_INTERSECT || a.op == LIST_UNION;
    const GlobalPtr<int32_t> offb = two ? as_global<int32_t>(a.offsets_b.values) + a.offsets_b.offset : offa;
    const GlobalPtr<T> vb = two ? as_global<T>(a.values_b.values) + a.values_b.offset : va;
    const int64_t nwaves = (a.n + 63) >> 6;
    for (int64_t wv = (int64_t)blockIdx.x * (kBlock / 64) + w; wv < nwaves; wv += (int64_t)gridDim.x * (kBlock / 64)) {
        const int64_t row = wv * 64 + lane;
        const bool inr = row < a.n;
        bool lvalid = inr;
        if (inr && a.offsets.validity) { const int64_t bi = a.offsets.offset + row; lvalid = (as_global<uint8_t>(a.offsets.validity)[bi >> 3] >> (bi & 7)) & 1; }
        // value_offsets are monotone, also across NULL rows: consecutive rows cover one contiguous span of the child arrays
        const int32_t la_lo = offa[inr ? row : a.n], la_hi = offa[inr ? row + 1 : a.n];
        const int32_t lb_lo = two ? offb[inr ? row : a.n] : 0, lb_hi = two ? offb[inr ? row + 1 : a.n] : 0;
        const int nrows = a.n - wv * 64 < 64 ? (int)(a.n - wv * 64) : 64;
        if (!a.scan && inr) a.kept[row] = 0;
        // the wave's rows are taken in runs of consecutive rows whose spans fit the staging areas (usually one run)
        for (int start = 0; start < nrows;) {
            const int32_t a0 = __shfl(la_lo, start), b0 = __shfl(lb_lo, start);
            const bool fits = lane >= start && inr && la_hi - a0 <= kSetStage && lb_hi - b0 <= kSetStage;
            const int cnt = __popcll(__ballot(fits));   // monotone in the lane: the run is [start, start + cnt)
            if (cnt == 0) {   // this row alone is too long: it goes to the row-per-wave kernel
                if (!a.scan && lane == start && lvalid && (la_hi > la_lo || lb_hi > lb_lo)) a.work[atomicAdd(a.work_count, 1u)] = (uint32_t)row;
                ++start;
                continue;
            }
            const int last = start + cnt - 1;
            const int32_t an = __shfl(la_hi, last) - a0, bn = __shfl(lb_hi, last) - b0;
            const bool mine = fits && lvalid;
            const int32_t ba = mine ? la_lo : 0, ea = mine ? la_hi : 0, bb = mine ? lb_lo : 0, eb = mine ? lb_hi : 0;
            for (int32_t i = lane; i < an; i += 64) sa[w][i] = va[a0 + i];
            for (int32_t i = lane; i < bn; i += 64) sb[w][i] = vb[b0 + i];
            __builtin_amdgcn_wave_barrier();
            auto in_a = [&](T v, int32_t upto) { for (int32_t k = ba; k < upto; ++k) if (sa[w][k - a0] == v) return true; return false; };
            auto in_b = [&](T v, int32_t upto) { for (int32_t k = bb; k < upto; ++k) if (sb[w][k - b0] == v) return true; return false; };
            int64_t out0 = 0, out_n = 0, o = 0;
            if (a.scan) { out0 = a.scan[wv * 64 + start]; out_n = a.scan[wv * 64 + last + 1] - out0; o = fits ? a.scan[row] : 0; }
            const bool ostaged = a.scan && out_n <= kSetStage;
            int64_t c = 0;
            auto emit = [&](T v) {
                if (a.scan) {
                    if (ostaged) so[w][o - out0] = v; else as_global_mut<T>(a.out.values)[o] = v;
                    ++o;
                } else ++c;
            };
            if (a.op == LIST_REPEAT) {
                if (!a.scan) c = (int64_t)(ea - ba) * a.count;
                else for (int32_t r = 0; r < a.count; ++r) for (int32_t i = ba; i < ea; ++i) emit(sa[w][i - a0]);
            } else {
                for (int32_t j = ba; j < ea; ++j) {
                    const T v = sa[w][j - a0];
                    bool keep = !in_a(v, j);
                    if (keep && a.op == LIST_EXCEPT) keep = !in_b(v, eb);
                    if (keep && a.op == LIST_INTERSECT) keep = in_b(v, eb);
                    if (keep) emit(v);
                }
                if (a.op == LIST_UNION)
                    for (int32_t j = bb; j < eb; ++j) {
                        const T v = sb[w][j - b0];
                        if (!in_a(v, ea) && !in_b(v, j)) emit(v);
                    }
            }
            if (!a.scan) { if (fits) a.kept[row] = c; }
            else if (ostaged) {
                __builtin_amdgcn_wave_barrier();
                for (int64_t i = lane; i < out_n; i += 64) as_global_mut<T>(a.out.values)[out0 + i] = so[w][i];
            }
            __builtin_amdgcn_wave_barrier();   // the next run refills the staging areas
            start += cnt;
        }
    }
}

template <class T>
__global__ __launch_bounds__(kBlock) void list_set_wave_kernel(const ListArgs a) {
    const int lane = threadIdx.x & 63;
    const GlobalPtr<int32_t> offa = as_global<int32_t>(a.offsets.values) + a.offsets.offset;
    const GlobalPtr<T> va = as_global<T>(a.values.values) + a.values.offset;
    const bool two = a.op == LIST_EXCEPT || a.op == LIST_INTERSECT || a.op == LIST_UNION;
    const GlobalPtr<int32_t> offb = two ? as_global<int32_t>(a.offsets_b.values) + a.offsets_b.offset : offa;
    const GlobalPtr<T> vb = two ? as_global<T>(a.values_b.values) + a.values_b.offset : va;
    const int64_t nwork = a.work ? (int64_t)*a.work_count : a.n;   // the worklist of list_set_kernel, or every row
    for (int64_t wi = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); wi < nwork; wi += (int64_t)gridDim.x * (kBlock / 64)) {
        const int64_t row = a.work ? (int64_t)a.work[wi] : wi;
        bool lvalid = true;
        if (a.offsets.validity) { const int64_t bi = a.offsets.offset + row; lvalid = (as_global<uint8_t>(a.offsets.validity)[bi >> 3] >> (bi & 7)) & 1; }
        const int32_t ba = lvalid ? offa[row] : 0, ea = lvalid ? offa[row + 1] : 0;
        const int32_t bb = lvalid && two ? offb[row] : 0, eb = lvalid && two ? offb[row + 1] : 0;
        int64_t o = a.scan ? a.scan[row] : 0, c = 0;
        if (a.op != LIST_REPEAT && ea - ba <= kSetBlockMax && eb - bb <= kSetBlockMax) continue;   // list_set_block_kernel took this row
        if (a.op == LIST_REPEAT) {
            const int64_t len = ea - ba, total = len * a.count;
            if (a.scan) for (int64_t t = lane; t < total; t += 64) as_global_mut<T>(a.out.values)[o + t] = va[ba + t % len];
            else if (lane == 0) a.kept[row] = total;
            continue;
        }
        const SetTable<T> ta{a.tab_a + 2 * (int64_t)ba, (uint32_t)(2 * (ea - ba)), va + ba};
        const SetTable<T> tb{a.tab_b + 2 * (int64_t)bb, (uint32_t)(2 * (eb - bb)), vb + bb};
        if (!a.scan) {   // pass 1 builds the tables (pass 2 finds them as they were left)
            for (uint32_t i = lane; i < ta.size; i += 64) ta.tab[i] = kSetEmpty;
            for (uint32_t i = lane; i < tb.size; i += 64) tb.tab[i] = kSetEmpty;
            __threadfence();
            __builtin_amdgcn_wave_barrier();
            for (int32_t i = ba + lane; i < ea; i += 64) { const T v = va[i]; if (!list_is_nan(v)) ta.insert((uint32_t)(i - ba), v); }
            for (int32_t i = bb + lane; i < eb; i += 64) { const T v = vb[i]; if (!list_is_nan(v)) tb.insert((uint32_t)(i - bb), v); }
            __threadfence();
            __builtin_amdgcn_wave_barrier();
        }
        auto flush = [&](bool keep, T v) {
            const uint64_t m = __ballot(keep);
            if (a.scan) {
                if (keep) as_global_mut<T>(a.out.values)[o + __popcll(m & ((1ull << lane) - 1))] = v;
                o += __popcll(m);
            } else c += __popcll(m);
        };
        for (int32_t i0 = ba; i0 < ea; i0 += 64) {   // candidates from a: first occurrence in a (+ the b membership rule)
            const int32_t i = i0 + lane;
            const bool have = i < ea;
            const T v = have ? va[i] : (T)0;
            bool keep = false;
            if (have) {
                const bool nan = list_is_nan(v);
                keep = nan || ta.find(v) == (uint32_t)(i - ba);
                if (a.op == LIST_EXCEPT) keep = keep && (nan || tb.find(v) == kSetEmpty);
                if (a.op == LIST_INTERSECT) keep = keep && !nan && tb.find(v) != kSetEmpty;
            }
            flush(keep, v);
        }
        if (a.op == LIST_UNION)
            for (int32_t i0 = bb; i0 < eb; i0 += 64) {   // candidates from b: not in a, first occurrence in b
                const int32_t i = i0 + lane;
                const bool have = i < eb;
                const T v = have ? vb[i] : (T)0;
                bool keep = false;
                if (have) keep = list_is_nan(v) || (tb.find(v) == (uint32_t)(i - bb) && ta.find(v) == kSetEmpty);
                flush(keep, v);
            }
        if (!a.scan && lane == 0) a.kept[row] = c;
    }
}
//   list_set_block_kernel: one row per 256-thread BLOCK for rows of up to kSetBlockMax elements per input — the row's values and
//                          its value -> smallest-index tables live in LDS (CAS / min there, no global atomics, no scratch
//                          traffic): rows of 1000 elements ran at 0.017 of the HBM roofline through the global-scratch tables
//                          of the row-per-wave kernel, which keeps the rows longer than this.

template <class T>
struct SetTableLds {
    uint32_t* tab;         // [2 * kSetBlockMax] LDS
    const T* vals;         // LDS copy of the row
    uint32_t size;         // power of two >= 2 * elements
    __device__ __forceinline__ uint32_t home(T v) const { return set_hash<T>(v) & (size - 1); }
    __device__ __forceinline__ void insert(uint32_t i, T v) const {
        uint32_t s = home(v);
        for (;;) {
            uint32_t cur = tab[s];
            if (cur == kSetEmpty) {
                cur = atomicCAS(tab + s, kSetEmpty, i);
                if (cur == kSetEmpty) return;
            }
            if (vals[cur] == v) { atomicMin(tab + s, i); return; }
            s = (s + 1) & (size - 1);
        }
    }
    __device__ __forceinline__ uint32_t find(T v) const {
        if (size == 0) return kSetEmpty;
        uint32_t s = home(v);
        for (;;) {
            const uint32_t cur = tab[s];
            if (cur == kSetEmpty || vals[cur] == v) return cur;
            s = (s + 1) & (size - 1);
        }
    }
};
template <class T>
__global__ __launch_bounds__(kBlock) void list_set_block_kernel(const ListArgs a) {
    // dynamic LDS: the row's values + its table, twice for the two-input functions (one-input calls get twice the blocks per CU)
    extern __shared__ __attribute__((aligned(16))) unsigned char lsm[];
    T* const sa = (T*)lsm;
    uint32_t* const ta_ = (uint32_t*)(sa + kSetBlockMax);
    T* const sb = (T*)(ta_ + 2 * kSetBlockMax);
    uint32_t* const tb_ = (uint32_t*)(sb + kSetBlockMax);
    __shared__ uint32_t wtot[kBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const GlobalPtr<int32_t> offa = as_global<int32_t>(a.offsets.values) + a.offsets.offset;
    const GlobalPtr<T> va = as_global<T>(a.values.values) + a.values.offset;
    const bool two = a.op == LIST_EXCEPT || a.op == LIST_INTERSECT || a.op == LIST_UNION;
    const GlobalPtr<int32_t> offb = two ? as_global<int32_t>(a.offsets_b.values) + a.offsets_b.offset : offa;
    const GlobalPtr<T> vb = two ? as_global<T>(a.values_b.values) + a.values_b.offset : va;
    const int64_t nwork = a.work ? (int64_t)*a.work_count : a.n;
    for (int64_t wi = blockIdx.x; wi < nwork; wi += gridDim.x) {
        const int64_t row = a.work ? (int64_t)a.work[wi] : wi;
        bool lvalid = true;
        if (a.offsets.validity) { const int64_t bi = a.offsets.offset + row; lvalid = (as_global<uint8_t>(a.offsets.validity)[bi >> 3] >> (bi & 7)) & 1; }
        const int32_t ba = lvalid ? offa[row] : 0, ea = lvalid ? offa[row + 1] : 0;
        const int32_t bb = lvalid && two ? offb[row] : 0, eb = lvalid && two ? offb[row + 1] : 0;
        const int32_t na = ea - ba, nb = eb - bb;
        if (na > kSetBlockMax || nb > kSetBlockMax) continue;           // the row-per-wave kernel's
        uint32_t sza = 2, szb = 2;
        while ((int32_t)sza < 2 * na) sza <<= 1;
        while ((int32_t)szb < 2 * nb) szb <<= 1;
        __syncthreads();                                                  // the previous row's tables are done with
        for (int32_t i = tid; i < na; i += kBlock) sa[i] = va[ba + i];
        for (int32_t i = tid; i < nb; i += kBlock) sb[i] = vb[bb + i];
        for (uint32_t i = tid; i < sza; i += kBlock) ta_[i] = kSetEmpty;
        if (two) for (uint32_t i = tid; i < szb; i += kBlock) tb_[i] = kSetEmpty;       // (one-input calls have no second table in their LDS)
        __syncthreads();
        const SetTableLds<T> ta{ta_, sa, na ? sza : 0u}, tb{tb_, sb, nb ? szb : 0u};
        for (int32_t i = tid; i < na; i += kBlock) { const T v = sa[i]; if (!list_is_nan(v)) ta.insert((uint32_t)i, v); }
        for (int32_t i = tid; i < nb; i += kBlock) { const T v = sb[i]; if (!list_is_nan(v)) tb.insert((uint32_t)i, v); }
        __syncthreads();
        // Pass 1 (this code): the keep bit of every candidate, as one ballot word per (chunk of kBlock candidates, wave), left in the
        // row's own slice of the scratch tables (8 bytes per element are reserved there, a row needs 1 bit per element) — pass 2
        // ranks and writes from those words alone: no table is built twice.
        unsigned long long* const mask_a = (unsigned long long*)(a.tab_a + 2 * (int64_t)ba);     // [ceil(na / 64)] words (<= 16)
        unsigned long long* const mask_b = (unsigned long long*)(a.tab_b + 2 * (int64_t)bb);
        uint32_t c = 0;
        for (int32_t i0 = 0; i0 < na; i0 += kBlock) {   // candidates from a: first occurrence in a (+ the b membership rule)
            const int32_t i = i0 + tid;
            const bool have = i < na;
            const T v = have ? sa[i] : (T)0;
            bool keep = false;
            if (have) {
                const bool nan = list_is_nan(v);
                keep = nan || ta.find(v) == (uint32_t)i;
                if (a.op == LIST_EXCEPT) keep = keep && (nan || tb.find(v) == kSetEmpty);
                if (a.op == LIST_INTERSECT) keep = keep && !nan && tb.find(v) != kSetEmpty;
            }
            const uint64_t m = __ballot(keep);
            if (lane == 0 && i0 + w * 64 < na) mask_a[(i0 >> 6) + w] = m;
            c += (uint32_t)__popcll(m);
        }
        if (a.op == LIST_UNION)
            for (int32_t i0 = 0; i0 < nb; i0 += kBlock) {   // candidates from b: not in a, first occurrence in b
                const int32_t i = i0 + tid;
                const bool have = i < nb;
                const T v = have ? sb[i] : (T)0;
                bool keep = false;
                if (have) keep = list_is_nan(v) || (tb.find(v) == (uint32_t)i && ta.find(v) == kSetEmpty);
                const uint64_t m = __ballot(keep);
                if (lane == 0 && i0 + w * 64 < nb) mask_b[(i0 >> 6) + w] = m;
                c += (uint32_t)__popcll(m);
            }
        if (lane == 0) wtot[w] = c;
        __syncthreads();
        if (tid == 0) { uint32_t t = 0; for (int k = 0; k < kBlock / 64; ++k) t += wtot[k]; a.kept[row] = t; }
    }
}
// pass 2 of the same rows: a wave per row, the keep words of pass 1 -> ranks -> the kept elements at the row's scanned offset
template <class T>
__global__ __launch_bounds__(kBlock) void list_set_block_write_kernel(const ListArgs a) {
    const int lane = threadIdx.x & 63;
    const GlobalPtr<int32_t> offa = as_global<int32_t>(a.offsets.values) + a.offsets.offset;
    const GlobalPtr<T> va = as_global<T>(a.values.values) + a.values.offset;
    const bool two = a.op == LIST_EXCEPT || a.op == LIST_INTERSECT || a.op == LIST_UNION;
    const GlobalPtr<int32_t> offb = two ? as_global<int32_t>(a.offsets_b.values) + a.offsets_b.offset : offa;
    const GlobalPtr<T> vb = two ? as_global<T>(a.values_b.values) + a.values_b.offset : va;
    const int64_t nwork = a.work ? (int64_t)*a.work_count : a.n;
    for (int64_t wi = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); wi < nwork; wi += (int64_t)gridDim.x * (kBlock / 64)) {
        const int64_t row = a.work ? (int64_t)a.work[wi] : wi;
        bool lvalid = true;
        if (a.offsets.validity) { const int64_t bi = a.offsets.offset + row; lvalid = (as_global<uint8_t>(a.offsets.validity)[bi >> 3] >> (bi & 7)) & 1; }
        const int32_t ba = lvalid ? offa[row] : 0, ea = lvalid ? offa[row + 1] : 0;
        const int32_t bb = lvalid && two ? offb[row] : 0, eb = lvalid && two ? offb[row + 1] : 0;
        const int32_t na = ea - ba, nb = eb - bb;
        if (na > kSetBlockMax || nb > kSetBlockMax) continue;
        int64_t o = a.scan[row];
        const unsigned long long* const mask_a = (const unsigned long long*)(a.tab_a + 2 * (int64_t)ba);
        const unsigned long long* const mask_b = (const unsigned long long*)(a.tab_b + 2 * (int64_t)bb);
        for (int32_t i0 = 0; i0 < na; i0 += 64) {
            const uint64_t m = mask_a[i0 >> 6];
            if ((m >> lane) & 1) as_global_mut<T>(a.out.values)[o + __popcll(m & ((1ull << lane) - 1))] = va[ba + i0 + lane];
            o += __popcll(m);
        }
        if (a.op == LIST_UNION)
            for (int32_t i0 = 0; i0 < nb; i0 += 64) {
                const uint64_t m = mask_b[i0 >> 6];
                if ((m >> lane) & 1) as_global_mut<T>(a.out.values)[o + __popcll(m & ((1ull << lane) - 1))] = vb[bb + i0 + lane];
                o += __popcll(m);
            }
    }
}
// wave_per_row: every row through the table kernel; otherwise the row-per-lane kernel first and the table kernel over
// the worklist it left behind (nothing to do when no group of 64 rows overflowed the staging area)
hipError_t launch_list_set(const ListArgs& a, bool wave_per_row, hipStream_t s) {
    if (a.n <= 0) return hipSuccess;
    if (!wave_per_row) {
        int64_t grid64 = ((a.n + 63) / 64 + (kBlock / 64) - 1) / (kBlock / 64);
        if (grid64 > eval_grid_limit()) grid64 = eval_grid_limit();
        const int grid = (int)grid64;
        RDF_LIST_DISPATCH(list_set_kernel)
    }
    if (a.op != LIST_REPEAT) {   // rows of up to kSetBlockMax elements per input (every row, or what the row-per-lane kernel left): tables in LDS
        if (!a.scan) {
            const bool two_in = a.op == LIST_EXCEPT || a.op == LIST_INTERSECT || a.op == LIST_UNION;
            const int es = a.dtype == RDF_I8 || a.dtype == RDF_U8 ? 1 : a.dtype == RDF_I16 || a.dtype == RDF_U16 ? 2 : a.dtype == RDF_I32 || a.dtype == RDF_U32 || a.dtype == RDF_F32 ? 4 : 8;
            const size_t lds = (size_t)(two_in ? 2 : 1) * (size_t)kSetBlockMax * (size_t)(es + 8);   // <= 32 KB
            const int per_cu = (int)std::min<size_t>(8, (size_t)(150 * 1024) / lds);
            int64_t grid64 = wave_per_row ? a.n : eval_grid_limit();
            if (grid64 > per_cu * (int64_t)(eval_grid_limit() / 8)) grid64 = per_cu * (int64_t)(eval_grid_limit() / 8);
            const int grid = (int)(grid64 < 1 ? 1 : grid64);
#define RDF_LIST_DISPATCH_LDS(KERNEL)                                                                                       \
    switch (a.dtype) {                                                                                                      \
        case RDF_I8: hipLaunchKernelGGL((KERNEL<int8_t>), dim3(grid), dim3(kBlock), lds, s, a); break;                      \
        case RDF_I16: hipLaunchKernelGGL((KERNEL<int16_t>), dim3(grid), dim3(kBlock), lds, s, a); break;                    \
        case RDF_I32: hipLaunchKernelGGL((KERNEL<int32_t>), dim3(grid), dim3(kBlock), lds, s, a); break;                    \
        case RDF_I64: hipLaunchKernelGGL((KERNEL<int64_t>), dim3(grid), dim3(kBlock), lds, s, a); break;                    \
        case RDF_U8: hipLaunchKernelGGL((KERNEL<uint8_t>), dim3(grid), dim3(kBlock), lds, s, a); break;                     \
        case RDF_U16: hipLaunchKernelGGL((KERNEL<uint16_t>), dim3(grid), dim3(kBlock), lds, s, a); break;                   \
        case RDF_U32: hipLaunchKernelGGL((KERNEL<uint32_t>), dim3(grid), dim3(kBlock), lds, s, a); break;                   \
        case RDF_U64: hipLaunchKernelGGL((KERNEL<uint64_t>), dim3(grid), dim3(kBlock), lds, s, a); break;                   \
        case RDF_F32: hipLaunchKernelGGL((KERNEL<float>), dim3(grid), dim3(kBlock), lds, s, a); break;                      \
        default: hipLaunchKernelGGL((KERNEL<double>), dim3(grid), dim3(kBlock), lds, s, a); break;                          \
    }
            RDF_LIST_DISPATCH_LDS(list_set_block_kernel)
#undef RDF_LIST_DISPATCH_LDS
        } else {
            int64_t grid64 = wave_per_row ? (a.n + (kBlock / 64) - 1) / (kBlock / 64) : eval_grid_limit();
            if (grid64 > eval_grid_limit()) grid64 = eval_grid_limit();
            const int grid = (int)(grid64 < 1 ? 1 : grid64);
            RDF_LIST_DISPATCH(list_set_block_write_kernel)
        }
    }
    {
        int64_t grid64 = wave_per_row ? (a.n + (kBlock / 64) - 1) / (kBlock / 64) : eval_grid_limit();
        if (grid64 > eval_grid_limit()) grid64 = eval_grid_limit();
        const int grid = (int)grid64;
        RDF_LIST_DISPATCH(list_set_wave_kernel)
    }
    return hipGetLastError();
}
// ------------------------------------------------------------------------------------------------
// array_sort (array.rs:328-354: every row's slice sorted ascending, value_offsets unchanged).  Rows are sorted where they
// lie — keys in LDS — instead of through a (row id, value) two-column radix sort of the whole child array plus a take
// (rows of 10 elements: 0.011 of the HBM roofline that way):
//   list_sort_lane_kernel  : a row per lane over runs of consecutive short rows (<= kSortLaneMax elements each) whose span fits
//                            the wave's staging area — loaded and written back coalesced, insertion-sorted in LDS by their lanes;
//                            anything longer goes to a worklist
//   list_sort_block_kernel : a row per 256-thread block, up to kSortBlockMax elements: bitonic network over the row's
//                            order-preserving keys (ListKey: IEEE total order for floats), padded with the largest key
// A row longer than that raises a flag and the host sends the whole call through the radix sort as before.
constexpr int kSortStage = 1024, kSortLaneMax = 40, kSortBlockMax = 4096;
template <class T> struct SortKeyOf { typedef uint32_t type; };
template <> struct SortKeyOf<double> { typedef uint64_t type; };
template <> struct SortKeyOf<int64_t> { typedef uint64_t type; };
template <> struct SortKeyOf<uint64_t> { typedef uint64_t type; };

template <class T>
__global__ __launch_bounds__(kBlock) void list_sort_lane_kernel(const ListArgs a) {
    using K = typename SortKeyOf<T>::type;
    __shared__ K st[kBlock / 64][kSortStage];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const GlobalPtr<int32_t> off = as_global<int32_t>(a.offsets.values) + a.offsets.offset;
    const int32_t first = a.count;                                   // child index of the slice's first element
    const GlobalPtr<T> va = as_global<T>(a.values.values) + a.values.offset - first;
    const GlobalMutPtr<T> out = as_global_mut<T>(a.out.values) - first;
    const int64_t nwaves = (a.n + 63) >> 6;
    for (int64_t wv = (int64_t)blockIdx.x * (kBlock / 64) + w; wv < nwaves; wv += (int64_t)gridDim.x * (kBlock / 64)) {
        const int64_t row = wv * 64 + lane;
        const bool inr = row < a.n;
        const int32_t lo = off[inr ? row : a.n], hi = off[inr ? row + 1 : a.n];
        const int nrows = a.n - wv * 64 < 64 ? (int)(a.n - wv * 64) : 64;
        for (int start = 0; start < nrows;) {
            const int32_t a0 = __shfl(lo, start);
            const bool fits = lane >= start && inr && hi - a0 <= kSortStage && hi - lo <= kSortLaneMax;
            const uint64_t m = __ballot(fits) >> start;
            const int cnt = m == ~0ull ? 64 - start : __builtin_ctzll(~m);       // the run [start, start + cnt)
            if (cnt == 0) {   // this row is too long for a lane: the block kernel's
                if (lane == start && hi - lo > 1) a.work[atomicAdd(a.work_count, 1u)] = (uint32_t)row;
                else if (lane == start && hi - lo == 1) out[lo] = va[lo];
                ++start;
                continue;
            }
            const int last = start + cnt - 1;
            const int32_t an = __shfl(hi, last) - a0;
            for (int32_t i = lane; i < an; i += 64) st[w][i] = (K)ListKey<T>::key(va[a0 + i]);
            __builtin_amdgcn_wave_barrier();
            if (lane >= start && lane <= last) {
                K* r = &st[w][lo - a0];
                const int len = hi - lo;
                for (int i = 1; i < len; ++i) {
                    const K x = r[i];
                    int j = i - 1;
                    while (j >= 0 && r[j] > x) { r[j + 1] = r[j]; --j; }
                    r[j + 1] = x;
                }
            }
            __builtin_amdgcn_wave_barrier();
            for (int32_t i = lane; i < an; i += 64) out[a0 + i] = ListKey<T>::unkey((uint64_t)st[w][i]);
            __builtin_amdgcn_wave_barrier();
            start += cnt;
        }
    }
}

template <class T>
__global__ __launch_bounds__(kBlock) void list_sort_block_kernel(const ListArgs a) {
    using K = typename SortKeyOf<T>::type;
    __shared__ K sk[kSortBlockMax];
    const int tid = threadIdx.x;
    const GlobalPtr<int32_t> off = as_global<int32_t>(a.offsets.values) + a.offsets.offset;
    const int32_t first = a.count;
    const GlobalPtr<T> va = as_global<T>(a.values.values) + a.values.offset - first;
    const GlobalMutPtr<T> out = as_global_mut<T>(a.out.values) - first;
    const int64_t nwork = a.work ? (int64_t)*a.work_count : a.n;
    for (int64_t wi = blockIdx.x; wi < nwork; wi += gridDim.x) {
        const int64_t row = a.work ? (int64_t)a.work[wi] : wi;
        const int32_t lo = off[row], hi = off[row + 1], len = hi - lo;
        if (len <= 0) continue;
        if (len > kSortBlockMax) { if (tid == 0) atomicOr(a.work_count + 1, 1u); continue; }     // the host falls back to the radix sort
        int N = 2;
        while (N < len) N <<= 1;
        __syncthreads();
        for (int i = tid; i < N; i += kBlock) sk[i] = i < len ? (K)ListKey<T>::key(va[lo + i]) : (K)~(K)0;
        __syncthreads();
        for (int k = 2; k <= N; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < N / 2; t += kBlock) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));      // the lower index of pair t at distance j
                    const int p = i | j;
                    const bool up = (i & k) == 0;
                    const K x = sk[i], y = sk[p];
                    if ((x > y) == up) { sk[i] = y; sk[p] = x; }
                }
                __syncthreads();
            }
        for (int i = tid; i < len; i += kBlock) out[lo + i] = ListKey<T>::unkey((uint64_t)sk[i]);
    }
}
// lane_first: short rows on average — the row-per-lane kernel first, the block kernel over its worklist; otherwise every row
// through the block kernel.  a.work: [n] row numbers, a.work_count: [0] worklist length, [1] flag "a row beyond kSortBlockMax"
hipError_t launch_list_sort(const ListArgs& a, bool lane_first, hipStream_t s) {
    if (a.n <= 0) return hipSuccess;
    if (lane_first) {
        int64_t grid64 = ((a.n + 63) / 64 + (kBlock / 64) - 1) / (kBlock / 64);
        if (grid64 > eval_grid_limit()) grid64 = eval_grid_limit();
        const int grid = (int)grid64;
        RDF_LIST_DISPATCH(list_sort_lane_kernel)
    }
    {
        ListArgs b = a;
        if (!lane_first) b.work = nullptr;
        const ListArgs& a = b;
        int64_t grid64 = lane_first ? eval_grid_limit() / 2 : a.n;
        if (grid64 > 5 * (int64_t)(eval_grid_limit() / 8)) grid64 = 5 * (int64_t)(eval_grid_limit() / 8);
        const int grid = (int)(grid64 < 1 ? 1 : grid64);
        RDF_LIST_DISPATCH(list_sort_block_kernel)
    }
    return hipGetLastError();
}
hipError_t launch_list_offsets(const int64_t* scan, int64_t n1, int32_t* out, hipStream_t s) {
    int64_t grid = (n1 + 255) / 256;
    if (grid > eval_grid_limit()) grid = eval_grid_limit();
    hipLaunchKernelGGL(list_offsets_kernel, dim3((unsigned)(grid < 1 ? 1 : grid)), dim3(256), 0, s, scan, n1, out);
    return hipGetLastError();
}

}  // namespace rdfk
