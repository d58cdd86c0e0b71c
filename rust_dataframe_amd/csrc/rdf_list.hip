// rdf_list.hip — ArrayFunctions over List<primitive> columns (src/functions/array.rs:15-399): the per-row kernels of the
// reference (one ListArray row = value_slice(value_offset(i), value_length(i)) of the child values) as segmented device
// kernels.  Child validity is ignored exactly like the reference's value_slice() does.
//
//   list_rows_kernel<T>  : one LIST ROW per lane (short lists: neighbouring lanes walk neighbouring slices, every
//                          64-byte sector is consumed within a few iterations), results leave as coalesced stores /
//                          ballot-built bitmap words
//   list_wave_kernel<T>  : one list row per WAVE (long lists): lanes stride the slice, butterfly reduction
//   list_row_ids_kernel  : element -> row number (for the sort / remove compositions in rdf_capi.cpp)
//   list_remove_kernel<T>: array_remove in two passes (count per row -> scan -> write at the scanned offsets)
#include "rdf_common.hip.h"

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
template <class T>
__global__ __launch_bounds__(kBlock) void list_rows_kernel(const ListArgs a) {
    __shared__ T stage[kBlock / 64][kListStage];
    const int lane = threadIdx.x & 63;
    const GlobalPtr<int32_t> off = as_global<int32_t>(a.offsets.values) + a.offsets.offset;
    const GlobalPtr<T> vals = as_global<T>(a.values.values) + a.values.offset;
    const T needle = list_needle<T>(a.needle);
    int nulls = 0;
    const int64_t nwaves = (a.n + 63) >> 6;
    for (int64_t wv = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); wv < nwaves; wv += (int64_t)gridDim.x * (kBlock / 64)) {
        const int64_t row = wv * 64 + lane;
        const bool inr = row < a.n;
        uint64_t lv = ~0ull;
        if (a.offsets.validity) lv = load_bits64(a.offsets.validity, a.offsets.offset + wv * 64, clamp64(a.n - wv * 64));
        const bool lvalid = inr && ((lv >> lane) & 1);
        int32_t b = 0, e = 0;
        if (lvalid) { b = off[row]; e = off[row + 1]; }
        bool found = false;
        int32_t pos = 0;
        uint64_t best = a.op == LIST_MAX ? 0 : ~0ull;
        bool any = false, nan_seen = false;
        // The slices of the wave's 64 consecutive rows are one contiguous span of the child array (value_offsets are
        // monotone, also across NULL rows): short spans are copied into LDS with coalesced loads and the lanes walk
        // their slices there; a lane reading its slice straight from HBM touches a different cache line than its
        // neighbours in every iteration (measured 0.20 of peak on rows of 10 f64).
        const int64_t rlast = wv * 64 + 64 < a.n ? wv * 64 + 64 : a.n;
        const int32_t span0 = off[wv * 64], span = off[rlast] - span0;
        const bool staged = span <= kListStage;
        if (staged) {
            for (int32_t i = lane; i < span; i += 64) stage[threadIdx.x >> 6][i] = vals[span0 + i];
            __builtin_amdgcn_wave_barrier();   // same-wave LDS operations execute in order; this pins the compiler
        }
        for (int32_t i = b; i < e; ++i) {
            const T v = staged ? stage[threadIdx.x >> 6][i - span0] : vals[i];
            if (a.op == LIST_CONTAINS || a.op == LIST_POSITION) {
                if (!found && v == needle) { found = true; pos = i - b + 1; }
            } else {
                if (list_is_nan(v)) { nan_seen = true; continue; }
                const uint64_t k = ListKey<T>::key(v);
                best = a.op == LIST_MAX ? (k > best ? k : best) : (k < best ? k : best);
                any = true;
            }
        }
        __builtin_amdgcn_wave_barrier();   // the next iteration refills the staging area
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
// pass 2 (scan != nullptr): the kept elements are written at the scanned offsets, order preserved
template <class T>
__global__ __launch_bounds__(kBlock) void list_remove_kernel(const ListArgs a) {
    __shared__ T stage_in[kBlock / 64][kListStage];
    __shared__ T stage_out[kBlock / 64][kListStage];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const GlobalPtr<int32_t> off = as_global<int32_t>(a.offsets.values) + a.offsets.offset;
    const GlobalPtr<T> vals = as_global<T>(a.values.values) + a.values.offset;
    const T needle = list_needle<T>(a.needle);
    const int64_t nwaves = (a.n + 63) >> 6;
    for (int64_t wv = (int64_t)blockIdx.x * (kBlock / 64) + w; wv < nwaves; wv += (int64_t)gridDim.x * (kBlock / 64)) {
        const int64_t row = wv * 64 + lane;
        const bool inr = row < a.n;
        bool lvalid = inr;
        if (inr && a.offsets.validity) { const int64_t bi = a.offsets.offset + row; lvalid = (as_global<uint8_t>(a.offsets.validity)[bi >> 3] >> (bi & 7)) & 1; }
        const int32_t b = lvalid ? off[row] : 0, e = lvalid ? off[row + 1] : 0;
        // the wave's 64 rows cover one contiguous span of the child array and of the output: both go through LDS when short
        const int64_t rlast = wv * 64 + 64 < a.n ? wv * 64 + 64 : a.n;
        const int32_t span0 = off[wv * 64], span = off[rlast] - span0;
        const bool staged = span <= kListStage;
        if (staged) {
            for (int32_t i = lane; i < span; i += 64) stage_in[w][i] = vals[span0 + i];
            __builtin_amdgcn_wave_barrier();
        }
        if (!a.scan) {
            int64_t c = 0;
            for (int32_t i = b; i < e; ++i) c += (staged ? stage_in[w][i - span0] : vals[i]) != needle;
            if (inr) a.kept[row] = c;
        } else {
            const int64_t out0 = a.scan[wv * 64], out_n = a.scan[rlast] - out0;
            int64_t o = inr ? a.scan[row] : 0;
            for (int32_t i = b; i < e; ++i) {
                const T v = staged ? stage_in[w][i - span0] : vals[i];
                if (v != needle) {
                    if (staged) stage_out[w][o - out0] = v; else as_global_mut<T>(a.out.values)[o] = v;
                    ++o;
                }
            }
            if (staged) {
                __builtin_amdgcn_wave_barrier();
                for (int64_t i = lane; i < out_n; i += 64) as_global_mut<T>(a.out.values)[out0 + i] = stage_out[w][i];
            }
        }
        __builtin_amdgcn_wave_barrier();   // the next iteration refills the staging areas
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
    if (wave_per_row) { RDF_LIST_DISPATCH(list_wave_kernel) } else { RDF_LIST_DISPATCH(list_rows_kernel) }
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
hipError_t launch_list_offsets(const int64_t* scan, int64_t n1, int32_t* out, hipStream_t s) {
    int64_t grid = (n1 + 255) / 256;
    if (grid > eval_grid_limit()) grid = eval_grid_limit();
    hipLaunchKernelGGL(list_offsets_kernel, dim3((unsigned)(grid < 1 ? 1 : grid)), dim3(256), 0, s, scan, n1, out);
    return hipGetLastError();
}

}  // namespace rdfk
