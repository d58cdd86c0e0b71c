// rdf_lanewin.hip.h — helpers shared by the compaction kernels (rdf_filter.hip, rdf_bfilter.hip): bitmap windows kept one
// 64-bit word per lane, 16-byte vector types, the f64 comparison of BooleanFilter (src/expression.rs:844-852).
#pragma once
#include "rdf_common.hip.h"

namespace rdfk {

template <class T> struct Vec16 { typedef T type __attribute__((ext_vector_type(16 / sizeof(T)))); };

typedef __attribute__((address_space(3))) void* LdsPtr;

// 64-bit windows of a bitmap kept ONE PER LANE (lane j < NW holds window j = bits [bitpos + 64 j, +64), rows past nbits
// cleared): sixteen windows as scalars are 32 SGPRs — with the validity words 64 of the wave's ~100 — and the spills they
// cause cost the kernel its occupancy; a window is read back with v_readlane where it is used.
template <int NW>
struct LaneWin { uint64_t w0, w1; int sh; int64_t nbits; };
template <int NW>
__device__ __forceinline__ LaneWin<NW> lane_windows_issue(const uint8_t* base, int64_t bitpos, int64_t nbits) {
    const int lane = threadIdx.x & 63;
    LaneWin<NW> r;
    r.w0 = 0; r.w1 = 0; r.sh = 0; r.nbits = nbits;
    if (nbits <= 0) return r;
    const uint64_t addr = uniform64((uint64_t)(uintptr_t)base + (uint64_t)(bitpos >> 3));
    const GlobalPtr<uint64_t> w = (GlobalPtr<uint64_t>)(uintptr_t)(addr & ~7ull);
    r.sh = __builtin_amdgcn_readfirstlane((int)(addr & 7) * 8 + (int)(bitpos & 7));
    const int64_t want = nbits < (int64_t)64 * NW ? nbits : (int64_t)64 * NW;
    const int last = __builtin_amdgcn_readfirstlane((int)((r.sh + want - 1) >> 6));   // index of the last word holding a requested bit
    if (lane < NW) { r.w0 = w[lane < last ? lane : last]; r.w1 = w[lane + 1 < last ? lane + 1 : last]; }
    return r;
}
template <int NW>
__device__ __forceinline__ uint64_t lane_windows_finish(const LaneWin<NW>& q) {
    const int lane = threadIdx.x & 63;
    if (q.nbits <= 0) return 0;
    uint64_t r = q.w0 >> q.sh;
    if (q.sh) r |= q.w1 << (64 - q.sh);
    const int64_t left = q.nbits - (int64_t)64 * lane;
    if (left < 64) r = left <= 0 ? 0 : (r & ((1ull << left) - 1));
    return lane < NW ? r : 0;
}
template <int NW>
__device__ __forceinline__ uint64_t lane_windows(const uint8_t* base, int64_t bitpos, int64_t nbits) {
    return lane_windows_finish<NW>(lane_windows_issue<NW>(base, bitpos, nbits));
}
__device__ __forceinline__ uint64_t rl64(uint64_t v, int i) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, i), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), i);
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ bool fcmp(int op, double x, double c) {
    switch (op) {
        case RDF_OP_GT: return x > c;
        case RDF_OP_GE: return x >= c;
        case RDF_OP_EQ: return x == c;
        case RDF_OP_NE: return x != c;
        case RDF_OP_LT: return x < c;
        default: return x <= c;
    }
}

}  // namespace rdfk
