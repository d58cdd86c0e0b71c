// rdf_probe.hip — bare streaming kernels: what the memory system of THIS device gives a read-only stream, a copy, and two reads +
// one write, with nothing but the cheapest possible consumer attached (rdf_probe_stream; bench.py's `roofline.peak_measured`,
// DESIGN.md's two-number memory model).  Measurement helpers: no operator of the path calls them.
// Loop shapes: U 16-byte vectors in flight per lane and iteration (1 / 4), grids of 4 / 8 / 16 blocks of 256 threads per CU,
// nontemporal loads and stores; the best one is reported with its name (which shape wins moves with the box by a few percent:
// profiles/r05_ubench_stream*.txt).
#include "rdf_common.hip.h"

namespace rdfk {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ __launch_bounds__(kBlock) void probe_read_kernel(const u32x4* __restrict__ src, int64_t nvec, uint32_t* out) {
    u32x4 x = {0, 0, 0, 0};
    const int64_t stride = (int64_t)gridDim.x * kBlock * U;
    const GlobalPtr<u32x4> s = (GlobalPtr<u32x4>)src;
    for (int64_t i = (int64_t)blockIdx.x * kBlock * U + threadIdx.x; i < nvec; i += stride) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int64_t j = i + u * kBlock; v[u] = j < nvec ? __builtin_nontemporal_load(s + j) : x; }
#pragma unroll
        for (int u = 0; u < U; ++u) x ^= v[u];
    }
    if ((x.x ^ x.y ^ x.z ^ x.w) == 0x12345677u) out[0] = 1;       // (keeps the loads alive; practically never true)
}

template <int U>
__global__ __launch_bounds__(kBlock) void probe_copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int64_t nvec) {
    const int64_t stride = (int64_t)gridDim.x * kBlock * U;
    const GlobalPtr<u32x4> s = (GlobalPtr<u32x4>)src;
    const GlobalMutPtr<u32x4> d = (GlobalMutPtr<u32x4>)dst;
    for (int64_t i = (int64_t)blockIdx.x * kBlock * U + threadIdx.x; i < nvec; i += stride) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int64_t j = i + u * kBlock; if (j < nvec) v[u] = __builtin_nontemporal_load(s + j); }
#pragma unroll
        for (int u = 0; u < U; ++u) { const int64_t j = i + u * kBlock; if (j < nvec) __builtin_nontemporal_store(v[u], d + j); }
    }
}

// c = a ^ b: two streams in, one out (the traffic shape of `add` into a new column)
template <int U>
__global__ __launch_bounds__(kBlock) void probe_triad_kernel(const u32x4* __restrict__ a, const u32x4* __restrict__ b, u32x4* __restrict__ c, int64_t nvec) {
    const int64_t stride = (int64_t)gridDim.x * kBlock * U;
    const GlobalPtr<u32x4> pa = (GlobalPtr<u32x4>)a, pb = (GlobalPtr<u32x4>)b;
    const GlobalMutPtr<u32x4> pc = (GlobalMutPtr<u32x4>)c;
    for (int64_t i = (int64_t)blockIdx.x * kBlock * U + threadIdx.x; i < nvec; i += stride) {
        u32x4 x[U], y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int64_t j = i + u * kBlock; if (j < nvec) { x[u] = __builtin_nontemporal_load(pa + j); y[u] = __builtin_nontemporal_load(pb + j); } }
#pragma unroll
        for (int u = 0; u < U; ++u) { const int64_t j = i + u * kBlock; if (j < nvec) __builtin_nontemporal_store(x[u] ^ y[u], pc + j); }
    }
}

// kind 0 read (a), 1 copy (a -> b), 2 two reads + one write (a, b -> c); shape = u * 100 + blocks per CU
hipError_t launch_probe(int kind, int u, int grid, const void* a, void* b, void* c, int64_t nvec, uint32_t* sink, hipStream_t s) {
    const dim3 g((unsigned)grid), t(kBlock);
    if (kind == 0) {
        if (u == 1) hipLaunchKernelGGL(probe_read_kernel<1>, g, t, 0, s, (const u32x4*)a, nvec, sink);
        else hipLaunchKernelGGL(probe_read_kernel<4>, g, t, 0, s, (const u32x4*)a, nvec, sink);
    } else if (kind == 1) {
        if (u == 1) hipLaunchKernelGGL(probe_copy_kernel<1>, g, t, 0, s, (const u32x4*)a, (u32x4*)b, nvec);
        else hipLaunchKernelGGL(probe_copy_kernel<4>, g, t, 0, s, (const u32x4*)a, (u32x4*)b, nvec);
    } else {
        if (u == 1) hipLaunchKernelGGL(probe_triad_kernel<1>, g, t, 0, s, (const u32x4*)a, (const u32x4*)b, (u32x4*)c, nvec);
        else hipLaunchKernelGGL(probe_triad_kernel<4>, g, t, 0, s, (const u32x4*)a, (const u32x4*)b, (u32x4*)c, nvec);
    }
    return hipGetLastError();
}

}  // namespace rdfk
