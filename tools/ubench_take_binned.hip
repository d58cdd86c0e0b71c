// ubench_take_binned.hip — round 6: would binning a take's indices by source page pay?  ("bin indices by 2 MB page
// (one radix pass on the high bits, carrying output positions), gather, and scatter back — or prove it loses".)
// Column::take (src/table.rs:218-241): out[j] = src[idx[j]], 2.5e8 random u32 indices into 1e9 f64 rows.  Timed here:
//   plain            the gather as rdf_take does it (one 128-byte line per element: 6.2 ms in the product)
//   binned_seq_out   the pairs (idx, pos) ALREADY sorted by 2 MB source page (the binning pass is NOT timed), values written in
//                    the sorted order — the read side's best case, not a take (the output order is wrong)
//   binned           the same pairs, out[pos] = src[idx]: the take through page-sorted pairs, binning pass not timed
//   scatter_only     out[pos] = (a sequential read): what the scattered 8-byte writes cost alone
// An XCD sweeps its eighth of the pair list with all its blocks side by side, so that about one page of the source is live in its L2.
// The binning pass itself would be one radix scatter pass over 2.5e8 (idx, pos) pairs: 653 us per 5e7 pairs in rdf_sort.hip
// (profiles/r05_sorts_box17.jsonl) = 3.3 ms.  binned + 3.3 ms against plain decides.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_take_binned.bin tools/ubench_take_binned.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
#define GAS __attribute__((address_space(1)))

constexpr int kPageShift = 18;       // 2^18 f64 rows = 2 MB

__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ void gen_src(double* x, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) x[i] = (double)(mix(i * 0x9E3779B97F4A7C15ull) >> 11);
}
__global__ void gen_idx(uint32_t* idx, long long m, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < m; i += (long long)gridDim.x * blockDim.x) idx[i] = (uint32_t)(mix(i * 0xD1B54A32D192ED03ull + 7) % (uint64_t)n);
}
// untimed preparation: counting sort of (idx, pos) by page (order inside a page as the atomics fall)
__global__ void bin_hist(const uint32_t* idx, long long m, unsigned long long* hist) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < m; i += (long long)gridDim.x * blockDim.x) atomicAdd(&hist[idx[i] >> kPageShift], 1ull);
}
__global__ void bin_scatter(const uint32_t* idx, long long m, unsigned long long* cursor, uint2* pairs) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < m; i += (long long)gridDim.x * blockDim.x) {
        const unsigned long long at = atomicAdd(&cursor[idx[i] >> kPageShift], 1ull);
        pairs[at] = make_uint2(idx[i], (uint32_t)i);
    }
}

__global__ __launch_bounds__(256) void take_plain(const double* src, const uint32_t* idx, double* out, long long m) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < m; i += (long long)gridDim.x * 256) out[i] = ((const GAS double*)src)[((const GAS uint32_t*)idx)[i]];
}
// chunk c of 256 pairs of XCD x's eighth is taken by that XCD's block (c mod slots) in round c / slots: the XCD's blocks walk side by side
template <int MODE>   // 0: out[sorted place] = src[idx]   1: out[pos] = src[idx]   2: out[pos] = src[sorted place] (sequential read)
__global__ __launch_bounds__(256) void take_binned(const double* src, const uint2* pairs, double* out, long long m) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const long long per = (m + 7) / 8, lo = per * xcd, hi = lo + per < m ? lo + per : m;
    for (long long c = slot; lo + c * 256 < hi; c += slots) {
        const long long i = lo + c * 256 + threadIdx.x;
        if (i >= hi) continue;
        const unsigned long long pw = ((const GAS unsigned long long*)pairs)[i];
        uint2 p;
        p.x = (uint32_t)pw; p.y = (uint32_t)(pw >> 32);
        const double v = MODE == 2 ? ((const GAS double*)src)[i] : ((const GAS double*)src)[p.x];
        ((GAS double*)out)[MODE == 0 ? i : (long long)p.y] = v;
    }
}
__global__ void checksum(const double* a, long long m, unsigned long long* acc) {
    unsigned long long s = 0;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < m; i += (long long)gridDim.x * blockDim.x) s += (unsigned long long)a[i] * (unsigned long long)(i | 1);
    atomicAdd(acc, s);
}

int main(int argc, char** argv) {
    const long long n = argc > 1 ? (long long)atof(argv[1]) : 1000000000ll, m = argc > 2 ? (long long)atof(argv[2]) : 250000000ll;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    double *src, *out, *ref;
    uint32_t* idx;
    uint2* pairs;
    unsigned long long *hist, *acc;
    const long long npages = (n >> kPageShift) + 2;
    CK(hipMalloc(&src, n * 8)); CK(hipMalloc(&out, m * 8)); CK(hipMalloc(&ref, m * 8)); CK(hipMalloc(&idx, m * 4)); CK(hipMalloc(&pairs, m * 8));
    CK(hipMalloc(&hist, npages * 8)); CK(hipMalloc(&acc, 16));
    hipLaunchKernelGGL(gen_src, dim3(ncu * 8), dim3(256), 0, 0, src, n);
    hipLaunchKernelGGL(gen_idx, dim3(ncu * 8), dim3(256), 0, 0, idx, m, n);
    CK(hipMemset(hist, 0, npages * 8));
    hipLaunchKernelGGL(bin_hist, dim3(ncu * 8), dim3(256), 0, 0, idx, m, hist);
    std::vector<unsigned long long> h(npages);
    CK(hipMemcpy(h.data(), hist, npages * 8, hipMemcpyDeviceToHost));
    unsigned long long run = 0;
    for (long long p = 0; p < npages; ++p) { const unsigned long long c = h[p]; h[p] = run; run += c; }
    CK(hipMemcpy(hist, h.data(), npages * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(bin_scatter, dim3(ncu * 8), dim3(256), 0, 0, idx, m, hist, pairs);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto sum_of = [&](const double* a) {
        CK(hipMemset(acc, 0, 8));
        hipLaunchKernelGGL(checksum, dim3(ncu * 8), dim3(256), 0, 0, a, m, acc);
        unsigned long long s = 0;
        CK(hipMemcpy(&s, acc, 8, hipMemcpyDeviceToHost));
        return s;
    };
    unsigned long long want = 0;
    auto run_one = [&](const char* name, auto launch, bool is_take, double alg_bytes) {
        std::vector<float> ms;
        for (int r = 0; r < 6; ++r) {
            CK(hipEventRecord(e0, 0));
            launch();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float t = 0;
            CK(hipEventElapsedTime(&t, e0, e1));
            if (r) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        const unsigned long long got = sum_of(out);
        if (want == 0 && is_take) want = got;
        printf("{\"variant\": \"%s\", \"rows\": %lld, \"indices\": %lld, \"ms_median\": %.3f, \"ms_min\": %.3f, \"alg_GBps\": %.1f, \"is_a_take\": %s, \"result_matches_plain\": %s}\n", name, n, m,
               ms[ms.size() / 2], ms[0], alg_bytes / ms[ms.size() / 2] / 1e6, is_take ? "true" : "false", is_take ? (got == want ? "true" : "false") : "null");
        fflush(stdout);
        CK(hipMemset(out, 0, m * 8));
    };
    const int gp = ncu * 8, gb = ncu * 8;      // (gb: a multiple of 8)
    const double alg = (double)m * 20.0;
    run_one("plain", [&] { hipLaunchKernelGGL(take_plain, dim3(gp), dim3(256), 0, 0, src, idx, out, m); }, true, alg);
    run_one("binned (pairs pre-sorted by 2 MB page, binning pass not timed)", [&] { hipLaunchKernelGGL(take_binned<1>, dim3(gb), dim3(256), 0, 0, src, pairs, out, m); }, true, alg);
    run_one("binned_seq_out (read side only: wrong output order)", [&] { hipLaunchKernelGGL(take_binned<0>, dim3(gb), dim3(256), 0, 0, src, pairs, out, m); }, false, alg);
    run_one("scatter_only (sequential read, out[pos] write)", [&] { hipLaunchKernelGGL(take_binned<2>, dim3(gb), dim3(256), 0, 0, src, pairs, out, m); }, false, alg);
    return 0;
}
