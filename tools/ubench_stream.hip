// ubench_stream.hip — what a read-only f64 stream with the headline's arithmetic reaches on this part, by loop shape and grid
// (the question behind bench.py's roofline.frac: is the last 10 % of the 8 TB/s the memory system's or the kernel's?).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_stream.bin tools/ubench_stream.hip
// Every kernel reads the same 8 GB column once: `xor` folds the raw words (the cheapest possible consumer), `agg` keeps
// {sum, min, max, count} of the rows with x > 0.5 like spec_kernel's headline program; U = 16-byte loads in flight per lane
// and iteration; grids of 4 / 8 / 16 blocks of 256 threads per CU; nontemporal and plain loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
typedef double f64x2 __attribute__((ext_vector_type(2)));

template <int U, bool NT, bool AGG>
__global__ __launch_bounds__(256) void stream_kernel(const f64x2* __restrict__ src, int64_t nvec, double* out) {
    double sum = 0.0, mn = 1e300, mx = -1e300;
    long long cnt = 0;
    uint64_t x = 0;
    const int64_t stride = (int64_t)gridDim.x * 256 * U;
    for (int64_t i = (int64_t)blockIdx.x * 256 * U + threadIdx.x; i < nvec; i += stride) {
        f64x2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t j = i + u * 256;
            v[u] = j < nvec ? (NT ? __builtin_nontemporal_load(src + j) : src[j]) : f64x2{0.0, 0.0};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (AGG) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const double t = v[u][e];
                    const bool k = t > 0.5;
                    sum += k ? t : 0.0;
                    mn = k ? fmin(mn, t) : mn;
                    mx = k ? fmax(mx, t) : mx;
                    cnt += k;
                }
            } else x ^= __double_as_longlong(v[u].x) ^ __double_as_longlong(v[u].y);
        }
    }
    if (AGG) { if (sum == 1.2345 && mn == mx && cnt == 77) out[0] = sum; }
    else if (x == 0x1234567) out[0] = 1.0;
}

int main() {
    const int64_t n = 1000000000, nvec = n / 2;
    f64x2* src = nullptr;
    double* out = nullptr;
    CK(hipMalloc((void**)&src, (size_t)n * 8 + 4096));
    CK(hipMalloc((void**)&out, 64));
    // values in [0, 1): a cheap fill
    {
        double* h = (double*)malloc(1 << 24);
        for (int i = 0; i < (1 << 21); ++i) h[i] = (double)((i * 2654435761u) >> 8) / 16777216.0;
        for (int64_t off = 0; off < n * 8; off += 1 << 24) CK(hipMemcpy((char*)src + off, h, (size_t)((n * 8 - off) < (1 << 24) ? (n * 8 - off) : (1 << 24)), hipMemcpyHostToDevice));
        free(h);
    }
    int ncu = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) == hipSuccess) ncu = prop.multiProcessorCount;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, int per_cu, auto kernel) {
        const int grid = ncu * per_cu;
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, (const f64x2*)src, nvec, out);
        CK(hipDeviceSynchronize());
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, (const f64x2*)src, nvec, out);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        printf("%-28s %2d blocks/CU  %7.3f ms  %7.1f GB/s  %.3f of 8 TB/s\n", name, per_cu, best, n * 8.0 / best / 1e6, n * 8.0 / best / 1e6 / 8000.0);
    };
    for (int per_cu : {4, 8, 16, 32}) {
        run("xor U=1 nontemporal", per_cu, stream_kernel<1, true, false>);
        run("xor U=4 nontemporal", per_cu, stream_kernel<4, true, false>);
        run("agg U=1 nontemporal", per_cu, stream_kernel<1, true, true>);
        run("agg U=2 nontemporal", per_cu, stream_kernel<2, true, true>);
        run("agg U=4 nontemporal", per_cu, stream_kernel<4, true, true>);
        run("agg U=4 plain loads", per_cu, stream_kernel<4, false, true>);
    }
    return 0;
}
