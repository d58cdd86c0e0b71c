// ubench_streams.hip — what several concurrent column streams of DIFFERENT widths reach with gspec_kernel's access shape and none of
// its arithmetic (the question behind Q1's 0.77 of peak next to C3's 0.87: the kernel, or seven streams of 8 / 4 / 1 bytes per row?).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_streams.bin tools/ubench_streams.hip
// A block takes tiles of 1024 rows (block-strided persistent grid), wave w rows [256 w, +256) of the tile, lane l rows 2l, 2l + 1 and
// 128 + 2l, 128 + 2l + 1 of every column: 16-byte loads for 8-byte columns, 8-byte loads for 4-byte ones, 2-byte loads for 1-byte ones —
// every wave instruction covers 128 consecutive rows of one column.  All loads of a tile are issued before any is consumed; the
// consumer is one xor per loaded word.  Layouts: C3 (4 x f64), Q1 (4 x f64 + date32 + 2 x i8), five and seven equal f64 columns,
// Q1 with its narrow columns widened to f64, and Q1's narrow columns alone.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
struct Cols { const char* p[8]; int w[8]; int n; };

template <int W> __device__ __forceinline__ uint64_t load2(const char* base, int64_t row) {     // rows row, row + 1 of a W-byte column
    if constexpr (W == 8) { const u64x2 v = __builtin_nontemporal_load((const u64x2*)(base + row * 8)); return v.x ^ v.y; }
    else if constexpr (W == 4) return __builtin_nontemporal_load((const uint64_t*)(base + row * 4));
    else if constexpr (W == 1) return __builtin_nontemporal_load((const uint16_t*)(base + row));
    else return 0;
}
template <int W0, int W1, int W2, int W3, int W4, int W5, int W6>
__global__ __launch_bounds__(256) void streams_kernel(const Cols c, int64_t ntiles, uint64_t* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t x = 0;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t r0 = t * 1024 + wave * 256 + 2 * lane;
        uint64_t v[7][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            v[0][u] = load2<W0>(c.p[0], r0 + 128 * u); v[1][u] = load2<W1>(c.p[1], r0 + 128 * u); v[2][u] = load2<W2>(c.p[2], r0 + 128 * u);
            v[3][u] = load2<W3>(c.p[3], r0 + 128 * u); v[4][u] = load2<W4>(c.p[4], r0 + 128 * u); v[5][u] = load2<W5>(c.p[5], r0 + 128 * u);
            v[6][u] = load2<W6>(c.p[6], r0 + 128 * u);
        }
#pragma unroll
        for (int k = 0; k < 7; ++k) x ^= v[k][0] ^ v[k][1];
    }
    if (x == 0x1234567) out[0] = 1;
}

int main() {
    const int64_t n = 500000000, ntiles = n / 1024;
    char* buf[7];
    for (int k = 0; k < 7; ++k) { CK(hipMalloc((void**)&buf[k], (size_t)n * 8 + 4096)); CK(hipMemset(buf[k], k + 1, (size_t)n * 8)); }
    uint64_t* out = nullptr;
    CK(hipMalloc((void**)&out, 64));
    int ncu = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) == hipSuccess) ncu = prop.multiProcessorCount;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, int per_cu, double bytes_per_row, auto kernel) {
        Cols c;
        for (int k = 0; k < 7; ++k) { c.p[k] = buf[k]; c.w[k] = 0; }
        c.n = 7;
        const int grid = ncu * per_cu;
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, c, ntiles, out);
        CK(hipDeviceSynchronize());
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, c, ntiles, out);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        const double gbs = n * bytes_per_row / best / 1e6;
        printf("{\"layout\": \"%s\", \"blocks_per_cu\": %d, \"bytes_per_row\": %.0f, \"ms\": %.3f, \"GBps\": %.1f, \"frac_of_8TBps\": %.3f}\n", name, per_cu, bytes_per_row, best, gbs, gbs / 8000.0);
    };
    for (int per_cu : {2, 3, 4, 8}) {
        run("1 x f64", per_cu, 8, streams_kernel<8, 0, 0, 0, 0, 0, 0>);
        run("4 x f64 (C3)", per_cu, 32, streams_kernel<8, 8, 8, 8, 0, 0, 0>);
        run("5 x f64", per_cu, 40, streams_kernel<8, 8, 8, 8, 8, 0, 0>);
        run("7 x f64", per_cu, 56, streams_kernel<8, 8, 8, 8, 8, 8, 8>);
        run("4 x f64 + i32 + 2 x i8 (Q1)", per_cu, 38, streams_kernel<8, 8, 8, 8, 4, 1, 1>);
        run("4 x f64 + i32", per_cu, 36, streams_kernel<8, 8, 8, 8, 4, 0, 0>);
        run("4 x f64 + 2 x i8", per_cu, 34, streams_kernel<8, 8, 8, 8, 0, 1, 1>);
        run("i32 + 2 x i8", per_cu, 6, streams_kernel<0, 0, 0, 0, 4, 1, 1>);
    }
    return 0;
}
