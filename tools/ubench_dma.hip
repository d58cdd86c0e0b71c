// ubench_dma.hip — what the LDS-DMA path (global_load_lds, 16 bytes per lane: HBM -> LDS without registers) reads per second when nothing
// consumes the data, next to the same bytes through 16-byte register loads: the ceiling behind the one-pass compaction kernels
// (rdf_filter.hip) and behind the staged variant of the grouped kernel that DESIGN.md 7.0 reports as slower than register loads.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_dma.bin tools/ubench_dma.hip
// A wave takes 2 KB (128 rows of 16 bytes) per step, block-strided persistent grid; DEPTH steps are kept in flight per wave
// (s_waitcnt vmcnt leaves the youngest DEPTH - 1 outstanding); LDS per wave = DEPTH x 2 KB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
typedef __attribute__((address_space(3))) void* LdsPtr;
typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));

template <int DEPTH>
__global__ __launch_bounds__(256) void dma_kernel(const char* __restrict__ src, int64_t nsteps, uint64_t* out) {
    __shared__ __attribute__((aligned(16))) unsigned char buf[4][DEPTH][2048];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t nw = (int64_t)gridDim.x * 4, w = (int64_t)blockIdx.x * 4 + wave;
    int slot = 0;
    for (int64_t s = w; s < nsteps; s += nw) {
        const char* p = src + s * 2048;
#pragma unroll
        for (int g = 0; g < 2; ++g)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + (g * 64 + lane) * 16), (LdsPtr)(buf[wave][slot] + g * 1024), 16, 0, 0);
        slot = slot + 1 == DEPTH ? 0 : slot + 1;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (DEPTH - 1)) : "memory");      // the step that used this slot DEPTH steps ago has landed
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (buf[wave][0][lane] == 0x5A && out) out[0] = 1;
}
template <int U>
__global__ __launch_bounds__(256) void reg_kernel(const char* __restrict__ src, int64_t nsteps, uint64_t* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t nw = (int64_t)gridDim.x * 4, w = (int64_t)blockIdx.x * 4 + wave;
    uint64_t x = 0;
    for (int64_t s = w; s + (U - 1) * nw < nsteps; s += U * nw) {
        u64x2 v[U][2];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int g = 0; g < 2; ++g) v[u][g] = __builtin_nontemporal_load((const u64x2*)(src + (s + u * nw) * 2048 + (g * 64 + lane) * 16));
#pragma unroll
        for (int u = 0; u < U; ++u) x ^= v[u][0].x ^ v[u][0].y ^ v[u][1].x ^ v[u][1].y;
    }
    if (x == 0x1234567 && out) out[0] = 1;
}

int main() {
    const int64_t bytes = 8000000000ll, nsteps = bytes / 2048;
    char* src = nullptr;
    uint64_t* out = nullptr;
    CK(hipMalloc((void**)&src, (size_t)bytes + 4096));
    CK(hipMemset(src, 1, (size_t)bytes));
    CK(hipMalloc((void**)&out, 64));
    int ncu = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) == hipSuccess) ncu = prop.multiProcessorCount;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, int per_cu, auto kernel) {
        const int grid = ncu * per_cu;
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, (const char*)src, nsteps, out);
        CK(hipDeviceSynchronize());
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, (const char*)src, nsteps, out);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        printf("{\"path\": \"%s\", \"blocks_per_cu\": %d, \"ms\": %.3f, \"GBps\": %.1f, \"frac_of_8TBps\": %.3f}\n", name, per_cu, best, bytes / best / 1e6, bytes / best / 1e6 / 8000.0);
    };
    for (int per_cu : {2, 3, 4, 8}) {
        run("lds-dma, 1 step in flight per wave", per_cu, dma_kernel<1>);
        run("lds-dma, 2 steps in flight per wave", per_cu, dma_kernel<2>);
        run("lds-dma, 4 steps in flight per wave", per_cu, dma_kernel<4>);
        run("lds-dma, 8 steps in flight per wave", per_cu, dma_kernel<8>);
        run("register loads, 1 step per iteration", per_cu, reg_kernel<1>);
        run("register loads, 2 steps per iteration", per_cu, reg_kernel<2>);
        run("register loads, 4 steps per iteration", per_cu, reg_kernel<4>);
    }
    return 0;
}
