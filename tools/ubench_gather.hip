// ubench_gather.hip — calibration of FETCH_SIZE on gather patterns; not on the product path.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_gather.bin tools/ubench_gather.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- tools/ubench_gather.bin      (its own pass)
// MI355X_MICROARCH.md calibrates FETCH_SIZE for ONE pattern — wide coalesced streams, 16 B per lane: the counter shows half
// the bytes — and calls everything else uncalibrated.  The take kernels' "a gathered 8-byte element costs a 128-byte
// transaction" was read off FETCH_SIZE x 2; this probe reads a KNOWN number of elements in patterns whose sector / line
// footprint is known by construction, so that the counter can be turned into bytes for exactly the patterns take and join use:
//   stream16 / stream8   every lane 16 / 8 consecutive bytes (the calibrated case and its 8-byte sibling)
//   stride64/128/256     ONE 8-byte element per 64- / 128- / 256-byte block: every element its own sector / line
//   random8              8-byte elements at uniformly random 8-byte slots of an 8 GiB buffer (a miss in every cache)
//   random32             32-byte records at random 32-byte slots (rows_gather_kernel's pattern)
//   random8_sorted4k     random slots, but a wave's 64 indices sorted inside 4 KiB pages of the source (what an index-bucketing
//                        pass would produce): several elements of a wave may share a 128-byte line
// Every kernel reads exactly N elements (printed) and folds them into a checksum; time and useful GB/s are printed too.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
typedef uint64_t u64x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
__device__ __forceinline__ void fold(uint64_t acc, uint64_t* out) {
    for (int d = 32; d; d >>= 1) acc += __shfl_xor(acc, d);
    if ((threadIdx.x & 63) == 0 && acc == 0x1234567) atomicAdd((unsigned long long*)out, 1ull);   // (never true: keeps the loads alive without a store per wave)
}

__global__ __launch_bounds__(256) void probe_stream16(const u64x2* __restrict__ src, int64_t nvec, uint64_t* out) {
    uint64_t acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) { const u64x2 v = __builtin_nontemporal_load(src + i); acc += v.x ^ v.y; }
    fold(acc, out);
}
__global__ __launch_bounds__(256) void probe_stream8(const uint64_t* __restrict__ src, int64_t n, uint64_t* out) {
    uint64_t acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc += __builtin_nontemporal_load(src + i);
    fold(acc, out);
}
// one 8-byte element per 64 / 128 / 256 bytes, n elements (three names so that the counter file tells them apart)
__global__ __launch_bounds__(256) void probe_stride_64B(const uint64_t* __restrict__ src, int64_t n, uint64_t* out) {
    uint64_t acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc += src[i * 8];
    fold(acc, out);
}
__global__ __launch_bounds__(256) void probe_stride_128B(const uint64_t* __restrict__ src, int64_t n, uint64_t* out) {
    uint64_t acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc += src[i * 16];
    fold(acc, out);
}
__global__ __launch_bounds__(256) void probe_stride_256B(const uint64_t* __restrict__ src, int64_t n, uint64_t* out) {
    uint64_t acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc += src[i * 32];
    fold(acc, out);
}
__global__ __launch_bounds__(256) void probe_random8(const uint64_t* __restrict__ src, int64_t slots, int64_t n, uint64_t* out) {
    uint64_t acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc += src[mix((uint64_t)i * 0x9E3779B97F4A7C15ull + 1) % (uint64_t)slots];
    fold(acc, out);
}
__global__ __launch_bounds__(256) void probe_random32(const u64x4* __restrict__ src, int64_t slots, int64_t n, uint64_t* out) {
    uint64_t acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const u64x4 v = src[mix((uint64_t)i * 0x9E3779B97F4A7C15ull + 7) % (uint64_t)slots];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    fold(acc, out);
}
// a wave's 64 elements fall into ONE random 4 KiB page (512 slots of 8 bytes): up to 64 elements share 32 lines of 128 bytes
__global__ __launch_bounds__(256) void probe_random8_page4k(const uint64_t* __restrict__ src, int64_t slots, int64_t n, uint64_t* out) {
    uint64_t acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const uint64_t page = mix((uint64_t)(i >> 6) * 0x9E3779B97F4A7C15ull + 3) % (uint64_t)(slots / 512);
        acc += src[page * 512 + (mix((uint64_t)i + 11) & 511)];
    }
    fold(acc, out);
}

int main() {
    const int64_t bytes = (int64_t)8 << 30;                  // 8 GiB source: far beyond L2 (32 MiB) and the Infinity Cache (256 MiB)
    uint64_t* src = nullptr;
    uint64_t* out = nullptr;
    CK(hipMalloc((void**)&src, (size_t)bytes + 4096));
    CK(hipMalloc((void**)&out, 64));
    CK(hipMemset(src, 1, (size_t)bytes));
    CK(hipMemset(out, 0, 64));
    const int grid = 256 * 8;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timed = [&](const char* name, int64_t elements, int elem_bytes, const char* footprint, auto launch) {
        launch();                                            // warm-up (code object, TLB)
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-22s elements %11lld  useful bytes %12lld  %8.3f ms  %8.1f GB/s useful  %9.2f G elements/s   footprint: %s\n", name, (long long)elements,
               (long long)elements * elem_bytes, ms, elements * (double)elem_bytes / ms / 1e6, elements / (double)ms / 1e6, footprint);
    };
    const int64_t n16 = bytes / 16, n8 = bytes / 8;
    timed("probe_stream16", n16, 16, "every byte once (8 GiB)", [&] { hipLaunchKernelGGL(probe_stream16, dim3(grid), dim3(256), 0, 0, (const u64x2*)src, n16, out); });
    timed("probe_stream8", n8, 8, "every byte once (8 GiB)", [&] { hipLaunchKernelGGL(probe_stream8, dim3(grid), dim3(256), 0, 0, src, n8, out); });
    timed("probe_stride_64B", bytes / 64, 8, "one 32 B sector of every 64 B block = 4 GiB of sectors, 8 GiB of 64 B blocks", [&] { hipLaunchKernelGGL(probe_stride_64B, dim3(grid), dim3(256), 0, 0, src, bytes / 64, out); });
    timed("probe_stride_128B", bytes / 128, 8, "one sector of every 128 B line = 2 GiB of sectors, 4 GiB of 64 B blocks, 8 GiB of lines", [&] { hipLaunchKernelGGL(probe_stride_128B, dim3(grid), dim3(256), 0, 0, src, bytes / 128, out); });
    timed("probe_stride_256B", bytes / 256, 8, "one sector of every other line = 1 GiB of sectors, 2 GiB of 64 B blocks, 4 GiB of lines", [&] { hipLaunchKernelGGL(probe_stride_256B, dim3(grid), dim3(256), 0, 0, src, bytes / 256, out); });
    const int64_t nr = (int64_t)1 << 28;                     // 2.7e8 gathered elements
    timed("probe_random8", nr, 8, "2^28 random 8 B slots: 8 GiB of sectors, 16 GiB of 64 B blocks, 32 GiB of lines if each is its own", [&] { hipLaunchKernelGGL(probe_random8, dim3(grid), dim3(256), 0, 0, src, n8, nr, out); });
    timed("probe_random32", nr, 32, "2^28 random 32 B records: 8 GiB of sectors, 16 GiB of 64 B blocks, 32 GiB of lines", [&] { hipLaunchKernelGGL(probe_random32, dim3(grid), dim3(256), 0, 0, (const u64x4*)src, bytes / 32, nr, out); });
    timed("probe_random8_page4k", nr, 8, "a wave's 64 elements inside one 4 KiB page (32 lines): <= 2^22 x 4 KiB = 16 GiB of lines, fewer where lanes share one", [&] { hipLaunchKernelGGL(probe_random8_page4k, dim3(grid), dim3(256), 0, 0, src, n8, nr, out); });
    CK(hipFree(src)); CK(hipFree(out));
    return 0;
}
