// ubench_scatter.hip — memory-system probes behind the GROUP BY scatter design; not on the product path.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_scatter.bin tools/ubench_scatter.hip
// Exp 1 (slab): does a slab of records written by one kernel stay on-die (Infinity Cache) for the next kernel?
//   per slab of S bytes: K1 reads S cold input bytes and writes S record bytes; K2 reads the S record bytes.
//   "fixed" re-uses one S-byte record buffer for every slab, "cold" gives every slab its own.
// Exp 2 (lines): write bandwidth of a radix scatter as a function of the flush unit: nb blocks x P streams,
//   every block appends one LINE-byte unit to each of its streams per iteration, aligned or shifted by 48 bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_rw(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int64_t n16, int nt_store) {
    const int64_t stride = (int64_t)gridDim.x * 256 * 4;
    for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < n16; i += stride) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = i + u * 256 < n16 ? __builtin_nontemporal_load(src + i + u * 256) : u32x4{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v[u].x ^= 0x9E3779B9u;
            if (i + u * 256 < n16) { if (nt_store) __builtin_nontemporal_store(v[u], dst + i + u * 256); else dst[i + u * 256] = v[u]; }
        }
    }
}
__global__ __launch_bounds__(256) void k_r(const u32x4* __restrict__ src, int64_t n16, uint32_t* out, int nt_load) {
    const int64_t stride = (int64_t)gridDim.x * 256 * 4;
    uint32_t acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < n16; i += stride) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = i + u * 256 < n16 ? (nt_load ? __builtin_nontemporal_load(src + i + u * 256) : src[i + u * 256]) : u32x4{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// Exp 2: block b appends to streams [b*P, (b+1)*P); a unit of LINE bytes is written by LINE/16 consecutive lanes.
template <int LINE>
__global__ __launch_bounds__(512) void k_lines(u32x4* __restrict__ out, int P, int64_t cap_lines, int iters, int shift16, int nt_store) {
    constexpr int G = LINE / 16;                 // lanes per unit
    const int upb = 512 / G;                     // units written per block per store instruction
    const int g = threadIdx.x / G, l = threadIdx.x % G;
    for (int it = 0; it < iters; ++it)
        for (int p = g; p < P; p += upb) {
            const int64_t stream = (int64_t)blockIdx.x * P + p;
            const int64_t a = (stream * cap_lines + it) * G + l + ((stream & 1) ? shift16 : 0);
            u32x4 v = {(uint32_t)a, (uint32_t)it, (uint32_t)p, 7u};
            if (nt_store) __builtin_nontemporal_store(v, out + a); else out[a] = v;
        }
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main(int argc, char** argv) {
    const bool only_lines = argc > 1;   // any argument: Exp 2 only
    const int64_t GB = 1ll << 30;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    char *in, *rec; uint32_t* sink;
    const int64_t T = 8 * GB;
    CK(hipMalloc(&in, T)); CK(hipMalloc(&rec, T + 2 * GB)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(in, 1, T)); CK(hipMemset(rec, 2, T));
    CK(hipDeviceSynchronize());
    const int grid = 2048;
    // ---- Exp 1
    for (int nt = 0; nt < (only_lines ? 0 : 2); ++nt)
        for (int64_t S : {16ll << 20, 32ll << 20, 64ll << 20, 128ll << 20, 256ll << 20, 1024ll << 20}) {
            for (int fixed = 1; fixed >= 0; --fixed) {
                const int64_t nslab = T / S;
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipEventRecord(e0));
                    for (int64_t k = 0; k < nslab; ++k) {
                        char* r = fixed ? rec : rec + k * S;
                        hipLaunchKernelGGL(k_rw, dim3(grid), dim3(256), 0, 0, (const u32x4*)(in + k * S), (u32x4*)r, S / 16, nt);
                        hipLaunchKernelGGL(k_r, dim3(grid), dim3(256), 0, 0, (const u32x4*)r, S / 16, sink, nt);
                    }
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    if (rep == 1) {
                        const float ms = time_ms(e0, e1);
                        printf("{\"exp\":\"slab\",\"nt\":%d,\"slab_MB\":%lld,\"fixed\":%d,\"ms_per_8GB\":%.3f,\"input_GBps\":%.1f,\"us_per_slab\":%.2f}\n",
                               nt, (long long)(S >> 20), fixed, ms, T / 1e9 / (ms * 1e-3), ms * 1e3 / nslab);
                    }
                }
            }
        }
    // read-only and copy denominators on the same buffers
    for (int rep = 0; rep < (only_lines ? 0 : 2); ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_r, dim3(grid), dim3(256), 0, 0, (const u32x4*)in, T / 16, sink, 1);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        if (rep) printf("{\"exp\":\"read8GB\",\"ms\":%.3f,\"GBps\":%.1f}\n", time_ms(e0, e1), T / 1e9 / (time_ms(e0, e1) * 1e-3));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_rw, dim3(grid), dim3(256), 0, 0, (const u32x4*)in, (u32x4*)rec, T / 16, 1);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        if (rep) printf("{\"exp\":\"copy8GB_nt\",\"ms\":%.3f,\"GBps_rw\":%.1f}\n", time_ms(e0, e1), 2 * T / 1e9 / (time_ms(e0, e1) * 1e-3));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_rw, dim3(grid), dim3(256), 0, 0, (const u32x4*)in, (u32x4*)rec, T / 16, 0);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        if (rep) printf("{\"exp\":\"copy8GB_plain\",\"ms\":%.3f,\"GBps_rw\":%.1f}\n", time_ms(e0, e1), 2 * T / 1e9 / (time_ms(e0, e1) * 1e-3));
    }
    CK(hipFree(in));
    // ---- Exp 2: total 8 GB written per configuration
    struct Cfg { int nb, P; };
    for (Cfg c : {Cfg{256, 256}, Cfg{512, 256}, Cfg{512, 512}, Cfg{2048, 512}, Cfg{256, 512}})
        for (int line : {64, 128, 256, 512, 1024})
            for (int shift : {0, 3})
                for (int nt = 0; nt < 2; ++nt) {
                    const int64_t total_lines = T / line;
                    const int iters = (int)(total_lines / ((int64_t)c.nb * c.P));
                    const int64_t cap_lines = iters + 1;
                    if (iters < 1) continue;
                    float ms = 0;
                    for (int rep = 0; rep < 2; ++rep) {
                        CK(hipEventRecord(e0));
                        if (line == 64) hipLaunchKernelGGL(k_lines<64>, dim3(c.nb), dim3(512), 0, 0, (u32x4*)rec, c.P, cap_lines, iters, shift, nt);
                        else if (line == 128) hipLaunchKernelGGL(k_lines<128>, dim3(c.nb), dim3(512), 0, 0, (u32x4*)rec, c.P, cap_lines, iters, shift, nt);
                        else if (line == 256) hipLaunchKernelGGL(k_lines<256>, dim3(c.nb), dim3(512), 0, 0, (u32x4*)rec, c.P, cap_lines, iters, shift, nt);
                        else if (line == 512) hipLaunchKernelGGL(k_lines<512>, dim3(c.nb), dim3(512), 0, 0, (u32x4*)rec, c.P, cap_lines, iters, shift, nt);
                        else hipLaunchKernelGGL(k_lines<1024>, dim3(c.nb), dim3(512), 0, 0, (u32x4*)rec, c.P, cap_lines, iters, shift, nt);
                        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                        ms = time_ms(e0, e1);
                    }
                    const double bytes = (double)c.nb * c.P * iters * line;
                    printf("{\"exp\":\"lines\",\"nb\":%d,\"P\":%d,\"line\":%d,\"shift16\":%d,\"nt\":%d,\"ms\":%.3f,\"write_GBps\":%.1f}\n",
                           c.nb, c.P, line, shift, nt, ms, bytes / 1e9 / (ms * 1e-3));
                }
    CK(hipGetLastError());
    return 0;
}
