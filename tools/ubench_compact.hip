// ubench_compact.hip — round 6: what a ONE-PASS order-preserving filter (predicate + compaction, DataFrame::filter of one long
// batch, src/dataframe.rs:178-189) can reach on MI355X, with the data path and the prefix (look-back) cost told apart:
//
//   data path   reg : a block's tile (4 waves x 1024 rows) is held in REGISTERS (8 x 16-byte loads per lane and column), the kept
//                     rows are staged in a block-wide LDS buffer at their rank and leave as aligned 16-byte stores
//               dma : the product's path of round 5 — a wave's 1024 rows go global -> LDS with global_load_lds, are compacted in
//                     place and leave from there (no data VGPRs)
//   prefix      pre : the tile offsets come from an exclusive scan computed beforehand (NOT timed): the ceiling of the data path
//               lb  : decoupled look-back between BLOCK tiles (4096 rows), 256 predecessors per round trip, by wave 0, while the
//                     other waves stage their rows; tiles are handed out by 64 ticket counters, the next ticket drawn one
//                     iteration ahead
//   PF (reg)        : the next tile's predicate column is requested before the current tile is staged and stored
//
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_compact.bin tools/ubench_compact.hip
// Run:   tools/ubench_compact.bin [rows=1e9] [reps=7]      (one JSON line per variant; every variant is checked against a
//                                                            three-kernel reference compaction on the device)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define GAS __attribute__((address_space(1)))
typedef double d2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* LdsPtr;

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s:%d: %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

constexpr int kWaves = 4, kBlock = 256, kWRows = 1024, kTile = kWaves * kWRows;
constexpr unsigned long long kAgg = 1ull << 62, kPre = 2ull << 62, kVal = (1ull << 62) - 1;
constexpr int kMaxCols = 4;

struct Args {
    const double* col[kMaxCols];
    double* out[kMaxCols];
    long long n, ntiles;
    double c;
    unsigned long long* state;   // [ntiles] (lb)
    unsigned int* ticket;        // 64 counters, 128 bytes apart (lb) / persistent walk (pre: unused)
    const long long* pre;        // [ntiles + 1] exclusive scan of the tile counts (pre)
    long long* out_len;
    int nclass;                  // (pipe) ticket counters in use: min(64, worker blocks)
    unsigned long long* stats;   // [2] scanner rounds, [3] of them idle; [0] wall-clock ticks (100 MHz) wave 0 spent waiting for its prefix, [1] ticks inside the tile loop, summed over blocks
};

__device__ __forceinline__ unsigned long long ld_state(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_state(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int mbcnt64(uint64_t m, int init) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, init)); }

// Called by all 64 lanes of ONE wave after the tile's row count has been published: finds the rows in front of the tile and
// publishes the inclusive prefix.
__device__ __forceinline__ long long lookback256(unsigned long long* state, long long T, long long cnt) {
    const int lane = threadIdx.x & 63;
    if (T == 0) return 0;            // (the caller has published the tile's count: kPre for tile 0, kAgg otherwise)
    long long excl = 0, hi = T - 1;     // hi: the nearest predecessor not added up yet
    for (;;) {
        unsigned long long w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long idx = hi - (k * 64 + lane);
            w[k] = idx >= 0 ? ld_state(state + idx) : kPre;      // in front of tile 0: a prefix of 0 rows
        }
        bool done = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint64_t ready = __ballot((w[k] >> 62) != 0), pref = __ballot((w[k] >> 62) == 2);
            const int pl = pref ? __builtin_ctzll(pref) : 63;
            const uint64_t need = pl == 63 ? ~0ull : ((2ull << pl) - 1);
            if ((ready & need) != need) break;                     // a predecessor has not published yet: read again from `hi`
            long long v = lane <= pl ? (long long)(w[k] & kVal) : 0;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);
            excl += v;
            hi -= 64;
            if (pref) { done = true; break; }
        }
        if (done) break;
        __builtin_amdgcn_s_sleep(1);
    }
    if (lane == 0) st_state(state + T, kPre | (unsigned long long)(excl + cnt));
    return excl;
}

// LB == 2: ONE wave of the grid (block 0) turns the tiles' row counts into prefixes, in tile order, as far as they have been
// published; a tile publishes its count and polls its OWN word until the prefix is there: 8 bytes of look-back traffic per tile and
// poll instead of a window of its predecessors' words.
__device__ __forceinline__ void scanner(const Args& a) {
    const int lane = threadIdx.x & 63;
    constexpr int K = 8;
    long long cur = 0;
    unsigned long long running = 0;
    while (cur < a.ntiles) {
        unsigned long long w[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const long long idx = cur + k * 64 + lane;
            w[k] = idx < a.ntiles ? ld_state(a.state + idx) : 0;
        }
        long long base = cur;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint64_t ready = __ballot((w[k] >> 62) != 0);
            const int f = ready == ~0ull ? 64 : __builtin_ctzll(~ready);
            if (f == 0) break;
            int val = lane < f ? (int)(w[k] & kVal) : 0, inc = val;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if (lane >= d) inc += t; }
            if (lane < f) st_state(a.state + base + lane, kPre | (running + (unsigned long long)(inc - val)));
            running += (unsigned long long)__shfl(inc, 63);
            base += f;
            if (f < 64) break;
        }
        if (base == cur) __builtin_amdgcn_s_sleep(1);
        cur = base;
    }
}

// aligned 16-byte stores of stage[0, cnt) to out[obase, obase + cnt); `nthreads` threads, this one is `t`
__device__ __forceinline__ void store_run(const double* stage, double* out, long long obase, int cnt, int t, int nthreads) {
    const int shift = (int)(obase & 1);
    GAS double* o = (GAS double*)out + obase;
    if (t == 0 && shift && cnt > 0) o[0] = stage[0];
    const int nv = (cnt - shift) >> 1;
    for (int q = t; q < nv; q += nthreads) {
        const int i0 = shift + 2 * q;
        d2 x;
        x.x = stage[i0]; x.y = stage[i0 + 1];
        __builtin_nontemporal_store(x, (GAS d2*)(o + i0));
    }
    if (t == 0 && cnt > shift && ((cnt - shift) & 1)) o[cnt - 1] = stage[cnt - 1];
}

// ---------------------------------------------------------------------------------------------------------------
// register data path
template <int LB, int PF, int M>
__global__ __launch_bounds__(kBlock, 4) void k_reg(const Args a) {
    __shared__ __attribute__((aligned(16))) double stage[kTile + 2];
    __shared__ int wcnt[2][kWaves];
    __shared__ long long sh_base;
    __shared__ long long sh_tile[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (LB == 2 && blockIdx.x == 0) { if (wave == 0) scanner(a); return; }
    const int ctr = (LB == 2 ? blockIdx.x - 1 : blockIdx.x) & 63;
    auto draw = [&]() -> long long {
        if (LB) return (long long)atomicAdd(a.ticket + ctr * 32, 1u) * 64 + ctr;
        return 0;
    };
    unsigned long long t_stall = 0, t_loop0 = wall_clock64();
    auto load_tile = [&](int k, long long T, d2 (&v)[8]) {
        const GAS d2* p = (const GAS d2*)a.col[k] + (T * (kTile / 2) + wave * (kWRows / 2) + lane);
#pragma unroll
        for (int g = 0; g < 8; ++g) v[g] = __builtin_nontemporal_load(p + g * 64);
    };
    // tiles: LB: by ticket (in order); pre: a static grid-stride walk
    long long T, Tn = 0;
    if (LB) {
        if (tid == 0) { sh_tile[0] = draw(); if (PF) sh_tile[1] = draw(); }
        __syncthreads();
        T = sh_tile[0]; if (PF) Tn = sh_tile[1];
        __syncthreads();
    } else { T = blockIdx.x; Tn = T + gridDim.x; }
    d2 v[8], vn[8];
    if (PF && T < a.ntiles) load_tile(0, T, vn);
    int it = 0;
    while (T < a.ntiles) {
        long long tk_next = 0;
        if (PF) {
#pragma unroll
            for (int g = 0; g < 8; ++g) v[g] = vn[g];
            if (Tn < a.ntiles) load_tile(0, Tn, vn);
        } else load_tile(0, T, v);
        if (LB && tid == 0) tk_next = draw();          // the ticket after the one(s) already held, under the loads just issued
        uint64_t B0[8], B1[8];
        int cnt = 0;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            B0[g] = __ballot(v[g].x > a.c);
            B1[g] = __ballot(v[g].y > a.c);
            cnt += __popcll(B0[g]) + __popcll(B1[g]);
        }
        if (lane == 0) wcnt[it & 1][wave] = cnt;
        __syncthreads();
        int wbase = 0, bcnt = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) { const int c = wcnt[it & 1][w]; if (w < wave) wbase += c; bcnt += c; }
        d2 u[2][8];
        if (M > 1) load_tile(1, T, u[1]);
        auto stage_col = [&](const d2 (&x)[8]) {
            int base = wbase;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int r0 = mbcnt64(B1[g], mbcnt64(B0[g], base));
                const bool k0 = (B0[g] >> lane) & 1, k1 = (B1[g] >> lane) & 1;
                if (k0) stage[r0] = x[g].x;
                if (k1) stage[r0 + (k0 ? 1 : 0)] = x[g].y;
                base += __popcll(B0[g]) + __popcll(B1[g]);
            }
        };
        if (LB == 2) {
            if (wave == 0) {
                if (lane == 0) st_state(a.state + T, kAgg | (unsigned long long)bcnt);
                stage_col(v);
                if (lane == 0) {
                    const unsigned long long t0 = wall_clock64();
                    unsigned long long w;
                    for (;;) { w = ld_state(a.state + T); if ((w >> 62) == 2) break; __builtin_amdgcn_s_sleep(2); }
                    t_stall += wall_clock64() - t0;
                    sh_base = (long long)(w & kVal);
                }
            } else stage_col(v);
        } else if (LB) {
            if (wave == 0) {
                if (lane == 0) st_state(a.state + T, (T == 0 ? kPre : kAgg) | (unsigned long long)bcnt);    // published before anything else
                stage_col(v);
                const unsigned long long t0 = wall_clock64();
                const long long e = lookback256(a.state, T, bcnt);
                if (lane == 0) { sh_base = e; t_stall += wall_clock64() - t0; }
            } else stage_col(v);
        } else {
            stage_col(v);
            if (tid == 0) sh_base = a.pre[T];
        }
        __syncthreads();
        const long long tbase = sh_base;
        store_run(stage, a.out[0], tbase, bcnt, tid, kBlock);
#pragma unroll
        for (int k = 1; k < M; ++k) {
            __syncthreads();                    // the previous column has left the staging buffer
            if (k + 1 < M) load_tile(k + 1, T, u[(k + 1) & 1]);
            stage_col(u[k & 1]);
            __syncthreads();
            store_run(stage, a.out[k], tbase, bcnt, tid, kBlock);
        }
        if (T == a.ntiles - 1 && tid == 0) *a.out_len = tbase + bcnt;
        // next tile
        if (LB) {
            if (tid == 0) sh_tile[it & 1] = tk_next;
            __syncthreads();                    // also: every thread has read its part of the staging buffer
            const long long drawn = sh_tile[it & 1];
            if (PF) { T = Tn; Tn = drawn; } else T = drawn;
        } else {
            __syncthreads();
            T = Tn; Tn += gridDim.x;
        }
        ++it;
    }
    if (tid == 0 && a.stats) { atomicAdd(a.stats, t_stall); atomicAdd(a.stats + 1, wall_clock64() - t_loop0); }
}


// ---------------------------------------------------------------------------------------------------------------
// "pipe": the register path with the count of tile n + 1 published one iteration BEFORE its prefix is asked for.  A block holds
// tile n (counted, its count published during the previous iteration) and tile n + 1 (loads in flight); per iteration:
//   a. wait for tile n + 1, count it, publish the count            (the scanner wave turns counts into prefixes meanwhile)
//   b. poll the prefix of tile n                                   (asked for a whole iteration after the count went out)
//   c. stage tile n at its ranks, store it, request tile n + 2 into the registers tile n has left
// The scanner (block 0, wave 0) does its 64-wide scans with DPP row shifts, not LDS permutes.
__device__ __forceinline__ int wave_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ void scanner_dpp(const Args& a) {
    const int lane = threadIdx.x & 63;
    constexpr int K = 8;
    long long cur = 0;
    unsigned long long running = 0;
    unsigned long long rounds = 0, idle = 0;
    while (cur < a.ntiles) {
        unsigned long long w[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const long long idx = cur + k * 64 + lane;
            w[k] = idx < a.ntiles ? ld_state(a.state + idx) : 0;
        }
        long long base = cur;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint64_t ready = __ballot((w[k] >> 62) != 0);
            const int f = ready == ~0ull ? 64 : __builtin_ctzll(~ready);
            if (f == 0) break;
            const int val = lane < f ? (int)(w[k] & kVal) : 0;
            const int inc = wave_incl_scan(val);
            if (lane < f) st_state(a.state + base + lane, kPre | (running + (unsigned long long)(inc - val)));
            running += (unsigned long long)(unsigned int)__builtin_amdgcn_readlane(inc, 63);
            base += f;
            if (f < 64) break;
        }
        ++rounds;
        if (base == cur) { ++idle; __builtin_amdgcn_s_sleep(1); }
        cur = base;
    }
    if (lane == 0 && a.stats) { a.stats[2] = rounds; a.stats[3] = idle; }
}

template <int M, int WAVES, int G>      // WAVES waves per block, G 16-byte loads per lane and column: a tile is WAVES * G * 128 rows
__global__ __launch_bounds__(WAVES * 64) void k_pipe(const Args a) {
    constexpr int kTile = WAVES * G * 128, kWRows = G * 128, kWaves = WAVES, kBlock = WAVES * 64;
    __shared__ __attribute__((aligned(16))) double stage[kTile + 2];
    __shared__ int wcnt[2][kWaves];
    __shared__ long long sh_base;
    __shared__ long long sh_tile[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (blockIdx.x == 0) { if (wave == 0) scanner_dpp(a); return; }
    const int nclass = a.nclass, ctr = (int)((blockIdx.x - 1) % nclass);
    auto draw = [&]() -> long long { return (long long)atomicAdd(a.ticket + ctr * 32, 1u) * nclass + ctr; };
    unsigned long long t_stall = 0, t_loop0 = wall_clock64();
    auto load_tile = [&](int k, long long T, d2 (&v)[G]) {
        const GAS d2* p = (const GAS d2*)a.col[k] + (T * (kTile / 2) + wave * (kWRows / 2) + lane);
#pragma unroll
        for (int g = 0; g < G; ++g) v[g] = __builtin_nontemporal_load(p + g * 64);
    };
    // count a tile and publish the block's count; returns this wave's rows in front of it inside the tile and the tile's rows
    auto count_publish = [&](const d2 (&x)[G], long long T, int par, int& wbase, int& bcnt) {
        int cnt = 0;
#pragma unroll
        for (int g = 0; g < G; ++g) cnt += __popcll(__ballot(x[g].x > a.c)) + __popcll(__ballot(x[g].y > a.c));
        if (lane == 0) wcnt[par][wave] = cnt;
        __syncthreads();
        wbase = 0; bcnt = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) { const int c = wcnt[par][w]; if (w < wave) wbase += c; bcnt += c; }
        if (tid == 0) st_state(a.state + T, kAgg | (unsigned long long)bcnt);
    };
    if (tid == 0) { sh_tile[0] = draw(); sh_tile[1] = draw(); }
    __syncthreads();
    long long Tc = sh_tile[0], Tn = sh_tile[1];
    __syncthreads();
    if (Tc >= a.ntiles) return;
    d2 A[G], B[G], U[G];
    int wbase_c, bcnt_c, wbase_n = 0, bcnt_n = 0;
    load_tile(0, Tc, A);
    if (Tn < a.ntiles) load_tile(0, Tn, B);
    count_publish(A, Tc, 0, wbase_c, bcnt_c);
    int par = 1;
    // one iteration: X = the current tile's registers, Y = the next tile's
    auto step = [&](d2 (&X)[G], d2 (&Y)[G]) -> bool {
        long long drawn = 0;
        if (tid == 0) drawn = draw();                               // the tile after the next one, under everything below
        if (M > 1) load_tile(1, Tc, U);
        if (Tn < a.ntiles) count_publish(Y, Tn, par, wbase_n, bcnt_n);   // a.
        else __syncthreads();                                      // (the staging buffer and sh_base are reused below)
        par ^= 1;
        if (tid == 0) {                                            // b.
            const unsigned long long t0 = wall_clock64();
            unsigned long long w;
            for (;;) { w = ld_state(a.state + Tc); if ((w >> 62) == 2) break; __builtin_amdgcn_s_sleep(2); }
            t_stall += wall_clock64() - t0;
            sh_base = (long long)(w & kVal);
            sh_tile[par] = drawn;
        }
        uint64_t B0[G], B1[G];                                     // c.
#pragma unroll
        for (int g = 0; g < G; ++g) { B0[g] = __ballot(X[g].x > a.c); B1[g] = __ballot(X[g].y > a.c); }
        auto stage_col = [&](const d2 (&x)[G]) {
            int base = wbase_c;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int r0 = mbcnt64(B1[g], mbcnt64(B0[g], base));
                const bool k0 = (B0[g] >> lane) & 1, k1 = (B1[g] >> lane) & 1;
                if (k0) stage[r0] = x[g].x;
                if (k1) stage[r0 + (k0 ? 1 : 0)] = x[g].y;
                base += __popcll(B0[g]) + __popcll(B1[g]);
            }
        };
        stage_col(X);
        __syncthreads();
        const long long tbase = sh_base, Tnn = sh_tile[par];
        store_run(stage, a.out[0], tbase, bcnt_c, tid, kBlock);
#pragma unroll
        for (int k = 1; k < M; ++k) {
            // columns 1, 3 travel in U, column 2 in X (the predicate column has left it)
            if (k + 1 < M) { if ((k + 1) & 1) load_tile(k + 1, Tc, U); else load_tile(k + 1, Tc, X); }
            __syncthreads();
            if (k & 1) stage_col(U); else stage_col(X);
            __syncthreads();
            store_run(stage, a.out[k], tbase, bcnt_c, tid, kBlock);
        }
        if (Tc == a.ntiles - 1 && tid == 0) *a.out_len = tbase + bcnt_c;
        if (Tnn < a.ntiles) load_tile(0, Tnn, X);                  // tile n + 2 into the registers tile n has left
        Tc = Tn; Tn = Tnn; wbase_c = wbase_n; bcnt_c = bcnt_n;
        return Tc < a.ntiles;
    };
    for (;;) {
        if (!step(A, B)) break;
        if (!step(B, A)) break;
    }
    if (tid == 0 && a.stats) { atomicAdd(a.stats, t_stall); atomicAdd(a.stats + 1, wall_clock64() - t_loop0); }
}

// ---------------------------------------------------------------------------------------------------------------
// LDS-DMA data path (the product's dma_tile / dma_compact of round 5, f64 only), block-level counts
__device__ __forceinline__ uint64_t rl64(uint64_t v, int i) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, i), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), i);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ void dma_tile(const double* col, long long r0, unsigned char* raw_bytes) {
    const int lane = threadIdx.x & 63;
    const GAS double* src = (const GAS double*)col + r0;
#pragma unroll
    for (int g = 0; g < 8; ++g)
        __builtin_amdgcn_global_load_lds((const GAS void*)(src + (g * 64 + lane) * 2), (LdsPtr)(raw_bytes + 16 + g * 1024), 16, 0, 0);
}
// raw tile at element 2.. of `raw`; kept rows end up at [0, cnt)
__device__ __forceinline__ void dma_compact_inplace(double* raw, uint64_t kwv) {
    const int lane = threadIdx.x & 63;
    int wb = 0;
#pragma unroll
    for (int b = 0; b < 16; b += 4) {
        double x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = raw[2 + (b + i) * 64 + lane];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint64_t w = rl64(kwv, b + i);
            if ((w >> lane) & 1) raw[wb + __popcll(w & ((1ull << lane) - 1))] = x[i];
            wb += __popcll(w);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

template <int LB, int M>
__global__ __launch_bounds__(kBlock, 4) void k_dma(const Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char stage[kWaves][kWRows * 8 + 32];
    __shared__ int wcnt[2][kWaves];
    __shared__ long long sh_base;
    __shared__ long long sh_tile[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ctr = blockIdx.x & 63;
    long long T;
    if (LB) {
        if (tid == 0) sh_tile[0] = (long long)atomicAdd(a.ticket + ctr * 32, 1u) * 64 + ctr;
        __syncthreads();
        T = sh_tile[0];
        __syncthreads();
    } else T = blockIdx.x;
    int it = 0;
    while (T < a.ntiles) {
        const long long r0 = T * kTile + wave * kWRows;
        double* raw = (double*)stage[wave];
        dma_tile(a.col[0], r0, stage[wave]);
        long long tk_next = 0;
        if (LB && tid == 0) tk_next = (long long)atomicAdd(a.ticket + ctr * 32, 1u) * 64 + ctr;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint64_t kwv = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint64_t w = __ballot(raw[2 + i * 64 + lane] > a.c);
            if (lane == i) kwv = w;
        }
        int cnt = __popcll(kwv);
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) cnt += __shfl_xor(cnt, d);
        cnt = __builtin_amdgcn_readfirstlane(cnt);
        if (lane == 0) wcnt[it & 1][wave] = cnt;
        __syncthreads();
        int wbase = 0, bcnt = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) { const int c = wcnt[it & 1][w]; if (w < wave) wbase += c; bcnt += c; }
        if (LB) {
            if (wave == 0) {
                if (lane == 0) st_state(a.state + T, (T == 0 ? kPre : kAgg) | (unsigned long long)bcnt);
                dma_compact_inplace(raw, kwv);
                const long long e = lookback256(a.state, T, bcnt);
                if (lane == 0) sh_base = e;
            } else dma_compact_inplace(raw, kwv);
        } else {
            dma_compact_inplace(raw, kwv);
            if (tid == 0) sh_base = a.pre[T];
        }
        __syncthreads();
        const long long tbase = sh_base;
        store_run(raw, a.out[0], tbase + wbase, cnt, lane, 64);
#pragma unroll 1
        for (int k = 1; k < M; ++k) {
            __builtin_amdgcn_wave_barrier();
            dma_tile(a.col[k], r0, stage[wave]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            dma_compact_inplace(raw, kwv);
            store_run(raw, a.out[k], tbase + wbase, cnt, lane, 64);
        }
        __builtin_amdgcn_wave_barrier();
        if (T == a.ntiles - 1 && tid == 0) *a.out_len = tbase + bcnt;
        if (LB) {
            if (tid == 0) sh_tile[it & 1] = tk_next;
            __syncthreads();
            T = sh_tile[it & 1];
        } else T += gridDim.x;
        ++it;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// data, reference, checks
__global__ void gen_kernel(double* x, long long n, unsigned long long seed) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        unsigned long long z = (unsigned long long)i * 0x9E3779B97F4A7C15ull + seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        x[i] = (double)(z >> 11) * (1.0 / 9007199254740992.0);
    }
}
__global__ void ref_count_kernel(const double* x, long long ntiles, double c, long long* counts) {
    __shared__ int acc;
    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        if (threadIdx.x == 0) acc = 0;
        __syncthreads();
        int k = 0;
        for (int i = threadIdx.x; i < kTile; i += blockDim.x) k += x[t * kTile + i] > c;
        atomicAdd(&acc, k);
        __syncthreads();
        if (threadIdx.x == 0) counts[t] = acc;
        __syncthreads();
    }
}
// one block per tile, serial order inside a wave-sized slice: simple and obviously right
__global__ void ref_scatter_kernel(const double* pred, const double* x, long long ntiles, double c, const long long* pre, double* out) {
    __shared__ int wtot[kBlock / 64];
    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        long long base = pre[t];
        for (int s = 0; s < kTile; s += kBlock) {
            const long long r = t * kTile + s + threadIdx.x;
            const bool k = pred[r] > c;
            const uint64_t b = __ballot(k);
            const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
            if (lane == 0) wtot[w] = __popcll(b);
            __syncthreads();
            int off = 0, tot = 0;
            for (int j = 0; j < kBlock / 64; ++j) { if (j < w) off += wtot[j]; tot += wtot[j]; }
            if (k) out[base + off + __popcll(b & ((1ull << lane) - 1))] = x[r];
            base += tot;
            __syncthreads();
        }
    }
}
__global__ void diff_kernel(const unsigned long long* a, const unsigned long long* b, long long n, unsigned long long* bad) {
    unsigned long long k = 0;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) k += a[i] != b[i];
    if (k) atomicAdd(bad, k);
}

int main(int argc, char** argv) {
    long long rows = argc > 1 ? (long long)atof(argv[1]) : 1000000000ll;
    const int reps = argc > 2 ? atoi(argv[2]) : 7;
    const long long ntiles = (rows + 8191) / 8192 * 2;   // reference tiles of kTile = 4096 rows; n is a multiple of 8192
    const long long n = ntiles * kTile;     // whole tiles (the product's kernels take the ragged end; this probe does not)
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    Args a;
    memset(&a, 0, sizeof a);
    a.n = n; a.ntiles = ntiles; a.c = 0.5;
    double* cols[kMaxCols];
    double* outs[kMaxCols];
    double* ref;
    const long long out_cap = n / 2 + n / 64 + 4096;
    for (int k = 0; k < kMaxCols; ++k) {
        CK(hipMalloc(&cols[k], n * 8));
        CK(hipMalloc(&outs[k], out_cap * 8));
        hipLaunchKernelGGL(gen_kernel, dim3(ncu * 8), dim3(256), 0, 0, cols[k], n, 0x1234567ull + 977 * k);
        a.col[k] = cols[k]; a.out[k] = outs[k];
    }
    CK(hipMalloc(&ref, out_cap * 8));
    long long *counts, *pre, *out_len;
    CK(hipMalloc(&counts, (ntiles + 1) * 8));
    CK(hipMalloc(&pre, (ntiles + 1) * 8));
    CK(hipMalloc(&out_len, 8));
    unsigned long long* bad;
    CK(hipMalloc(&bad, 8));
    CK(hipMalloc(&a.state, ntiles * 4 * 8 + 64));
    CK(hipMalloc(&a.ticket, 64 * 128));
    a.pre = pre; a.out_len = out_len;
    CK(hipMalloc(&a.stats, 32));
    hipLaunchKernelGGL(ref_count_kernel, dim3(ncu * 8), dim3(256), 0, 0, cols[0], ntiles, a.c, counts);
    std::vector<long long> hc(ntiles + 1), hp(ntiles + 1);
    CK(hipMemcpy(hc.data(), counts, ntiles * 8, hipMemcpyDeviceToHost));
    long long kept = 0;
    for (long long t = 0; t < ntiles; ++t) { hp[t] = kept; kept += hc[t]; }
    hp[ntiles] = kept;
    if (kept > out_cap) { fprintf(stderr, "output capacity\n"); return 1; }
    CK(hipMemcpy(pre, hp.data(), (ntiles + 1) * 8, hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    auto run = [&](const char* name, int lb, int pf, int m, auto kernel, int block = kBlock, int tile_rows = kTile) {
        int per_cu = 0;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, 0));
        a.ntiles = n / tile_rows;
        const long long grid = std::min<long long>((long long)ncu * per_cu + (lb >= 2 ? 1 : 0), a.ntiles);
        CK(hipMemset(a.stats, 0, 32));
        a.nclass = (int)std::min<long long>(64, grid - 1);
        std::vector<float> ms;
        for (int r = 0; r < reps + 1; ++r) {
            CK(hipEventRecord(e0, 0));
            if (lb) {
                CK(hipMemsetAsync(a.state, 0, a.ntiles * 8, 0));
                CK(hipMemsetAsync(a.ticket, 0, 64 * 128, 0));
            }
            hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(block), 0, 0, a);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float t = 0;
            CK(hipEventElapsedTime(&t, e0, e1));
            if (r) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        // check every column against the reference
        unsigned long long nbad = 0;
        long long got_len = 0;
        CK(hipMemcpy(&got_len, out_len, 8, hipMemcpyDeviceToHost));
        for (int k = 0; k < m; ++k) {
            hipLaunchKernelGGL(ref_scatter_kernel, dim3(ncu * 8), dim3(kBlock), 0, 0, cols[0], cols[k], ntiles, a.c, pre, ref);
            CK(hipMemset(bad, 0, 8));
            hipLaunchKernelGGL(diff_kernel, dim3(ncu * 8), dim3(256), 0, 0, (const unsigned long long*)ref, (const unsigned long long*)outs[k], kept, bad);
            unsigned long long b = 0;
            CK(hipMemcpy(&b, bad, 8, hipMemcpyDeviceToHost));
            nbad += b;
            CK(hipMemset(outs[k], 0xff, out_cap * 8));
        }
        unsigned long long hs[4] = {0, 0, 0, 0};
        CK(hipMemcpy(hs, a.stats, 32, hipMemcpyDeviceToHost));
        const double med = ms[ms.size() / 2];
        const double bytes = (double)n * 8 * m + (double)kept * 8 * m;
        printf("{\"variant\": \"%s\", \"path\": \"%s\", \"prefix\": \"%s\", \"next_tile_prefetch\": %d, \"cols\": %d, \"rows\": %lld, \"kept\": %lld, "
               "\"tile_rows\": %d, \"block_threads\": %d, \"blocks_per_cu\": %d, \"ms_median\": %.4f, \"ms_min\": %.4f, \"alg_bytes\": %.0f, \"TBps\": %.3f, \"frac_of_8TBps\": %.3f, "
               "\"prefix_wait_share_of_loop\": %.3f, \"prefix_wait_us_per_tile\": %.2f, \"scanner_rounds\": %llu, \"scanner_idle_rounds\": %llu, \"mismatches\": %llu, \"len_ok\": %s}\n",
               name, strncmp(name, "dma", 3) == 0 ? "lds-dma" : "registers", lb == 3 ? "scanner wave, counts published one iteration ahead" : lb == 2 ? "scanner wave" : lb ? "lookback" : "precomputed", pf, m, n, kept, tile_rows, block, per_cu, med, ms[0], bytes,
               bytes / med / 1e9, bytes / med / 1e9 / 8.0, hs[1] ? (double)hs[0] / (double)hs[1] : 0.0, (double)hs[0] * 0.01 / (double)(reps + 1) / (double)a.ntiles, hs[2], hs[3], nbad,
               got_len == kept ? "true" : "false");
        fflush(stdout);
    };
    run("reg_pre", 0, 0, 1, k_reg<0, 0, 1>);
    run("reg_pre_pf", 0, 1, 1, k_reg<0, 1, 1>);
    run("pipe_w4g8", 3, 1, 1, k_pipe<1, 4, 8>, 256, 4096);
    run("pipe_w8g4", 3, 1, 1, k_pipe<1, 8, 4>, 512, 4096);
    run("pipe_w4g4", 3, 1, 1, k_pipe<1, 4, 4>, 256, 2048);
    run("pipe_w8g8", 3, 1, 1, k_pipe<1, 8, 8>, 512, 8192);
    run("pipe_w16g4", 3, 1, 1, k_pipe<1, 16, 4>, 1024, 8192);
    run("dma_pre", 0, 0, 1, k_dma<0, 1>);
    run("pipe_4col_w4g8", 3, 1, 4, k_pipe<4, 4, 8>, 256, 4096);
    run("pipe_4col_w8g4", 3, 1, 4, k_pipe<4, 8, 4>, 512, 4096);
    run("pipe_4col_w4g4", 3, 1, 4, k_pipe<4, 4, 4>, 256, 2048);
    run("reg_pre_4col", 0, 0, 4, k_reg<0, 0, 4>);
    run("dma_pre_4col", 0, 0, 4, k_dma<0, 4>);
    return 0;
}
