// rdf_sort.hip — second-generation radix passes of DataFrame::sort (-> arrow::compute::lexsort_to_indices,
// src/dataframe.rs:194-214): ONE kernel per 8-bit digit that reads the (key, row) pairs once and writes them once.
//
// The first generation (sort_hist_kernel -> scan -> sort_scatter_kernel, rdf_kernels.hip) reads the pairs twice per digit
// (histogram, then scatter) with a scan launch in between, and ranks a tile through 16 KB of per-(row, wave) LDS counters and
// seven block barriers.  Here:
//   os_hist_kernel     one read of the keys builds the digit histograms of EVERY pass of the column at once
//   os_bases_kernel    their exclusive scans = where each digit's run starts in every pass
//   os_scatter_kernel  per pass: a block takes the next tile (ticket), ranks it with per-WAVE digit counters (a wave's lanes
//                      that share a digit are found with 8 ballots; the run's first lane bumps the counter — no atomics),
//                      publishes the tile's digit counts, finds its global offsets by DECOUPLED LOOK-BACK over the tiles
//                      before it (each digit's thread walks back until it meets a tile whose inclusive prefix is published),
//                      and writes the locally sorted tile out in digit runs.
// Stable: tiles are ticketed in index order and a tile's offsets are the prefix over lower-numbered tiles only.
//
// Cross-XCD visibility: the per-XCD L2s are not coherent, so a tile's state word is ONE naturally aligned 8-byte
// {sequence | flag | value} granule written and read with agent-scope atomics (write-through / L2-bypassing on gfx950,
// MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup visibility": an 8-byte granule needs no further
// ordering).  The sequence field makes words of earlier passes read as "not published", so the state array is zeroed once
// per sort, not once per pass.
#include "rdf_common.hip.h"

namespace rdfk {

// Phase timers of os_scatter_kernel (build with -DRDF_SORT_TIMERS, run with RDF_DEBUG_SORT=1): reading the clock drains the
// memory counters, so the timed build is ~8 % slower and is not the one that ships.
#ifdef RDF_SORT_TIMERS
#define OS_TIMERS_DECL unsigned long long tph[6] = {0, 0, 0, 0, 0, 0}
#define OS_TICK(c) const unsigned long long c = __builtin_readcyclecounter()
#define OS_TIMERS_ADD do { tph[0] += c1 - c0; tph[1] += c2 - c1; tph[2] += c3 - c2; tph[3] += c4 - c3; tph[4] += c5 - c4; tph[5] += 1; } while (0)
#define OS_TIMERS_FLUSH do { if (a.debug && threadIdx.x == 0) for (int i = 0; i < 6; ++i) atomicAdd(a.debug + i, tph[i]); } while (0)
#else
#define OS_TIMERS_DECL
#define OS_TICK(c)
#define OS_TIMERS_ADD
#define OS_TIMERS_FLUSH
#endif

constexpr int kOsWaves = kBlock / 64;
constexpr int kOsLook = 8;                  // predecessors read per look-back step
constexpr uint64_t kOsValueMask = (1ull << 48) - 1;
constexpr uint64_t kOsLocal = 1ull << 48, kOsInclusive = 2ull << 48;

__device__ __forceinline__ int os_digit(uint64_t key, uint64_t bias, int shift, int mask = 255) { return (int)(((key - bias) >> shift) & (uint64_t)mask); }
// every kernel that maps keys holds the plan's segment table (OsBucket::seg, rdf_sort_map.h) in LDS: as a dependent global load per
// key it cost the histogram and boundary kernels 0.1 - 0.15 ms each per 5e7 keys
__device__ __forceinline__ void os_load_segs(const OsBucket& f, uint2* lds) {      // whole block
    if (!f.seg || f.flat) return;
    for (int i = threadIdx.x; i < kOsSegs; i += blockDim.x) lds[i] = make_uint2(as_global<uint32_t>(f.seg)[2 * i], as_global<uint32_t>(f.seg)[2 * i + 1]);
    __syncthreads();
}
// the value bucket of an f64 key (OsBucket): order-preserving key bits -> the double -> os_map_value
__device__ __forceinline__ uint32_t os_value_bucket(uint64_t key, const OsBucket& f, const uint2* segs) {
    const uint64_t ord = f.flip ? ~key : key;
    const uint64_t b = (ord >> 63) ? (ord ^ 0x8000000000000000ull) : ~ord;
    const uint32_t k = os_map_value(u2d(b), (b >> 63) != 0, f, segs);
    return f.flip ? ((1u << f.bits) - 1) - k : k;
}

// Histograms of all `npass` digits of (key - bias) in one read of the keys (+ the NULL count for the nulls-last pass).
__global__ __launch_bounds__(kBlock) void os_hist_kernel(const OsHistArgs a) {
    __shared__ unsigned int h[9][256];
    __shared__ uint2 segs[kOsSegs];
    os_load_segs(a.fb, segs);
    for (int p = 0; p < 9; ++p) h[p][threadIdx.x] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * kBlock) {
        uint64_t k = __builtin_nontemporal_load(as_global<uint64_t>(a.keys) + i);
        k = a.fb.bits ? (uint64_t)os_value_bucket(k, a.fb, segs) : k - a.bias;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            if (p >= a.npass) break;
            const int d = a.generic ? (int)((k >> a.shift[p]) & (uint64_t)a.mask[p]) : (int)((k >> (8 * p)) & 255);
            // a wave whose rows all share the digit (the upper bytes of a narrow range, dictionary codes): one add, not 64 serialised ones
            const int d0 = __builtin_amdgcn_readfirstlane(d);
            const uint64_t same = __ballot(d == d0);
            if (same == __ballot(1)) { if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(same)) atomicAdd(&h[p][d0], (unsigned)__popcll(same)); }
            else atomicAdd(&h[p][d], 1u);
        }
        if (a.nullflags) { if (as_global<uint8_t>(a.nullflags)[i]) atomicAdd(&h[8][1], 1u); }
    }
    __syncthreads();
    for (int p = 0; p < a.npass; ++p) { const unsigned int c = h[p][threadIdx.x]; if (c) atomicAdd((unsigned long long*)&a.hist[p * 256 + threadIdx.x], (unsigned long long)c); }
    if (a.nullflags && threadIdx.x == 1 && h[8][1]) atomicAdd((unsigned long long*)&a.hist[8 * 256 + 1], (unsigned long long)h[8][1]);
}

// exclusive scan of each pass's 256 counts (block p = pass p; pass 8 = the nulls-last pass: bin 0 = n - nulls)
__global__ __launch_bounds__(256) void os_bases_kernel(int64_t* hist, int64_t n) {
    __shared__ int64_t s[256];
    const int p = blockIdx.x;
    int64_t c = hist[p * 256 + threadIdx.x];
    if (p == 8 && threadIdx.x == 0) c = n - hist[8 * 256 + 1];
    s[threadIdx.x] = c;
    __syncthreads();
    int64_t run = 0;
    for (int i = 0; i < (int)threadIdx.x; ++i) run += s[i];
    hist[p * 256 + threadIdx.x] = run;
}

template <int ITEMS>
__global__ __launch_bounds__(kBlock) void os_scatter_kernel(const OsPassArgs a) {
    constexpr int TILE = kBlock * ITEMS;
    __shared__ uint64_t lkeys[TILE];
    __shared__ uint32_t lidx[TILE];
    __shared__ uint8_t ldig[TILE];
    __shared__ unsigned int whist[kOsWaves][256];      // per-wave digit counts, then the wave's exclusive offset inside the digit's run
    __shared__ unsigned int dbase[256];                // tile-local start of each digit's run
    __shared__ int64_t gbase[256];                     // global start of this tile's part of each digit's run
    __shared__ unsigned int wsum[kOsWaves];
    __shared__ unsigned int thist[256];                // the tile's digit counts, taken before the ranking so that they can be published early
    __shared__ int64_t tile_s;
    __shared__ uint2 segs[kOsSegs];
    os_load_segs(a.fb, segs);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t seq = (uint64_t)a.seq << 50;
    OS_TIMERS_DECL;
    for (;;) {
        OS_TICK(c0);
        if (threadIdx.x == 0) tile_s = (int64_t)atomicAdd((unsigned long long*)a.ticket, 1ull);
#pragma unroll
        for (int w = 0; w < kOsWaves; ++w) whist[w][threadIdx.x] = 0;
        thist[threadIdx.x] = 0;
        __syncthreads();
        const int64_t tile = tile_s;
        if (tile >= a.ntiles) break;
        const int64_t base = tile * TILE;
        const int count = (int)((a.n - base) < (int64_t)TILE ? (a.n - base) : (int64_t)TILE);
        // ---- load (a wave owns ITEMS consecutive rows of 64 items) and rank inside the wave
        uint64_t key[ITEMS];
        uint32_t idx[ITEMS];
        int digit[ITEMS], rank[ITEMS];
        OS_TICK(c1);
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int64_t i = base + (wave * ITEMS + j) * 64 + lane;
            const bool in = i < a.n;
            key[j] = in ? __builtin_nontemporal_load(as_global<uint64_t>(a.keys_in) + i) : 0;
            idx[j] = in ? (a.idx_in ? __builtin_nontemporal_load(as_global<uint32_t>(a.idx_in) + i) : (uint32_t)i) : 0;
        }
        // The tile's digit counts first, with plain LDS adds, and PUBLISHED at once: every tile ticketed after this one waits for
        // them in its look-back, and the ranking below (~6 us of dependent LDS round trips) then runs while this tile in turn
        // waits for the tiles before it, instead of in front of everybody's wait.
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int64_t i = base + (wave * ITEMS + j) * 64 + lane;
            int d = 0;
            if (i < a.n) {
                d = a.nullflags ? (int)as_global<uint8_t>(a.nullflags)[idx[j]]
                                : a.fb.bits ? (int)((os_value_bucket(key[j], a.fb, segs) >> a.shift) & (uint32_t)a.mask) : os_digit(key[j], a.bias, a.shift, a.mask);
                atomicAdd(&thist[d], 1u);
            }
            digit[j] = d;
        }
        __syncthreads();
        const unsigned int total_d = thist[threadIdx.x];
        unsigned long long* st = a.state + tile * 256 + threadIdx.x;
        if (tile > 0) __hip_atomic_store(st, seq | kOsLocal | total_d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ranks inside the wave: the lanes of a row that share my digit (8 ballots); the run's first lane bumps the wave's counter
        // of the digit (plain read + write: the LDS serves a wave's instructions in order, rows are taken in order: stable).
        // (Measured alternatives, both slower on the whole sort: one returning LDS add per row issued back to back for all rows
        // — it keeps four more values per item live, 5.97 against 5.49 ms per 5e7 full-range keys — and 2048-pair tiles at
        // five blocks per CU, 996 against 653 us per pass: more tiles in flight, longer waits in the look-back.)
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int64_t i = base + (wave * ITEMS + j) * 64 + lane;
            const bool in = i < a.n;
            const int d = digit[j];
            uint64_t peers = __ballot(in);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const uint64_t m = __ballot((d >> b) & 1);
                peers &= ((d >> b) & 1) ? m : ~m;
            }
            const int leader = __builtin_ctzll(peers | (1ull << 63));
            unsigned int before = 0;
            if (in && lane == leader) { before = whist[wave][d]; whist[wave][d] = before + (unsigned)__popcll(peers); }
            before = __shfl(before, leader);
            rank[j] = (int)before + __popcll(peers & ((1ull << lane) - 1));
        }
        OS_TICK(c2);
        __syncthreads();
        OS_TICK(c3);
        // ---- thread d: the waves' counts of digit d -> their offsets inside the run
        {
            unsigned int run = 0;
#pragma unroll
            for (int w = 0; w < kOsWaves; ++w) { const unsigned int c = whist[w][threadIdx.x]; whist[w][threadIdx.x] = run; run += c; }
        }
        // look back for what lies before the tile
        unsigned int inc = total_d;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) { const unsigned int o = __shfl_up(inc, dd); if (lane >= dd) inc += o; }
        if (lane == 63) wsum[wave] = inc;
        // kOsLook predecessors are read at once (independent loads) and consumed nearest first
        int64_t excl = 0;
        for (int64_t t = tile - 1; t >= 0;) {
            unsigned long long w[kOsLook];
#pragma unroll
            for (int u = 0; u < kOsLook; ++u)
                w[u] = t - u >= 0 ? __hip_atomic_load(a.state + (t - u) * 256 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (seq | kOsInclusive);
            int used = 0;
            bool done = false, stalled = false;
#pragma unroll
            for (int u = 0; u < kOsLook; ++u) {
                if (done || stalled) continue;
                if ((w[u] >> 50) != (unsigned long long)a.seq) { stalled = true; continue; }   // not published yet in THIS pass
                excl += (int64_t)(w[u] & kOsValueMask);
                ++used;
                if (w[u] & kOsInclusive) done = true;
            }
            if (done) break;
            t -= used;
            if (stalled) __builtin_amdgcn_s_sleep(1);
        }
        __hip_atomic_store(st, seq | kOsInclusive | (unsigned long long)(excl + total_d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        gbase[threadIdx.x] = a.bases[threadIdx.x] + excl;
        OS_TICK(c4);
        __syncthreads();
        unsigned int wb = 0;
        for (int w = 0; w < wave; ++w) wb += wsum[w];
        dbase[threadIdx.x] = wb + inc - total_d;
        __syncthreads();
        // ---- local stable sort by digit into LDS
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int64_t i = base + (wave * ITEMS + j) * 64 + lane;
            if (i < a.n) {
                const int pos = (int)(dbase[digit[j]] + whist[wave][digit[j]]) + rank[j];
                lkeys[pos] = key[j];
                lidx[pos] = idx[j];
                ldig[pos] = (uint8_t)digit[j];
            }
        }
        __syncthreads();
        // ---- write-out: consecutive threads hold consecutive members of a digit run
        if (count == TILE) {
            // a full tile: all ITEMS rounds at once — the digit reads, then the two base reads that depend on them, then the pair
            // reads and the stores (one round at a time each store waited for a chain of three LDS round trips)
            int dd[ITEMS];
            int64_t dst[ITEMS];
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) dd[j] = ldig[threadIdx.x + j * kBlock];
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) dst[j] = gbase[dd[j]] + ((int)(threadIdx.x + j * kBlock) - (int)dbase[dd[j]]);
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const int t = threadIdx.x + j * kBlock;
                __builtin_nontemporal_store(lkeys[t], as_global_mut<uint64_t>(a.keys_out) + dst[j]);
                __builtin_nontemporal_store(lidx[t], as_global_mut<uint32_t>(a.idx_out) + dst[j]);
            }
        } else {
            for (int t = threadIdx.x; t < count; t += kBlock) {
                const int d = ldig[t];
                const int64_t dst = gbase[d] + (t - (int)dbase[d]);
                __builtin_nontemporal_store(lkeys[t], as_global_mut<uint64_t>(a.keys_out) + dst);
                __builtin_nontemporal_store(lidx[t], as_global_mut<uint32_t>(a.idx_out) + dst);
            }
        }
        __syncthreads();
        OS_TICK(c5);
        OS_TIMERS_ADD;
    }
    OS_TIMERS_FLUSH;
}

// ------------------------------------------------------------------------------------------------
// Round 6, second form (os_scatter4_kernel; rdf_set_option("sort_super", K) with K > 1 — NOT the default: measured level with the
// kernel above on 1e9 i64 keys (51.8 - 59.2 against 57.7 - 58.0 ms, box noise) and 8 - 12 % slower on 5e7 f64 keys and two-key sorts,
// profiles/r06_sort_super_tiles_ab.jsonl: the second read of the keys and the count sweep cost what the shorter wait saves, as the
// static-range passes had shown — a pass is bound by the WORK a CU does per tile, 15 us that two resident blocks do not overlap, and
// the waiting sits under the other block's work already; with 16 loads of every lane in flight in the count sweep instead of one
// the whole sort came out 25 - 40 % SLOWER still, r06_sort_super_tiles_unrolled_sweep_ab.jsonl: not the sweep's latency either).
// A block draws a SUPER-TILE of K consecutive tiles.  It first counts the digits of all K tiles (one sweep over their keys: plain LDS adds, nothing
// else), publishes those counts as ONE participant of the look-back, finds its offsets once — and then ranks, sorts and writes its K
// tiles one after the other, thread d carrying digit d's running offset in a register.  What that buys (phase timers of the kernel
// above, 1e9 pairs: 15 us of work and 12 us of waiting per 4096-pair tile, 2 blocks per CU):
//   * the look-back's walk is as long as before — its length is the number of blocks in flight, not the tile size — but it is paid
//     once per K tiles;
//   * K times fewer state words and ticket draws (one ticket counter hands out 43 tiles per us at best: 5.6 ms per pass of 1e9 pairs);
//   * the per-tile part loses its count-and-publish step (a tile's digit counts fall out of the per-wave counters the ranking fills).
// The price is a second read of the keys, 8 of the pass's 24 bytes per pair; the K tiles of a block were read microseconds
// earlier (K x 32 KB per block, 128 MB over the grid at K = 8: within the 256 MB of memory-side cache).
// What the per-range variant of this idea cannot do: hand every range of tiles its own chain of offsets — a range's start inside
// digit d's run is the count of d in ALL earlier ranges in the CURRENT order of the rows, which the one histogram taken before the
// passes does not know (only the global counts survive a permutation).
template <int ITEMS>
__global__ __launch_bounds__(kBlock) void os_scatter4_kernel(const OsPassArgs a) {
    constexpr int TILE = kBlock * ITEMS;
    __shared__ uint64_t lkeys[TILE];
    __shared__ uint32_t lidx[TILE];
    __shared__ uint8_t ldig[TILE];
    __shared__ unsigned int whist[kOsWaves][256];
    __shared__ unsigned int dbase[256];
    __shared__ int64_t gbase[256];
    __shared__ unsigned int wsum[kOsWaves];
    __shared__ unsigned int thist[256];                // the super-tile's digit counts
    __shared__ int64_t tile_s;
    __shared__ uint2 segs[kOsSegs];
    os_load_segs(a.fb, segs);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t seq = (uint64_t)a.seq << 50;
    const int K = a.super_tiles;
    const int64_t nsuper = (a.ntiles + K - 1) / K;
    auto digit_of = [&](uint64_t key, uint32_t row) __attribute__((always_inline)) -> int {
        return a.nullflags ? (int)as_global<uint8_t>(a.nullflags)[row]
                           : a.fb.bits ? (int)((os_value_bucket(key, a.fb, segs) >> a.shift) & (uint32_t)a.mask) : os_digit(key, a.bias, a.shift, a.mask);
    };
    for (;;) {
        if (threadIdx.x == 0) tile_s = (int64_t)atomicAdd((unsigned long long*)a.ticket, 1ull);
        thist[threadIdx.x] = 0;
        __syncthreads();
        const int64_t sup = tile_s;
        if (sup >= nsuper) break;
        const int64_t sbase = sup * K * TILE;
        const int64_t send = (a.n - sbase) < (int64_t)K * TILE ? a.n : sbase + (int64_t)K * TILE;
        // ---- the digit counts of all K tiles, published as one participant
        if (a.nullflags) {
            for (int64_t i = sbase + threadIdx.x; i < send; i += kBlock) {
                const uint32_t row = a.idx_in ? __builtin_nontemporal_load(as_global<uint32_t>(a.idx_in) + i) : (uint32_t)i;
                atomicAdd(&thist[as_global<uint8_t>(a.nullflags)[row]], 1u);
            }
        } else {
            // (two keys per lane and load: TILE is even, so a super-tile starts on a 16-byte boundary of the key array)
            const int64_t pairs_end = sbase + ((send - sbase) & ~(int64_t)1);
            for (int64_t i = sbase + 2 * (int64_t)threadIdx.x; i < pairs_end; i += 2 * kBlock) {
                typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
                const u64x2 kk = *(GlobalPtr<u64x2>)(as_global<uint64_t>(a.keys_in) + i);
                atomicAdd(&thist[digit_of(kk[0], 0)], 1u);
                atomicAdd(&thist[digit_of(kk[1], 0)], 1u);
            }
            if (threadIdx.x == 0 && pairs_end < send) atomicAdd(&thist[digit_of(as_global<uint64_t>(a.keys_in)[pairs_end], 0)], 1u);
        }
        __syncthreads();
        const unsigned int total_d = thist[threadIdx.x];
        unsigned long long* st = a.state + sup * 256 + threadIdx.x;
        if (sup > 0) __hip_atomic_store(st, seq | kOsLocal | total_d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ---- look back over the super-tiles before this one (thread d: digit d)
        int64_t excl = 0;
        for (int64_t t = sup - 1; t >= 0;) {
            unsigned long long w[kOsLook];
#pragma unroll
            for (int u = 0; u < kOsLook; ++u)
                w[u] = t - u >= 0 ? __hip_atomic_load(a.state + (t - u) * 256 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (seq | kOsInclusive);
            int used = 0;
            bool done = false, stalled = false;
#pragma unroll
            for (int u = 0; u < kOsLook; ++u) {
                if (done || stalled) continue;
                if ((w[u] >> 50) != (unsigned long long)a.seq) { stalled = true; continue; }
                excl += (int64_t)(w[u] & kOsValueMask);
                ++used;
                if (w[u] & kOsInclusive) done = true;
            }
            if (done) break;
            t -= used;
            if (stalled) __builtin_amdgcn_s_sleep(1);
        }
        __hip_atomic_store(st, seq | kOsInclusive | (unsigned long long)(excl + total_d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int64_t run_base = a.bases[threadIdx.x] + excl;         // where digit d's rows of the NEXT tile of this super-tile go
        // ---- the K tiles
        for (int k = 0; k < K; ++k) {
            const int64_t base = sbase + (int64_t)k * TILE;
            if (base >= a.n) break;
            const int count = (int)((a.n - base) < (int64_t)TILE ? (a.n - base) : (int64_t)TILE);
#pragma unroll
            for (int w = 0; w < kOsWaves; ++w) whist[w][threadIdx.x] = 0;
            uint64_t key[ITEMS];
            uint32_t idx[ITEMS];
            int digit[ITEMS], rank[ITEMS];
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const int64_t i = base + (wave * ITEMS + j) * 64 + lane;
                const bool in = i < a.n;
                key[j] = in ? as_global<uint64_t>(a.keys_in)[i] : 0;          // (read a few microseconds ago by the count sweep: not a streaming load)
                idx[j] = in ? (a.idx_in ? __builtin_nontemporal_load(as_global<uint32_t>(a.idx_in) + i) : (uint32_t)i) : 0;
            }
            __syncthreads();               // whist is clear
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const int64_t i = base + (wave * ITEMS + j) * 64 + lane;
                const bool in = i < a.n;
                const int d = in ? digit_of(key[j], idx[j]) : 0;
                digit[j] = d;
                uint64_t peers = __ballot(in);
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const uint64_t m = __ballot((d >> b) & 1);
                    peers &= ((d >> b) & 1) ? m : ~m;
                }
                const int leader = __builtin_ctzll(peers | (1ull << 63));
                unsigned int before = 0;
                if (in && lane == leader) { before = whist[wave][d]; whist[wave][d] = before + (unsigned)__popcll(peers); }
                before = __shfl(before, leader);
                rank[j] = (int)before + __popcll(peers & ((1ull << lane) - 1));
            }
            __syncthreads();
            // thread d: the waves' counts of digit d -> their offsets inside the run; their sum is the tile's count of d
            unsigned int tile_d = 0;
#pragma unroll
            for (int w = 0; w < kOsWaves; ++w) { const unsigned int c = whist[w][threadIdx.x]; whist[w][threadIdx.x] = tile_d; tile_d += c; }
            unsigned int inc = tile_d;
#pragma unroll
            for (int dd = 1; dd < 64; dd <<= 1) { const unsigned int o = __shfl_up(inc, dd); if (lane >= dd) inc += o; }
            if (lane == 63) wsum[wave] = inc;
            gbase[threadIdx.x] = run_base;
            run_base += tile_d;
            __syncthreads();
            unsigned int wb = 0;
            for (int w = 0; w < wave; ++w) wb += wsum[w];
            dbase[threadIdx.x] = wb + inc - tile_d;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const int64_t i = base + (wave * ITEMS + j) * 64 + lane;
                if (i < a.n) {
                    const int pos = (int)(dbase[digit[j]] + whist[wave][digit[j]]) + rank[j];
                    lkeys[pos] = key[j];
                    lidx[pos] = idx[j];
                    ldig[pos] = (uint8_t)digit[j];
                }
            }
            __syncthreads();
            for (int t = threadIdx.x; t < count; t += kBlock) {
                const int d = ldig[t];
                const int64_t dst = gbase[d] + (t - (int)dbase[d]);
                __builtin_nontemporal_store(lkeys[t], as_global_mut<uint64_t>(a.keys_out) + dst);
                __builtin_nontemporal_store(lidx[t], as_global_mut<uint32_t>(a.idx_out) + dst);
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Round 6: the same pass with the waiting taken out (os_scatter3_kernel; rdf_set_option("sort_pipe", 0) brings the kernel above
// back for A/B).  By the phase timers a tile above spends 12 of its 27 us waiting in the look-back: its offsets need the counts of
// EVERY tile ticketed before it, those tiles are being counted at the same moment, and a count is one memory round trip away
// from its reader.  The compaction kernel of rdf_bfilter.hip met the same wall and got past it with two changes, taken over here:
//   * the counts of tile n + 1 are published a whole iteration BEFORE its offsets are asked for: a block holds two tiles in
//     registers — the current one (counted last iteration) and the next one (loads in flight while the current one is ranked,
//     sorted in LDS and written out);
//   * nobody walks back over the tiles in flight (with counts out that early a flat look-back would cross ~1000 tiles): the
//     first kOsScanBlocks blocks of the grid are SCANNERS.  A scanner wave owns 8 of the 256 digits, reads the count words of
//     256 tiles x 8 digits per round, adds them up in tile order and writes the exclusive prefixes back over the counts;
//     thread d of a tile polls its own word.
// Tiles are handed out by up to 64 ticket counters (one counter serialises its draws at ~23 ns each: 5.6 ms per pass of 1e9 keys).


// A scanner wave owns DPW = 8 digits.  Lanes (g, d) = (lane & 7, lane >> 3): digit d0 + d of the 32 CONSECUTIVE tiles 32 g .. 32 g + 31
// of a 256-tile window — a lane adds up its own run serially, the eight runs of a digit are joined by one scan over the lanes of the
// digit (three DPP row shifts), and the first tile whose words are not all out is a wave minimum.  (First form: 16 digits per wave,
// the four tiles of a row in a quad, a cross-lane step per row: 1100 wave instructions per 128 tiles — the scanners, not the memory,
// were what every tile waited for: 10 us per round, 13 tiles per us where the pass needs 35.)
constexpr int kOsScanDigits = 8;                    // digits per scanner wave
constexpr int kOsScanBlocks = 256 / kOsScanDigits / kOsWaves;      // 8 blocks = 32 scanner waves
__device__ __forceinline__ void os_scanner_wave(const OsPassArgs& a, int d0) {
    constexpr int K = 32, G = 64 / kOsScanDigits, W = G * K;      // 8 groups x 32 tiles
    const int lane = threadIdx.x & 63, g = lane & (G - 1), d = lane >> 3;
    const unsigned long long seq = (unsigned long long)a.seq;
    unsigned long long running = 0;             // rows of digit d0 + d in front of tile `cur`
    int64_t cur = 0;
    unsigned long long w[K], wn[K];
    // (no end-of-array tests: the state array carries kOsStatePadTiles tiles of zeroed words behind the last tile — words that are
    // never published; with a test per load the loop spilled its 64-bit tile numbers and ran 15 us per round)
    auto fetch = [&](int64_t at, unsigned long long (&x)[K]) __attribute__((always_inline)) {
        const unsigned long long* p = a.state + (at + K * g) * 256 + d0 + d;
#pragma unroll
        for (int k = 0; k < K; ++k) x[k] = __hip_atomic_load(p + k * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    unsigned long long n_rounds = 0, n_idle = 0, n_part = 0;
    fetch(cur, w);
    while (cur < a.ntiles) {
        ++n_rounds;
        // this lane's first word that is not out yet, as a tile of the window; the window's: the minimum over the wave
        int bad = W;
#pragma unroll
        for (int k = K - 1; k >= 0; --k) if (!((w[k] >> 50) == seq && (w[k] & kOsLocal))) bad = K * g + k;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) { const int o = __shfl_xor(bad, m); bad = o < bad ? o : bad; }
        const int f = __builtin_amdgcn_readfirstlane(bad);          // (tiles past the end are never published: f stops there)
        if (f == 0) { ++n_idle; __builtin_amdgcn_s_sleep(2); fetch(cur, w); continue; }
        const bool whole = f == W;
        if (!whole) ++n_part;
        if (whole) fetch(cur + W, wn);                              // the next window travels while this one is written
        // counts of my run that take part (tiles below f)
        const int mine_n = f - K * g;                               // how many of my 32 tiles
        unsigned int tot = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) tot += k < mine_n ? (unsigned int)(w[k] & kOsValueMask) : 0u;
        // the runs of the digit in front of mine: lanes g' < g of the same digit = the lanes before me in my row of 8
        unsigned int inc = tot;
        { const unsigned int o = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xf, 0xf, false); if (g >= 1) inc += o; }
        { const unsigned int o = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xf, 0xf, false); if (g >= 2) inc += o; }
        { const unsigned int o = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xf, 0xf, false); if (g >= 4) inc += o; }
        unsigned long long at = running + (inc - tot);
        unsigned long long* q = a.state + (cur + K * g) * 256 + d0 + d;
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (k < mine_n) {
                __hip_atomic_store(q + k * 256, (seq << 50) | kOsInclusive | at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                at += w[k] & kOsValueMask;
            }
        running += (unsigned int)__shfl((int)inc, (lane & ~(G - 1)) | (G - 1));       // the digit's total over the window: its last lane's inclusive sum
        cur += f;
        if (cur >= a.ntiles) break;
        if (whole) {
#pragma unroll
            for (int k = 0; k < K; ++k) w[k] = wn[k];
        } else fetch(cur, w);
    }
    if (a.debug && lane == 0 && d0 == 0) { a.debug[2] = n_rounds; a.debug[3] = n_idle; a.debug[4] = n_part; }
}

template <int ITEMS>
__global__ __launch_bounds__(kBlock, 2) void os_scatter3_kernel(const OsPassArgs a) {     // (two blocks per CU: the LDS's limit)
    constexpr int TILE = kBlock * ITEMS;
    __shared__ uint64_t lkeys[TILE];
    __shared__ uint32_t lidx[TILE];
    __shared__ uint8_t ldig[TILE];
    __shared__ unsigned int whist[kOsWaves][256];
    __shared__ unsigned int dbase[256];
    __shared__ int64_t gbase[256];
    __shared__ unsigned int wsum[kOsWaves];
    __shared__ unsigned int thist[2][256];              // the counts of the tile being published, and of the one published before
    __shared__ int64_t tile_s[2];
    __shared__ uint2 segs[kOsSegs];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (blockIdx.x < kOsScanBlocks) { os_scanner_wave(a, (int)(blockIdx.x * kOsWaves + wave) * kOsScanDigits); return; }
    os_load_segs(a.fb, segs);
    const unsigned long long seq = (unsigned long long)a.seq << 50;
    const int ctr = (int)((blockIdx.x - kOsScanBlocks) % (unsigned)a.nclass);
    auto draw = [&]() __attribute__((always_inline)) -> int64_t { return (int64_t)atomicAdd(a.class_tickets + ctr * 32, 1u) * a.nclass + ctr; };
    struct Regs { uint64_t key[ITEMS]; uint32_t idx[ITEMS]; };
    // A tile's rows are addressed as (tile's first row: wave-uniform) + (32-bit place in the tile): the base goes in scalar registers
    // and the loads take a 32-bit lane offset — with 64-bit row numbers per item the two tiles in flight did not fit 256 registers.
    const int r0 = wave * ITEMS * 64 + lane;              // place of this lane's item 0 in a tile; item j: r0 + 64 j
    auto rows_of = [&](int64_t tile) __attribute__((always_inline)) -> int {
        const int64_t left = a.n - tile * TILE;
        return (int)(left < (int64_t)TILE ? left : (int64_t)TILE);
    };
    auto load = [&](int64_t tile, Regs& r) __attribute__((always_inline)) {
        const int64_t base = (int64_t)uniform64((uint64_t)(tile * TILE));
        const int count = __builtin_amdgcn_readfirstlane(rows_of(tile));
        const GlobalPtr<uint64_t> kp = as_global<uint64_t>(a.keys_in) + base;
        const GlobalPtr<uint32_t> ip = as_global<uint32_t>(a.idx_in) + base;
        const bool has_idx = a.idx_in != nullptr;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int rel = r0 + 64 * j;
            const bool in = rel < count;
            r.key[j] = in ? __builtin_nontemporal_load(kp + rel) : 0;
            r.idx[j] = in ? (has_idx ? __builtin_nontemporal_load(ip + rel) : (uint32_t)base + (uint32_t)rel) : 0;
        }
    };
    auto digit_of = [&](const Regs& r, int j) __attribute__((always_inline)) -> int {
        return a.nullflags ? (int)as_global<uint8_t>(a.nullflags)[r.idx[j]]
                           : a.fb.bits ? (int)((os_value_bucket(r.key[j], a.fb, segs) >> a.shift) & (uint32_t)a.mask) : os_digit(r.key[j], a.bias, a.shift, a.mask);
    };
    // the tile's digit counts, published (thread d: digit d); thist[par ^ 1] is cleared for the tile after this one
    auto count_publish = [&](const Regs& r, int64_t tile, int par) __attribute__((always_inline)) -> unsigned int {
        const int count = __builtin_amdgcn_readfirstlane(rows_of(tile));
        thist[par ^ 1][threadIdx.x] = 0;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j)
            if (r0 + 64 * j < count) atomicAdd(&thist[par][digit_of(r, j)], 1u);
        __syncthreads();
        const unsigned int total_d = thist[par][threadIdx.x];
        __hip_atomic_store(a.state + tile * 256 + threadIdx.x, seq | kOsLocal | total_d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return total_d;
    };
    thist[0][threadIdx.x] = 0;
    if (threadIdx.x == 0) { tile_s[0] = draw(); tile_s[1] = draw(); }
    __syncthreads();
    int64_t T = (int64_t)uniform64((uint64_t)tile_s[0]), Tn = (int64_t)uniform64((uint64_t)tile_s[1]);
    __syncthreads();
    if (T >= a.ntiles) return;
    Regs A, B;
    load(T, A);
    if (Tn < a.ntiles) load(Tn, B);
    unsigned int total_c = count_publish(A, T, 0), total_n = 0;
    int par = 1;
    unsigned long long t_poll = 0, t_loop = wall_clock64(), n_tiles = 0;
    auto step = [&](Regs& X, Regs& Y) __attribute__((always_inline)) -> bool {
        int64_t drawn = 0;
        if (threadIdx.x == 0) drawn = draw();
#pragma unroll
        for (int w = 0; w < kOsWaves; ++w) whist[w][threadIdx.x] = 0;
        // a. the next tile: count, publish (its offsets are asked for one iteration from now)
        if (Tn < a.ntiles) total_n = count_publish(Y, Tn, par);
        else __syncthreads();
        par ^= 1;
        const int count = __builtin_amdgcn_readfirstlane(rows_of(T));
        // b. ranks inside the wave (as os_scatter_kernel)
        uint32_t dr[ITEMS];       // digit | rank inside the wave << 8 (a tile holds 4096 pairs)
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const bool in = r0 + 64 * j < count;
            const int d = in ? digit_of(X, j) : 0;
            uint64_t peers = __ballot(in);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const uint64_t m = __ballot((d >> b) & 1);
                peers &= ((d >> b) & 1) ? m : ~m;
            }
            const int leader = __builtin_ctzll(peers | (1ull << 63));
            unsigned int before = 0;
            if (in && lane == leader) { before = whist[wave][d]; whist[wave][d] = before + (unsigned)__popcll(peers); }
            before = __shfl(before, leader);
            dr[j] = (uint32_t)d | (((uint32_t)before + (uint32_t)__popcll(peers & ((1ull << lane) - 1))) << 8);
        }
        __syncthreads();
        {
            unsigned int run = 0;
#pragma unroll
            for (int w = 0; w < kOsWaves; ++w) { const unsigned int c = whist[w][threadIdx.x]; whist[w][threadIdx.x] = run; run += c; }
        }
        unsigned int inc = total_c;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) { const unsigned int o = __shfl_up(inc, dd); if (lane >= dd) inc += o; }
        if (lane == 63) wsum[wave] = inc;
        // c. rows of digit d in front of the tile: the scanner's word
        {
            // (a scanner wave writes 16 digits of a tile with one store: while the word is not there only one thread in 16 asks again —
            // 512 blocks x 256 polling threads would be a terabyte per second of 8-byte reads in front of the scanners' own)
            unsigned long long w;
            const unsigned long long tp0 = a.debug ? wall_clock64() : 0;
            for (;;) {
                w = __hip_atomic_load(a.state + T * 256 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool ok = (w >> 50) == (unsigned long long)a.seq && (w & kOsInclusive);
                if (__ballot(!ok) == 0) break;
                // lanes 15, 31, 47, 63 keep asking for their groups; the others wait for them
                for (;;) {
                    bool gok = true;
                    if ((lane & 15) == 15) {
                        const unsigned long long v = __hip_atomic_load(a.state + T * 256 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        gok = (v >> 50) == (unsigned long long)a.seq && (v & kOsInclusive);
                    }
                    if (__ballot(!gok) == 0) break;
                    __builtin_amdgcn_s_sleep(4);
                }
            }
            gbase[threadIdx.x] = a.bases[threadIdx.x] + (int64_t)(w & kOsValueMask);
            if (a.debug) { t_poll += wall_clock64() - tp0; ++n_tiles; }
        }
        if (threadIdx.x == 0) tile_s[par] = drawn;
        __syncthreads();
        unsigned int wb = 0;
        for (int w = 0; w < wave; ++w) wb += wsum[w];
        dbase[threadIdx.x] = wb + inc - total_c;
        const int64_t Tnn = (int64_t)uniform64((uint64_t)tile_s[par]);
        __syncthreads();
        // d. local stable sort by digit into LDS
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            if (r0 + 64 * j < count) {
                const int dg = (int)(dr[j] & 255u);
                const int pos = (int)(dbase[dg] + whist[wave][dg]) + (int)(dr[j] >> 8);
                lkeys[pos] = X.key[j];
                lidx[pos] = X.idx[j];
                ldig[pos] = (uint8_t)dg;
            }
        }
        __syncthreads();
        // the tile after the next one: into the registers the current tile has left; in flight during the write-out and the next count
        if (Tnn < a.ntiles) load(Tnn, X);
        for (int t = threadIdx.x; t < count; t += kBlock) {
            const int d = ldig[t];
            const int64_t dst = gbase[d] + (t - (int)dbase[d]);
            __builtin_nontemporal_store(lkeys[t], as_global_mut<uint64_t>(a.keys_out) + dst);
            __builtin_nontemporal_store(lidx[t], as_global_mut<uint32_t>(a.idx_out) + dst);
        }
        __syncthreads();
        T = Tn; Tn = Tnn; total_c = total_n;
        return T < a.ntiles;
    };
    for (;;) {
        if (!step(A, B)) break;
        if (!step(B, A)) break;
    }
    if (a.debug && threadIdx.x == 0) { atomicAdd(a.debug + 0, t_poll); atomicAdd(a.debug + 1, wall_clock64() - t_loop); atomicAdd(a.debug + 5, n_tiles); }
}

// ------------------------------------------------------------------------------------------------
// Static ranges: block b owns the contiguous tiles [b * tpb, (b + 1) * tpb).  A count pass gives every block its digit counts
// (sr_hist_kernel, keys only: 8 bytes per row), one scan turns them into where each block's part of each digit's run starts,
// and the scatter walks its tiles in order carrying the running offsets in LDS: no block ever waits for another one; the next
// tile's pairs are loaded while the current tile is ranked, sorted and written (registers double-buffered by unrolling the
// tile loop by two).  Kept as the A/B partner of the look-back kernel (rdf_set_option("sort_gen", 2)): per pass of 5e7 pairs
// it measured 708 + 215 us (scatter + count) against 653 us — double buffering the tile costs the kernel its second block per
// CU (256 VGPRs), and the count pass re-reads the keys.  What the look-back kernel pays instead is waiting: a tile cannot
// resolve its offsets before EVERY tile ticketed before it has published its counts (12 us of a 27 us tile by the phase
// timers, RDF_DEBUG_SORT=1, whatever the look-back width).
__global__ __launch_bounds__(kBlock) void sr_hist_kernel(const OsPassArgs a, int64_t* hist) {
    __shared__ unsigned int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    constexpr int TILE = kBlock * kOsItems;
    const int64_t tpb = (a.ntiles + gridDim.x - 1) / gridDim.x;
    const int64_t t0 = (int64_t)blockIdx.x * tpb, t1 = t0 + tpb < a.ntiles ? t0 + tpb : a.ntiles;
    const int64_t lo = t0 * TILE, hi = t1 * TILE < a.n ? t1 * TILE : a.n;
    for (int64_t i = lo + threadIdx.x; i < hi; i += kBlock) {
        int d;
        if (a.nullflags) d = as_global<uint8_t>(a.nullflags)[a.idx_in ? (int64_t)as_global<uint32_t>(a.idx_in)[i] : i];
        else d = os_digit(__builtin_nontemporal_load(as_global<uint64_t>(a.keys_in) + i), a.bias, a.shift);
        const int d0 = __builtin_amdgcn_readfirstlane(d);
        const uint64_t same = __ballot(d == d0);
        if (same == __ballot(1)) { if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(same)) atomicAdd(&h[d0], (unsigned)__popcll(same)); }
        else atomicAdd(&h[d], 1u);
    }
    __syncthreads();
    hist[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = h[threadIdx.x];
}

template <int ITEMS>
struct SrTile { uint64_t key[ITEMS]; uint32_t idx[ITEMS]; };

template <int ITEMS>
__device__ __forceinline__ void sr_load(const OsPassArgs& a, int64_t tile, int wave, int lane, SrTile<ITEMS>& r) {
    const int64_t base = tile * (kBlock * ITEMS);
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int64_t i = base + (wave * ITEMS + j) * 64 + lane;
        const bool in = i < a.n;
        r.key[j] = in ? __builtin_nontemporal_load(as_global<uint64_t>(a.keys_in) + i) : 0;
        r.idx[j] = in ? (a.idx_in ? __builtin_nontemporal_load(as_global<uint32_t>(a.idx_in) + i) : (uint32_t)i) : 0;
    }
}

template <int ITEMS>
struct SrLds {
    uint64_t lkeys[kBlock * ITEMS];
    uint32_t lidx[kBlock * ITEMS];
    uint8_t ldig[kBlock * ITEMS];
    unsigned int whist[kOsWaves][256];
    unsigned int dbase[256];
    int64_t gbase[256];
    unsigned int wsum[kOsWaves];
};

// one tile: rank (as os_scatter_kernel), local sort, write-out at the block's running offsets
template <int ITEMS>
__device__ __forceinline__ void sr_tile(const OsPassArgs& a, SrLds<ITEMS>& L, int64_t tile, int wave, int lane, const SrTile<ITEMS>& r) {
    constexpr int TILE = kBlock * ITEMS;
    const int64_t base = tile * TILE;
    const int count = (int)((a.n - base) < (int64_t)TILE ? (a.n - base) : (int64_t)TILE);
#pragma unroll
    for (int w = 0; w < kOsWaves; ++w) L.whist[w][threadIdx.x] = 0;
    __syncthreads();
    // per item ONE register: digit (8 bits) | first peer lane (6) | peers below me (6) | peers (7)
    unsigned int pk[ITEMS], before[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int64_t i = base + (wave * ITEMS + j) * 64 + lane;
        const bool in = i < a.n;
        int d = 0;
        if (in) d = a.nullflags ? (int)as_global<uint8_t>(a.nullflags)[r.idx[j]] : os_digit(r.key[j], a.bias, a.shift);
        uint64_t peers = __ballot(in);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint64_t m = __ballot((d >> b) & 1);
            peers &= ((d >> b) & 1) ? m : ~m;
        }
        const unsigned int leader = (unsigned)__builtin_ctzll(peers | (1ull << 63));
        pk[j] = (unsigned)d | (leader << 8) | ((unsigned)__popcll(peers & ((1ull << lane) - 1)) << 14) | ((unsigned)__popcll(peers) << 20);
        __builtin_amdgcn_sched_barrier(0);     // one row's ballots at a time: sixteen rows' masks in flight spill hundreds of scalar registers
    }
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int64_t i = base + (wave * ITEMS + j) * 64 + lane;
        unsigned int old = 0;
        if (i < a.n && (unsigned)lane == ((pk[j] >> 8) & 63)) old = atomicAdd(&L.whist[wave][pk[j] & 255], pk[j] >> 20);
        before[j] = old;
    }
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) before[j] = __shfl(before[j], (int)((pk[j] >> 8) & 63)) + ((pk[j] >> 14) & 63);   // rank inside the wave's run of the digit
    __syncthreads();
    unsigned int total_d = 0;
#pragma unroll
    for (int w = 0; w < kOsWaves; ++w) { const unsigned int c = L.whist[w][threadIdx.x]; L.whist[w][threadIdx.x] = total_d; total_d += c; }
    unsigned int inc = total_d;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) { const unsigned int o = __shfl_up(inc, dd); if (lane >= dd) inc += o; }
    if (lane == 63) L.wsum[wave] = inc;
    __syncthreads();
    unsigned int wb = 0;
    for (int w = 0; w < wave; ++w) wb += L.wsum[w];
    L.dbase[threadIdx.x] = wb + inc - total_d;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int64_t i = base + (wave * ITEMS + j) * 64 + lane;
        if (i < a.n) {
            const int d = (int)(pk[j] & 255);
            const int pos = (int)(L.dbase[d] + L.whist[wave][d] + before[j]);
            L.lkeys[pos] = r.key[j];
            L.lidx[pos] = r.idx[j];
            L.ldig[pos] = (uint8_t)d;
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < count; t += kBlock) {
        const int d = L.ldig[t];
        const int64_t dst = L.gbase[d] + (t - (int)L.dbase[d]);
        __builtin_nontemporal_store(L.lkeys[t], as_global_mut<uint64_t>(a.keys_out) + dst);
        __builtin_nontemporal_store(L.lidx[t], as_global_mut<uint32_t>(a.idx_out) + dst);
    }
    __syncthreads();
    L.gbase[threadIdx.x] += total_d;      // the block's next tile continues each digit's run
}

template <int ITEMS>
__global__ __launch_bounds__(kBlock) void sr_scatter_kernel(const OsPassArgs a, const int64_t* hist) {
    __shared__ SrLds<ITEMS> L;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    L.gbase[threadIdx.x] = hist[(int64_t)threadIdx.x * gridDim.x + blockIdx.x];
    const int64_t tpb = (a.ntiles + gridDim.x - 1) / gridDim.x;
    const int64_t t0 = (int64_t)blockIdx.x * tpb, t1 = t0 + tpb < a.ntiles ? t0 + tpb : a.ntiles;
    if (t0 >= t1) return;
    SrTile<ITEMS> ra, rb;
    sr_load<ITEMS>(a, t0, wave, lane, ra);
    for (int64_t tile = t0; tile < t1; tile += 2) {
        if (tile + 1 < t1) sr_load<ITEMS>(a, tile + 1, wave, lane, rb);
        sr_tile<ITEMS>(a, L, tile, wave, lane, ra);
        if (tile + 1 >= t1) break;
        if (tile + 2 < t1) sr_load<ITEMS>(a, tile + 2, wave, lane, ra);
        sr_tile<ITEMS>(a, L, tile + 1, wave, lane, rb);
    }
}

int sr_grid(int64_t ntiles) {
    const int64_t lim = (int64_t)(eval_grid_limit() / 8) * 2;      // ~59 KB of LDS per block: two blocks per CU
    const int64_t g = ntiles < lim ? ntiles : lim;
    return g < 1 ? 1 : (int)g;
}
hipError_t launch_sr_hist(const OsPassArgs& a, int64_t* hist, hipStream_t s) {
    hipLaunchKernelGGL(sr_hist_kernel, dim3(sr_grid(a.ntiles)), dim3(kBlock), 0, s, a, hist);
    return hipGetLastError();
}
hipError_t launch_sr_scatter(const OsPassArgs& a, const int64_t* hist, hipStream_t s) {
    hipLaunchKernelGGL((sr_scatter_kernel<kOsItems>), dim3(sr_grid(a.ntiles)), dim3(kBlock), 0, s, a, hist);
    return hipGetLastError();
}

hipError_t launch_os_hist(const OsHistArgs& a, hipStream_t s) {
    int64_t grid = (a.n + (int64_t)kBlock * 16 - 1) / ((int64_t)kBlock * 16);
    if (grid > eval_grid_limit()) grid = eval_grid_limit();
    if (grid <= 0) return hipSuccess;
    hipLaunchKernelGGL(os_hist_kernel, dim3((unsigned)grid), dim3(kBlock), 0, s, a);
    hipLaunchKernelGGL(os_bases_kernel, dim3(9), dim3(256), 0, s, a.hist, a.n);
    return hipGetLastError();
}
int os_tile_items() { return kBlock * kOsItems; }
hipError_t launch_os_scatter(const OsPassArgs& a, hipStream_t s) {
    // every block must be RESIDENT (a ticketed tile waits for its predecessors): 3 blocks of ~57 KB LDS per CU
    int64_t grid = (int64_t)(eval_grid_limit() / 8) * (kOsItems >= 16 ? 2 : 5);
    if (grid > a.ntiles) grid = a.ntiles;
    if (grid <= 0) return hipSuccess;
    OsPassArgs b = a;
    if (b.mask == 0) b.mask = 255;
    if (b.class_tickets) {      // round 6: counts published an iteration ahead, offsets from scanner blocks
        b.nclass = (int32_t)(grid < 64 ? grid : 64);
        hipLaunchKernelGGL((os_scatter3_kernel<kOsItems>), dim3((unsigned)(grid + kOsScanBlocks)), dim3(kBlock), 0, s, b);
        return hipGetLastError();
    }
    if (b.super_tiles > 1) {    // round 6: K tiles per ticket — counted together, one look-back, then ranked and written one by one
        const int64_t nsuper = (b.ntiles + b.super_tiles - 1) / b.super_tiles;
        if (grid > nsuper) grid = nsuper;
        hipLaunchKernelGGL((os_scatter4_kernel<kOsItems>), dim3((unsigned)grid), dim3(kBlock), 0, s, b);
        return hipGetLastError();
    }
    hipLaunchKernelGGL((os_scatter_kernel<kOsItems>), dim3((unsigned)grid), dim3(kBlock), 0, s, b);
    return hipGetLastError();
}
// tiles per ticket for a pass over `ntiles` tiles: as many as `max_k` while every resident block still gets several super-tiles
int os_super_tiles(int64_t ntiles, int max_k) {
    const int64_t grid = (int64_t)(eval_grid_limit() / 8) * (kOsItems >= 16 ? 2 : 5);
    int64_t k = ntiles / (grid * 4);
    if (k > max_k) k = max_k;
    return k < 2 ? 1 : (int)k;
}

// ------------------------------------------------------------------------------------------------
// The finish of the most-significant-digits-first order (see OsLocalArgs): bucket boundaries, then one block per bucket.

// bstart[b] = first row whose bucket is >= b (rows are sorted by bucket); thread i writes the entries of the buckets that END in
// front of row i — every entry of bstart[0 .. nbuckets] exactly once, empty buckets included
__global__ __launch_bounds__(kBlock) void os_bounds_kernel(const uint64_t* keys, int64_t n, uint64_t bias, int rbits, int nbuckets, const OsBucket fb, uint32_t* bstart) {
    __shared__ uint2 segs[kOsSegs];
    os_load_segs(fb, segs);
    auto bucket = [&](int64_t i) -> int64_t {
        const uint64_t k = as_global<uint64_t>(keys)[i];
        return fb.bits ? (int64_t)os_value_bucket(k, fb, segs) : (int64_t)((k - bias) >> rbits);
    };
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i <= n; i += (int64_t)gridDim.x * kBlock) {
        const int64_t cur = i < n ? bucket(i) : (int64_t)nbuckets;
        const int64_t prev = i > 0 ? bucket(i - 1) : -1;
        for (int64_t b = prev + 1; b <= cur; ++b) bstart[b] = (uint32_t)i;
    }
}
__global__ __launch_bounds__(kBlock) void os_bucket_max_kernel(const uint32_t* bstart, int nbuckets, unsigned int* out) {
    unsigned int m = 0;
    for (int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x; b < nbuckets; b += (int64_t)gridDim.x * kBlock) {
        const unsigned int len = bstart[b + 1] - bstart[b];
        m = len > m ? len : m;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const unsigned int o = (unsigned int)__shfl_xor((int)m, d); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// Bitonic network over P 64-bit words in LDS, ONE WAVE per bucket.  A word is (the key's low rbits) << 12 | (the row's place in
// the bucket): the places make the words distinct, so the order of equal keys is the order the rows came in — the sort is
// stable, which the column-by-column lexicographic order and the NULLs-last pass rely on.
// A lane takes 2^MB words whose indices differ in MB consecutive stride bits and runs the MB sub-stages of those strides in
// registers: one LDS round trip per four sub-stages (18 for 1024 words instead of 55 — the first version, one sub-stage per
// round trip and block barrier, was bound by LDS bandwidth: 1.04 ms per 5e7 rows), no block barrier at all.  Words are padded by
// one per 16 (a lane's 16 consecutive words would otherwise put all 64 lanes on the same banks).
__device__ __forceinline__ int os_pad(int i) { return i + (i >> 4); }
template <int MB>
__device__ __forceinline__ void os_chunk(uint64_t* s, int P, int k, int jl_log) {
    constexpr int E = 1 << MB;
    for (int q = threadIdx.x; q < (P >> MB); q += 64) {
        const int i = ((q >> jl_log) << (jl_log + MB)) | (q & ((1 << jl_log) - 1));
        uint64_t v[E];
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = s[os_pad(i | (e << jl_log))];
        const bool asc = (i & k) == 0;
#pragma unroll
        for (int b = MB - 1; b >= 0; --b) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if (e & (1 << b)) continue;
                const uint64_t x = v[e], y = v[e | (1 << b)];
                const bool sw = (x > y) == asc;
                v[e] = sw ? y : x;
                v[e | (1 << b)] = sw ? x : y;
            }
        }
#pragma unroll
        for (int e = 0; e < E; ++e) s[os_pad(i | (e << jl_log))] = v[e];
    }
}
// ... over (whole key, place) pairs: value buckets (f64 keys) — the rows of a bucket share no key bits
template <int MB>
__device__ __forceinline__ void os_chunk_wide(uint64_t* s, uint32_t* sp, int P, int k, int jl_log) {
    constexpr int E = 1 << MB;
    for (int q = threadIdx.x; q < (P >> MB); q += 64) {
        const int i = ((q >> jl_log) << (jl_log + MB)) | (q & ((1 << jl_log) - 1));
        uint64_t v[E];
        uint32_t w[E];
#pragma unroll
        for (int e = 0; e < E; ++e) { v[e] = s[os_pad(i | (e << jl_log))]; w[e] = sp[os_pad(i | (e << jl_log))]; }
        const bool asc = (i & k) == 0;
#pragma unroll
        for (int b = MB - 1; b >= 0; --b) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if (e & (1 << b)) continue;
                const int f = e | (1 << b);
                const bool gt = v[e] > v[f] || (v[e] == v[f] && w[e] > w[f]);
                const bool sw = gt == asc;
                const uint64_t x = v[e], y = v[f];
                const uint32_t px = w[e], py = w[f];
                v[e] = sw ? y : x; v[f] = sw ? x : y;
                w[e] = sw ? py : px; w[f] = sw ? px : py;
            }
        }
#pragma unroll
        for (int e = 0; e < E; ++e) { s[os_pad(i | (e << jl_log))] = v[e]; sp[os_pad(i | (e << jl_log))] = w[e]; }
    }
}
__global__ __launch_bounds__(64) void os_local_wide_kernel(const OsLocalArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint64_t s[];
    const int64_t bucket = blockIdx.x;
    const uint32_t start = a.bstart[bucket], len = a.bstart[bucket + 1] - start;
    if (len <= a.len_lo || len > a.len_hi) return;
    if (len == 1) {
        if (threadIdx.x == 0) {
            as_global_mut<uint64_t>(a.keys_out)[start] = as_global<uint64_t>(a.keys_in)[start];
            as_global_mut<uint32_t>(a.idx_out)[start] = a.idx_in ? as_global<uint32_t>(a.idx_in)[start] : start;
        }
        return;
    }
    int P = 2, plog = 1;
    while (P < (int)len) { P <<= 1; ++plog; }
    uint32_t* sp = (uint32_t*)(s + os_pad(a.lds_items) + 1);
    for (int i = threadIdx.x; i < P; i += 64) {
        // padding sorts behind every row: the largest key with a place no row has
        s[os_pad(i)] = i < (int)len ? __builtin_nontemporal_load(as_global<uint64_t>(a.keys_in) + start + i) : ~0ull;
        sp[os_pad(i)] = i < (int)len ? (uint32_t)i : 0xFFFFFFFFu;
    }
    __syncthreads();
    for (int ks = 1; ks <= plog; ++ks) {
        for (int jlog = ks - 1; jlog >= 0;) {
            const int mb = jlog + 1 < 3 ? jlog + 1 : 3;      // 8 pairs per lane: the registers of 16 words
            const int jl_log = jlog - mb + 1;
            switch (mb) {
                case 3: os_chunk_wide<3>(s, sp, P, 1 << ks, jl_log); break;
                case 2: os_chunk_wide<2>(s, sp, P, 1 << ks, jl_log); break;
                default: os_chunk_wide<1>(s, sp, P, 1 << ks, jl_log); break;
            }
            __syncthreads();
            jlog = jl_log - 1;
        }
    }
    for (int j = threadIdx.x; j < (int)len; j += 64) {
        const uint32_t from = start + sp[os_pad(j)];
        __builtin_nontemporal_store(s[os_pad(j)], as_global_mut<uint64_t>(a.keys_out) + start + j);
        __builtin_nontemporal_store(a.idx_in ? as_global<uint32_t>(a.idx_in)[from] : from, as_global_mut<uint32_t>(a.idx_out) + start + j);
    }
}
__global__ __launch_bounds__(64) void os_local_kernel(const OsLocalArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint64_t s[];
    const int64_t bucket = blockIdx.x;
    const uint32_t start = a.bstart[bucket], len = a.bstart[bucket + 1] - start;
    if (len <= a.len_lo || len > a.len_hi) return;
    if (len == 1) {
        if (threadIdx.x == 0) {
            as_global_mut<uint64_t>(a.keys_out)[start] = as_global<uint64_t>(a.keys_in)[start];
            as_global_mut<uint32_t>(a.idx_out)[start] = a.idx_in ? as_global<uint32_t>(a.idx_in)[start] : start;
        }
        return;
    }
    int P = 2, plog = 1;
    while (P < (int)len) { P <<= 1; ++plog; }
    const uint64_t low = (1ull << a.rbits) - 1;
    for (int i = threadIdx.x; i < P; i += 64) {
        uint64_t w = ~0ull;
        if (i < (int)len) w = (((__builtin_nontemporal_load(as_global<uint64_t>(a.keys_in) + start + i) - a.bias) & low) << 12) | (uint64_t)i;
        s[os_pad(i)] = w;
    }
    __syncthreads();
    for (int ks = 1; ks <= plog; ++ks) {
        for (int jlog = ks - 1; jlog >= 0;) {
            const int mb = jlog + 1 < 4 ? jlog + 1 : 4;
            const int jl_log = jlog - mb + 1;
            switch (mb) {
                case 4: os_chunk<4>(s, P, 1 << ks, jl_log); break;
                case 3: os_chunk<3>(s, P, 1 << ks, jl_log); break;
                case 2: os_chunk<2>(s, P, 1 << ks, jl_log); break;
                default: os_chunk<1>(s, P, 1 << ks, jl_log); break;
            }
            __syncthreads();   // one wave: orders the LDS traffic for the compiler, costs nothing
            jlog = jl_log - 1;
        }
    }
    const uint64_t top = (uint64_t)bucket << a.rbits;
    for (int j = threadIdx.x; j < (int)len; j += 64) {
        const uint64_t w = s[os_pad(j)];
        const uint32_t from = start + (uint32_t)(w & 4095);
        __builtin_nontemporal_store((top | (w >> 12)) + a.bias, as_global_mut<uint64_t>(a.keys_out) + start + j);
        __builtin_nontemporal_store(a.idx_in ? as_global<uint32_t>(a.idx_in)[from] : from, as_global_mut<uint32_t>(a.idx_out) + start + j);
    }
}
// A sample of the keys for the host's bucket plan (value range without the outliers, densest region): one key from every stride,
// at a hashed place inside it (a column with a period would alias with fixed places).
__global__ __launch_bounds__(kBlock) void os_sample_kernel(const uint64_t* keys, int64_t n, int64_t stride, int nsamp, uint64_t* out) {
    const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
    if (i >= nsamp) return;
    uint64_t h = (uint64_t)i * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
    int64_t pos = (int64_t)i * stride + (int64_t)(h % (uint64_t)stride);
    if (pos >= n) pos = n - 1;
    out[i] = as_global<uint64_t>(keys)[pos];
}
hipError_t launch_os_sample(const uint64_t* keys, int64_t n, int nsamp, uint64_t* out, hipStream_t s) {
    if (nsamp < 1 || n < 1) return hipSuccess;
    const int64_t stride = n / nsamp > 0 ? n / nsamp : 1;
    hipLaunchKernelGGL(os_sample_kernel, dim3((unsigned)((nsamp + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, keys, n, stride, nsamp, out);
    return hipGetLastError();
}
hipError_t launch_os_bounds(const uint64_t* keys, int64_t n, uint64_t bias, int rbits, int nbuckets, const OsBucket& fb, uint32_t* bstart, unsigned int* maxlen, hipStream_t s) {
    int64_t grid = (n + 1 + kBlock - 1) / kBlock;
    if (grid > eval_grid_limit()) grid = eval_grid_limit();
    hipLaunchKernelGGL(os_bounds_kernel, dim3((unsigned)grid), dim3(kBlock), 0, s, keys, n, bias, rbits, nbuckets, fb, bstart);
    int64_t g2 = ((int64_t)nbuckets + kBlock - 1) / kBlock;
    if (g2 > eval_grid_limit()) g2 = eval_grid_limit();
    hipLaunchKernelGGL(os_bucket_max_kernel, dim3((unsigned)g2), dim3(kBlock), 0, s, bstart, nbuckets, maxlen);
    return hipGetLastError();
}
static hipError_t launch_os_local_class(const OsLocalArgs& a, hipStream_t s);
hipError_t launch_os_local(const OsLocalArgs& a0, hipStream_t s) {
    // the LDS a block asks for decides how many blocks (of one wave) a CU holds: the few buckets beyond 1024 rows (tails,
    // outliers, a crowded value) get launches of their own (classes of <= 512, <= 1024, more rows) instead of setting the LDS
    // size of every bucket's block
    OsLocalArgs a = a0;
    a.len_lo = 0; a.len_hi = 0xFFFFFFFFu;
    if (a0.lds_items <= 512) return launch_os_local_class(a, s);
    a.lds_items = 512; a.len_hi = 512;
    hipError_t e = launch_os_local_class(a, s);
    if (e != hipSuccess) return e;
    if (a0.lds_items > 1024) {
        a.lds_items = 1024; a.len_lo = 512; a.len_hi = 1024;
        e = launch_os_local_class(a, s);
        if (e != hipSuccess) return e;
    }
    a.lds_items = a0.lds_items; a.len_lo = a0.lds_items > 1024 ? 1024 : 512; a.len_hi = 0xFFFFFFFFu;
    return launch_os_local_class(a, s);
}
static hipError_t launch_os_local_class(const OsLocalArgs& a, hipStream_t s) {
    const size_t words = (size_t)(a.lds_items + a.lds_items / 16 + 1);
    if (a.wide) {
        (void)hipFuncSetAttribute((const void*)os_local_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(words * 12 + 16));
        hipLaunchKernelGGL(os_local_wide_kernel, dim3((unsigned)a.nbuckets), dim3(64), words * 12 + 16, s, a);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(os_local_kernel, dim3((unsigned)a.nbuckets), dim3(64), words * 8, s, a);
    return hipGetLastError();
}

}  // namespace rdfk
