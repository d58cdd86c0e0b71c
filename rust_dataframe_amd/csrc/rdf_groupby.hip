// rdf_groupby.hip — hash GROUP BY key -> {sum | min | max, count}, second generation.
//
//   Transformation::GroupAggregate(groups, [AggregateFunction::{Sum, Min, Max, Count, Avg}]) is planned by
//   Dataset::try_aggregate (src/expression.rs:114-221, :696-711) and never executed by the reference
//   (src/evaluation.rs:73 panics): SQL semantics, parity unpinned by the reference.
//
// Three shapes, chosen by the promised number of groups:
//   gb2_stream_kernel     <= 2048 groups: every block folds its rows straight from the columns into an LDS table
//                         (the probe loop of the partition aggregation, fed by 16-byte-record-free column loads) and
//                         merges its few groups into the global table once.
//   gb2_scatter_kernel -> gb2_aggregate_kernel   <= 1.3 M groups: ONE scatter pass on the top 9 bits of an invertible
//                         hash, NO histogram pass: every (partition, block) owns a fixed-capacity region and the block
//                         writes nothing but whole, aligned 128-byte lines (8 records) — a partition's records wait in an
//                         LDS carry until a line is full.  Measured on MI355X (tools/ubench_scatter.hip): aligned 128-byte
//                         lines to 131 072 streams go out at 5.4-5.5 TB/s, the unaligned runs of the first-generation
//                         scatter at 2.5-2.9.  Traffic: 16 (read) + 16 (lines) + 16 (aggregate) = 48 B/row.
//                         A region that overflows (heavily skewed keys) sets a flag and the host re-runs the
//                         first-generation histogram + combining path (rdf_kernels.hip).
//   gb2_table_rows_kernel / gb2_merge_kernel   any number of groups: one open-addressing table in HBM, 64-bit CAS +
//                         hardware atomics; also the merge step of the multi-GPU exchange of partial groups.
#include <algorithm>

#include "rdf_common.hip.h"

namespace rdfk {

typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));

constexpr int kP = 1 << kG2PartBits;                 // 256 partitions
constexpr unsigned long long kFree = ~0ull;          // LDS free marker in hashed-key space
constexpr uint64_t kKeyMask = (1ull << (64 - kG2PartBits)) - 1;
constexpr uint64_t kDead = ~0ull;
constexpr int kMaxFlushLines = (7 * kP + kG2Super) / 8 + 32;   // lines a tile can flush: every carry full but one record, plus the tile (16-record lines: fewer)

__device__ __forceinline__ uint64_t g2_hash(uint64_t x) { x ^= x >> 32; x *= 0x9E3779B97F4A7C15ull; return x ^ (x >> 32); }
__device__ __forceinline__ uint64_t g2_unhash(uint64_t x) { x ^= x >> 32; x *= 0xF1DE83E19937733Dull; return x ^ (x >> 32); }
// Compact records: keys inside [base, base + 2^39) are hashed to 39 bits — 8 partition bits, implied by where a record lies,
// and the 31 bits a 4-byte key word holds next to the "value is not NULL" flag — by an invertible multiplication mod 2^39.
// The multiplier matters: a dense key range must land on the table's slots as evenly as under the 64-bit golden-ratio hash (no
// collisions at all; a wave waits for its unluckiest lane, so the tail of the probe lengths is what an LDS table costs).  The
// golden ratio rounded to 39 bits stops being golden past ~5e5 points (continued fraction 1,1,...,1 x 27, then 4, 2, 2, 1, 3, 3,
// 1, 1, 1, 22, ...); 0x4F1BBD2385 / 2^39 = [0; 1 x 17, 3, 1, 3, 1, 1, 1, 2, 3, ...] has no partial quotient above 3 and spreads
// 1e5 ... 1.3e6 consecutive (and strided) keys without a single first-probe collision (simulated per partition; the first try,
// 0x9E3779B1 mod 2^32, measured 50 ms per 1e9 rows in the aggregate pass against 2.8).
// The 64-bit image keeps the layout the tables expect: partition bits on top, the slot taken from the bits below them, mixed
// low bits for the probe step; it is a function of the key alone, and g2c_unhash gives the key back from its top 39 bits.
constexpr uint64_t kC39 = (1ull << 39) - 1;
__device__ __forceinline__ uint64_t g2c_image(uint64_t h39) { return (h39 << 25) | (((h39 * 0x2545F491ull) >> 7) & ((1ull << 25) - 1)); }
__device__ __forceinline__ uint64_t g2c_hash(uint64_t key, uint64_t base) { return g2c_image(((key - base) * kG2cMul) & kC39); }
__device__ __forceinline__ uint64_t g2c_unhash(uint64_t hk, uint64_t base) { return base + (((hk >> 25) * kG2cInv) & kC39); }
__device__ __forceinline__ uint64_t mix64b(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// order-preserving 64-bit image of a value of class `cls` (unsigned compare == the class's compare; IEEE total order for f64)
__device__ __forceinline__ uint64_t ord_bits(int cls, uint64_t b) {
    if (cls == CLS_F64) return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
    if (cls == CLS_SIGNED) return b ^ 0x8000000000000000ull;
    return b;
}
__device__ __forceinline__ uint64_t unord_bits(int cls, uint64_t o) {
    if (cls == CLS_F64) return (o >> 63) ? (o & 0x7FFFFFFFFFFFFFFFull) : ~o;
    if (cls == CLS_SIGNED) return o ^ 0x8000000000000000ull;
    return o;
}
__device__ __forceinline__ uint64_t agg_identity(int op) { return op == AGG_MIN ? ~0ull : 0ull; }

// what a row contributes to its group's accumulator, in table form
__device__ __forceinline__ uint64_t to_table_form(int op, int cls, uint64_t native_bits, bool is_null) {
    if (op == AGG_SUM) return is_null ? 0ull : native_bits;
    if (is_null) return agg_identity(op);
    if (cls == CLS_F64) { const double d = u2d(native_bits); if (d != d) return agg_identity(op); }   // NaN never wins (the column aggregates' rule)
    return ord_bits(cls, native_bits);
}

template <class P>
__device__ __forceinline__ void acc_apply(P* p, int op, int cls, uint64_t v) {
    if (op == AGG_SUM) {
        if (cls == CLS_F64) unsafeAtomicAdd((double*)p, u2d(v)); else atomicAdd(p, (unsigned long long)v);
    } else if (op == AGG_MIN) atomicMin(p, (unsigned long long)v);
    else atomicMax(p, (unsigned long long)v);
}

__device__ __forceinline__ int g2_dtype_size(int dt) {
    switch (dt) {
        case RDF_I8: case RDF_U8: return 1;
        case RDF_I16: case RDF_U16: return 2;
        case RDF_I32: case RDF_U32: case RDF_F32: return 4;
        default: return 8;
    }
}
__device__ __forceinline__ uint64_t g2_load_raw(const void* base, int size, int64_t idx, bool pred) {
    uint64_t x = 0;
    if (size == 8) { if (pred) x = __builtin_nontemporal_load(as_global<uint64_t>(base) + idx); }
    else if (size == 4) { if (pred) x = __builtin_nontemporal_load(as_global<uint32_t>(base) + idx); }
    else if (size == 2) { if (pred) x = as_global<uint16_t>(base)[idx]; }
    else { if (pred) x = as_global<uint8_t>(base)[idx]; }
    return x;
}

// ------------------------------------------------------------------------------------------------
// rows of one super-tile: thread t holds row t of each of NT consecutive tiles of kEvalTile rows (a tile lies inside one chunk)

template <int NT>
struct G2Raw {
    uint64_t key[NT], val[NT];
    uint32_t kb[NT], vb[NT];       // the validity BYTE holding the row's bit (0xFF: no bitmap)
    uint32_t kbit, vbit;           // 3 bits per row: bit position inside that byte
    uint32_t exists;
};

// All table lookups first (block-uniform, scalar unit), then every data load back to back; nothing consumes a load here.
// FAST (host-checked): 8-byte integer keys, 8-byte values (or none), no validity bitmap anywhere, 16-byte aligned chunks.
// A thread then holds rows 2t, 2t + 1 of each tile: ONE 16-byte load per tile and column, no bitmap bytes, no dtype switches.
template <int NT, int BLOCK, bool FAST = false>
__device__ __forceinline__ void g2_load(const Gb2Args& a, int64_t st, int tid, G2Raw<NT>& r) {
    static_assert(BLOCK == kEvalTile || BLOCK * 2 == kEvalTile, "a block covers a tile in one or two rows per thread");
    constexpr int RPT = kEvalTile / BLOCK;   // rows per thread per tile
    static_assert(NT % RPT == 0, "NT counts rows per thread");
    const int ksz = g2_dtype_size(a.key_dtype), vsz = g2_dtype_size(a.value_dtype);
    DevChunkCol kc[NT / RPT], vc[NT / RPT];
    int64_t r0[NT / RPT], clen[NT / RPT];
#pragma unroll
    for (int tt = 0; tt < NT / RPT; ++tt) {
        const int64_t tile = st + tt;
        kc[tt] = DevChunkCol{nullptr, nullptr, 0}; vc[tt] = kc[tt]; r0[tt] = 0; clen[tt] = 0;
        if (tile >= a.ntiles) continue;
        if (a.nchunks == 1) { kc[tt] = a.key0; vc[tt] = a.val0; r0[tt] = tile * kEvalTile; clen[tt] = a.len0; }
        else {
            const ConstPtr<int64_t> tile_start = as_const<int64_t>(a.chunk_tile_start);
            const int64_t c = find_chunk_tile(tile_start, a.nchunks, tile);
            r0[tt] = (tile - tile_start[c]) * kEvalTile;
            clen[tt] = as_const<int64_t>(a.chunk_len)[c];
            kc[tt] = const_col(a.keys, c);
            if (a.value_dtype >= 0) vc[tt] = const_col(a.values, c);
        }
    }
    r.exists = 0; r.kbit = 0; r.vbit = 0;
    if constexpr (FAST) {
        static_assert(RPT == 2, "the fast loader takes two rows per thread and tile");
#pragma unroll
        for (int tt = 0; tt < NT / 2; ++tt) {
            const int64_t row = r0[tt] + 2 * tid;
            const GlobalPtr<uint64_t> kp = as_global<uint64_t>(kc[tt].values) + kc[tt].offset + row;
            const GlobalPtr<uint64_t> vp = as_global<uint64_t>(vc[tt].values) + vc[tt].offset + row;
            u64x2 k = {0, 0}, v = {0, 0};
            // every register is the target of ONE load on either side of this block-uniform branch: a second load into a
            // register with one possibly in flight makes the compiler wait for vmcnt(0), i.e. for the prefetched batches too
            if (r0[tt] + kEvalTile <= clen[tt]) {
                r.exists |= 3u << (2 * tt);
                k = __builtin_nontemporal_load((GlobalPtr<u64x2>)kp);
                if (a.value_dtype >= 0) v = __builtin_nontemporal_load((GlobalPtr<u64x2>)vp);
            } else {   // the last tile of a chunk
                const bool e0 = row < clen[tt], e1 = row + 1 < clen[tt];
                r.exists |= ((uint32_t)e0 | (uint32_t)e1 << 1) << (2 * tt);
                if (e0) k[0] = __builtin_nontemporal_load(kp);
                if (e1) k[1] = __builtin_nontemporal_load(kp + 1);
                if (a.value_dtype >= 0) {
                    if (e0) v[0] = __builtin_nontemporal_load(vp);
                    if (e1) v[1] = __builtin_nontemporal_load(vp + 1);
                }
            }
            r.key[2 * tt] = k[0]; r.key[2 * tt + 1] = k[1];
            r.val[2 * tt] = v[0]; r.val[2 * tt + 1] = v[1];
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int tt = j / RPT;
        const int64_t row = r0[tt] + (int64_t)(j % RPT) * BLOCK + tid;
        const bool e = row < clen[tt];
        r.exists |= (uint32_t)e << j;
        r.key[j] = g2_load_raw(kc[tt].values, ksz, kc[tt].offset + row, e);
        r.kb[j] = 0xFFu;
        if (kc[tt].validity) {
            const int64_t b = kc[tt].offset + row;
            if (e) r.kb[j] = as_global<uint8_t>(kc[tt].validity)[b >> 3];
            r.kbit |= (uint32_t)(b & 7) << (3 * j);
        }
        r.val[j] = 0;
        r.vb[j] = 0xFFu;
        if (a.value_dtype >= 0) {
            r.val[j] = g2_load_raw(vc[tt].values, vsz, vc[tt].offset + row, e);
            if (vc[tt].validity) {
                const int64_t b = vc[tt].offset + row;
                if (e) r.vb[j] = as_global<uint8_t>(vc[tt].validity)[b >> 3];
                r.vbit |= (uint32_t)(b & 7) << (3 * j);
            }
        }
    }
}

// Raw rows -> (hashed key, accumulator contribution, count).  live bit j: the row becomes a record / a table update;
// NULL keys and the one key whose hash is the free marker go straight to two global accumulators.
template <int NT>
struct G2Rows { uint64_t hk[NT], val[NT]; uint32_t cnt, live, bad; };   // cnt bit j: the value is not NULL; bad (compact hashing): a key outside the 32-bit window

// Where the two groups that never enter a hash table (the NULL key; the key whose hash is the free marker) accumulate.
// The stream kernel keeps them in LDS and flushes once per block: a global store / atomic inside the streaming loop — even
// one that practically never executes — puts a WRITE next to the prefetch loads in the vector-memory counter, reads and
// writes return out of order, and the compiler then has to wait for vmcnt(0) wherever it waits at all.
struct LdsSpecial { unsigned long long* acc; unsigned int* cnt; unsigned int* flag; };   // [2] each; acc == nullptr: global

template <int NT, bool FAST = false, bool COMPACT = false>
__device__ __forceinline__ void g2_prepare(const Gb2Args& a, const G2Raw<NT>& r, G2Rows<NT>& o, const LdsSpecial ls = LdsSpecial{nullptr, nullptr, nullptr},
                                           const uint32_t* hot_lds = nullptr) {
    o.cnt = 0; o.live = 0; o.bad = 0;
    const bool counts_rows = a.value_dtype < 0;
    // heavy-hitter split: this pass takes the rows of the hot hash prefixes (mode 2) or all the others (mode 1); the two special
    // groups (NULL key, the key that hashes to the free marker) belong to the pass over the others.  The class bitmap sits in
    // LDS: read from global memory it put a wait for EVERY load in flight — the prefetched batches too — behind each tile
    // (7.5 ms per 1e9 rows for the hot pass alone).
    auto hot_filter = [&]() {
        if (!a.hot_mode || !hot_lds) return;
        const uint32_t want = a.hot_mode == 2 ? 1u : 0u;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (!((o.live >> j) & 1)) continue;
            const uint32_t bin = (uint32_t)(o.hk[j] >> 48);
            uint32_t is_hot = (hot_lds[bin >> 5] >> (bin & 31)) & 1u;          // the key's class holds a heavy hitter ...
            if (is_hot) {                                                          // ... and the key is one: 256-slot table of the hashed hot keys
                const unsigned long long* kt = (const unsigned long long*)(hot_lds + 2048);
                uint32_t sl = (uint32_t)(o.hk[j] >> 24) & 255u;
                is_hot = 0;
                for (;;) {
                    const unsigned long long k = kt[sl];
                    if (k == (unsigned long long)o.hk[j]) { is_hot = 1; break; }
                    if (k == kFree) break;
                    sl = (sl + 1) & 255u;
                }
            }
            if (is_hot != want) o.live &= ~(1u << j);
        }
    };
    const bool skip_special = a.hot_mode == 2;
    if constexpr (FAST) {
        o.cnt = r.exists;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const uint64_t v = counts_rows ? 0ull : to_table_form(a.op, a.vcls, r.val[j], false);
            const uint64_t hk = COMPACT ? g2c_hash(r.key[j], a.key_base) : g2_hash(r.key[j]);
            o.hk[j] = hk;
            o.val[j] = v;
            if (!((r.exists >> j) & 1)) continue;
            if (COMPACT && ((r.key[j] - a.key_base) >> 39)) { o.bad = 1; continue; }
            if (hk == kFree) {   // the one key whose hash is the free marker
                if (skip_special) continue;
                if (ls.acc) {
                    ls.flag[0] = 1;
                    if (!counts_rows) acc_apply(&ls.acc[0], a.op, a.vcls, v);
                    atomicAdd(&ls.cnt[0], 1u);
                } else {
                    a.special[0] = 1;
                    if (!counts_rows) acc_apply(&a.special_sums[0], a.op, a.vcls, v);
                    atomicAdd(&a.special_counts[0], 1ull);
                }
                continue;
            }
            o.live |= 1u << j;
        }
        hot_filter();
        return;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const bool e = (r.exists >> j) & 1;
        const bool knull = ((r.kb[j] >> ((r.kbit >> (3 * j)) & 7)) & 1u) == 0;
        const bool vnull = !counts_rows && ((r.vb[j] >> ((r.vbit >> (3 * j)) & 7)) & 1u) == 0;
        uint64_t v = r.val[j];
        if (a.value_dtype == RDF_F32) v = d2u((double)__uint_as_float((uint32_t)v));
        else if (a.value_dtype >= 0 && a.value_dtype != RDF_F64) v = normalize_int(a.value_dtype, v);
        v = counts_rows ? 0ull : to_table_form(a.op, a.vcls, v, vnull);
        const uint64_t nk = normalize_int(a.key_dtype, r.key[j]);
        const uint64_t hk = COMPACT ? g2c_hash(nk, a.key_base) : g2_hash(nk);
        o.hk[j] = hk;
        o.val[j] = v;
        const bool c1 = !vnull;
        if (e && c1) o.cnt |= 1u << j;
        if (!e) continue;
        if (COMPACT && !knull && ((nk - a.key_base) >> 39)) { o.bad = 1; continue; }
        if (knull || hk == kFree) {
            const int s = knull ? 1 : 0;
            if (skip_special) continue;
            if (ls.acc) {
                ls.flag[s] = 1;
                if (c1) {
                    if (!counts_rows) acc_apply(&ls.acc[s], a.op, a.vcls, v);
                    atomicAdd(&ls.cnt[s], 1u);
                }
            } else {
                a.special[s] = 1;
                if (c1) {
                    if (!counts_rows) acc_apply(&a.special_sums[s], a.op, a.vcls, v);
                    atomicAdd(&a.special_counts[s], 1ull);
                }
            }
            continue;
        }
        o.live |= 1u << j;
    }
    hot_filter();
}

// ------------------------------------------------------------------------------------------------
// LDS table: open addressing over hashed keys, double hashing inside a (sub-)table of a prime number of slots

struct LdsTab {
    unsigned long long* keys;   // [slots] hashed key, kFree = empty
    unsigned long long* acc;    // [slots]
    unsigned int*       cnt;    // [slots]
    unsigned int*       ngroups;
};

// Up to B pending (hashed key, value, count) updates per lane, probes interleaved: one LDS round trip serves all pending
// updates of the lane (the probe's dependent latency, not the atomics, is what a table costs).
// PB = partition bits above the slot bits.  PB == 0: the table holds every key (stream kernel) and the slot comes from the
// hash's top 32 bits — Fibonacci hashing proper, collision-free for dense key ranges.  PB > 0: inside a partition the top
// PB bits are the same for all keys and the slot comes from the 32 bits below them.  (Taking bits 23..54 over a WHOLE key
// range turns the golden-ratio rotation into one close to 13/30: 2000 dense keys then build probe chains of up to 52
// steps, and a wave waits for its unluckiest lane — measured 16.2 ms instead of 5.2 ms per 1e9 rows.)
template <int B, int PB>
__device__ __forceinline__ void tab_upsert(const LdsTab& t, int op, int cls, bool has_values, uint32_t base, uint32_t slots,
                                           const uint64_t (&hk)[B], const uint64_t (&val)[B], const uint32_t (&cnt)[B], uint32_t pending, uint32_t& err,
                                           uint32_t full_flag = 4u) {
    uint32_t s[B], step[B];
#pragma unroll
    for (int u = 0; u < B; ++u) {
        s[u] = (uint32_t)(((uint64_t)(uint32_t)(hk[u] >> (32 - PB)) * (uint64_t)slots) >> 32);
        step[u] = 1u + (uint32_t)(((uint64_t)(uint32_t)(PB ? hk[u] >> 3 : hk[u]) * (uint64_t)(slots - 1)) >> 32);
    }
    uint32_t guard = 0;
    while (__any(pending != 0)) {
        unsigned long long old[B];
#pragma unroll
        for (int u = 0; u < B; ++u) old[u] = ((pending >> u) & 1) ? t.keys[base + s[u]] : 0;
#pragma unroll
        for (int u = 0; u < B; ++u) {
            if (!((pending >> u) & 1)) continue;
            unsigned long long o = old[u];
            if (o == kFree) {
                o = atomicCAS(&t.keys[base + s[u]], kFree, (unsigned long long)hk[u]);
                if (o == kFree) { atomicAdd(t.ngroups, 1u); o = hk[u]; }
            }
            if (o == hk[u]) {
                pending &= ~(1u << u);
                if (cnt[u]) {
                    if (has_values) acc_apply(&t.acc[base + s[u]], op, cls, val[u]);
                    atomicAdd(&t.cnt[base + s[u]], cnt[u]);
                }
            } else {
                s[u] += step[u];
                if (s[u] >= slots) s[u] -= slots;
            }
        }
        if (++guard > slots) { if (pending) err |= full_flag; break; }   // table full: more groups than promised (stream) / than one partition's table holds (aggregate)
    }
}

// The partition tables of the aggregate pass, four keys per 32-byte BUCKET: one LDS round trip looks at all four, and a lane
// moves on only when all are somebody else's.  With one key per probe a table at load 0.5 costs a wave the chain of its
// unluckiest lane — 256 look-ups per batch, 8-10 dependent round trips for keys that hash like random numbers (scattered 64-bit
// key values: 7.9 ms per 1e9 rows in this pass against 2.8 ms for dense keys, whose multiplicative hash never collides; buckets
// of two: 6.6 ms).  A key's second bucket comes from other hash bits (a step of its own), so a crowded neighbourhood is left
// in one jump.  A bucket fills its slots in order and keys never leave, so two lanes with one key cannot both insert it:
// whoever loses a CAS looks at the bucket again.
constexpr uint32_t kG2Buckets = 1973;                 // prime; 4 x 1973 <= kG2Slots
static_assert(4 * kG2Buckets <= kG2Slots, "bucketed table larger than the LDS arrays");
template <int B, int PB>
__device__ __forceinline__ void tab_upsert_b4(const LdsTab& t, int op, int cls, bool has_values, const uint64_t (&hk)[B], const uint64_t (&val)[B],
                                              const uint32_t (&cnt)[B], uint32_t pending, uint32_t& err, uint32_t full_flag) {
    uint32_t b[B], step[B];
#pragma unroll
    for (int u = 0; u < B; ++u) {
        b[u] = (uint32_t)(((uint64_t)(uint32_t)(hk[u] >> (32 - PB)) * (uint64_t)kG2Buckets) >> 32);
        step[u] = 1u + (uint32_t)(((uint64_t)(uint32_t)(hk[u] >> 3) * (uint64_t)(kG2Buckets - 1)) >> 32);
    }
    uint32_t guard = 0;
    while (__any(pending != 0)) {
        u64x2 ka[B], kb[B];
#pragma unroll
        for (int u = 0; u < B; ++u) {
            ka[u][0] = 0; ka[u][1] = 0; kb[u] = ka[u];
            if ((pending >> u) & 1) { ka[u] = *(const u64x2*)&t.keys[4 * b[u]]; kb[u] = *(const u64x2*)&t.keys[4 * b[u] + 2]; }
        }
#pragma unroll
        for (int u = 0; u < B; ++u) {
            if (!((pending >> u) & 1)) continue;
            const unsigned long long h = (unsigned long long)hk[u];
            const unsigned long long k[4] = {ka[u][0], ka[u][1], kb[u][0], kb[u][1]};
            int slot = -1, free_at = -1;
#pragma unroll
            for (int j = 3; j >= 0; --j) { if (k[j] == kFree) free_at = j; }
#pragma unroll
            for (int j = 0; j < 4; ++j) if (k[j] == h) slot = (int)(4 * b[u]) + j;
            if (slot < 0) {
                if (free_at >= 0) {
                    const uint32_t at = 4 * b[u] + (uint32_t)free_at;
                    const unsigned long long o = atomicCAS(&t.keys[at], kFree, h);
                    if (o == kFree) { atomicAdd(t.ngroups, 1u); slot = (int)at; }
                    else if (o == h) slot = (int)at;
                    // else: another key took the slot meanwhile — the bucket is read again
                } else {
                    b[u] += step[u];
                    if (b[u] >= kG2Buckets) b[u] -= kG2Buckets;
                }
            }
            if (slot >= 0) {
                pending &= ~(1u << u);
                if (cnt[u]) {
                    if (has_values) acc_apply(&t.acc[slot], op, cls, val[u]);
                    atomicAdd(&t.cnt[slot], cnt[u]);
                }
            }
        }
        if (++guard > 6 * kG2Buckets) { if (pending) err |= full_flag; break; }   // this partition's table is full
    }
}

// Insert-or-find `key` (raw key bits) in the global table and fold (v, cnt) into its slot.  False on overflow.
__device__ __forceinline__ bool g2_global_upsert(const GroupTable& t, uint64_t key, int op, int cls, bool has_values, uint64_t v, uint64_t cnt) {
    const uint64_t mask = (uint64_t)t.capacity - 1;
    uint64_t s = mix64b(key) & mask;
    int64_t probes = 0;
    for (;;) {
        const unsigned long long old = atomicCAS(&t.keys[s], kGroupEmpty, (unsigned long long)key);
        if (old == kGroupEmpty) { atomicAdd(t.ngroups, 1u); break; }
        if (old == key) break;
        s = (s + 1) & mask;
        if (++probes > t.capacity) return false;
    }
    if (cnt) {
        if (has_values) acc_apply(&t.sums[s], op, cls, v);
        atomicAdd(&t.counts[s], (unsigned long long)cnt);
    }
    return true;
}
__device__ __forceinline__ void g2_global_special(const GroupTable& t, int which, int op, int cls, bool has_values, uint64_t v, uint64_t cnt) {
    const int64_t gs = t.capacity + which;
    t.special[which] = 1;
    if (cnt) {
        if (has_values) acc_apply(&t.sums[gs], op, cls, v);
        atomicAdd(&t.counts[gs], (unsigned long long)cnt);
    }
}

// A heavy partition is heavy because of a few keys: almost every record of such a partition carries the same hashed key, and
// 64 lanes adding into ONE LDS slot serialise (the hot-key input measured 12.9 ms in this kernel against 2.8 ms for uniform
// keys).  Before the table is touched, the lanes that hold the key of the wave's first pending lane reduce their contributions
// with a butterfly and leave ONE record; two rounds per batch element take care of the two heaviest keys of a wave.
__device__ __forceinline__ uint64_t g2_combine(int op, int cls, uint64_t a, uint64_t b) {
    if (op == AGG_SUM) return cls == CLS_F64 ? d2u(u2d(a) + u2d(b)) : a + b;
    if (op == AGG_MIN) return a < b ? a : b;
    return a > b ? a : b;
}
template <int B>
__device__ __forceinline__ void wave_combine(int op, int cls, const uint64_t (&hk)[B], uint64_t (&val)[B], uint32_t (&cnt)[B], uint32_t& pending) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int u = 0; u < B; ++u) {
        bool active = (pending >> u) & 1;
#pragma unroll 1
        for (int round = 0; round < 2; ++round) {
            const uint64_t m = __ballot(active);
            if (__popcll(m) < 8) break;
            const int leader = __builtin_ctzll(m);
            const uint64_t lk = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(hk[u] >> 32), leader) << 32) | (uint32_t)__shfl((int)(uint32_t)hk[u], leader);
            const bool same = active && hk[u] == lk;
            if (__popcll(__ballot(same)) >= 8) {
                uint64_t v = same ? val[u] : agg_identity(op);
                uint32_t c = same ? cnt[u] : 0u;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) {
                    const uint64_t ov = shfl_xor64(v, d);
                    const uint32_t oc = (uint32_t)__shfl_xor((int)c, d);
                    v = g2_combine(op, cls, v, ov);
                    c += oc;
                }
                if (lane == leader) { val[u] = v; cnt[u] = c; }
                else if (same) pending &= ~(1u << u);
            }
            active = active && !same;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// <= 2048 groups: columns -> per-block LDS table -> global table

constexpr int kStreamBlock = kGbBlock;       // 512 threads, 2 blocks per CU (80 KB of LDS each)
constexpr int kStreamRows = 4;               // rows per thread per batch = two tiles per block iteration
// The kernel is latency-bound (PMC at 4.3 TB/s: waves waiting 74 % of their cycles, VALU 29 %, LDS 24 % busy): what it needs
// is an unbroken load pipeline (see the loop below) and few registers; the fast variant is held to 80 VGPRs.
template <bool FAST>
__global__ __launch_bounds__(kStreamBlock, FAST ? 6 : 4) void gb2_stream_kernel(const Gb2Args a) {
    extern __shared__ __attribute__((aligned(16))) uint64_t gsm[];
    const int slots = a.table_slots;
    LdsTab t;
    t.keys = (unsigned long long*)gsm;
    t.acc = t.keys + slots;
    t.cnt = (unsigned int*)(t.acc + slots);
    t.ngroups = t.cnt + slots;
    LdsSpecial ls;
    ls.acc = (unsigned long long*)(gsm + ((size_t)slots * 20 + 4 + 7) / 8);   // behind keys, acc, cnt, ngroups (8-byte aligned)
    ls.cnt = (unsigned int*)(ls.acc + 2);
    ls.flag = ls.cnt + 2;
    uint32_t* hot_lds = a.hot_mode ? (uint32_t*)(ls.acc + 6) : nullptr;          // [2048] the hot classes' bitmap + [256] hashed hot keys, behind the special groups (48 bytes)
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned long long ident = agg_identity(a.op);
    if (hot_lds) for (int i = tid; i < 2048 + 512; i += kStreamBlock) hot_lds[i] = as_global<uint32_t>(a.hot_bitmap)[i];
    for (int i = tid; i < slots; i += kStreamBlock) { t.keys[i] = kFree; t.acc[i] = ident; t.cnt[i] = 0; }
    if (tid == 0) *t.ngroups = 0;
    if (tid < 2) { ls.acc[tid] = ident; ls.cnt[tid] = 0; ls.flag[tid] = 0; }
    __syncthreads();
    const bool has_values = a.value_dtype >= 0;
    const uint32_t base = (uint32_t)(lane & (a.replicas - 1)) * (uint32_t)a.sub_slots;
    uint32_t err = 0;
    constexpr int TPI = kStreamRows * kStreamBlock / kEvalTile;   // tiles per block iteration
    const int64_t stride = (int64_t)gridDim.x * TPI;
    int64_t st = (int64_t)blockIdx.x * TPI;
    // FAST: two batches of column loads stay in flight behind the one being folded.  With 16 waves per CU (the LDS tables
    // allow no more) one batch ahead is 64 KB per CU in flight, which is what bounded the kernel (PMC: waves waiting 74 % of
    // their cycles, VALU 29 % and LDS 24 % busy at 4.3 TB/s); two are 128 KB.  (The general loader keeps one batch ahead: a
    // second one would cost it the registers that let two blocks share a CU.)
    auto fold = [&](const G2Raw<kStreamRows>& raw) {
        G2Rows<kStreamRows> rows;
        g2_prepare<kStreamRows, FAST>(a, raw, rows, ls, hot_lds);
        uint32_t cnt[kStreamRows];
#pragma unroll
        for (int j = 0; j < kStreamRows; ++j) cnt[j] = (rows.cnt >> j) & 1;
        // (hot_mode 2, the heavy hitters' pass: at most ~100 keys, which the replicated sub-tables take like any small GROUP BY; a
        // full table — cannot happen with the host's sizes — raises flag 128 and the host runs the call again without the split)
        tab_upsert<kStreamRows, 0>(t, a.op, a.vcls, has_values, base, (uint32_t)a.sub_slots, rows.hk, rows.val, cnt, rows.live, err, a.hot_mode == 2 ? 128u : 4u);
    };
    // Software pipeline: the next batch's loads fly while this one is folded.  The batches ROTATE through two register sets
    // by unrolling, never by copying (`cur = nxt` is a read of nxt: a wait for the loads just issued), and the prefetch is
    // issued UNCONDITIONALLY (past the end the block's last batch is loaded again and never folded).  Deeper prefetch does
    // not pay: the compiler waits with vmcnt(0) at every fold whatever the depth (measured: two batches ahead = one).
    const int64_t last_st = ((a.ntiles - 1) / TPI) * TPI;
    auto prefetch = [&](G2Raw<kStreamRows>& dst, int64_t at) { g2_load<kStreamRows, kStreamBlock, FAST>(a, at < a.ntiles ? at : last_st, tid, dst); };
    if (st < a.ntiles) {
        G2Raw<kStreamRows> ra, rb;
        prefetch(ra, st);
        for (;;) {
            prefetch(rb, st + stride); fold(ra); st += stride; if (st >= a.ntiles) break;
            prefetch(ra, st + stride); fold(rb); st += stride; if (st >= a.ntiles) break;
        }
    }
    __syncthreads();
    if (tid < 2 && ls.flag[tid]) {   // the block's share of the two special groups
        a.special[tid] = 1;
        if (ls.cnt[tid]) {
            if (has_values) acc_apply(&a.special_sums[tid], a.op, a.vcls, ls.acc[tid]);
            atomicAdd(&a.special_counts[tid], (unsigned long long)ls.cnt[tid]);
        }
    }
    // the block's groups -> the global table (keys un-hashed: the global table is addressed by raw key bits)
    for (int i = tid; i < slots; i += kStreamBlock) {
        const unsigned long long hk = t.keys[i];
        if (hk == kFree) continue;
        const uint64_t key = g2_unhash(hk);
        if (key == kGroupEmpty) g2_global_special(a.t, 0, a.op, a.vcls, has_values, t.acc[i], t.cnt[i]);
        else if (!g2_global_upsert(a.t, key, a.op, a.vcls, has_values, t.acc[i], t.cnt[i])) err |= 4u;
        // a group whose every value was NULL still exists: the upsert above inserted its key with cnt == 0
    }
    if (err) atomicOr(a.flags, err);
}

// ------------------------------------------------------------------------------------------------
// Skew probe in front of the scatter pass: the partition histogram of a sample of the keys (64 rows out of every sampled
// tile, tiles spread evenly over the input, about 1 M keys in all).  The scatter's (partition, block) regions hold 1.19 x
// their mean: a partition whose share is above that overflows them near the END of the pass, so without the probe heavily
// skewed keys paid for a whole wasted scatter (~9 ms per 1e9 rows) before the combining path ran.
// minmax (optional): the smallest / largest sampled key in the key type's order (sign bit flipped for the signed types), NULL
// keys left out — what the host sizes the 32-bit window of the compact records from.
__global__ __launch_bounds__(256) void gb2_skew_probe_kernel(const Gb2Args a, int64_t tile_step, unsigned int* hist, unsigned long long* minmax, unsigned int* hbins, unsigned long long* ktab) {
    // hist[0 .. kP): partitions of the 64-bit hash; hist[kP .. 2 kP): partitions of the compact hash taken with base 0 — the host
    // picks window bases that are multiples of 2^31, for which the real partition numbers are these rotated by a constant
    __shared__ unsigned int h[2 * kP];
    __shared__ unsigned long long mm[2];
    // the block's own counters for the classes it meets (class -> slot by its low bits, first come first kept): a skewed input
    // sends a quarter of the samples to ONE class, and a million same-address global atomics took 3 ms
    constexpr int kCache = 4096;
    __shared__ unsigned int cls_of[kCache], cls_n[kCache];
    for (int i = threadIdx.x; i < kCache; i += 256) { cls_of[i] = 0xFFFFFFFFu; cls_n[i] = 0; }
    for (int i = threadIdx.x; i < 2 * kP; i += 256) h[i] = 0;
    if (threadIdx.x == 0) { mm[0] = ~0ull; mm[1] = 0ull; }
    const uint64_t flip = (a.key_dtype == RDF_I64 || a.key_dtype == RDF_I32 || a.key_dtype == RDF_I16 || a.key_dtype == RDF_I8) ? 0x8000000000000000ull : 0ull;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ksz = g2_dtype_size(a.key_dtype);
    for (int64_t i = (int64_t)blockIdx.x * 4 + wave; i * tile_step < a.ntiles; i += (int64_t)gridDim.x * 4) {
        const int64_t tile = i * tile_step;
        DevChunkCol kc = a.key0;
        int64_t r0 = tile * kEvalTile, clen = a.len0;
        if (a.nchunks > 1) {
            const ConstPtr<int64_t> ts = as_const<int64_t>(a.chunk_tile_start);
            const int64_t c = find_chunk_tile(ts, a.nchunks, tile);
            r0 = (tile - ts[c]) * kEvalTile;
            clen = as_const<int64_t>(a.chunk_len)[c];
            kc = const_col(a.keys, c);
        }
        const int64_t row = r0 + lane * 16 + (int)((tile * 7) & 15);
        uint32_t hbin = 0xFFFFFFFFu;
        uint64_t khot = 0;
        bool kact = false;
        if (row < clen) {
            const uint64_t nk = normalize_int(a.key_dtype, g2_load_raw(kc.values, ksz, kc.offset + row, true));
            const uint64_t hk = g2_hash(nk);
            atomicAdd(&h[(uint32_t)(hk >> (64 - kG2PartBits))], 1u);
            atomicAdd(&h[kP + (uint32_t)(g2c_hash(nk, 0) >> (64 - kG2PartBits))], 1u);
            hbin = (uint32_t)(hk >> 48);
            if (ktab && hk != kFree && ((as_global<uint32_t>(a.hot_bitmap)[hbin >> 5] >> (hbin & 31)) & 1u)) { khot = hk; kact = true; }
            if (minmax) {
                bool valid = true;
                if (kc.validity) { const int64_t b = kc.offset + row; valid = (as_global<uint8_t>(kc.validity)[b >> 3] >> (b & 7)) & 1; }
                if (valid) { atomicMin(&mm[0], (unsigned long long)(nk ^ flip)); atomicMax(&mm[1], (unsigned long long)(nk ^ flip)); }
            }
        }
        if (hbins && hbin != 0xFFFFFFFFu) {
            // 65 536 classes by the hash's top 16 bits: where the heavy hitters are
            const uint32_t sl = hbin & (kCache - 1);
            const unsigned int o = atomicCAS(&cls_of[sl], 0xFFFFFFFFu, hbin);
            if (o == 0xFFFFFFFFu || o == hbin) atomicAdd(&cls_n[sl], 1u);
            else atomicAdd(&hbins[hbin], 1u);
        }
        if (ktab) {
            // second run, over the same sample: the KEYS inside the hot classes, counted in a 4096-slot table (hashed key, samples) —
            // a class is 1 / 65 536 of the key space, the heavy hitter in it is one key
#pragma unroll 1
            for (int round = 0; round < 64; ++round) {
                const uint64_t mact = __ballot(kact);
                if (!mact) break;
                const int leader = __builtin_ctzll(mact);
                const uint64_t lk = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(khot >> 32), leader) << 32) | (uint32_t)__shfl((int)(uint32_t)khot, leader);
                const bool same = kact && khot == lk;
                const unsigned long long c = (unsigned long long)__popcll(__ballot(same));
                if (lane == leader) {
                    uint32_t sl = (uint32_t)(lk >> 20) & 4095u;
                    for (int probe = 0; probe < 4096; ++probe) {
                        const unsigned long long o = atomicCAS(&ktab[2 * sl], 0ull, (unsigned long long)lk);
                        if (o == 0ull || o == (unsigned long long)lk) { atomicAdd(&ktab[2 * sl + 1], c); break; }
                        sl = (sl + 1) & 4095u;
                    }
                }
                kact = kact && !same;
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * kP; i += 256) if (h[i]) atomicAdd(&hist[i], h[i]);
    if (hbins) for (int i = threadIdx.x; i < kCache; i += 256) if (cls_n[i]) atomicAdd(&hbins[cls_of[i]], cls_n[i]);
    if (minmax && threadIdx.x == 0 && mm[0] <= mm[1]) { atomicMin(&minmax[0], mm[0]); atomicMax(&minmax[1], mm[1]); }
}

// ------------------------------------------------------------------------------------------------
// <= 1.3 M groups, pass 1: scatter 16-byte records into (partition, block) regions, whole aligned lines only.
// Record: word 0 = (cnt << 56) | (hashed key & (2^56 - 1)) — the partition bits are implied by where the record lies and
// make room for cnt (0: the value was NULL, the group must still exist) —, word 1 = accumulator contribution; ~0 = dead.
//
// Two blocks per CU (72 KB of LDS each): measured on the one-block-per-CU version (tools/abl_gb2.py) a tile's phases did
// not overlap at all — loads + hash 3.8 ms, ranking + staging 2.3 ms, flush 3.3 ms per 1e9 rows added up to the
// kernel's 9.5 ms — so the second block is what fills the memory pipe while the first one is in its LDS phases.
// Per flushed line ONE 64-bit descriptor (destination line, staging index, carry length, partition) is all the flush
// phase reads before it moves the line: the partition's owner thread writes it while it does the bookkeeping.

template <bool FAST, bool COMPACT>
__global__ __launch_bounds__(kG2Block, 2 * kG2BlocksPerCU) void gb2_scatter_kernel(const Gb2Args a) {
    // COMPACT: 12-byte records in units of 16 (a 128-byte line of values in `recs`, 64 bytes of key words in `recs_k`), the
    // LDS staging split the same way; a "line" below is a unit of L records in either layout.
    constexpr int L = COMPACT ? kG2LineC : kG2Line, LS = COMPACT ? 4 : 3;
    constexpr int CS = L - 1;                 // carry slots per partition (a carry never holds a whole line)
    extern __shared__ __attribute__((aligned(16))) uint64_t gsm[];
    u64x2* stage = (u64x2*)gsm;                           // [kG2Super] this tile's records, grouped by partition
    u64x2* carry = stage + kG2Super;                      // [kP * 7] records waiting for their line to fill
    uint64_t* stage_v = gsm;                              // COMPACT: [kG2Super] values, [kP * 15] carried values,
    uint64_t* carry_v = stage_v + kG2Super;               //          [kG2Super] key words, [kP * 15] carried key words
    uint32_t* stage_k = (uint32_t*)(carry_v + kP * CS);
    uint32_t* carry_k = stage_k + kG2Super;
    uint64_t* ldesc = COMPACT ? (uint64_t*)(carry_k + kP * CS + (kP * CS & 1)) : (uint64_t*)(carry + kP * CS);  // [kMaxFlushLines] per flush line: dst line | (stage index + L) << 32 | c << 48 | j0 << (48 + LS) | d << (49 + LS)
    uint32_t* tcnt = (uint32_t*)(ldesc + kMaxFlushLines); // [kP] rank counters of the tile
    uint32_t* ccnt = tcnt + kP;                           // [kP] records in the carry
    uint32_t* written = ccnt + kP;                        // [kP] lines of the region already written
    uint32_t* lstart = written + kP;                      // [kP] first staging slot of the partition
    uint32_t* tail = lstart + kP;                         // [kP] staging index of the partition's new carry (partitions that flush)
    uint32_t* hot_lds = a.hot_mode ? tail + kP : nullptr; // [2048] heavy-hitter split: bitmap of the hash classes another pass takes
    __shared__ uint32_t wtot_t[kP / 64], wtot_k[kP / 64], ltot, abort_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (hot_lds) for (int i = tid; i < 2048 + 512; i += kG2Block) hot_lds[i] = as_global<uint32_t>(a.hot_bitmap)[i];
    if (tid < kP) { tcnt[tid] = 0; ccnt[tid] = 0; written[tid] = 0; }
    if (tid == 0) abort_s = 0;
    __syncthreads();
    const uint32_t nb = gridDim.x, bid = blockIdx.x;
    const uint32_t cap = tid < kP && a.part_cap ? as_const<uint32_t>(a.part_cap)[tid] : (uint32_t)a.cap_lines;   // lines of this thread's partition's region
    const uint32_t region0 = tid < kP ? (a.part_off ? as_const<uint32_t>(a.part_off)[tid] + bid * cap : ((uint32_t)tid * nb + bid) * cap) : 0;   // its first line
    u64x2* const recs = (u64x2*)a.recs;
    uint64_t* const recs_v = a.recs;
    uint32_t* const recs_k = a.recs_k;
    const bool has_values = a.value_dtype >= 0;
    uint32_t err = 0;
    constexpr int TPI = kG2Super / kEvalTile;
    const int64_t stride = (int64_t)nb * TPI;
    int64_t st = (int64_t)bid * TPI;
    G2Raw<kG2Rows> raw;
    G2Rows<kG2Rows> rows;
    if (st < a.ntiles) { g2_load<kG2Rows, kG2Block, FAST>(a, st, tid, raw); g2_prepare<kG2Rows, FAST, COMPACT>(a, raw, rows, LdsSpecial{nullptr, nullptr, nullptr}, hot_lds); }
    for (; st < a.ntiles; st += stride) {
        const bool more = st + stride < a.ntiles;
        if (more) g2_load<kG2Rows, kG2Block, FAST>(a, st + stride, tid, raw);   // the next tile's loads fly during the LDS phases
        if (a.ablate == 24) { uint64_t x = 0; for (int j = 0; j < kG2Rows; ++j) x ^= rows.hk[j] ^ rows.val[j]; if (x == 0x1234567) a.special[0] = 1; if (more) g2_prepare<kG2Rows, FAST, COMPACT>(a, raw, rows, LdsSpecial{nullptr, nullptr, nullptr}, hot_lds); continue; }   // loads + hash only
        if (COMPACT && rows.bad) { err |= 64u; atomicOr(a.flags, 64u); }   // a key outside the 32-bit window: every block stops at its next tile
        // (B) rank inside the partition
        uint32_t rank[kG2Rows];
#pragma unroll
        for (int j = 0; j < kG2Rows; ++j)
            if ((rows.live >> j) & 1) rank[j] = atomicAdd(&tcnt[(uint32_t)(rows.hk[j] >> (64 - kG2PartBits))], 1u);
        __syncthreads();
        // (C) per partition: staging offset, number of whole lines it can now flush, their place in the flush list
        uint32_t tc = 0, cc = 0, kk = 0, nd = 0, inc_t = 0, inc_k = 0;
        if (tid < kP) {
            tc = tcnt[tid]; cc = ccnt[tid];
            kk = (cc + tc) >> LS;
            nd = kk;
            inc_t = tc; inc_k = nd;
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) {
                const uint32_t yt = (uint32_t)__shfl_up((int)inc_t, m), yk = (uint32_t)__shfl_up((int)inc_k, m);
                if (lane >= m) { inc_t += yt; inc_k += yk; }
            }
            if (lane == 63) { wtot_t[wave] = inc_t; wtot_k[wave] = inc_k; }
        }
        if (tid == 0) abort_s = __hip_atomic_load(a.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (16u | 64u);
        __syncthreads();
        if (abort_s) break;     // some block overflowed a region / met a key outside the window: the host re-runs another path
        if (tid < kP) {
            uint32_t off_t = 0, off_k = 0;
#pragma unroll
            for (int w = 0; w < kP / 64; ++w) if (w < wave) { off_t += wtot_t[w]; off_k += wtot_k[w]; }
            const uint32_t ex_t = off_t + inc_t - tc, ex_k = off_k + inc_k - nd;
            const uint32_t w0 = written[tid];
            // no line to flush: the tile's records go straight to the carry in phase (D) (bit 31: lstart is a carry index) — a loop
            // in this thread, or descriptors of their own in phase (E), sat on every tile's critical path
            lstart[tid] = kk == 0 ? (0x80000000u | ((uint32_t)tid * CS + cc)) : ex_t;
            tail[tid] = ex_t + L * kk - cc;
            if (w0 + kk > cap) err |= 16u;
            for (uint32_t j = 0; j < kk; ++j) {
                const uint32_t dst = w0 + j < cap ? region0 + w0 + j : 0xFFFFFFFFu;
                // stage index of the line's lane 0, biased by L (line 0 starts c slots before the partition's staging run)
                ldesc[ex_k + j] = (uint64_t)dst | ((uint64_t)(ex_t + L * j + L - cc) << 32) | ((uint64_t)(j == 0 ? cc : 0) << 48)
                                  | ((uint64_t)(j == 0) << (48 + LS)) | ((uint64_t)tid << (49 + LS));
            }
            ccnt[tid] = cc + tc - L * kk;
            written[tid] = w0 + kk;
            tcnt[tid] = 0;
            if (tid == kP - 1) ltot = ex_k + nd;
        }
        __syncthreads();
        // (D) stage the tile's records grouped by partition
#pragma unroll
        for (int j = 0; j < kG2Rows; ++j)
            if (((rows.live >> j) & 1) && a.ablate != 23) {
                const uint32_t d = (uint32_t)(rows.hk[j] >> (64 - kG2PartBits));
                const uint32_t at = lstart[d] + rank[j], ci = at & 0x7FFFFFFFu;
                const bool to_carry = at >> 31;
                if constexpr (COMPACT) {
                    const uint32_t kw = (((rows.cnt >> j) & 1u) << 31) | ((uint32_t)(rows.hk[j] >> 25) & 0x7FFFFFFFu);
                    if (to_carry) { carry_k[ci] = kw; if (has_values) carry_v[ci] = rows.val[j]; }
                    else { stage_k[ci] = kw; if (has_values) stage_v[ci] = rows.val[j]; }
                } else {
                    u64x2 rec;
                    rec[0] = ((uint64_t)((rows.cnt >> j) & 1) << (64 - kG2PartBits)) | (rows.hk[j] & kKeyMask);
                    rec[1] = rows.val[j];
                    if (to_carry) carry[ci] = rec; else stage[ci] = rec;
                }
            }
        __syncthreads();
        // the next tile's rows: waiting for its loads HERE keeps the flush stores below out of that wait
        if (more) g2_prepare<kG2Rows, FAST, COMPACT>(a, raw, rows, LdsSpecial{nullptr, nullptr, nullptr}, hot_lds);
        // (E) flush whole lines: L consecutive lanes write one aligned line of a region (COMPACT: 128 bytes of values + 64 of key
        // words).  An iteration is a chain of dependent LDS round trips (descriptor -> record -> new carry); the chains of
        // different lines are independent (a line reads its partition's OLD carry, only that partition's first line writes the
        // new one), so the compact layout — half as many lines per pass of the block — takes two lines per lane group and
        // iteration, every LDS read of both before the first store.  (The 16-byte layout spills registers when it does that:
        // measured 10.1 against 9.6 ms per 1e9 rows.)
        const uint32_t nl = (a.ablate == 22 || a.ablate == 23) ? 0u : ltot;
        struct Line { uint64_t v, nv; uint32_t kw, nkw, dst, slot; u64x2 rec, nrec; bool live, carry_w; };
        const uint32_t l8 = (uint32_t)tid & (L - 1);
        auto line_load = [&](uint32_t i, Line& ln) {
            ln.live = i < nl; ln.carry_w = false;
            if (!ln.live) return;
            const uint64_t ds = ldesc[i];
            const uint32_t src = (uint32_t)(ds >> 32) & 0xFFFFu, c = (uint32_t)(ds >> 48) & (uint32_t)(L - 1), d = (uint32_t)(ds >> (49 + LS));
            ln.dst = (uint32_t)ds;
            ln.slot = d * CS + l8;
            const bool fc = l8 < c;
            if constexpr (COMPACT) {
                ln.kw = fc ? carry_k[ln.slot] : stage_k[src + l8 - L];
                ln.v = 0;
                if (has_values) ln.v = fc ? carry_v[ln.slot] : stage_v[src + l8 - L];
            } else ln.rec = fc ? carry[ln.slot] : stage[src + l8 - L];
            if (((ds >> (48 + LS)) & 1) && l8 < ccnt[d]) {   // the partition's new carry: the tail of its tile records
                ln.carry_w = true;
                const uint32_t t = tail[d] + l8;
                if constexpr (COMPACT) { ln.nkw = stage_k[t]; if (has_values) ln.nv = stage_v[t]; }
                else ln.nrec = stage[t];
            }
        };
        auto line_store = [&](const Line& ln) {
            if (!ln.live) return;
            if constexpr (COMPACT) {
                if (ln.dst != 0xFFFFFFFFu) {
                    if (a.ablate != 21) { recs_k[(uint64_t)ln.dst * L + l8] = ln.kw; if (has_values) recs_v[(uint64_t)ln.dst * L + l8] = ln.v; }
                    else if (ln.kw == 0x1234567u && ln.v == 1) a.special[0] = 1;
                }
                if (ln.carry_w) { carry_k[ln.slot] = ln.nkw; if (has_values) carry_v[ln.slot] = ln.nv; }
            } else {
                if (ln.dst != 0xFFFFFFFFu) { if (a.ablate != 21) recs[(uint64_t)ln.dst * kG2Line + l8] = ln.rec; else if (ln.rec[0] == 0x1234567) a.special[0] = 1; }
                if (ln.carry_w) carry[ln.slot] = ln.nrec;
            }
        };
        constexpr uint32_t G = kG2Block / L;   // lines per pass of the block
        if constexpr (COMPACT) {
            for (uint32_t i = (uint32_t)tid >> LS; i < nl; i += 2 * G) {
                Line l0, l1;
                line_load(i, l0);
                line_load(i + G, l1);
                line_store(l0);
                line_store(l1);
            }
        } else {
            for (uint32_t i = (uint32_t)tid >> LS; i < nl; i += G) {
                Line l0;
                line_load(i, l0);
                line_store(l0);
            }
        }
        if (err & 16u) atomicOr(a.flags, 16u);
        __syncthreads();
    }
    // what is left in the carries goes out as one last line per partition, padded with dead records
    __syncthreads();
    for (uint32_t d = (uint32_t)tid >> LS; d < (uint32_t)kP; d += kG2Block / L) {
        const uint32_t c = ccnt[d], l8 = (uint32_t)tid & (L - 1);
        if (c == 0) continue;
        const uint32_t line = written[d];
        const uint32_t dcap = a.part_cap ? as_const<uint32_t>(a.part_cap)[d] : (uint32_t)a.cap_lines;
        const uint64_t dreg = a.part_off ? (uint64_t)as_const<uint32_t>(a.part_off)[d] + (uint64_t)bid * dcap : (uint64_t)(d * nb + bid) * dcap;
        if constexpr (COMPACT) {
            // (no dead marker in a 4-byte key word: nlines holds the region's RECORD count in this layout)
            if (line >= dcap) err |= 16u;
            else if (l8 < c) { recs_k[(dreg + line) * L + l8] = carry_k[d * CS + l8]; if (has_values) recs_v[(dreg + line) * L + l8] = carry_v[d * CS + l8]; }
        } else {
            u64x2 rec;
            rec[0] = kDead; rec[1] = 0;
            if (l8 < c) rec = carry[d * CS + l8];
            if (line < dcap) recs[(dreg + line) * kG2Line + l8] = rec;
            else err |= 16u;
        }
    }
    __syncthreads();
    if (tid < kP) a.nlines[(int64_t)tid * nb + bid] = COMPACT ? written[tid] * L + ccnt[tid] : written[tid] + (ccnt[tid] ? 1u : 0u);
    if (err) atomicOr(a.flags, err);
}

// pass 2: one block per partition; its records are the nb line ranges the scatter blocks wrote
constexpr int kAggBatch = 4;
constexpr int kAggMaxRegions = kG2MaxBlocks;      // scatter blocks (two per CU of the 256): what an item's region list may hold in LDS
__global__ __launch_bounds__(kG2AggBlock) void gb2_aggregate_kernel(const Gb2AggArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint64_t gsm[];
    LdsTab t;
    t.keys = (unsigned long long*)gsm;
    t.acc = t.keys + kG2Slots;
    t.cnt = (unsigned int*)(t.acc + kG2Slots);
    t.ngroups = t.cnt + kG2Slots;
    unsigned int* misc = t.ngroups + 1;   // [0] output base, [1] emit cursor
    unsigned int* s_nrec = misc + 3;      // [kAggMaxRegions] records of the item's regions
    unsigned int* s_pre = s_nrec + kAggMaxRegions;   // [kAggMaxRegions + 1] exclusive prefix of their batch counts
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = wave_id();
    constexpr int NW = kG2AggBlock / 64;
    const unsigned long long ident = agg_identity(a.op);
    uint32_t err = 0;
    const u64x2* const recs = (const u64x2*)a.recs;
    const bool compact = (a.compact & 1) != 0, single = (a.compact & 256) != 0;
    const int L = compact ? kG2LineC : kG2Line;
    const int nitems = a.nwork > 0 ? a.nwork : kP;
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        int p = item, rb0 = 0, rb1 = (int)a.nb, multi = 0;
        if (a.nwork > 0) { const Gb2Work w = a.work[item]; p = w.p; rb0 = w.b0; rb1 = w.b1; multi = w.multi; }
        const int64_t pcap = a.part_cap ? (int64_t)a.part_cap[p] : a.cap_lines;
        const int64_t pline0 = a.part_off ? (int64_t)a.part_off[p] : (int64_t)p * a.nb * a.cap_lines;
        for (int i = tid; i < kG2Slots; i += kG2AggBlock) { t.keys[i] = kFree; t.acc[i] = ident; t.cnt[i] = 0; }
        if (tid == 0) *t.ngroups = 0;
        __syncthreads();
        const uint64_t ptop = (uint64_t)p << (64 - kG2PartBits);
        // The item's records are the line ranges ("regions") the scatter blocks rb0 .. rb1 wrote for this partition.  Their record
        // counts are staged in LDS once per item together with an exclusive prefix of their BATCH counts (a batch = kAggBatch x 64
        // consecutive records of one region), and wave w takes the batches w, w + NW, ... of the whole item: every wave gets the
        // same number of batches whatever the regions' lengths, and stepping from one region to the next costs two LDS reads.
        // (Until round 5 every wave walked every region and read each region's count from memory when it got there — a dependent
        // global load per region and wave, ~1 ms per launch whatever the input's size: 512 regions x a memory latency; and a
        // region of 954 records, 1.25e8 rows, fed 4 of the 16 waves.)  The next batch's loads are issued before the current one
        // is folded into the table.
        const int nreg = rb1 - rb0;
        for (int i = tid; i < nreg; i += kG2AggBlock) {
            const int64_t nl = (int64_t)as_const<uint32_t>(a.nlines)[(int64_t)p * a.nb + rb0 + i];
            s_nrec[i] = (uint32_t)(compact ? nl : nl * L);
        }
        __syncthreads();
        if (wave == 0) {   // exclusive prefix of ceil(nrec / batch) over <= kAggMaxRegions regions: 8 per lane + a wave scan
            constexpr int PER = kAggMaxRegions / 64;
            uint32_t loc[PER], sum = 0;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int i = lane * PER + j;
                loc[j] = sum;
                sum += i < nreg ? (s_nrec[i] + kAggBatch * 64 - 1) / (kAggBatch * 64) : 0u;
            }
            uint32_t incl = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d); if (lane >= d) incl += o; }
            const uint32_t excl = incl - sum;
#pragma unroll
            for (int j = 0; j < PER; ++j) { const int i = lane * PER + j; if (i <= nreg) s_pre[i] = excl + loc[j]; }
            if (lane == 63 && nreg == kAggMaxRegions) s_pre[nreg] = incl;
        }
        __syncthreads();
        const int64_t vtotal = (int64_t)s_pre[nreg];
        int64_t vb = wave;
        int reg = 0;
        uint32_t reg_end = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_pre[1 < nreg ? 1 : nreg]);
        auto load_batch = [&](u64x2 (&r)[kAggBatch]) -> bool {   // false: this wave's share of the item is exhausted
            if (vb >= vtotal) return false;
            while ((int64_t)reg_end <= vb) { ++reg; reg_end = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_pre[reg + 1]); }
            const int64_t nrec = (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)s_nrec[reg]);
            const int64_t off = (vb - (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)s_pre[reg])) * (kAggBatch * 64);
            const int64_t base_i = (pline0 + (int64_t)(rb0 + reg) * pcap) * L;
            if (compact) {
                // 12-byte records: the key WORD travels as it was read (zero-extended: never kDead) and becomes the 16-byte form in
                // the fold.  (Converted here, the words were waited for right after their loads were issued — the next batch was
                // never in flight under the current one.  Round 6, read off the ISA.)
#pragma unroll
                for (int u = 0; u < kAggBatch; ++u) {
                    const int64_t i = off + u * 64 + lane;
                    r[u][0] = kDead; r[u][1] = 0;
                    if (i < nrec) {
                        r[u][0] = (uint64_t)__builtin_nontemporal_load(a.recs_k + base_i + i);
                        if (a.has_values) r[u][1] = __builtin_nontemporal_load(a.recs + base_i + i);
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < kAggBatch; ++u) {
                    const int64_t i = off + u * 64 + lane;
                    r[u][0] = kDead; r[u][1] = 0;
                    if (i < nrec) r[u] = __builtin_nontemporal_load(recs + base_i + i);
                }
            }
            vb += NW;
            return true;
        };
        u64x2 cur[kAggBatch], nxt[kAggBatch];
        bool have = load_batch(cur);
        while (have) {
            const bool nhave = load_batch(nxt);
            uint64_t hk[kAggBatch], val[kAggBatch];
            uint32_t cnt[kAggBatch], pending = 0;
            if (compact) {   // key word -> cnt : 8 | the key's image below its partition bits
#pragma unroll
                for (int u = 0; u < kAggBatch; ++u)
                    if (cur[u][0] != kDead) {
                        const uint32_t kw = (uint32_t)cur[u][0];
                        const uint64_t h39 = ((uint64_t)p << 31) | (kw & 0x7FFFFFFFu);
                        cur[u][0] = ((uint64_t)(kw >> 31) << 56) | (g2c_image(h39) & kKeyMask);
                    }
            }
#pragma unroll
            for (int u = 0; u < kAggBatch; ++u) {
                hk[u] = (cur[u][0] & kKeyMask) | ptop;
                val[u] = cur[u][1];
                cnt[u] = (uint32_t)(cur[u][0] >> (64 - kG2PartBits));
                if (cur[u][0] != kDead) pending |= 1u << u;
            }
            if (a.nwork > 0) wave_combine<kAggBatch>(a.op, a.vcls, hk, val, cnt, pending);   // skewed input: also the partitions that were not cut hold keys that fill a third of a wave
            if (single) tab_upsert<kAggBatch, kG2PartBits>(t, a.op, a.vcls, a.has_values != 0, 0u, (uint32_t)kG2Slots, hk, val, cnt, pending, err, 32u);
            else tab_upsert_b4<kAggBatch, kG2PartBits>(t, a.op, a.vcls, a.has_values != 0, hk, val, cnt, pending, err, 32u);   // 32: this partition's LDS table is full — says nothing about max_groups, the host retries on the HBM table
            if (nhave) {
#pragma unroll
                for (int u = 0; u < kAggBatch; ++u) cur[u] = nxt[u];
            }
            have = nhave;
        }
        __syncthreads();
        if (multi) {   // one of several items of a big partition: its groups meet the other items' in the global table
            for (int k = tid; k < kG2Slots; k += kG2AggBlock) {
                if (t.keys[k] == kFree) continue;
                const uint64_t key = compact ? g2c_unhash(t.keys[k], a.key_base) : g2_unhash(t.keys[k]);
                if (key == kGroupEmpty) g2_global_special(a.t, 0, a.op, a.vcls, a.has_values != 0, t.acc[k], t.cnt[k]);
                else if (!g2_global_upsert(a.t, key, a.op, a.vcls, a.has_values != 0, t.acc[k], t.cnt[k])) err |= 4u;
            }
            __syncthreads();
            continue;
        }
        if (tid == 0) { misc[0] = atomicAdd(a.cursor, *t.ngroups); misc[1] = 0; }
        __syncthreads();
        for (int k = tid; k < kG2Slots; k += kG2AggBlock) {
            if (t.keys[k] == kFree) continue;
            const unsigned idx = misc[0] + atomicAdd(&misc[1], 1u);
            if ((int64_t)idx >= a.max_out) { err |= 4u; continue; }
            const uint64_t key = compact ? g2c_unhash(t.keys[k], a.key_base) : g2_unhash(t.keys[k]);
            switch (a.key_dtype) {
                case RDF_I32: case RDF_U32: ((uint32_t*)a.out_keys)[idx] = (uint32_t)key; break;
                case RDF_I16: case RDF_U16: ((uint16_t*)a.out_keys)[idx] = (uint16_t)key; break;
                case RDF_I8: case RDF_U8: ((uint8_t*)a.out_keys)[idx] = (uint8_t)key; break;
                default: ((uint64_t*)a.out_keys)[idx] = key; break;
            }
            a.out_acc[idx] = t.acc[k];
            a.out_counts[idx] = (int64_t)t.cnt[k];
        }
        __syncthreads();
    }
    if (err) atomicOr(a.flags, err);
}

// ------------------------------------------------------------------------------------------------
// one table in HBM: rows -> table (any number of groups), and (key, accumulator, count) triples -> table (merge)

__global__ __launch_bounds__(kStreamBlock) void gb2_table_rows_kernel(const Gb2Args a) {
    const int tid = threadIdx.x;
    const bool has_values = a.value_dtype >= 0;
    uint32_t err = 0;
    constexpr int TPI = kStreamRows * kStreamBlock / kEvalTile;
    for (int64_t st = (int64_t)blockIdx.x * TPI; st < a.ntiles; st += (int64_t)gridDim.x * TPI) {
        G2Raw<kStreamRows> raw;
        g2_load<kStreamRows, kStreamBlock>(a, st, tid, raw);
        G2Rows<kStreamRows> rows;
        g2_prepare<kStreamRows>(a, raw, rows);
#pragma unroll
        for (int j = 0; j < kStreamRows; ++j)
            if ((rows.live >> j) & 1) {
                const uint64_t key = g2_unhash(rows.hk[j]);
                const uint64_t c = (rows.cnt >> j) & 1;
                if (key == kGroupEmpty) g2_global_special(a.t, 0, a.op, a.vcls, has_values, rows.val[j], c);
                else if (!g2_global_upsert(a.t, key, a.op, a.vcls, has_values, rows.val[j], c)) err |= 4u;
            }
    }
    if (err) atomicOr(a.flags, err);
}
__global__ __launch_bounds__(kBlock) void gb2_merge_kernel(const Gb2MergeArgs a) {
    uint32_t err = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * kBlock) {
        const uint64_t key = a.keys[i], v = a.acc ? a.acc[i] : 0, c = a.counts ? (uint64_t)a.counts[i] : 1;
        const bool knull = a.key_validity && !((a.key_validity[i >> 3] >> (i & 7)) & 1);
        if (knull) g2_global_special(a.t, 1, a.op, a.vcls, a.acc != nullptr, v, c);
        else if (key == kGroupEmpty) g2_global_special(a.t, 0, a.op, a.vcls, a.acc != nullptr, v, c);
        else if (!g2_global_upsert(a.t, key, a.op, a.vcls, a.acc != nullptr, v, c)) err |= 4u;
    }
    if (err) atomicOr(a.t.flags, err);
}
// occupied slots -> dense (keys, raw accumulators, counts); the two special groups last
__global__ __launch_bounds__(kBlock) void gb2_emit_kernel(const GroupEmitArgs a) {
    const int64_t n = a.t.capacity + 2;
    for (int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x; s < n; s += (int64_t)gridDim.x * kBlock) {
        bool occ; uint64_t key = 0; bool knull = false;
        if (s < a.t.capacity) { key = a.t.keys[s]; occ = key != kGroupEmpty; }
        else if (s == a.t.capacity) { key = kGroupEmpty; occ = a.t.special[0] != 0; }
        else { knull = true; occ = a.t.special[1] != 0; }
        if (!occ) continue;
        const unsigned idx = atomicAdd(a.cursor, 1u);
        switch (a.key_dtype) {
            case RDF_I32: case RDF_U32: ((uint32_t*)a.out_keys)[idx] = (uint32_t)key; break;
            case RDF_I16: case RDF_U16: ((uint16_t*)a.out_keys)[idx] = (uint16_t)key; break;
            case RDF_I8: case RDF_U8: ((uint8_t*)a.out_keys)[idx] = (uint8_t)key; break;
            default: ((uint64_t*)a.out_keys)[idx] = key; break;
        }
        if (a.out_keys_validity && !knull) atomicOr((unsigned int*)a.out_keys_validity + (idx >> 5), 1u << (idx & 31));
        ((uint64_t*)a.out_sums)[idx] = a.t.sums[s];
        a.out_counts[idx] = (int64_t)a.t.counts[s];
    }
}

// raw accumulators <-> values: MIN / MAX un-order their images, a group without a non-NULL value is NULL
__global__ __launch_bounds__(kBlock) void gb2_finish_kernel(const Gb2FinishArgs a) {
    const int lane = threadIdx.x & 63;
    unsigned long long nulls = 0;
    const int64_t n64 = (a.n + 63) & ~(int64_t)63;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n64; i += (int64_t)gridDim.x * kBlock) {
        const bool in = i < a.n;
        bool valid = in;
        if (in) {
            const int64_t c = a.counts ? a.counts[i] : 1;
            uint64_t v = a.acc[i];
            if (a.native_in) v = to_table_form(a.op, a.vcls, v, c == 0);
            else if (a.op != AGG_SUM) { valid = c > 0; v = valid ? unord_bits(a.vcls, v) : 0; }
            a.acc[i] = v;
        }
        const uint64_t vb = __ballot(valid);
        if (a.validity && !a.native_in && lane == 0 && i < a.n) ((uint64_t*)a.validity)[i >> 6] = vb;
        if (lane == 0 && !a.native_in) nulls += (unsigned long long)__popcll(__ballot(in) & ~vb);
    }
    if (lane == 0 && nulls && a.nulls) atomicAdd(a.nulls, nulls);
}
__global__ void gb2_fill_kernel(uint64_t* p, int64_t n, uint64_t v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

// ------------------------------------------------------------------------------------------------
// multi-GPU exchange of partial groups (the one real collective of the path): owner = hash(key) % world

__device__ __forceinline__ uint32_t gx_owner(uint64_t key, int world) { return (uint32_t)(((key * 0x9E3779B97F4A7C15ull) >> 33) % (uint64_t)world); }
__global__ __launch_bounds__(kBlock) void gx_pack_kernel(const GxPackArgs a) {
    __shared__ unsigned int lc[64];
    __shared__ unsigned long long lbase[64];
    if (threadIdx.x < 64) lc[threadIdx.x] = 0;
    __syncthreads();
    // a block handles a contiguous slice so that phase 1 needs one global atomic per (block, owner)
    const int64_t per = (a.n + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < a.n ? lo + per : a.n;
    for (int64_t i = lo + threadIdx.x; i < hi; i += kBlock) atomicAdd(&lc[gx_owner(a.keys[i], a.world)], 1u);
    __syncthreads();
    if (a.phase == 0) {
        if (threadIdx.x < a.world && lc[threadIdx.x]) atomicAdd(&a.owner_counts[threadIdx.x], (unsigned long long)lc[threadIdx.x]);
        return;
    }
    if (threadIdx.x < a.world) { lbase[threadIdx.x] = lc[threadIdx.x] ? atomicAdd(&a.cursors[threadIdx.x], (unsigned long long)lc[threadIdx.x]) : 0; lc[threadIdx.x] = 0; }
    __syncthreads();
    for (int64_t i = lo + threadIdx.x; i < hi; i += kBlock) {
        const uint64_t k = a.keys[i];
        const uint32_t o = gx_owner(k, a.world);
        const unsigned long long pos = lbase[o] + atomicAdd(&lc[o], 1u);
        a.packed[a.words * pos] = k;
        a.packed[a.words * pos + 1] = a.acc[i];
        if (a.words == 3) a.packed[3 * pos + 2] = (uint64_t)a.counts[i];
    }
}
// exclusive scan of the per-owner counts into the pack's write cursors — on the device, so that the scatter phase can be queued
// behind the count phase without the host looking at the counts first (rdf_groupby_agg_dist: the counts travel to the peers
// meanwhile).  counts[world] is left intact: it is this rank's row of the all-gathered split-size matrix.
__global__ void gx_scan_kernel(const unsigned long long* counts, unsigned long long* cursors, int world) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        unsigned long long run = 0;
        for (int r = 0; r < world; ++r) { cursors[r] = run; run += counts[r]; }
    }
}
__global__ __launch_bounds__(kBlock) void gx_unpack_kernel(const uint64_t* packed, int64_t n, uint64_t* keys, uint64_t* acc, int64_t* counts, int words) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        keys[i] = packed[words * i]; acc[i] = packed[words * i + 1];
        if (words == 3) counts[i] = (int64_t)packed[3 * i + 2];
    }
}

// ------------------------------------------------------------------------------------------------
// several grouping columns <-> one packed 64-bit key

__device__ __forceinline__ uint64_t key_order_bits(int dt, uint64_t raw) {   // order-preserving unsigned image of an integer key
    return dt <= RDF_I64 ? normalize_int(dt, raw) ^ 0x8000000000000000ull : normalize_int(dt, raw);
}
__global__ __launch_bounds__(kBlock) void key_pack_kernel(const KeyPackArgs a) {
    const double inv = chunk_lookup_scale(a.chunk_row_start, a.nchunks);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * kBlock) {
        const int64_t c = find_chunk_row(a.chunk_row_start, a.nchunks, i, inv);
        const int64_t r = i - a.chunk_row_start[c];
        uint64_t packed = 0;
        for (int k = 0; k < a.nkeys; ++k) {
            const DevChunkCol cc = a.cols[(int64_t)k * a.nchunks + c];
            const int64_t e = cc.offset + r;
            const bool valid = !cc.validity || ((as_global<uint8_t>(cc.validity)[e >> 3] >> (e & 7)) & 1);
            uint64_t f = 0;
            if (valid) {
                const uint64_t ob = key_order_bits(a.dtype[k], g2_load_raw(cc.values, g2_dtype_size(a.dtype[k]), e, true));
                if (a.dict[k]) {   // rank in the column's sorted dictionary (every value of the column is in it)
                    int64_t lo = 0, hi = a.dict_n[k] - 1;
                    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (a.dict[k][mid] < ob) lo = mid + 1; else hi = mid; }
                    f = (uint64_t)lo + (uint64_t)a.nullable[k];
                } else {
                    f = ob - a.bias[k] + (uint64_t)a.nullable[k];
                }
            }
            packed |= f << a.shift[k];
        }
        a.packed[i] = packed;
    }
}
__global__ __launch_bounds__(kBlock) void key_unpack_kernel(const KeyPackArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t n64 = (a.n + 63) & ~(int64_t)63;
    unsigned long long nulls[kMaxKeyCols] = {0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n64; i += (int64_t)gridDim.x * kBlock) {
        const bool in = i < a.n;
        const uint64_t packed = in ? a.packed[i] : 0;
#pragma unroll
        for (int k = 0; k < kMaxKeyCols; ++k) {
            if (k >= a.nkeys) break;
            const uint64_t f = (packed >> a.shift[k]) & a.mask[k];
            const bool valid = in && !(a.nullable[k] && f == 0);
            uint64_t ob = 0;
            if (valid) ob = a.dict[k] ? a.dict[k][f - (uint64_t)a.nullable[k]] : f - (uint64_t)a.nullable[k] + a.bias[k];
            if (valid && a.dtype[k] <= RDF_I64) ob ^= 0x8000000000000000ull;
            if (in) switch (g2_dtype_size(a.dtype[k])) {
                case 8: ((uint64_t*)a.out_values[k])[i] = ob; break;
                case 4: ((uint32_t*)a.out_values[k])[i] = (uint32_t)ob; break;
                case 2: ((uint16_t*)a.out_values[k])[i] = (uint16_t)ob; break;
                default: ((uint8_t*)a.out_values[k])[i] = (uint8_t)ob; break;
            }
            const uint64_t vb = __ballot(valid);
            if (lane == 0 && i < a.n) {
                if (a.out_validity[k]) ((uint64_t*)a.out_validity[k])[i >> 6] = vb;
                nulls[k] += (unsigned long long)__popcll(__ballot(in) & ~vb);
            }
        }
    }
    if (lane == 0)
        for (int k = 0; k < a.nkeys; ++k) if (nulls[k]) atomicAdd(&a.out_nulls[k], nulls[k]);
}

// ------------------------------------------------------------------------------------------------
// launchers

size_t gb2_scatter_lds_bytes(bool compact) {
    const size_t bookkeeping = (size_t)kMaxFlushLines * 8 + (size_t)kP * 4 * 5;
    if (compact) return ((size_t)kG2Super + (size_t)kP * (kG2LineC - 1)) * 12 + 8 + bookkeeping;   // 78 KB: still two blocks per CU
    return (size_t)kG2Super * 16 + (size_t)kP * (kG2Line - 1) * 16 + bookkeeping;
}
hipError_t launch_gb2_stream(const Gb2Args& a, int grid, hipStream_t s) {
    const size_t lds = (size_t)a.table_slots * 20 + 16 + 48 + (a.hot_mode ? 8192 + 2048 + 16 : 0);   // table, group counter, the two special groups (, the hot classes' bitmap and the hot keys)
    if (a.fast) {
        (void)hipFuncSetAttribute((const void*)gb2_stream_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(gb2_stream_kernel<true>, dim3(grid), dim3(kStreamBlock), lds, s, a);
    } else {
        (void)hipFuncSetAttribute((const void*)gb2_stream_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(gb2_stream_kernel<false>, dim3(grid), dim3(kStreamBlock), lds, s, a);
    }
    return hipGetLastError();
}
hipError_t launch_gb2_skew_probe(const Gb2Args& a, int64_t tile_step, unsigned int* hist, unsigned long long* minmax, unsigned int* hbins, unsigned long long* ktab, hipStream_t s) {
    const int64_t sampled = (a.ntiles + tile_step - 1) / tile_step;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((sampled + 3) / 4, 1024));
    hipLaunchKernelGGL(gb2_skew_probe_kernel, dim3(grid), dim3(256), 0, s, a, tile_step, hist, minmax, hbins, ktab);
    return hipGetLastError();
}
hipError_t launch_gb2_scatter(const Gb2Args& a, int grid, hipStream_t s) {
    const size_t lds = gb2_scatter_lds_bytes(a.compact != 0) + (a.hot_mode ? 8192 + 2048 : 0);   // (+ the hot classes' bitmap and the hot keys: 79 KB, still two blocks per CU)
    auto go = [&](auto kernel) {
        (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(kG2Block), lds, s, a);
    };
    if (a.compact) { if (a.fast) go(gb2_scatter_kernel<true, true>); else go(gb2_scatter_kernel<false, true>); }
    else { if (a.fast) go(gb2_scatter_kernel<true, false>); else go(gb2_scatter_kernel<false, false>); }
    return hipGetLastError();
}
hipError_t launch_gb2_aggregate(const Gb2AggArgs& a, hipStream_t s) {
    if (a.nb > kAggMaxRegions) return hipErrorInvalidValue;   // (the host plans at most two scatter blocks per CU)
    const size_t lds = (size_t)kG2Slots * 20 + 32 + (size_t)(2 * kAggMaxRegions + 1) * 4;
    (void)hipFuncSetAttribute((const void*)gb2_aggregate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    // work list (skewed keys): one block per CU walks the items with the grid's stride — the list is sorted largest first
    hipLaunchKernelGGL(gb2_aggregate_kernel, dim3(a.nwork > 0 ? std::min(a.nwork, eval_grid_limit() / 8) : kP), dim3(kG2AggBlock), lds, s, a);
    return hipGetLastError();
}
static int g2_rows_grid(int64_t n) {
    int64_t g = (n + kBlock - 1) / kBlock;
    if (g > eval_grid_limit()) g = eval_grid_limit();
    return g < 1 ? 1 : (int)g;
}
hipError_t launch_gb2_merge(const Gb2MergeArgs& a, hipStream_t s) {
    if (a.n > 0) hipLaunchKernelGGL(gb2_merge_kernel, dim3(g2_rows_grid(a.n)), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_gb2_table_rows(const Gb2Args& a, hipStream_t s) {
    constexpr int TPI = kStreamRows * kStreamBlock / kEvalTile;
    int64_t grid = (a.ntiles + TPI - 1) / TPI;
    if (grid > eval_grid_limit() / 2) grid = eval_grid_limit() / 2;
    if (grid > 0) hipLaunchKernelGGL(gb2_table_rows_kernel, dim3((unsigned)grid), dim3(kStreamBlock), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_gb2_emit(const GroupEmitArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(gb2_emit_kernel, dim3(g2_rows_grid(a.t.capacity + 2)), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_gb2_finish(const Gb2FinishArgs& a, hipStream_t s) {
    if (a.n > 0) hipLaunchKernelGGL(gb2_finish_kernel, dim3(g2_rows_grid(a.n)), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_gb2_fill(uint64_t* p, int64_t n, uint64_t v, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(gb2_fill_kernel, dim3(g2_rows_grid(n)), dim3(kBlock), 0, s, p, n, v);
    return hipGetLastError();
}
hipError_t launch_gx_pack(const GxPackArgs& a, hipStream_t s) {
    if (a.n > 0) {
        int64_t grid = (a.n + 4 * kBlock - 1) / (4 * kBlock);
        if (grid > 1024) grid = 1024;
        hipLaunchKernelGGL(gx_pack_kernel, dim3((unsigned)grid), dim3(kBlock), 0, s, a);
    }
    return hipGetLastError();
}
hipError_t launch_gx_scan(const unsigned long long* counts, unsigned long long* cursors, int world, hipStream_t s) {
    hipLaunchKernelGGL(gx_scan_kernel, dim3(1), dim3(64), 0, s, counts, cursors, world);
    return hipGetLastError();
}
hipError_t launch_gx_unpack(const uint64_t* packed, int64_t n, uint64_t* keys, uint64_t* acc, int64_t* counts, int words, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(gx_unpack_kernel, dim3(g2_rows_grid(n)), dim3(kBlock), 0, s, packed, n, keys, acc, counts, words);
    return hipGetLastError();
}
hipError_t launch_key_pack(const KeyPackArgs& a, hipStream_t s) {
    if (a.n > 0) hipLaunchKernelGGL(key_pack_kernel, dim3(g2_rows_grid(a.n)), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_key_unpack(const KeyPackArgs& a, hipStream_t s) {
    if (a.n > 0) hipLaunchKernelGGL(key_unpack_kernel, dim3(g2_rows_grid(a.n)), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

}  // namespace rdfk
