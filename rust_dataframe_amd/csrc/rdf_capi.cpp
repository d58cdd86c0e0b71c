// rdf_capi.cpp — host side of librdf_mi355x.so: the extern "C" boundary of include/rdf_mi355x.h.
//
// Per calling thread: one context (device, stream, HBM arena, pinned staging buffer, last error).
// RDF_MEM_HOST arrays are staged into the arena (small chunks packed through one pinned buffer and
// ONE H2D copy — the reference reads CSV/JSON/Parquet in 1024-row batches, src/dataframe.rs:352 —
// large chunks DMA'd directly), computed by the kernels of rdf_kernels.hip, and copied back.
// RDF_MEM_DEVICE arrays are used in place.  There is no CPU compute path in this file.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <chrono>
#include <functional>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <condition_variable>
#include <string>
#include <thread>
#include <vector>
#include <sys/syscall.h>
#include <unistd.h>

#include "rdf_device.h"
#include "rdf_stage_copy.h"

using namespace rdfk;

namespace {

// ------------------------------------------------------------------------------------------------
// context

struct Arena {
    char*  base = nullptr;
    size_t cap = 0, used = 0, wanted = 0;
    std::vector<void*> overflow;
};

struct Ctx {
    bool        ready = false;
    int         device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;  // own_stream or the caller's
    hipStream_t copy_stream = nullptr;   // ingestion uploads (rdf_copy_h2d_async): overlap with parsing on the host and with kernels
    bool copy_pending = false;
    Arena       arena;
    char*       pinned = nullptr;
    size_t      pinned_cap = 0;
    std::string err;
    std::string last_kernel;       // name of the dominant kernel of the last compute call
    // tunables (rdf_set_option)
    bool   opt_spec = true;        // specialised straight-line kernels (rdf_spec.hip)
    bool   opt_fast_filter = true; // filter_agg_f64_kernel (handles 8-byte-misaligned columns)
    bool   opt_vec_bitmap = false; // validity words of the specialised kernels: scalar loads (default since the wave-granular tiles: 1.30 vs 1.355 ms on the 1e9-row filter->sum with nulls) or one vector load by lanes 0..NW + readlane (A/B; it was the faster one with block-wide tiles)
    bool   opt_filter_one = true;   // one-chunk compaction kernel with its descriptors in the kernel arguments (A/B)
    int    opt_filter_gen = 2;      // compaction kernels: 2 = wave-granular tiles (rdf_filter.hip, default), 1 = first-generation block tiles (A/B)
    int    opt_filter_block = 1;    // one-pass compaction of LONG batches on block tiles held in registers, prefixes from a scanner wave (rdf_bfilter.hip, round 6; default); 0 = the wave-tile kernels only (A/B)
    int    opt_interp_lean = 1;             // interpreted aggregate programs whose every step has a lean handler run on eval_lean_kernel (rdf_eval_lean.hip); 0: always eval_kernel (the A/B)
    int    opt_filter_owned = 1;    // long batches, many of them, none a large share of the frame: a block draws whole batches and adds up its own offsets (round 6; default); 0 = tiles by ticket, offsets from the scanner wave (A/B)
    int    opt_filter_ends = 1;     // the wave-tile one-pass kernel: tiles at the END of a batch take the LDS-DMA path too (clamped addresses; frames whose batch lengths are not multiples of 1024 rows; round 6, default); 0 = they load row by row (A/B)
    int    opt_filter_mixed = 0;    // rdf_filter_frame over frames of 8- AND 4-byte columns: the block kernel twice (the predicate's width first, writing the kept rows into the frame's mask; the other width by that mask; round 6: built, parity-tested, measured 2 - 20 % BEHIND the wave-tile kernel on multiples of 1024 rows and — once that kernel's end-of-batch tiles took the DMA path, filter_ends — behind it on ragged lengths too: not the default) — 0 = never (default); 1 = for short batches whose rows mostly lie in partial 1024-row tiles; 2 = wherever the block kernel's forms apply (tests, A/B)
    int    opt_filter_short = 1;    // batches / chunks no longer than a block tile on the block kernel's short-batch mode (round 6; default); 0 = the wave-tile kernels (A/B)
    int    opt_filter_block_rows = 8192;    // ... for frames whose mean batch length is at least this many rows (one block tile of a single 8-byte column; measured ahead of the wave-tile kernel from 8192-row batches up: profiles/r06_filter_frame_batch_length_sweep.jsonl)
    int    opt_filter_fused = 1;    // rdf_filter_frame: `col CMP literal [AND|OR col CMP literal]` predicates evaluated inside the compaction kernel, one pass (1, default); 0 = predicate -> mask, count, compact (A/B)
    int    opt_filter_lookback = 3; // one-pass rdf_filter_frame, batches longer than a tile: 3 = a super-tile's first tile finds the rows in front of the super-tile for all 64, from the nearest super-tiles' tile counts and the older ones' totals (default); 2 = from totals only; 1 = every tile walks the totals (round 4; batches of at most 1024 tiles) — A/B
    int    opt_filter_tile = 0;     // 0: compaction tile chosen from the mean chunk length; 1024 / 4096 force one (A/B)
    int    opt_gb_debug = 0;        // ablations of the partitioned GROUP BY (tools/bench_kernels.py): 1 = aggregate without LDS work, 2 = scatter without stores
    int    opt_sort_gen = 3;        // radix passes: 3 = one read + one write of the pairs per digit, decoupled look-back between 4096-pair tiles (rdf_sort.hip, default); 2 = count -> scan -> scatter over static tile ranges with the same wave-ranked tiles (A/B: slower, see rdf_sort.hip); 1 = first generation (rdf_kernels.hip)
    bool   sort_used_local = false; // the last sort finished at least one column with os_local_kernel
    int    opt_sort_super = 1;      // the digit passes of rdf_sort.hip: 1 = a tile per ticket, decoupled look-back between tiles (os_scatter_kernel, default); K > 1 = a ticket is up to K consecutive tiles, counted together, ONE look-back, then ranked and written one by one (os_scatter4_kernel, round 6: built, correct, measured level at 1e9 i64 keys and 8-12 % SLOWER on 5e7 f64 / two-key sorts — profiles/r06_sort_super_tiles_ab.jsonl — kept as the A/B: the second read of the keys costs what the shorter wait saves)
    int    opt_sort_super_force = 0;    // tests: this many tiles per ticket whatever the input's size (rdf_set_option("sort_super", 100 + K))
    int    opt_sort_pipe = 0;       // the digit passes of rdf_sort.hip: 0 = decoupled look-back between the tiles (os_scatter_kernel, default); 1 = a tile's digit counts are published one iteration before its offsets are asked for and scanner blocks turn counts into offsets (os_scatter3_kernel, round 6: built, correct, measured 4-16 % SLOWER — profiles/r06_sort_digit_pass_ab.jsonl — kept as the A/B)
    int    opt_sort_msd = 1;        // sort keys that vary in more than 32 bits: passes over the top bits, then every bucket sorted in LDS (1, default); 0 = one pass per byte (A/B)
    int    opt_sort_sample = 1;     // doubles: value buckets planned from a sample of the keys (range without outliers, bucket bits from the densest region); 0 = [min, max] and ~500 rows per bucket (round 3, A/B)
    int    opt_join_table = 2;      // equi-join on one key column: probe a table of the distinct build keys — 2 (default, round 5): the build side sorted by hash, the table placed by a scan (no atomics); 1: sorted by key, slots claimed by compare-and-swap (round 3); 0 = the bucket index over the sorted build keys (A/B)
    int    opt_jit = 1;             // a program shape outside the catalogs: 1 = compile spec_kernel<Prog> for it on a helper thread (the interpreter answers until the kernel is ready; a code object in the cache directory is loaded at once), default; 2 = the call waits for the compiler; 0 = always the interpreter
    int    opt_take_rows = 1;       // take over a frame through interleaved row records: 1 = when the transaction model says so (default), 0 = never, 2 = always (tests, A/B)
    int    opt_gb_skew_plan = 1;    // skewed keys: per-partition region sizes + big partitions cut into several aggregate items (1 = when the probe finds skew, default; 0 = the first-generation combining path instead, A/B; 2 = always, tests)
    int    opt_gb_compact = 1;      // partition path, keys inside a window of 2^39: 1 = 4-byte records when only rows are counted (default); 2 = also 12-byte records (key word + value) for the other aggregates (measured slower than 16-byte records, kept for A/B); 0 = 16-byte records always
    int64_t opt_comm_max_bytes = 0;  // group-by exchange: most bytes one ncclSend / peer copy moves (0 = 256 MiB); larger shares travel in several rounds
    int    opt_spec_blocks = 0;     // resident blocks of the specialised kernels per CU (persistent grid = CUs x this); 0 = by the program (4 / 5 / 8, see run_program)
    int    opt_spec_tile_rot = -1;  // specialised kernels' tile walk: wave p of row i takes tile i * S + (p + i * 4 * this) mod S (S = the grid's waves); 0 = plain grid stride (SpecArgs::tile_rot); -1 = by the program (run_program)
    int    opt_spec_xcd_swz = -1;   // 1: XCD x (block index mod 8) walks the x-th contiguous eighth of every row of tiles (SpecArgs::xcd_swz); 0: plain; -1 = by the program
    int    opt_spec_grid_adj = 0;   // added to the specialised kernels' persistent grid (A/B of grids that are not a multiple of the CU count)
    int    opt_gspec_blocks = 0;    // resident blocks per CU of the grouped register-accumulator kernel (0 = 2: what its ~220 VGPRs allow)
    int    opt_gb_hot = 1;          // skewed keys: 1 = heavy-hitter split (the hot hash classes through gb2_stream_kernel, the scatter path over the rest), default; 0 = capacity plan / first-generation path as in round 3 (A/B)
    int    opt_gb_bucket = 0;       // partition tables of the aggregate pass: 4 = four keys per 32-byte bucket, 1 = one key per probe, 0 = by the sampled key range (default: one key per probe for keys packed into <= 4 x max_groups values, buckets otherwise)
    int    opt_gb_partition = 3;    // hash GROUP BY: 3 = second generation (rdf_groupby.hip: stream / line-aligned scatter / table by max_groups, default), 4 = its partition path whatever max_groups says, 1 = first-generation histogram + scatter, 2 = first-generation radix sort, 0 = one table in HBM
    // kernel timing (bench.py roofline leg)
    bool   timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t events_used = 0;
    // buffers of released frames, kept for the next frame-level operator (size -> pointer): the outputs of
    // rdf_filter_frame / rdf_take_frame / ... are GBs, and a hipMalloc + hipFree pair per call costs more than the kernels
    std::multimap<size_t, void*> pool_free;
    size_t pool_cached = 0;
    // streamed batch loop over host-resident frames (rdf_capi_stream.inc): two slab buffers in HBM, two page-locked staging buffers
    int64_t opt_stream_slab = 0;    // bytes per slab (0 = 256 MiB); rdf_pipeline streams host inputs above one slab; -1 = never
    void*  sbuf_dev[2] = {nullptr, nullptr};
    void*  sbuf_pin[2] = {nullptr, nullptr};
    size_t sbuf_dev_cap = 0, sbuf_pin_cap = 0;
    hipEvent_t sbuf_ev[2] = {nullptr, nullptr};
    // the way back of the streamed batch loop (sinks that materialise on the host): two output slabs in HBM, two page-locked
    // staging buffers, a third stream for the device-to-host copies
    void*  dbuf_dev[2] = {nullptr, nullptr};
    void*  dbuf_pin[2] = {nullptr, nullptr};
    size_t dbuf_dev_cap = 0, dbuf_pin_cap = 0;
    hipEvent_t dbuf_ev[2] = {nullptr, nullptr};
    hipStream_t d2h_stream = nullptr;
    int64_t stream_slabs = 0, stream_bytes_staged = 0, stream_bytes_direct = 0;   // what the last rdf_pipeline call streamed
    bool in_stream = false;          // a streamed call is running its per-slab calls on this thread (they are not streamed again)
    bool streaming = false;          // stream_run is active: slabs are in flight on the copy engines (frame_table_upload)
    int  opt_stream_table_kernel = 1;   // 1 = while streaming, a frame's tables reach the device through a kernel instead of the copy engine (A/B: 0)
    char* tab_pin = nullptr; size_t tab_pin_cap = 0, tab_pin_used = 0;   // page-locked ring the tables are staged in for that kernel
    bool   agg_comm_entered = false;  // ... and this call has entered agg_dist_finish's collectives (a rank that fails before them still has to: pipeline_dist)
    ::rdf_comm* agg_comm = nullptr;   // rdf_pipeline_dist: the next aggregate's device-resident partials are all-gathered and folded on the device before the host reads anything
    ~Ctx();
};

thread_local Ctx g_ctx;

// A worker thread that exits (the reference runs its kernels on rayon workers, src/functions/scalar.rs:28,99) gives back
// its arena, pinned staging buffer, events and stream.  The main thread's context is left to process teardown: its
// destructor runs inside exit(), where calling into a HIP runtime that may already be shutting down is not worth the risk.
Ctx::~Ctx() {
    if (!ready || (long)syscall(SYS_gettid) == (long)getpid()) return;
    (void)hipSetDevice(device);
    if (stream) (void)hipStreamSynchronize(stream);
    for (void* p : arena.overflow) (void)hipFree(p);
    if (arena.base) (void)hipFree(arena.base);
    if (pinned) (void)hipHostFree(pinned);
    for (auto& kv : pool_free) (void)hipFree(kv.second);
    for (int b = 0; b < 2; ++b) { if (sbuf_dev[b]) (void)hipFree(sbuf_dev[b]); if (sbuf_pin[b]) (void)hipHostFree(sbuf_pin[b]); if (sbuf_ev[b]) (void)hipEventDestroy(sbuf_ev[b]); }
    if (tab_pin) (void)hipHostFree(tab_pin);
    if (d2h_stream) (void)hipStreamSynchronize(d2h_stream);
    for (int b = 0; b < 2; ++b) { if (dbuf_dev[b]) (void)hipFree(dbuf_dev[b]); if (dbuf_pin[b]) (void)hipHostFree(dbuf_pin[b]); if (dbuf_ev[b]) (void)hipEventDestroy(dbuf_ev[b]); }
    if (d2h_stream) (void)hipStreamDestroy(d2h_stream);
    for (auto& ev : events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    if (copy_stream) { (void)hipStreamSynchronize(copy_stream); (void)hipStreamDestroy(copy_stream); }
    if (own_stream) (void)hipStreamDestroy(own_stream);
    ready = false;
}

rdf_status fail(rdf_status st, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_ctx.err = buf;
    // an error return may leave asynchronous copies out of the pinned staging buffer (or into caller memory) queued: drain
    // them, so that the next call — or the caller freeing its buffers — cannot race them
    if (g_ctx.ready && g_ctx.stream) (void)hipStreamSynchronize(g_ctx.stream);
    return st;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) return fail(RDF_DEVICE_ERROR, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)
#define RDF_TRY(expr)                  \
    do {                               \
        rdf_status _s = (expr);        \
        if (_s != RDF_OK) return _s;   \
    } while (0)

rdf_status ensure_ready() {
    Ctx& c = g_ctx;
    if (c.ready) return RDF_OK;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(RDF_DEVICE_ERROR, "no HIP device visible (%s); librdf_mi355x has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    HIP_TRY(hipGetDevice(&c.device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, c.device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RDF_DEVICE_ERROR, "device %d is %s; this library carries gfx950 (MI355X) code only", c.device, prop.gcnArchName);
    HIP_TRY(hipStreamCreateWithFlags(&c.own_stream, hipStreamNonBlocking));
    c.stream = c.own_stream;
    if (const char* e = getenv("RDF_SPEC_BLOCKS_PER_CU")) { const int v = atoi(e); if (v >= 0 && v <= 8) c.opt_spec_blocks = v; }   // (A/B without touching the caller)
    if (const char* e = getenv("RDF_SPEC_TILE_ROT")) c.opt_spec_tile_rot = atoi(e);
    if (const char* e = getenv("RDF_SPEC_XCD_SWZ")) c.opt_spec_xcd_swz = atoi(e);
    if (const char* e = getenv("RDF_SPEC_GRID_ADJ")) c.opt_spec_grid_adj = atoi(e);
    if (const char* e = getenv("RDF_STREAM_TABLE_KERNEL")) c.opt_stream_table_kernel = atoi(e) != 0;
    if (const char* e = getenv("RDF_GSPEC_BLOCKS_PER_CU")) { const int v = atoi(e); if (v >= 0 && v <= 8) c.opt_gspec_blocks = v; }
    c.ready = true;
    return RDF_OK;
}

// ---- arena: bump allocator over one HBM block, regrown between calls ----
void arena_begin() {
    Arena& a = g_ctx.arena;
    for (void* p : a.overflow) (void)hipFree(p);
    a.overflow.clear();
    if (a.wanted > a.cap) {
        if (a.base) (void)hipFree(a.base);
        a.base = nullptr;
        a.cap = 0;
        size_t want = a.wanted + a.wanted / 4 + (1u << 20);
        void* p = nullptr;
        if (hipMalloc(&p, want) == hipSuccess) { a.base = (char*)p; a.cap = want; }
    }
    a.used = 0;
    a.wanted = 0;
}
rdf_status arena_alloc(size_t bytes, void** out) {
    Arena& a = g_ctx.arena;
    bytes = (bytes + 16 + 255) & ~(size_t)255;  // 16-byte tail pad for 8-byte bitmap window reads
    a.wanted += bytes;
    if (a.used + bytes <= a.cap) {
        *out = a.base + a.used;
        a.used += bytes;
        return RDF_OK;
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return fail(RDF_MEMORY_ERROR, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    a.overflow.push_back(p);
    *out = p;
    return RDF_OK;
}
rdf_status pinned_reserve(size_t bytes) {
    Ctx& c = g_ctx;
    if (bytes <= c.pinned_cap) return RDF_OK;
    // growing: copies still in flight may be reading the old buffer
    if (c.pinned && c.stream) (void)hipStreamSynchronize(c.stream);
    if (c.pinned) (void)hipHostFree(c.pinned);
    c.pinned = nullptr;
    c.pinned_cap = 0;
    size_t want = bytes + bytes / 2 + 4096;
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e != hipSuccess) return fail(RDF_MEMORY_ERROR, "hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    c.pinned = (char*)p;
    c.pinned_cap = want;
    return RDF_OK;
}

// ---- pool: device buffers owned by frames ----
constexpr size_t kPoolGranule = (size_t)2 << 20;
constexpr size_t kPoolMaxCached = (size_t)96 << 30;   // of 288 GB
rdf_status pool_alloc(size_t bytes, void** out, size_t* got) {
    Ctx& c = g_ctx;
    const size_t gran = bytes < ((size_t)1 << 20) ? (size_t)4096 : kPoolGranule;   // descriptor tables of small frames stay small
    const size_t want = (bytes + 256 + gran - 1) / gran * gran;
    auto it = c.pool_free.lower_bound(want);
    if (it != c.pool_free.end() && it->first <= want + want / 4 + gran) {
        *out = it->second; *got = it->first;
        c.pool_cached -= it->first;
        c.pool_free.erase(it);
        return RDF_OK;
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess && !c.pool_free.empty()) {   // give the cache back and retry
        if (c.stream) (void)hipStreamSynchronize(c.stream);
        for (auto& kv : c.pool_free) (void)hipFree(kv.second);
        c.pool_free.clear();
        c.pool_cached = 0;
        e = hipMalloc(&p, want);
    }
    if (e != hipSuccess) return fail(RDF_MEMORY_ERROR, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    *out = p; *got = want;
    return RDF_OK;
}
void pool_release(void* p, size_t bytes) {
    Ctx& c = g_ctx;
    if (!p) return;
    if (c.pool_cached + bytes > kPoolMaxCached) { (void)hipFree(p); return; }
    c.pool_free.emplace(bytes, p);
    c.pool_cached += bytes;
}

// ---- kernel timing ----
struct KernelTimer {
    bool on;
    size_t idx = 0;
    explicit KernelTimer() : on(g_ctx.timing) {
        if (!on) return;
        Ctx& c = g_ctx;
        if (c.events_used == c.events.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
            c.events.emplace_back(a, b);
        }
        idx = c.events_used++;
        (void)hipEventRecord(c.events[idx].first, c.stream);
    }
    void stop() {
        if (on) (void)hipEventRecord(g_ctx.events[idx].second, g_ctx.stream);
    }
};

// ------------------------------------------------------------------------------------------------
// dtype helpers

int dtype_size(int dt) {
    switch (dt) {
        case RDF_I8: case RDF_U8: return 1;
        case RDF_I16: case RDF_U16: return 2;
        case RDF_I32: case RDF_U32: case RDF_F32: return 4;
        case RDF_I64: case RDF_U64: case RDF_F64: return 8;
        default: return 0;
    }
}
bool is_float(int dt) { return dt == RDF_F32 || dt == RDF_F64; }
bool is_numeric(int dt) { return dt >= RDF_I8 && dt <= RDF_F64; }
bool is_signed_int(int dt) { return dt >= RDF_I8 && dt <= RDF_I64; }
bool is_unsigned_int(int dt) { return dt >= RDF_U8 && dt <= RDF_U64; }
int value_class(int dt) { return is_float(dt) ? CLS_F64 : is_signed_int(dt) ? CLS_SIGNED : CLS_UNSIGNED; }

uint64_t h_normalize_int(int dt, uint64_t x) {
    switch (dt) {
        case RDF_I8: return (uint64_t)(int64_t)(int8_t)x;
        case RDF_I16: return (uint64_t)(int64_t)(int16_t)x;
        case RDF_I32: return (uint64_t)(int64_t)(int32_t)x;
        case RDF_U8: return x & 0xFFull;
        case RDF_U16: return x & 0xFFFFull;
        case RDF_U32: return x & 0xFFFFFFFFull;
        default: return x;
    }
}

// ------------------------------------------------------------------------------------------------
// staging of RDF_MEM_HOST buffers: small items are packed through the pinned buffer and moved with
// ONE copy; large items are copied directly.

constexpr size_t kSmallCopy = 256 * 1024;

struct Region {
    std::vector<StageItem> items;
    char* dev = nullptr;
    size_t small_bytes = 0, total = 0;

    int add(const void* host, size_t bytes) {
        items.push_back(StageItem{host, bytes, 0, bytes < kSmallCopy});
        return (int)items.size() - 1;
    }
    rdf_status layout() {
        size_t off = 0;
        for (auto& it : items) if (it.small) { it.off = off; off += (it.bytes + 16 + 63) & ~(size_t)63; }
        small_bytes = off;
        off = (off + 255) & ~(size_t)255;
        for (auto& it : items) if (!it.small) { it.off = off; off += (it.bytes + 16 + 255) & ~(size_t)255; }
        total = off;
        void* p = nullptr;
        RDF_TRY(arena_alloc(total ? total : 256, &p));
        dev = (char*)p;
        return RDF_OK;
    }
    char* ptr(int idx) const { return dev + items[(size_t)idx].off; }

    rdf_status upload(size_t pinned_off, size_t* pinned_used) {
        Ctx& c = g_ctx;
        if (small_bytes) {
            char* pin = c.pinned + pinned_off;
            packed_copy(items, pin, true, small_bytes);
            HIP_TRY(hipMemcpyAsync(dev, pin, small_bytes, hipMemcpyHostToDevice, c.stream));
        }
        *pinned_used = small_bytes;
        for (auto& it : items)
            if (!it.small && it.bytes) HIP_TRY(hipMemcpyAsync(dev + it.off, it.src, it.bytes, hipMemcpyHostToDevice, c.stream));
        return RDF_OK;
    }
    // D2H: `src` fields are the host destinations.  Ends with a stream sync.
    rdf_status download(size_t pinned_off) {
        Ctx& c = g_ctx;
        if (small_bytes) HIP_TRY(hipMemcpyAsync(c.pinned + pinned_off, dev, small_bytes, hipMemcpyDeviceToHost, c.stream));
        for (auto& it : items)
            if (!it.small && it.bytes) HIP_TRY(hipMemcpyAsync((void*)it.src, dev + it.off, it.bytes, hipMemcpyDeviceToHost, c.stream));
        HIP_TRY(hipStreamSynchronize(c.stream));
        if (small_bytes) {
            packed_copy(items, c.pinned + pinned_off, false, small_bytes);
        }
        return RDF_OK;
    }
};

// Plan of one staged input array: which byte ranges to move and the resulting device view.
struct InPlan {
    int values_item = -1, validity_item = -1;
    int64_t dev_offset = 0;
};

// Stage (or alias) a list of arrays.  After finish(), dev[i] is the HBM view of arrays[i].
struct DbgTimer {   // RDF_DEBUG=1: host-side phase times of a call (stderr)
    bool on;
    std::chrono::steady_clock::time_point t0;
    DbgTimer() : on(getenv("RDF_DEBUG") != nullptr), t0(std::chrono::steady_clock::now()) {}
    void mark(const char* what) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[rdf] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

struct InputStager {
    Region region;
    std::vector<InPlan> plans;
    std::vector<const rdf_array*> arrays;
    std::vector<DevChunkCol> dev;
    bool host = false;

    void add(const rdf_array* a) {
        arrays.push_back(a);
        if (a->mem != RDF_MEM_HOST && !host) return;   // device-resident calls (one memory space per call) need no staging plan
        if (plans.size() + 1 < arrays.size()) plans.resize(arrays.size() - 1);   // (device arrays ahead of a host one: plans stay index-aligned)
        InPlan p;
        if (a->mem == RDF_MEM_HOST) {
            host = true;
            const int64_t k = a->offset & 7;
            const int64_t s0 = a->offset - k;
            p.dev_offset = k;
            if (a->length > 0) {
                if (a->dtype == RDF_BOOL) {
                    const size_t b0 = (size_t)(s0 >> 3), b1 = (size_t)((a->offset + a->length + 7) >> 3);
                    p.values_item = region.add((const char*)a->values + b0, b1 - b0);
                } else {
                    const size_t es = (size_t)dtype_size(a->dtype);
                    p.values_item = region.add((const char*)a->values + (size_t)s0 * es, (size_t)(a->length + k) * es);
                }
                if (a->validity) {
                    const size_t b0 = (size_t)(s0 >> 3), b1 = (size_t)((a->offset + a->length + 7) >> 3);
                    p.validity_item = region.add(a->validity + b0, b1 - b0);
                }
            }
        }
        plans.push_back(p);
    }
    rdf_status finish(size_t pinned_off, size_t* pinned_used) {
        *pinned_used = 0;
        dev.resize(arrays.size());
        if (host) {
            RDF_TRY(region.layout());
            RDF_TRY(pinned_reserve(pinned_off + region.small_bytes));
            RDF_TRY(region.upload(pinned_off, pinned_used));
        }
        for (size_t i = 0; i < arrays.size(); ++i) {
            const rdf_array* a = arrays[i];
            if (a->mem == RDF_MEM_DEVICE) dev[i] = DevChunkCol{a->values, a->validity, a->offset};
            else {
                const InPlan& p = plans[i];
                dev[i].values = p.values_item >= 0 ? region.ptr(p.values_item) : region.dev;
                dev[i].validity = p.validity_item >= 0 ? (const uint8_t*)region.ptr(p.validity_item) : nullptr;
                dev[i].offset = p.dev_offset;
            }
        }
        return RDF_OK;
    }
};

// Small host tables (chunk descriptors, prefix arrays) uploaded in one copy.
struct TableBuilder {   // reserve() every table, bind() to a place in the pinned staging buffer, fill through at(), alloc() + upload()
    size_t size = 0;
    char* base = nullptr;   // the tables are written straight into pinned memory: a frame of a million 1024-row batches has 40 MB of them
    char* dev = nullptr;
    size_t reserve(size_t bytes) {
        size_t off = (size + 15) & ~(size_t)15;
        size = off + bytes;
        return off;
    }
    rdf_status bind(size_t pinned_off) {
        RDF_TRY(pinned_reserve(pinned_off + size + 16));
        base = g_ctx.pinned + pinned_off;
        return RDF_OK;
    }
    template <typename T> T* at(size_t off) { return (T*)(base + off); }
    template <typename T> T* dev_at(size_t off) const { return (T*)(dev + off); }
    rdf_status alloc() {
        void* p = nullptr;
        RDF_TRY(arena_alloc(size ? size : 16, &p));
        dev = (char*)p;
        return RDF_OK;
    }
    rdf_status upload(size_t pinned_off) {
        Ctx& c = g_ctx;
        if (size == 0) return RDF_OK;
        if (base != c.pinned + pinned_off) return fail(RDF_COMPUTE_ERROR, "internal: table builder not bound to its staging offset");
        HIP_TRY(hipMemcpyAsync(dev, base, size, hipMemcpyHostToDevice, c.stream));
        return RDF_OK;
    }
};

rdf_status check_mem(const rdf_array* arrs, int64_t n, int32_t* mem_io) {
    for (int64_t i = 0; i < n; ++i) {
        if (arrs[i].mem != RDF_MEM_HOST && arrs[i].mem != RDF_MEM_DEVICE) return fail(RDF_INVALID_ARGUMENT, "bad mem tag %d", arrs[i].mem);
        if (*mem_io < 0) *mem_io = arrs[i].mem;
        else if (*mem_io != arrs[i].mem) return fail(RDF_INVALID_ARGUMENT, "all arrays of one call must share one memory space");
        if (arrs[i].length < 0 || arrs[i].offset < 0) return fail(RDF_INVALID_ARGUMENT, "negative length/offset");
        if (arrs[i].length > 0 && arrs[i].values == nullptr) return fail(RDF_INVALID_ARGUMENT, "null values pointer");
    }
    return RDF_OK;
}
rdf_status check_out_mem(const rdf_out* outs, int64_t n, int32_t mem) {
    for (int64_t i = 0; i < n; ++i)
        if (outs[i].mem != mem) return fail(RDF_INVALID_ARGUMENT, "outputs must live in the same memory space as the inputs");
    return RDF_OK;
}

// ------------------------------------------------------------------------------------------------
// expression compiler: rdf_expr_node tree -> accumulator-machine bytecode

bool op_is_arith(int op) { return op >= RDF_OP_ADD && op <= RDF_OP_DIV; }
bool op_is_fbinary(int op) { return op >= RDF_OP_ATAN2 && op <= RDF_OP_LOG; }
bool op_is_unary_math(int op) { return (op >= RDF_OP_ABS && op <= RDF_OP_TANH) || (op >= RDF_OP_COT && op <= RDF_OP_CSC); }
bool op_is_cmp(int op) { return op >= RDF_OP_GT && op <= RDF_OP_LE; }
bool op_is_hour(int op) { return op >= RDF_OP_HOUR_S && op <= RDF_OP_HOUR_DAY; }
bool op_is_binary(int op) { return op_is_arith(op) || op_is_fbinary(op) || op_is_cmp(op) || op == RDF_OP_AND || op == RDF_OP_OR; }
bool op_is_heavy(int op) {
    if (op_is_fbinary(op)) return true;
    if (!op_is_unary_math(op)) return false;
    switch (op) {
        case RDF_OP_ABS: case RDF_OP_CEIL: case RDF_OP_FLOOR: case RDF_OP_ROUND: case RDF_OP_SQRT:
        case RDF_OP_DEGREES: case RDF_OP_RADIANS: return false;
        default: return true;
    }
}

struct Compiler {
    const rdf_expr_node* nodes;
    int nnodes;
    const int* col_dtype;
    int ncols;
    std::vector<Instr> code;
    std::vector<int> memo;
    int tmp_used = 0, tmp_max = 0;
    bool heavy = false, intdiv = false;
    bool lossy_cast = false;   // the program holds a cast that can turn a valid value into NULL (arrow::compute::cast: None -> NULL)
    rdf_status st = RDF_OK;
    int feat() const { return heavy ? 2 : intdiv ? 1 : 0; }

    Compiler(const rdf_expr_node* n, int nn, const int* cd, int nc) : nodes(n), nnodes(nn), col_dtype(cd), ncols(nc), memo((size_t)(nn > 0 ? nn : 0), -2) {}

    int bad(rdf_status s, const char* fmt, ...) {
        if (st == RDF_OK) {
            char buf[256];
            va_list ap;
            va_start(ap, fmt);
            vsnprintf(buf, sizeof buf, fmt, ap);
            va_end(ap);
            st = fail(s, "%s", buf);
        }
        return -1;
    }

    // result dtype of node idx (or -1 on error)
    int infer(int idx, int depth = 0) {
        if (idx < 0 || idx >= nnodes) return bad(RDF_INVALID_ARGUMENT, "bad node index %d", idx);
        if (depth > 64) return bad(RDF_INVALID_ARGUMENT, "expression too deep");
        if (memo[(size_t)idx] != -2) return memo[(size_t)idx];
        const rdf_expr_node& nd = nodes[idx];
        int r = -1;
        if (nd.kind == RDF_NODE_COLUMN) {
            if (nd.column < 0 || nd.column >= ncols) r = bad(RDF_COMPUTE_ERROR, "Cannot find column %d", nd.column);
            else r = col_dtype[nd.column];
        } else if (nd.kind == RDF_NODE_SCALAR) {
            if (nd.dtype == RDF_NULLTYPE) r = RDF_BOOL;
            else if (is_numeric(nd.dtype) || nd.dtype == RDF_BOOL) r = nd.dtype;
            else r = bad(RDF_INVALID_ARGUMENT, "unsupported scalar type %d", nd.dtype);
        } else if (nd.kind == RDF_NODE_OP) {
            const int op = nd.op;
            const int l = infer(nd.lhs, depth + 1);
            if (l < 0) return memo[(size_t)idx] = -1;
            if (op_is_binary(op)) {
                const int rr = infer(nd.rhs, depth + 1);
                if (rr < 0) return memo[(size_t)idx] = -1;
                if (op_is_arith(op) || op_is_fbinary(op)) {
                    if (l != rr) r = bad(RDF_INVALID_ARGUMENT, "binary op %d: operand types differ (%d vs %d); insert a Cast", op, l, rr);
                    else if (!is_numeric(l)) r = bad(RDF_INVALID_ARGUMENT, "binary op %d: numeric type required", op);
                    else if (op_is_fbinary(op) && !is_float(l)) r = bad(RDF_INVALID_ARGUMENT, "math_op: float type required");
                    else r = l;
                } else r = RDF_BOOL;  // comparisons, and, or
            } else if (op_is_unary_math(op)) {
                if (op == RDF_OP_ABS) {
                    if (!(is_float(l) || is_signed_int(l))) r = bad(RDF_INVALID_ARGUMENT, "abs: signed numeric type required");
                    else r = l;
                } else if (!is_float(l)) r = bad(RDF_INVALID_ARGUMENT, "float type required");
                else r = l;
            } else if (op_is_hour(op)) {
                if (l != RDF_I32 && l != RDF_I64) r = bad(RDF_INVALID_ARGUMENT, "hour: Int32 / Int64 temporal storage required");
                else r = l;
            } else if (op == RDF_OP_CAST) {
                if (!(is_numeric(nd.dtype) || nd.dtype == RDF_BOOL)) r = bad(RDF_INVALID_ARGUMENT, "cast: unsupported type");
                else r = nd.dtype;
            } else if (op == RDF_OP_NOT) r = RDF_BOOL;
            else r = bad(RDF_INVALID_ARGUMENT, "unsupported op %d", op);
        } else r = bad(RDF_INVALID_ARGUMENT, "bad node kind %d", nd.kind);
        return memo[(size_t)idx] = r;
    }

    bool is_leaf(int idx) const { return nodes[idx].kind != RDF_NODE_OP; }

    void push(Instr in) {
        if ((int)code.size() >= kMaxCode) { bad(RDF_INVALID_ARGUMENT, "expression too long (more than %d steps)", kMaxCode); return; }
        code.push_back(in);
    }
    static Instr mk(uint8_t bc) {
        Instr in;
        memset(&in, 0, sizeof in);
        in.bc = bc;
        return in;
    }

    // literal payload converted on the host into domain `dom`
    uint64_t imm_for(const rdf_expr_node& nd, int dom) {
        const int lt = nd.dtype == RDF_NULLTYPE ? RDF_BOOL : nd.dtype;
        const bool flit = is_float(lt);
        const double f = lt == RDF_F32 ? (double)(float)nd.f64 : nd.f64;
        const uint64_t iv = nd.dtype == RDF_NULLTYPE ? 0 : (lt == RDF_BOOL ? (uint64_t)(nd.i64 != 0) : h_normalize_int(lt, (uint64_t)nd.i64));
        if (dom == RDF_BOOL) return flit ? (uint64_t)(f != 0.0) : (uint64_t)(iv != 0);
        if (dom == RDF_F64) {
            double d = flit ? f : (is_signed_int(lt) ? (double)(int64_t)iv : (double)iv);
            uint64_t u; memcpy(&u, &d, 8); return u;
        }
        if (dom == RDF_F32) {
            float d = flit ? (float)f : (is_signed_int(lt) ? (float)(int64_t)iv : (float)iv);
            uint32_t u; memcpy(&u, &d, 4); return u;
        }
        if (flit) {
            if (f != f) return 0;
            if (is_signed_int(dom)) {
                double lo = dom == RDF_I8 ? -128.0 : dom == RDF_I16 ? -32768.0 : dom == RDF_I32 ? -2147483648.0 : -9223372036854775808.0;
                double hi = dom == RDF_I8 ? 127.0 : dom == RDF_I16 ? 32767.0 : dom == RDF_I32 ? 2147483647.0 : 9223372036854775807.0;
                if (f <= lo) return (uint64_t)(int64_t)lo;
                if (f >= hi) return dom == RDF_I64 ? (uint64_t)INT64_MAX : (uint64_t)(int64_t)hi;
                return (uint64_t)(int64_t)f;
            }
            double hi = dom == RDF_U8 ? 255.0 : dom == RDF_U16 ? 65535.0 : dom == RDF_U32 ? 4294967295.0 : 18446744073709551615.0;
            if (f <= 0.0) return 0;
            if (f >= hi) return dom == RDF_U64 ? ~0ull : (uint64_t)hi;
            return (uint64_t)f;
        }
        return h_normalize_int(dom, iv);
    }

    void operand_fields(Instr& in, int leaf_idx, int dom) {
        const rdf_expr_node& nd = nodes[leaf_idx];
        if (nd.kind == RDF_NODE_COLUMN) {
            in.src_kind = SRC_COL;
            in.src = (uint16_t)nd.column;
            in.src_dtype = (uint8_t)col_dtype[nd.column];
        } else {
            in.src_kind = SRC_IMM;
            in.src_dtype = (uint8_t)dom;
            in.imm = imm_for(nd, dom);
        }
        in.dtype = (uint8_t)dom;
    }
    static bool cast_can_null(int from, int to) {
        if (from == to || to == RDF_BOOL || to == RDF_F32 || to == RDF_F64 || from == RDF_BOOL) return false;
        if (is_float(from)) return true;
        const bool fs = is_signed_int(from), ts = is_signed_int(to);
        const int fb = dtype_size(from), tb = dtype_size(to);
        if (fs == ts) return tb < fb;
        if (fs) return true;
        return tb <= fb;
    }
    void cast_acc(int from, int to) {
        if (from == to) return;
        lossy_cast |= cast_can_null(from, to);
        Instr in = mk(BC_CAST);
        in.src_dtype = (uint8_t)from;
        in.dtype = (uint8_t)to;
        push(in);
    }

    // generate code leaving node idx in the accumulator, in the domain of its inferred dtype
    void gen(int idx) {
        if (st != RDF_OK) return;
        const int dt = infer(idx);
        if (dt < 0) return;
        const rdf_expr_node& nd = nodes[idx];
        if (nd.kind != RDF_NODE_OP) {
            Instr in = mk(BC_LOAD);
            operand_fields(in, idx, dt);
            push(in);
            return;
        }
        const int op = nd.op;
        if (op_is_heavy(op)) heavy = true;
        if ((op == RDF_OP_DIV && !is_float(dt)) || op_is_hour(op)) intdiv = true;
        if (!op_is_binary(op)) {
            gen(nd.lhs);
            const int l = infer(nd.lhs);
            if (op == RDF_OP_CAST) { cast_acc(l, nd.dtype); return; }
            Instr in = mk(BC_UN);
            in.op = (uint8_t)op;
            if (op == RDF_OP_NOT) { cast_acc(l, RDF_BOOL); in.dtype = RDF_BOOL; }
            else in.dtype = (uint8_t)l;
            push(in);
            return;
        }
        const int l = infer(nd.lhs), r = infer(nd.rhs);
        const int dom = op_is_cmp(op) ? RDF_F64 : (op == RDF_OP_AND || op == RDF_OP_OR) ? RDF_BOOL : l;
        Instr in = mk(BC_BIN);
        in.op = (uint8_t)op;
        if (is_leaf(nd.rhs)) {
            gen(nd.lhs);
            cast_acc(l, dom);
            operand_fields(in, nd.rhs, dom);
        } else if (is_leaf(nd.lhs)) {
            gen(nd.rhs);
            cast_acc(r, dom);
            operand_fields(in, nd.lhs, dom);
            in.swapped = 1;
        } else {
            gen(nd.rhs);
            cast_acc(r, dom);
            if (tmp_used >= kMaxTmp) { bad(RDF_INVALID_ARGUMENT, "expression needs more than %d temporaries", kMaxTmp); return; }
            const int slot = tmp_used++;
            if (tmp_used > tmp_max) tmp_max = tmp_used;
            Instr stt = mk(BC_STORE_TMP);
            stt.src = (uint16_t)slot;
            push(stt);
            gen(nd.lhs);
            cast_acc(l, dom);
            in.src_kind = SRC_TMP;
            in.src = (uint16_t)slot;
            in.src_dtype = (uint8_t)dom;
            in.dtype = (uint8_t)dom;
            --tmp_used;
        }
        push(in);
    }
};

// ------------------------------------------------------------------------------------------------
// specialised-kernel lookup: canonical signature of a program (same grammar as rdf_spec.hip's sig())

struct SpecPlan {
    std::string sig;
    int col_map[kGSpecCols];      // canonical column -> program column
    int ncols = 0;
    uint64_t imm[kGSpecImm];
    char imm_tag[kGSpecImm];
    int nimm = 0;
    int width = 0;       // element width of the columns (gspec: of the first one); spec_kernel with `widest`: the program's widest element
    bool widest = false;
    bool ok = true;
    // spec_kernel: 4 columns of one width, 4 literals.  gspec_kernel (grouped sink): 8 columns of any widths,
    // 8 literals, equal literals share one slot
    int max_cols = 4, max_imm = 4;
    bool mixed = false, dedup = false;
};

char spec_tag(int dt) {
    switch (dt) { case RDF_F64: return 'd'; case RDF_I64: return 'l'; case RDF_U64: return 'u'; case RDF_F32: return 'f';
                  case RDF_I32: return 'i'; case RDF_U32: return 'j'; case RDF_BOOL: return 'b';
                  case RDF_I8: return 'a'; case RDF_U8: return 'h'; case RDF_I16: return 's'; case RDF_U16: return 't'; default: return 0; }
}

struct SpecSigBuilder {
    Compiler& cc;
    SpecPlan& sp;
    SpecSigBuilder(Compiler& c, SpecPlan& s) : cc(c), sp(s) {}

    std::string leaf(int idx, int dom) {
        const rdf_expr_node& nd = cc.nodes[idx];
        if (nd.kind == RDF_NODE_COLUMN) {
            const int dt = cc.col_dtype[nd.column];
            if (!spec_tag(dt) || dt == RDF_BOOL) { sp.ok = false; return "?"; }
            if (sp.width == 0 || sp.widest) sp.width = std::max(sp.width, dtype_size(dt));   // spec_kernel: the widest element decides the row layout
            else if (sp.width != dtype_size(dt) && !sp.mixed) { sp.ok = false; return "?"; }  // one width per program
            int id = -1;
            for (int i = 0; i < sp.ncols; ++i) if (sp.col_map[i] == nd.column) id = i;
            if (id < 0) {
                if (sp.ncols >= sp.max_cols) { sp.ok = false; return "?"; }
                id = sp.ncols;
                sp.col_map[sp.ncols++] = nd.column;
            }
            return std::string("c") + char('0' + id) + spec_tag(dt);
        }
        // scalar: payload converted to `dom`
        if (!spec_tag(dom) || dom == RDF_BOOL) { sp.ok = false; return "?"; }
        if (nd.dtype == RDF_NULLTYPE) { sp.ok = false; return "?"; }
        const uint64_t bits = cc.imm_for(nd, dom);
        int id = -1;
        if (sp.dedup)
            for (int i = 0; i < sp.nimm; ++i) if (sp.imm[i] == bits && sp.imm_tag[i] == spec_tag(dom)) id = i;
        if (id < 0) {
            if (sp.nimm >= sp.max_imm) { sp.ok = false; return "?"; }
            id = sp.nimm;
            sp.imm_tag[sp.nimm] = spec_tag(dom);
            sp.imm[sp.nimm++] = bits;
        }
        return std::string("k") + char('0' + id) + spec_tag(dom);
    }

    std::string node(int idx, int dom_for_scalar) {
        if (!sp.ok) return "?";
        const rdf_expr_node& nd = cc.nodes[idx];
        if (nd.kind != RDF_NODE_OP) return leaf(idx, dom_for_scalar);
        const int op = nd.op;
        if (op_is_binary(op)) {
            int l = nd.lhs, r = nd.rhs, o = op;
            const int lt = cc.infer(l), rt = cc.infer(r);
            int dom = op_is_cmp(op) ? RDF_F64 : lt;
            if (op_is_cmp(op) && cc.nodes[l].kind == RDF_NODE_SCALAR && cc.nodes[r].kind != RDF_NODE_SCALAR) {
                std::swap(l, r);  // c CMP x  ==  x CMP' c
                o = op == RDF_OP_GT ? RDF_OP_LT : op == RDF_OP_GE ? RDF_OP_LE : op == RDF_OP_LT ? RDF_OP_GT : op == RDF_OP_LE ? RDF_OP_GE : op;
            }
            (void)rt;
            const std::string a = node(l, dom), b = node(r, dom);
            return "(" + std::to_string(o) + " " + a + " " + b + ")";
        }
        if (op == RDF_OP_CAST) {
            const int from = cc.infer(nd.lhs);
            if (from == nd.dtype) return node(nd.lhs, dom_for_scalar);
            if (!spec_tag(nd.dtype) || nd.dtype == RDF_BOOL) { sp.ok = false; return "?"; }
            return "{" + std::to_string(nd.dtype) + " " + node(nd.lhs, from) + "}";
        }
        return "[" + std::to_string(op) + " " + node(nd.lhs, cc.infer(nd.lhs)) + "]";
    }
};

// Builds the plan; returns true when a specialised kernel exists for this program.
bool build_spec_plan(Compiler& cc, int filter_root, int nvalues, const int* value_roots, int sink, SpecPlan& sp) {
    if (nvalues > 2 || (sink == RDF_SINK_STORE && nvalues != 1)) return false;
    sp.widest = true;
    sp.max_cols = kSpecCols; sp.max_imm = kSpecImm;   // (the catalogs' programs stop at 4 and 4; rdf_jit.cpp compiles up to these)
    SpecSigBuilder b(cc, sp);
    std::string s = "P:";
    s += filter_root >= 0 ? b.node(filter_root, RDF_F64) : std::string("-");
    s += ";V:" + b.node(value_roots[0], cc.infer(value_roots[0]));
    s += ";" + (nvalues > 1 ? b.node(value_roots[1], cc.infer(value_roots[1])) : std::string("-"));
    s += ";S:" + std::to_string(sink == RDF_SINK_AGG ? SINK_AGG : SINK_STORE);
    if (!sp.ok) return false;
    if (sink == RDF_SINK_STORE && cc.infer(value_roots[0]) != RDF_BOOL) sp.width = std::max(sp.width, dtype_size(cc.infer(value_roots[0])));
    sp.sig = s;
    return spec_available(s.c_str()) || (g_ctx.opt_jit && jit_find(s.c_str()) != nullptr);   // in the catalog, or compiled earlier in this process (rdf_jit.cpp)
}

// Second-level lookup: kernels specialised on the tree SHAPE with runtime operators (rdf_expr.hip.h *RT nodes; the 8- and
// 4-byte numeric types).  Every leaf occurrence gets its own canonical column / literal slot, operator slots are numbered
// in pre-order (predicate first).  Canonical operand order, reached by setting the operator's swap bit: the deeper
// subtree first, a subtree before a leaf, a column before a literal (rdf_spec_kernel.hip.h lists the compiled shapes:
// up to three levels of arithmetic, sin / cos / tan over up to two levels, behind no predicate, x CMP c, or
// x CMP c AND|OR y CMP d).
struct ShapeSigBuilder {
    Compiler& cc;
    SpecPlan& sp;
    int* rt;
    int nslots = 0;
    int width = 0;      // element width of the program's columns (one width per program)
    int pred_dt = -1;
    ShapeSigBuilder(Compiler& c, SpecPlan& s, int* r) : cc(c), sp(s), rt(r) {}
    bool is_scalar(int idx) const { return cc.nodes[idx].kind == RDF_NODE_SCALAR; }
    bool is_column(int idx) const { return cc.nodes[idx].kind == RDF_NODE_COLUMN; }
    static bool shape_dtype(int dt) { return dt == RDF_F64 || dt == RDF_I64 || dt == RDF_U64 || dt == RDF_F32 || dt == RDF_I32 || dt == RDF_U32 || dt == RDF_I16 || dt == RDF_U16 || dt == RDF_I8 || dt == RDF_U8; }
    int strip(int idx) {   // skip no-op casts
        while (cc.nodes[idx].kind == RDF_NODE_OP && cc.nodes[idx].op == RDF_OP_CAST && cc.infer(cc.nodes[idx].lhs) == cc.nodes[idx].dtype) idx = cc.nodes[idx].lhs;
        return idx;
    }
    int depth(int idx) {   // levels of operators under (and including) idx; anything the shapes do not hold counts as too deep
        idx = strip(idx);
        const rdf_expr_node& nd = cc.nodes[idx];
        if (nd.kind != RDF_NODE_OP) return 0;
        if (nd.op == RDF_OP_CAST && is_column(strip(nd.lhs))) return 0;   // a cast column is a leaf (the plan builders' cast of an operand)
        if (nd.op == RDF_OP_SIN || nd.op == RDF_OP_COS || nd.op == RDF_OP_TAN) return 1 + depth(nd.lhs);
        if (nd.op >= RDF_OP_ADD && nd.op <= RDF_OP_DIV) return 1 + std::max(depth(nd.lhs), depth(nd.rhs));
        return 100;
    }
    // a leaf in domain `dom`: a column of exactly that dtype, or a literal converted to it
    std::string leaf(int idx, int dom) {
        const rdf_expr_node& nd = cc.nodes[idx];
        const char tag = spec_tag(dom);
        if (nd.kind == RDF_NODE_COLUMN) {
            if (cc.col_dtype[nd.column] != dom || sp.ncols >= 4) { sp.ok = false; return "?"; }
            width = std::max(width, dtype_size(dom));   // the program's widest element; narrower columns are read with narrower vectors
            sp.col_map[sp.ncols] = nd.column;
            return std::string("c") + char('0' + sp.ncols++) + tag;
        }
        if (nd.dtype == RDF_NULLTYPE || sp.nimm >= 4) { sp.ok = false; return "?"; }
        sp.imm[sp.nimm] = cc.imm_for(nd, dom);
        return std::string("k") + char('0' + sp.nimm++) + tag;
    }
    std::string node(int idx, int dom) {
        if (!sp.ok) return "?";
        idx = strip(idx);
        const rdf_expr_node& nd = cc.nodes[idx];
        if (nd.kind != RDF_NODE_OP) return leaf(idx, dom);
        if (nslots >= 8) { sp.ok = false; return "?"; }
        const int op = nd.op;
        if (op == RDF_OP_CAST) {   // cast(column) to the domain's type: a leaf that keeps its own dtype and width in memory
            const int child = strip(nd.lhs);
            if (!is_column(child) || nd.dtype != dom) { sp.ok = false; return "?"; }
            const int from = cc.col_dtype[cc.nodes[child].column];
            if (!shape_dtype(from) || from == dom) { sp.ok = false; return "?"; }
            return "{" + std::to_string(dom) + " " + leaf(child, from) + "}";
        }
        if (op == RDF_OP_SIN || op == RDF_OP_COS || op == RDF_OP_TAN) {
            if (!(dom == RDF_F64 || dom == RDF_F32) || cc.infer(nd.lhs) != dom) { sp.ok = false; return "?"; }
            const int slot = nslots++;
            rt[slot] = op;
            return "[T" + std::to_string(slot) + " " + node(nd.lhs, dom) + "]";
        }
        const bool arith = op >= RDF_OP_ADD && op <= RDF_OP_DIV, cmp = op_is_cmp(op), logic = op == RDF_OP_AND || op == RDF_OP_OR;
        if (!(arith || cmp || logic)) { sp.ok = false; return "?"; }
        int l = strip(nd.lhs), r = strip(nd.rhs);
        if (arith && (cc.infer(idx) != dom || cc.infer(l) != dom || cc.infer(r) != dom)) { sp.ok = false; return "?"; }
        if (cmp && !((is_column(l) && is_scalar(r)) || (is_scalar(l) && is_column(r)))) { sp.ok = false; return "?"; }
        bool swap = false;
        if (!logic) {
            const int dl = depth(l), dr = depth(r);
            if (is_scalar(l) && is_scalar(r)) { sp.ok = false; return "?"; }
            if (dl < dr || (dl == 0 && dr == 0 && is_scalar(l) && is_column(r))) swap = true;
        }
        const int slot = nslots++;
        if (swap && cmp) {          // x CMP c with the literal first: the mirrored operator, operands in canonical order
            const int m = op == RDF_OP_GT ? RDF_OP_LT : op == RDF_OP_GE ? RDF_OP_LE : op == RDF_OP_LT ? RDF_OP_GT : op == RDF_OP_LE ? RDF_OP_GE : op;
            rt[slot] = m;
        } else if (swap && (op == RDF_OP_ADD || op == RDF_OP_MUL)) {
            rt[slot] = op;          // commutative (IEEE addition / multiplication and the wrapping integer ones): no swap needed
        } else {
            rt[slot] = op | (swap ? 0x100 : 0);
        }
        if (swap) std::swap(l, r);
        std::string a, b;
        if (cmp) {   // the column keeps its own dtype, the literal is compared in f64 (src/expression.rs:844-845)
            const int cdt = cc.col_dtype[cc.nodes[l].column];
            if (!shape_dtype(cdt)) { sp.ok = false; return "?"; }
            if (pred_dt >= 0 && pred_dt != cdt) { sp.ok = false; return "?"; }
            pred_dt = cdt;
            a = leaf(l, cdt);
            b = leaf(r, RDF_F64);
        } else { a = node(l, dom); b = node(r, dom); }
        return std::string("(") + (arith ? 'A' : cmp ? 'C' : 'G') + std::to_string(slot) + " " + a + " " + b + ")";
    }
};
bool build_shape_plan(Compiler& cc, int filter_root, int nvalues, const int* value_roots, int sink, SpecPlan& sp, int* rt) {
    if (nvalues != 1) return false;
    ShapeSigBuilder b(cc, sp, rt);
    const int vdt = cc.infer(value_roots[0]);
    const int dom = vdt == RDF_BOOL ? RDF_F64 : vdt;   // a predicate as the value (mask output): its columns pick their own dtype
    if (!ShapeSigBuilder::shape_dtype(dom)) return false;
    std::string s = "P:";
    s += filter_root >= 0 ? b.node(filter_root, RDF_F64) : std::string("-");
    s += ";V:" + b.node(value_roots[0], dom) + ";-;S:" + std::to_string(sink == RDF_SINK_AGG ? SINK_AGG : SINK_STORE);
    if (!sp.ok || b.nslots == 0 || b.width == 0) return false;
    sp.width = sink == RDF_SINK_STORE && vdt != RDF_BOOL ? std::max(b.width, dtype_size(dom)) : b.width;   // a stored value counts too
    sp.sig = s;
    return spec_available(s.c_str());
}

// Grouped sink: signature "G<g>;P:<pred|->;K:<group id>;V:<v0>;<v1>;..." for the smallest catalog G >= ngroups.
bool build_gspec_plan(Compiler& cc, int filter_root, int group_root, int ngroups, int nvalues, const int* value_roots, SpecPlan& sp) {
    sp.max_cols = kGSpecCols; sp.max_imm = kGSpecImm; sp.mixed = true; sp.dedup = true;
    SpecSigBuilder b(cc, sp);
    std::string s = ";P:";
    s += filter_root >= 0 ? b.node(filter_root, RDF_F64) : std::string("-");
    s += ";K:" + b.node(group_root, cc.infer(group_root)) + ";V:";
    for (int v = 0; v < nvalues; ++v) s += b.node(value_roots[v], cc.infer(value_roots[v])) + ";";
    if (!sp.ok) return false;
    for (int g : {2, 4, 6, 8}) {
        if (g < ngroups) continue;
        const std::string full = "G" + std::to_string(g) + s;
        if (gspec_available(full.c_str()) || (g_ctx.opt_jit && jit_find(full.c_str()))) { sp.sig = full; return true; }
    }
    if (g_ctx.opt_jit)   // no catalog holds the program: the grouped kernel template compiled for it at run time (rdf_jit.cpp)
        for (int g : {2, 4, 6, 8}) {
            if (g < ngroups) continue;
            const std::string full = "G" + std::to_string(g) + s;
            if (jit_spec_kernel(full.c_str(), g_ctx.opt_jit >= 2)) { sp.sig = full; return true; }
            break;
        }
    return false;
}

// ------------------------------------------------------------------------------------------------
// the fused evaluator driver

struct ProgramSpec {
    const rdf_expr_node* nodes;
    int nnodes;
    int filter_root;
    int nvalues;
    int value_roots[kMaxGroupValues];
    int sink;
    // RDF_SINK_GROUP (internal): rdf_group_pipeline
    int group_root = -1, ngroups = 0;
    rdf_group_result* gout = nullptr;
    int64_t* grows = nullptr;
    bool casts_always_fit = false;   // internal programs whose narrowing cast cannot fail (rdf_hour: 0..23 into Int32): no output bitmap needed for it
    // SINK_STORE into buffers a frame owns (frame-level operators): the outputs' descriptors are a DEVICE table laid out
    // [nvalues * nchunks], 16-byte aligned values / 8-byte aligned bitmaps per chunk, every chunk with room for its batch;
    // no rdf_out list is walked, lengths / null counts are not reported.  frame_out0 = chunk 0's descriptors (nchunks == 1).
    const DevOutChunk* frame_outs = nullptr;
    DevOutChunk frame_out0[kMaxValues] = {};
    int frame_out_dtype[kMaxValues] = {0, 0, 0, 0};
};
constexpr int RDF_SINK_GROUP = 2;

// a kernel compiled at run time could not be launched (it is marked failed): the caller runs the program again, interpreted
const rdf_status kRetryInterpreted = static_cast<rdf_status>(77);

// Which handler of eval_lean_kernel runs each step of an aggregate program, if every step has one (LH_* in rdf_device.h, the
// kernel's header says what it covers): columns of 8-byte types all held in registers, f64 comparisons, f64 / 64-bit integer
// arithmetic, Boolean connectives, i64 / u64 -> f64 casts, the filter, aggregate sinks of 8-byte values.  The indices go into
// bits 1..7 of Instr::swapped of `lean`, a copy — the program eval_kernel would run is not touched.
bool lean_assign(const EvalArgs& ea, EvalArgs& lean, int sink = SINK_AGG) {
    if (ea.ncols < 1 || ea.ncols > kPreCols || ea.nvalues < 1 || ea.nvalues > kMaxValues || ea.ncode < 1) return false;
    auto wide = [](int dt) { return dt == RDF_F64 || dt == RDF_I64 || dt == RDF_U64; };
    for (int c = 0; c < ea.ncols; ++c) if (!wide(ea.col_dtype[c])) return false;
    lean = ea;
    for (int i = 0; i < ea.ncode; ++i) {
        const Instr& in = ea.code[i];
        const bool same = in.src_dtype == in.dtype;
        int h = LH_NONE;
        switch (in.bc) {
            case BC_LOAD:
                if (in.src_kind == SRC_COL && same && wide(in.dtype) && in.src < ea.ncols) h = LH_LOAD;
                else if (in.src_kind == SRC_IMM && same) h = LH_LOAD;   // (the payload is already in the step's domain)
                break;
            case BC_STORE_TMP: h = LH_STORE_TMP; break;
            case BC_FILTER: h = LH_FILTER; break;
            case BC_EMIT: if ((wide(in.dtype) || (sink == SINK_STORE && in.dtype == RDF_BOOL)) && in.src < ea.nvalues) h = LH_EMIT; break;   // (stored: 8-byte values, or a predicate's bitmap)
            case BC_UN: if (in.op == RDF_OP_NOT) h = LH_NOT; break;
            case BC_CAST:
                if (in.dtype == RDF_F64 && in.src_dtype == RDF_I64) h = LH_CAST_I2F;
                else if (in.dtype == RDF_F64 && in.src_dtype == RDF_U64) h = LH_CAST_U2F;
                break;
            case BC_BIN: {
                // (a 64-bit integer column as the operand of an f64 step is converted on the way in)
                if (!same && !(in.src_kind == SRC_COL && in.dtype == RDF_F64 && (in.src_dtype == RDF_I64 || in.src_dtype == RDF_U64))) break;
                if (in.src_kind == SRC_COL && !(in.src < ea.ncols)) break;
                if (in.src_kind != SRC_COL && in.src_kind != SRC_IMM && in.src_kind != SRC_TMP) break;
                const bool sw = in.swapped & 1;
                const int op = in.op;
                if (op >= RDF_OP_GT && op <= RDF_OP_LE) {
                    // acc CMP b; swapped (b CMP acc) is the mirrored comparison of acc with b
                    static const int fwd[6] = {LH_F_GT, LH_F_GE, LH_F_EQ, LH_F_NE, LH_F_LT, LH_F_LE};
                    static const int mir[6] = {LH_F_LT, LH_F_LE, LH_F_EQ, LH_F_NE, LH_F_GT, LH_F_GE};
                    h = (sw ? mir : fwd)[op - RDF_OP_GT];
                } else if (op == RDF_OP_AND) h = LH_AND;
                else if (op == RDF_OP_OR) h = LH_OR;
                else if (in.dtype == RDF_F64) {
                    if (op == RDF_OP_ADD) h = LH_F_ADD;
                    else if (op == RDF_OP_MUL) h = LH_F_MUL;
                    else if (op == RDF_OP_SUB) h = sw ? LH_F_RSUB : LH_F_SUB;
                    else if (op == RDF_OP_DIV) h = sw ? LH_F_RDIV : LH_F_DIV;
                } else if (in.dtype == RDF_I64 || in.dtype == RDF_U64) {
                    if (op == RDF_OP_ADD) h = LH_I_ADD;
                    else if (op == RDF_OP_MUL) h = LH_I_MUL;
                    else if (op == RDF_OP_SUB) h = sw ? LH_I_RSUB : LH_I_SUB;
                }
            } break;
            default: break;
        }
        if (h == LH_NONE) return false;
        lean.code[i].swapped = (uint8_t)((in.swapped & 1) | (h << 1));
    }
    return true;
}

rdf_status launch_agg_pair(const EvalArgs* ea, const FilterAggF64Args* fa, int cmp, int feat, int grid, int nvalues,
                           const int* cls, AggPartial* partials, AggPartial* result, const char* spec_sig = nullptr,
                           const SpecArgs* sa = nullptr) {
    Ctx& c = g_ctx;
    KernelTimer kt;
    if (sa) {
        const bool jit = jit_find(spec_sig) != nullptr;
        c.last_kernel = std::string("spec_kernel<") + spec_sig + ">" + (jit ? " [compiled at run time]" : "");
        const hipError_t le = launch_spec(spec_sig, *sa, grid, c.stream);
        if (le != hipSuccess && jit) { (void)hipGetLastError(); return kRetryInterpreted; }
        if (le != hipSuccess) return fail(RDF_DEVICE_ERROR, "launch_spec: %s", hipGetErrorString(le));
    }
    else if (fa) { c.last_kernel = "filter_agg_f64_kernel"; HIP_TRY(launch_filter_agg_f64(*fa, cmp, grid, c.stream)); }
    else {
        EvalArgs lean;
        if (c.opt_interp_lean && feat == 0 && lean_assign(*ea, lean)) { c.last_kernel = "eval_kernel<AGG, lean>"; HIP_TRY(launch_eval_lean(lean, SINK_AGG, grid, c.stream, c.opt_interp_lean == 2)); }
        else { c.last_kernel = "eval_kernel<AGG>"; HIP_TRY(launch_eval(*ea, SINK_AGG, feat, grid, c.stream)); }
    }
    kt.stop();
    AggFinalArgs f;
    memset(&f, 0, sizeof f);
    f.partials = partials;
    f.result = result;
    f.nblocks = grid;
    f.nvalues = nvalues;
    for (int k = 0; k < nvalues; ++k) f.value_cls[k] = cls[k];
    HIP_TRY(launch_agg_final(f, c.stream));
    return RDF_OK;
}

void fill_agg_result(rdf_agg_result* r, int dt, const AggPartial& p) {
    memset(r, 0, sizeof *r);
    r->dtype = dt;
    r->count = p.cnt;
    r->is_some = p.cnt > 0;
    if (is_float(dt)) {
        double s, a, b;
        memcpy(&s, &p.sum, 8); memcpy(&a, &p.mn, 8); memcpy(&b, &p.mx, 8);
        r->sum_f64 = dt == RDF_F32 ? (double)(float)s : s;
        if (p.cnt > 0) { r->min_f64 = a; r->max_f64 = b; }
    } else {
        r->sum_i64 = (int64_t)h_normalize_int(dt, p.sum);
        if (p.cnt > 0) { r->min_i64 = (int64_t)p.mn; r->max_i64 = (int64_t)p.mx; }
    }
}

}  // namespace

// rdf_frame (rdf_frame_pin): what a call over a frame of many RecordBatches would otherwise re-derive from the caller's
// rdf_array list every time — validated dtypes and batch lengths, the device descriptors, tile prefix tables per tile
// size, descriptor tables in a program's canonical column order, alignment of every column — kept on the host and in HBM.
struct rdf_frame {
    int device = 0;
    int ncols = 0;
    int64_t nchunks = 0, total_rows = 0;
    int col_dtype[kMaxFrameCols];
    bool col_aligned16[kMaxFrameCols];       // every non-empty chunk of the column starts on a 16-byte boundary
    bool col_nullable[kMaxFrameCols];        // some chunk of the column carries a validity bitmap
    bool col_contig[kMaxFrameCols];          // the column's batches are consecutive slices of ONE buffer (values and bitmap): row -> element
                                             // needs no batch lookup (take / sort read it as one chunk through dev[k * nchunks])
    bool host_valid = true;                  // clen / dev / chunk_nullable below mirror the device tables (frames built by an operator
                                             // fill them on first need: frame_host)
    bool owned = false;                      // the column buffers belong to the frame (outputs of frame-level operators)
    int64_t uniform_len = 0;                 // > 0: every chunk but the last holds this many rows (the readers' batches)
    std::vector<int64_t> clen;
    std::vector<DevChunkCol> dev;            // [ncols * nchunks]
    std::vector<uint8_t> chunk_nullable;     // [nchunks]: some column of the batch carries a validity bitmap
    DevChunkCol* d_cols = nullptr;           // device copies
    int64_t* d_clen = nullptr;
    struct Tiles { int64_t ntiles = 0; uint64_t tile_inv = 0; int64_t* d_start = nullptr; };
    std::map<int, Tiles> tiles;              // rows per tile -> prefix table
    std::map<std::vector<int>, DevChunkCol*> col_tabs;   // canonical column order of a specialised program -> descriptor table
    std::vector<void*> allocs;
    int64_t* d_row_start = nullptr;          // [nchunks + 1] prefix of the batch lengths (take / sort: row -> chunk)
    // rdf_filter_frame: the predicate's mask lives in the frame (bit-packed, batch c at bit mask_pos[c], 64-bit aligned)
    uint8_t* mask_values = nullptr;
    uint8_t* mask_validity = nullptr;
    DevOutChunk* d_mask_outs = nullptr;
    DevChunkCol* d_mask_cols = nullptr;
    DevOutChunk mask_out0 = {nullptr, nullptr};
    int64_t* d_mask_pos = nullptr;           // [nchunks + 1] bit position of every batch's mask (multiples of 64)
    int64_t mask_padded_rows = 0;            // d_mask_pos[nchunks]: rows of a buffer that holds every batch at its mask position
    int64_t max_clen = 0;                    // longest batch
    std::vector<std::pair<void*, size_t>> pooled;   // buffers taken from the per-thread pool, returned at release
    // projections onto <= kMaxCols columns (a predicate over a wide frame reads a few of its columns): frames that share this
    // frame's buffers and own only their descriptor tables; built once per column list
    std::map<std::vector<int>, rdf_frame*> projections;
};

namespace {

rdf_status frame_host(rdf_frame& f);   // rdf_capi_frame.inc: host mirrors of a frame an operator built on the device

// A frame's small device tables (descriptors, batch lengths, tile prefixes) come out of the thread's buffer pool like its column
// buffers: hipFree waits for the whole device — every stream, the streamed batch loop's copies included — and a frame per slab
// was a wait per slab.
rdf_status frame_table_alloc(rdf_frame& f, size_t bytes, void** out) {
    size_t got = 0;
    RDF_TRY(pool_alloc(bytes, out, &got));
    f.pooled.emplace_back(*out, got);
    return RDF_OK;
}

// A frame's tables, host vectors -> HBM.  Plain frames: hipMemcpy.  Frames made per slab by the streamed batch loop: the copy engine is
// busy with the next slab's 256 MiB for milliseconds and a hipMemcpy queues behind it — the tables are staged in a page-locked ring and
// fetched by a kernel on the compute stream instead (rdf_set_option("stream_table_kernel", 0): A/B).
rdf_status frame_table_upload(void* dst, const void* src, size_t bytes) {
    Ctx& c = g_ctx;
    if (bytes == 0) return RDF_OK;
    constexpr size_t kRing = (size_t)32 << 20;
    if (c.streaming && c.opt_stream_table_kernel && bytes <= kRing / 4) {
        if (!c.tab_pin) {
            void* p = nullptr;
            hipError_t e = hipHostMalloc(&p, kRing, hipHostMallocDefault);
            if (e != hipSuccess) return fail(RDF_MEMORY_ERROR, "hipHostMalloc(%zu) for the table ring failed: %s", kRing, hipGetErrorString(e));
            c.tab_pin = (char*)p; c.tab_pin_cap = kRing; c.tab_pin_used = 0;
        }
        const size_t need = (bytes + 255) & ~(size_t)255;
        if (c.tab_pin_used + need > c.tab_pin_cap) { HIP_TRY(hipStreamSynchronize(c.stream)); c.tab_pin_used = 0; }     // the kernels that read the ring's old contents are done
        char* pin = c.tab_pin + c.tab_pin_used;
        c.tab_pin_used += need;
        memcpy(pin, src, bytes);
        HIP_TRY(launch_copy_small(pin, dst, bytes, c.stream));
        return RDF_OK;
    }
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return RDF_OK;
}

rdf_status frame_tiles(rdf_frame& f, int rows_per_tile, const rdf_frame::Tiles** out) {
    RDF_TRY(frame_host(f));
    auto it = f.tiles.find(rows_per_tile);
    if (it == f.tiles.end()) {
        std::vector<int64_t> ts((size_t)f.nchunks + 1, 0);
        for (int64_t c = 0; c < f.nchunks; ++c) ts[(size_t)c + 1] = ts[(size_t)c] + (f.clen[(size_t)c] + rows_per_tile - 1) / rows_per_tile;
        rdf_frame::Tiles t;
        t.ntiles = ts[(size_t)f.nchunks];
        if (f.nchunks > 1 && ts[(size_t)f.nchunks - 1] > 0 && f.nchunks - 1 < ((int64_t)1 << 31))
            t.tile_inv = (uint64_t)(((unsigned __int128)(uint64_t)(f.nchunks - 1) << 32) / (unsigned __int128)(uint64_t)ts[(size_t)f.nchunks - 1]);
        void* p = nullptr;
        RDF_TRY(frame_table_alloc(f, ts.size() * 8 + 64, &p));
        RDF_TRY(frame_table_upload(p, ts.data(), ts.size() * 8));
        t.d_start = (int64_t*)p;
        it = f.tiles.emplace(rows_per_tile, t).first;
    }
    *out = &it->second;
    return RDF_OK;
}
rdf_status frame_col_tab(rdf_frame& f, const int* col_map, int n, DevChunkCol** out) {
    if (n <= 0) { *out = f.d_cols; return RDF_OK; }
    std::vector<int> key(col_map, col_map + n);
    auto it = f.col_tabs.find(key);
    if (it == f.col_tabs.end()) {
        std::vector<DevChunkCol> tab((size_t)n * (size_t)f.nchunks);
        for (int k = 0; k < n; ++k) memcpy(tab.data() + (size_t)k * (size_t)f.nchunks, f.dev.data() + (size_t)col_map[k] * (size_t)f.nchunks, sizeof(DevChunkCol) * (size_t)f.nchunks);
        void* p = nullptr;
        RDF_TRY(frame_table_alloc(f, tab.size() * sizeof(DevChunkCol) + 64, &p));
        RDF_TRY(frame_table_upload(p, tab.data(), tab.size() * sizeof(DevChunkCol)));
        it = f.col_tabs.emplace(key, (DevChunkCol*)p).first;
    }
    *out = it->second;
    return RDF_OK;
}

// rdf_capi_comm.inc: this rank's device-resident partial aggregates -> all ranks' -> folded on the device -> host (one wait)
rdf_status agg_dist_finish(::rdf_comm& c, const AggPartial* d_result, const uint32_t* d_flags, int nvalues, const int* cls, AggPartial* h_out, uint32_t* h_flags);

rdf_status run_program(const ProgramSpec& ps, const rdf_array* cols, int ncols, int64_t nchunks, rdf_out* outs,
                       rdf_agg_result* aggs, const char* len_mismatch_msg, rdf_frame* fc = nullptr) {
    if (nchunks < 0) return fail(RDF_INVALID_ARGUMENT, "negative chunk count");
    if (ncols < 0 || ncols > kMaxCols) return fail(RDF_INVALID_ARGUMENT, "a fused program reads at most %d columns", kMaxCols);
    const bool grouped = ps.sink == RDF_SINK_GROUP;
    if (ps.nvalues < 1 || ps.nvalues > (grouped ? kMaxGroupValues : kMaxValues)) return fail(RDF_INVALID_ARGUMENT, "nvalues out of range");
    if (grouped) {
        if (ps.ngroups < 1 || (int64_t)(ps.ngroups + 1) * ps.nvalues > RDF_MAX_GROUP_SLOTS)
            return fail(RDF_INVALID_ARGUMENT, "grouped aggregation: (ngroups + 1) * nvalues must be in [2, %d] (large key domains: rdf_groupby_sum)", RDF_MAX_GROUP_SLOTS);
        if (!ps.gout) return fail(RDF_INVALID_ARGUMENT, "null output pointer");
    }
    if (ps.sink == RDF_SINK_STORE && ps.filter_root >= 0)
        return fail(RDF_INVALID_ARGUMENT, "SINK_STORE with a filter: use rdf_predicate + rdf_filter_columns");
    int32_t mem = fc ? RDF_MEM_DEVICE : -1;
    if (fc) RDF_TRY(frame_host(*fc));     // a frame returned by an operator mirrors its tables on first need
    if (!fc) RDF_TRY(check_mem(cols, (int64_t)ncols * nchunks, &mem));
    if (mem < 0) mem = ps.sink == RDF_SINK_STORE && outs && nchunks > 0 ? outs[0].mem : RDF_MEM_HOST;
    const bool frame_store = ps.sink == RDF_SINK_STORE && ps.frame_outs != nullptr;
    if (frame_store && !fc) return fail(RDF_INVALID_ARGUMENT, "frame-owned outputs need a frame");
    if (ps.sink == RDF_SINK_STORE && nchunks > 0 && !frame_store) {
        if (!outs) return fail(RDF_INVALID_ARGUMENT, "outs is null");
        RDF_TRY(check_out_mem(outs, (int64_t)ps.nvalues * nchunks, mem));
    }
    if (ps.sink == RDF_SINK_AGG && !aggs) return fail(RDF_INVALID_ARGUMENT, "aggs is null");

    DbgTimer dbg;
    // column dtypes: chunk 0 decides, every chunk must agree (ChunkedArray::from_arrays, src/table.rs:24-40)
    int col_dtype[kMaxCols];
    std::vector<int64_t> clen_own;
    int64_t total_rows = 0;
    if (fc) {
        for (int k = 0; k < ncols; ++k) col_dtype[k] = fc->col_dtype[k];
        total_rows = fc->total_rows;
    } else {
        for (int k = 0; k < ncols; ++k) {
            col_dtype[k] = nchunks > 0 ? cols[(int64_t)k * nchunks].dtype : RDF_F64;
            if (!(is_numeric(col_dtype[k]) || col_dtype[k] == RDF_BOOL)) return fail(RDF_INVALID_ARGUMENT, "column %d: unsupported dtype %d", k, col_dtype[k]);
            for (int64_t c = 0; c < nchunks; ++c)
                if (cols[(int64_t)k * nchunks + c].dtype != col_dtype[k]) return fail(RDF_INVALID_ARGUMENT, "column %d: chunks differ in dtype", k);
        }
        // batch lengths: all columns of RecordBatch c have one length
        clen_own.assign((size_t)nchunks, 0);
        for (int64_t c = 0; c < nchunks; ++c) {
            clen_own[(size_t)c] = ncols > 0 ? cols[c].length : 0;
            for (int k = 1; k < ncols; ++k)
                if (cols[(int64_t)k * nchunks + c].length != clen_own[(size_t)c]) return fail(RDF_COMPUTE_ERROR, "%s", len_mismatch_msg);
            total_rows += clen_own[(size_t)c];
        }
    }
    const std::vector<int64_t>& clen = fc ? fc->clen : clen_own;
    dbg.mark("validate");
    // compile
    Compiler cc(ps.nodes, ps.nnodes, col_dtype, ncols);
    int value_dtype[kMaxGroupValues];
    if (ps.filter_root >= 0) {
        const int ft = cc.infer(ps.filter_root);
        if (cc.st != RDF_OK) return cc.st;
        if (ft != RDF_BOOL) return fail(RDF_INVALID_ARGUMENT, "predicate root must be boolean");
        cc.gen(ps.filter_root);
        cc.push(Compiler::mk(BC_FILTER));
    }
    if (grouped) {
        const int gt = cc.infer(ps.group_root);
        if (cc.st != RDF_OK) return cc.st;
        if (!(gt == RDF_BOOL || (gt >= RDF_I8 && gt <= RDF_U64))) return fail(RDF_INVALID_ARGUMENT, "group id expression must be integer-valued");
        cc.gen(ps.group_root);
        Instr g = Compiler::mk(BC_GROUP);
        g.dtype = (uint8_t)gt;
        cc.push(g);
    }
    for (int v = 0; v < ps.nvalues; ++v) {
        value_dtype[v] = cc.infer(ps.value_roots[v]);
        if (cc.st != RDF_OK) return cc.st;
        if (ps.sink != RDF_SINK_STORE && !(is_numeric(value_dtype[v]) || value_dtype[v] == RDF_BOOL))
            return fail(RDF_INVALID_ARGUMENT, "aggregate of a non-numeric value");
        cc.gen(ps.value_roots[v]);
        Instr e = Compiler::mk(BC_EMIT);
        e.src = (uint16_t)v;
        e.dtype = (uint8_t)value_dtype[v];
        cc.push(e);
    }
    if (cc.st != RDF_OK) return cc.st;
    if (ps.casts_always_fit) cc.lossy_cast = false;

    // output validation (SINK_STORE)
    if (frame_store) {
        for (int v = 0; v < ps.nvalues; ++v)
            if (ps.frame_out_dtype[v] != value_dtype[v]) return fail(RDF_INVALID_ARGUMENT, "output dtype %d != expression dtype %d", ps.frame_out_dtype[v], value_dtype[v]);
    } else if (ps.sink == RDF_SINK_STORE) {
        for (int v = 0; v < ps.nvalues; ++v)
            for (int64_t c = 0; c < nchunks; ++c) {
                rdf_out& o = outs[(int64_t)v * nchunks + c];
                if (o.dtype != value_dtype[v]) return fail(RDF_INVALID_ARGUMENT, "output dtype %d != expression dtype %d", o.dtype, value_dtype[v]);
                if (o.capacity < clen[(size_t)c]) return fail(RDF_MEMORY_ERROR, "output capacity too small");
                bool nullable = cc.lossy_cast;   // a cast to a narrower / differently signed / integer type yields NULL where the value does not fit
                if (fc) nullable |= fc->chunk_nullable[(size_t)c] != 0;
                else for (int k = 0; k < ncols; ++k) nullable |= cols[(int64_t)k * nchunks + c].validity != nullptr;
                if (nullable && !o.validity) return fail(RDF_INVALID_ARGUMENT, "output validity buffer required");
                if (clen[(size_t)c] > 0 && !o.values) return fail(RDF_INVALID_ARGUMENT, "null output values pointer");
            }
    }

    // nothing to launch
    if (total_rows == 0) {
        if (frame_store) return RDF_OK;
        if (ps.sink == RDF_SINK_STORE) {
            for (int64_t i = 0; i < (int64_t)ps.nvalues * nchunks; ++i) { outs[i].length = 0; outs[i].null_count = 0; }
        } else if (grouped) {
            for (int v = 0; v < ps.nvalues; ++v)
                for (int g = 0; g <= ps.ngroups; ++g) {
                    rdf_group_result& r = ps.gout[(size_t)v * (size_t)(ps.ngroups + 1) + (size_t)g];
                    memset(&r, 0, sizeof r);
                    r.dtype = value_dtype[v];
                }
            if (ps.grows) memset(ps.grows, 0, sizeof(int64_t) * (size_t)(ps.ngroups + 1));
        } else if (g_ctx.agg_comm) {
            // rdf_pipeline_dist over an EMPTY shard: the other ranks are about to enter the all-gathers of agg_dist_finish, so this
            // one must too (returning its local zeros left them in the collective until the watchdog aborted the communicator).
            // Its contribution is the fold's identity, built on the host and handed over like a kernel's partials.
            RDF_TRY(ensure_ready());
            Ctx& cx = g_ctx;
            arena_begin();
            void* scr = nullptr;
            const size_t pb = sizeof(AggPartial) * (size_t)ps.nvalues;
            RDF_TRY(arena_alloc(64 + pb, &scr));
            RDF_TRY(pinned_reserve(64 + pb));
            memset(cx.pinned, 0, 64 + pb);
            int cls[kMaxValues];
            for (int v = 0; v < ps.nvalues; ++v) {
                cls[v] = value_class(value_dtype[v]);
                AggPartial id;
                memset(&id, 0, sizeof id);
                if (cls[v] == CLS_F64) id.mn = id.mx = 0x7FF8000000000000ull;                     // (agg_init, rdf_common.hip.h)
                else if (cls[v] == CLS_SIGNED) { id.mn = (uint64_t)INT64_MAX; id.mx = (uint64_t)INT64_MIN; }
                else { id.mn = ~0ull; id.mx = 0; }
                memcpy(cx.pinned + 64 + sizeof(AggPartial) * (size_t)v, &id, sizeof id);
            }
            HIP_TRY(hipMemcpyAsync(scr, cx.pinned, 64 + pb, hipMemcpyHostToDevice, cx.stream));
            AggPartial hp[kMaxValues];
            uint32_t flags = 0;
            RDF_TRY(agg_dist_finish(*cx.agg_comm, (const AggPartial*)((char*)scr + 64), (const uint32_t*)scr, ps.nvalues, cls, hp, &flags));
            if (flags & 2u) return fail(RDF_COMPUTE_ERROR, "pipeline_dist: another rank failed before the combine");
            if (flags & 1u) return fail(RDF_DIVIDE_BY_ZERO, "Divide by zero error");
            for (int v = 0; v < ps.nvalues; ++v) fill_agg_result(&aggs[v], value_dtype[v], hp[v]);
        } else {
            for (int v = 0; v < ps.nvalues; ++v) { memset(&aggs[v], 0, sizeof aggs[v]); aggs[v].dtype = value_dtype[v]; }
        }
        return RDF_OK;
    }

    RDF_TRY(ensure_ready());
    Ctx& ctx = g_ctx;
    arena_begin();
    size_t pin_off = 0, pin_used = 0;

    // inputs
    InputStager in;
    if (!fc) {
        in.arrays.reserve((size_t)ncols * (size_t)nchunks);
        in.plans.reserve((size_t)ncols * (size_t)nchunks);
        for (int64_t i = 0; i < (int64_t)ncols * nchunks; ++i) in.add(&cols[i]);
        RDF_TRY(in.finish(pin_off, &pin_used));
        pin_off += (pin_used + 255) & ~(size_t)255;
    }
    const std::vector<DevChunkCol>& in_dev = fc ? fc->dev : in.dev;
    dbg.mark("stage inputs");

    // outputs (SINK_STORE)
    Region outr;
    std::vector<DevOutChunk> dev_outs;
    std::vector<int> out_val_item, out_vld_item;
    if (frame_store) {   // descriptors live on the device; chunk 0's are mirrored for the one-chunk kernels
        if (nchunks == 1) { dev_outs.resize((size_t)ps.nvalues); for (int v = 0; v < ps.nvalues; ++v) dev_outs[(size_t)v] = ps.frame_out0[v]; }
    } else if (ps.sink == RDF_SINK_STORE) {
        dev_outs.resize((size_t)ps.nvalues * nchunks);
        if (mem == RDF_MEM_HOST) {
            out_val_item.assign(dev_outs.size(), -1);
            out_vld_item.assign(dev_outs.size(), -1);
            for (int v = 0; v < ps.nvalues; ++v)
                for (int64_t c = 0; c < nchunks; ++c) {
                    const size_t i = (size_t)((int64_t)v * nchunks + c);
                    const int64_t n = clen[(size_t)c];
                    if (n == 0) continue;
                    const size_t vb = value_dtype[v] == RDF_BOOL ? (size_t)((n + 7) / 8) : (size_t)n * (size_t)dtype_size(value_dtype[v]);
                    out_val_item[i] = outr.add(outs[i].values, vb);
                    if (outs[i].validity) out_vld_item[i] = outr.add(outs[i].validity, (size_t)((n + 7) / 8));
                }
            // word-granular bitmap stores need the items padded to 8 bytes: Region pads every item by >= 16
            RDF_TRY(outr.layout());
            for (size_t i = 0; i < dev_outs.size(); ++i) {
                dev_outs[i].values = out_val_item[i] >= 0 ? outr.ptr(out_val_item[i]) : nullptr;
                dev_outs[i].validity = out_vld_item[i] >= 0 ? (uint8_t*)outr.ptr(out_vld_item[i]) : nullptr;
            }
        } else {
            for (size_t i = 0; i < dev_outs.size(); ++i) dev_outs[i] = DevOutChunk{outs[i].values, outs[i].validity};
        }
    }

    // tiles
    std::vector<int64_t> tile_start;
    const rdf_frame::Tiles* eval_tiles = nullptr;
    if (fc) RDF_TRY(frame_tiles(*fc, kEvalTile, &eval_tiles));
    else {
        tile_start.assign((size_t)nchunks + 1, 0);
        for (int64_t c = 0; c < nchunks; ++c) tile_start[(size_t)c + 1] = tile_start[(size_t)c] + (clen[(size_t)c] + kEvalTile - 1) / kEvalTile;
    }
    const int64_t ntiles = fc ? eval_tiles->ntiles : tile_start[(size_t)nchunks];
    int grid = (int)(ntiles < (int64_t)eval_grid_limit() ? ntiles : (int64_t)eval_grid_limit());

    // device scratch: flags | null counts | partials | result
    const size_t n_nc = ps.sink == RDF_SINK_STORE ? (size_t)ps.nvalues * (size_t)nchunks : 0;
    const int gwords = grouped ? group_words(ps.ngroups, ps.nvalues) : 0;
    const size_t scratch_bytes = 16 + n_nc * 8 + ((size_t)grid + 4) * (grouped ? (size_t)gwords * 8 : (size_t)ps.nvalues * sizeof(AggPartial));   // + 4: the specialised kernels' grid (wave-granular tiles) may round up past this one
    void* scratch = nullptr;
    RDF_TRY(arena_alloc(scratch_bytes, &scratch));
    uint32_t* d_flags = (uint32_t*)scratch;
    int64_t* d_nullc = (int64_t*)((char*)scratch + 16);
    AggPartial* d_partials = (AggPartial*)((char*)scratch + 16 + n_nc * 8);
    AggPartial* d_result = d_partials + (size_t)grid * (size_t)ps.nvalues;
    HIP_TRY(hipMemsetAsync(scratch, 0, 16 + n_nc * 8, ctx.stream));

    EvalArgs ea;
    memset(&ea, 0, sizeof ea);
    ea.nchunks = nchunks;
    ea.ntiles = ntiles;
    ea.ncols = ncols;
    ea.nvalues = ps.nvalues;
    ea.ncode = (int)cc.code.size();
    ea.ntmp = cc.tmp_max;
    ea.flags = d_flags;
    ea.out_null_counts = d_nullc;
    ea.partials = d_partials;
    for (int k = 0; k < ncols; ++k) ea.col_dtype[k] = col_dtype[k];
    int cls[kMaxGroupValues] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int v = 0; v < ps.nvalues; ++v) { cls[v] = value_class(value_dtype[v]); ea.value_cls[v] = cls[v]; }
    memcpy(ea.code, cc.code.data(), cc.code.size() * sizeof(Instr));

    // The general evaluator's chunk tables (descriptors, tile prefix, lengths): built only when that kernel (or the grouped
    // sink, which shares them) is going to run — for a frame of a million 1024-row batches they are 40 MB of host work
    // and upload that a specialised kernel, which carries its own tables, never reads.
    TableBuilder tb;
    auto build_eval_tables = [&]() -> rdf_status {
        if (nchunks == 1) {
            for (int k = 0; k < ncols; ++k) ea.inline_cols[k] = in_dev[(size_t)k];
            for (int v = 0; v < ps.nvalues && ps.sink == RDF_SINK_STORE; ++v) ea.inline_outs[v] = dev_outs[(size_t)v];
            ea.inline_len = clen[0];
        } else if (fc && frame_store) {
            ea.cols = fc->d_cols;
            ea.chunk_tile_start = eval_tiles->d_start;
            ea.chunk_len = fc->d_clen;
            ea.outs = const_cast<DevOutChunk*>(ps.frame_outs);
        } else if (fc) {   // the frame's own tables; only the outputs' descriptors are per call
            ea.cols = fc->d_cols;
            ea.chunk_tile_start = eval_tiles->d_start;
            ea.chunk_len = fc->d_clen;
            const size_t o_outs = tb.reserve(sizeof(DevOutChunk) * (dev_outs.size() + 1));
            RDF_TRY(tb.bind(pin_off));
            if (!dev_outs.empty()) memcpy(tb.at<char>(o_outs), dev_outs.data(), sizeof(DevOutChunk) * dev_outs.size());
            RDF_TRY(tb.alloc());
            RDF_TRY(tb.upload(pin_off));
            pin_off += (tb.size + 255) & ~(size_t)255;
            ea.outs = tb.dev_at<DevOutChunk>(o_outs);
        } else {
            const size_t o_cols = tb.reserve(sizeof(DevChunkCol) * in_dev.size());
            const size_t o_ts = tb.reserve(sizeof(int64_t) * tile_start.size());
            const size_t o_len = tb.reserve(sizeof(int64_t) * clen.size());
            const size_t o_outs = tb.reserve(sizeof(DevOutChunk) * (dev_outs.size() + 1));
            RDF_TRY(tb.bind(pin_off));
            memcpy(tb.at<char>(o_cols), in_dev.data(), sizeof(DevChunkCol) * in_dev.size());
            memcpy(tb.at<char>(o_ts), tile_start.data(), sizeof(int64_t) * tile_start.size());
            memcpy(tb.at<char>(o_len), clen.data(), sizeof(int64_t) * clen.size());
            if (!dev_outs.empty()) memcpy(tb.at<char>(o_outs), dev_outs.data(), sizeof(DevOutChunk) * dev_outs.size());
            RDF_TRY(tb.alloc());
            RDF_TRY(tb.upload(pin_off));
            pin_off += (tb.size + 255) & ~(size_t)255;
            ea.cols = tb.dev_at<DevChunkCol>(o_cols);
            ea.chunk_tile_start = tb.dev_at<int64_t>(o_ts);
            ea.chunk_len = tb.dev_at<int64_t>(o_len);
            ea.outs = tb.dev_at<DevOutChunk>(o_outs);
        }
        return RDF_OK;
    };
    if (grouped) RDF_TRY(build_eval_tables());
    dbg.mark("interpreter tables");
    // specialised straight-line kernel for this program shape? (8-byte columns, every chunk 16-byte aligned)
    SpecPlan sp;
    SpecArgs sa;
    TableBuilder stb;
    bool use_spec = false;
    int spec_rpb = 0;
    int rt_ops[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool have_plan = ctx.opt_spec && !grouped && build_spec_plan(cc, ps.filter_root, ps.nvalues, ps.value_roots, ps.sink, sp);
    if (!have_plan && ctx.opt_spec && !grouped) {   // exact shape not in the catalog: a shape-specialised kernel with runtime operators?
        sp = SpecPlan();
        have_plan = build_shape_plan(cc, ps.filter_root, ps.nvalues, ps.value_roots, ps.sink, sp, rt_ops);
    }
    if (!have_plan && ctx.opt_spec && ctx.opt_jit && !grouped) {
        // neither: the same kernel template instantiated for exactly this program at run time (rdf_jit.cpp) — up to 4 columns and
        // 4 literals, any tree over them; a second or so the first time a process meets the shape, then cached
        sp = SpecPlan();
        for (int k = 0; k < 8; ++k) rt_ops[k] = 0;
        (void)build_spec_plan(cc, ps.filter_root, ps.nvalues, ps.value_roots, ps.sink, sp);
        if (getenv("RDF_DEBUG_JIT")) fprintf(stderr, "[rdf] jit: candidate %s (ok %d)\n", sp.sig.c_str(), (int)sp.ok);
        have_plan = sp.ok && !sp.sig.empty() && jit_spec_kernel(sp.sig.c_str(), ctx.opt_jit >= 2) != nullptr;
    }
    if (have_plan) {
        memset(&sa, 0, sizeof sa);
        use_spec = true;
        spec_rpb = spec_rows_per_tile(sp.sig.c_str());
        for (int k = 0; k < sp.ncols && use_spec; ++k) {
            if (fc) { use_spec = fc->col_aligned16[sp.col_map[k]]; continue; }
            const int es = dtype_size(col_dtype[sp.col_map[k]]);                       // a vector slot of this column: 16 / width of its elements
            const uintptr_t amask = (uintptr_t)(16 / std::max(sp.width, 1)) * (uintptr_t)es - 1;
            for (int64_t c = 0; c < nchunks; ++c) {
                const DevChunkCol& d = in_dev[(size_t)((int64_t)sp.col_map[k] * nchunks + c)];
                if (clen[(size_t)c] > 0 && ((uintptr_t)((const char*)d.values + d.offset * es) & amask) != 0) { use_spec = false; break; }
            }
        }
        if (ps.sink == RDF_SINK_STORE && !frame_store)
            for (int64_t c = 0; c < nchunks && use_spec; ++c)
                if (clen[(size_t)c] > 0 && (((uintptr_t)dev_outs[(size_t)c].values & 15) != 0 || ((uintptr_t)dev_outs[(size_t)c].validity & 7) != 0)) use_spec = false;
    }
    if (use_spec) {
        for (int k = 0; k < sp.nimm; ++k) sa.imm[k] = sp.imm[k];
        for (int k = 0; k < 8; ++k) sa.rt[k] = rt_ops[k];
        for (int k = 0; k < kSpecCols; ++k) {   // a repeated program column is loaded once
            sa.alias[k] = -1;
            for (int j = 0; j < k && k < sp.ncols; ++j) if (sp.col_map[j] == sp.col_map[k]) { sa.alias[k] = j; break; }
        }
        sa.partials = d_partials;
        sa.flags = d_flags;
        sa.vec_bitmap = ctx.opt_vec_bitmap ? 1 : 0;
        sa.out_null_count = d_nullc;
        sa.nchunks = nchunks;
        std::vector<int64_t> sts;
        const rdf_frame::Tiles* spec_tiles = nullptr;
        if (fc) {
            RDF_TRY(frame_tiles(*fc, spec_rpb, &spec_tiles));
            sa.ntiles = spec_tiles->ntiles;
            sa.tile_inv = spec_tiles->tile_inv;
        } else {
            sts.assign((size_t)nchunks + 1, 0);
            for (int64_t c = 0; c < nchunks; ++c) sts[(size_t)c + 1] = sts[(size_t)c] + (clen[(size_t)c] + spec_rpb - 1) / spec_rpb;
            sa.ntiles = sts[(size_t)nchunks];
            sa.tile_inv = 0;
            if (nchunks > 1 && sts[(size_t)nchunks - 1] > 0 && nchunks - 1 < ((int64_t)1 << 31))
                sa.tile_inv = (uint64_t)(((unsigned __int128)(uint64_t)(nchunks - 1) << 32) / (unsigned __int128)(uint64_t)sts[(size_t)nchunks - 1]);
        }
        if (nchunks == 1) {
            for (int k = 0; k < sp.ncols; ++k) sa.cols[k] = in_dev[(size_t)sp.col_map[k]];
            sa.n = clen[0];
            if (ps.sink == RDF_SINK_STORE) sa.out = dev_outs[0];
        } else if (fc) {
            DevChunkCol* tab = nullptr;
            RDF_TRY(frame_col_tab(*fc, sp.col_map, sp.ncols, &tab));
            sa.cols_tab = tab;
            sa.chunk_tile_start = spec_tiles->d_start;
            sa.chunk_len = fc->d_clen;
            if (frame_store) sa.outs_tab = ps.frame_outs;
            else if (ps.sink == RDF_SINK_STORE) {   // the outputs' descriptors are per call
                const size_t o_o = stb.reserve(sizeof(DevOutChunk) * ((size_t)nchunks + 1));
                RDF_TRY(stb.bind(pin_off));
                memcpy(stb.at<char>(o_o), dev_outs.data(), sizeof(DevOutChunk) * (size_t)nchunks);
                RDF_TRY(stb.alloc());
                RDF_TRY(stb.upload(pin_off));
                pin_off += (stb.size + 255) & ~(size_t)255;
                sa.outs_tab = stb.dev_at<DevOutChunk>(o_o);
            }
        } else {
            const size_t o_c = stb.reserve(sizeof(DevChunkCol) * (size_t)(sp.ncols > 0 ? sp.ncols : 1) * (size_t)nchunks);
            const size_t o_t = stb.reserve(sizeof(int64_t) * sts.size());
            const size_t o_l = stb.reserve(sizeof(int64_t) * clen.size());
            const size_t o_o = stb.reserve(sizeof(DevOutChunk) * ((size_t)nchunks + 1));
            RDF_TRY(stb.bind(pin_off));
            for (int k = 0; k < sp.ncols; ++k)
                memcpy(stb.at<DevChunkCol>(o_c) + (size_t)k * (size_t)nchunks, in_dev.data() + (size_t)sp.col_map[k] * (size_t)nchunks, sizeof(DevChunkCol) * (size_t)nchunks);
            memcpy(stb.at<char>(o_t), sts.data(), sizeof(int64_t) * sts.size());
            memcpy(stb.at<char>(o_l), clen.data(), sizeof(int64_t) * clen.size());
            if (ps.sink == RDF_SINK_STORE) memcpy(stb.at<char>(o_o), dev_outs.data(), sizeof(DevOutChunk) * (size_t)nchunks);
            RDF_TRY(stb.alloc());
            RDF_TRY(stb.upload(pin_off));
            pin_off += (stb.size + 255) & ~(size_t)255;
            sa.cols_tab = stb.dev_at<DevChunkCol>(o_c);
            sa.chunk_tile_start = stb.dev_at<int64_t>(o_t);
            sa.chunk_len = stb.dev_at<int64_t>(o_l);
            sa.outs_tab = stb.dev_at<DevOutChunk>(o_o);
        }
        const int64_t btiles = (sa.ntiles + kBlock / 64 - 1) / (kBlock / 64);   // a block's waves take consecutive tiles
        // Resident blocks per CU of the persistent grid.  More is not always better on this part: a one-column filter -> aggregate
        // over a column without a bitmap keeps 128 KB per CU in flight with eight blocks and runs at 0.826-0.844 of the HBM peak;
        // with three (48 KB in flight) at 0.876-0.879, with four 0.866-0.868, with five 0.824-0.830 (bench.py, same box, three
        // alternating rounds; tools/ubench_stream's bare loop shows the same: 4 blocks x 64 B per lane 0.86-0.88, 8 blocks 0.80).
        // Aggregates over several columns follow it: a*b+c -> min / max / count over four columns (config C3) 0.824-0.840 with three
        // against 0.786-0.813 with eight (two boxes, alternating rounds), `x > c AND y < d -> sum` 2.32 against 2.62 ms per 1e9
        // rows; EVEN counts are the bad ones (four: 0.767 on C3, six: 0.770 — the strides between the blocks' tiles then line up
        // with the memory channels' interleave).  Store sinks (new columns): seven — a + b 4.2-4.4 ms per 1e9 rows against 4.6-4.8
        // with eight, with a validity bitmap 4.0-4.1 against 4.8-4.9, a*b+c 5.7-5.95 against 6.2-6.3 (two boxes).  Aggregates over
        // columns with bitmaps measured best at eight (1.34 against 1.35 / 1.46 ms with seven / three) and keep it.  The same program over the readers' 1024-row batches (a descriptor per two tiles): four blocks 0.825-0.830,
        // five 0.810, eight 0.788-0.796, three 0.764, six 0.740-0.745 (two boxes, alternating rounds).
        // rdf_set_option("spec_blocks_per_cu", n) pins a value (A/B).
        // Round 5 (tools/exp_tilewalk.py, three boxes, profiles/r05_tilewalk_*.jsonl): with the chunk tables in SGPRs and the
        // full-tile bitmap path, aggregates run best with 48-64 KB per CU in flight — TWO blocks per CU for the one-column
        // headline shape (0.886-0.894 against 0.869-0.877 with three), three where a bitmap or a batch table adds scalar work
        // per tile (0.858-0.866 with 10 % NULLs, 0.76 in round 4; 1024-row batches 0.863-0.875, 0.82) — and with a tile walk that
        // does not depend on the grid: XCD x takes the x-th contiguous eighth of every row of tiles (xcd_swz), rows rotated by
        // one block per iteration where several columns or batches are walked (tile_rot).  The plain grid stride swung
        // 0.78-0.83 from box to box on C3; rotated / swizzled walks and 8 rows per lane hold 0.827-0.842 on every grid tried.
        int blocks_per_cu = ctx.opt_spec_blocks;
        int walk_rot = ctx.opt_spec_tile_rot, walk_swz = ctx.opt_spec_xcd_swz;     // (< 0: by the program, below)
        {
            bool any_bitmap = false;
            for (int k = 0; k < sp.ncols; ++k) any_bitmap |= fc ? fc->col_nullable[sp.col_map[k]] : (nchunks > 0 && in_dev[(size_t)((int64_t)sp.col_map[k] * nchunks)].validity != nullptr);
            bool heavy = false;
            for (int i = 0; i < ps.nnodes; ++i) heavy |= ps.nodes[i].kind == RDF_NODE_OP && op_is_heavy(ps.nodes[i].op);
            const bool agg = ps.sink == RDF_SINK_AGG && !heavy;
            if (blocks_per_cu <= 0) {
                if (agg) blocks_per_cu = (nchunks == 1 && !any_bitmap) ? 2 : 3;
                else if (ps.sink == RDF_SINK_STORE && nchunks == 1) blocks_per_cu = 7;
                else blocks_per_cu = 8;
            }
            if (walk_swz < 0) walk_swz = agg ? 1 : 0;
            if (walk_rot < 0) walk_rot = agg && (nchunks > 1 || sp.ncols > 1) ? 1 : 0;
        }
        const int64_t spec_limit = (int64_t)(eval_grid_limit() / 8) * std::max(1, std::min(8, blocks_per_cu));
        grid = (int)(btiles < spec_limit ? btiles : spec_limit);
        if (grid == spec_limit && ctx.opt_spec_grid_adj != 0) grid = std::max(1, std::min(eval_grid_limit(), grid + ctx.opt_spec_grid_adj));
        if (grid < 1) grid = 1;
        {
            const int64_t nwv = (int64_t)grid * (kBlock / 64);
            sa.tile_rot = ((int64_t)walk_rot * (kBlock / 64)) % nwv;
            sa.xcd_swz = walk_swz ? 1 : 0;
        }
        d_result = d_partials + (size_t)grid * (size_t)ps.nvalues;
    }

    dbg.mark("specialised tables");
    if (!grouped && !use_spec) RDF_TRY(build_eval_tables());
    if (grouped) {
        ea.ngroups = ps.ngroups;
        // LDS copies of the table: as many (power of two, <= 32) as fit in 32 KB next to the TMP spill area
        const size_t tmp_bytes = (size_t)cc.tmp_max * (kVPT * kBlock * 8 + kBlock * 4);
        int reps = 32;
        while (reps > 1 && ((size_t)reps * (size_t)gwords * 8 > 32768 || tmp_bytes + (size_t)reps * (size_t)gwords * 8 > 65536)) reps >>= 1;
        ea.group_replicas = reps;
        uint64_t* d_gpart = (uint64_t*)d_partials;
        uint64_t* d_gres = d_gpart + (size_t)grid * (size_t)gwords;
        ea.group_partials = d_gpart;
        // a specialised register-accumulator kernel for this shape?  (every column aligned for 2-row vector loads)
        SpecPlan gp;
        bool use_gspec = ctx.opt_spec && ps.ngroups <= 8 &&
                         build_gspec_plan(cc, ps.filter_root, ps.group_root, ps.ngroups, ps.nvalues, ps.value_roots, gp);
        for (int k = 0; k < gp.ncols && use_gspec; ++k) {
            const int w = dtype_size(col_dtype[gp.col_map[k]]);
            for (int64_t c = 0; c < nchunks; ++c) {
                const DevChunkCol& d = in_dev[(size_t)((int64_t)gp.col_map[k] * nchunks + c)];
                if (clen[(size_t)c] > 0 && ((uintptr_t)((const char*)d.values + d.offset * w) & (uintptr_t)(2 * w - 1)) != 0) { use_gspec = false; break; }
            }
        }
        if (use_gspec) {
            GSpecArgs ga;
            memset(&ga, 0, sizeof ga);
            ga.cols_tab = ea.cols; ga.chunk_tile_start = ea.chunk_tile_start; ga.chunk_len = ea.chunk_len;
            ga.nchunks = nchunks; ga.ntiles = ntiles; ga.n = clen[0];
            for (int k = 0; k < gp.ncols; ++k) { ga.col_map[k] = gp.col_map[k]; if (nchunks == 1) ga.cols[k] = in_dev[(size_t)gp.col_map[k]]; }
            for (int k = 0; k < gp.nimm; ++k) ga.imm[k] = gp.imm[k];
            ga.group_partials = d_gpart; ga.flags = d_flags;
            ga.ngroups = ps.ngroups; ga.nvalues = ps.nvalues; ga.vec_bitmap = ctx.opt_vec_bitmap ? 1 : 0;
            // the kernel keeps G x NV accumulators per lane (~220 VGPRs for Q1): two waves per SIMD = two blocks per CU are resident
            // whatever is launched; launching just those keeps the grid persistent (one prologue / epilogue per block, and the
            // next-tile prefetch of rdf_gspec_kernel.hip.h never runs dry at a block's end)
            {
                const int per_cu = ctx.opt_gspec_blocks > 0 ? ctx.opt_gspec_blocks : 2;
                const int64_t lim = (int64_t)(eval_grid_limit() / 8) * per_cu;
                if (grid > lim) grid = (int)lim;
                // (the walks that help the ungrouped aggregates do not help here — Q1, same box: plain 0.769 of peak, swizzled 0.755,
                // rotated 0.752, both 0.748 — so the grouped kernel keeps the plain grid stride unless an option asks)
                ga.xcd_swz = ctx.opt_spec_xcd_swz > 0 ? 1 : 0;
                ga.tile_rot = (ctx.opt_spec_tile_rot > 0 ? ctx.opt_spec_tile_rot : 0) % std::max(1, grid);
            }
            KernelTimer kt;
            ctx.last_kernel = "gspec_kernel<" + gp.sig + ">" + (jit_find(gp.sig.c_str()) ? " [compiled at run time]" : "");
            const hipError_t le = launch_gspec(gp.sig.c_str(), ga, grid, ctx.stream);
            if (le != hipSuccess && jit_find(gp.sig.c_str()) == nullptr && !gspec_available(gp.sig.c_str())) {     // a run-time kernel that could not be launched (now marked failed)
                (void)hipGetLastError();
                return run_program(ps, cols, ncols, nchunks, outs, aggs, len_mismatch_msg, fc);
            }
            if (le != hipSuccess) return fail(RDF_DEVICE_ERROR, "launch_gspec: %s", hipGetErrorString(le));
            kt.stop();
        } else {
            KernelTimer kt;
            ctx.last_kernel = "eval_kernel<GROUP>";
            HIP_TRY(launch_eval(ea, SINK_GROUP, cc.feat(), grid, ctx.stream));
            kt.stop();
        }
        GroupFinalArgs gf;
        memset(&gf, 0, sizeof gf);
        gf.partials = d_gpart; gf.result = d_gres; gf.nblocks = grid; gf.words = gwords; gf.ngroups = ps.ngroups; gf.nvalues = ps.nvalues;
        for (int v = 0; v < ps.nvalues; ++v) gf.value_cls[v] = cls[v];
        HIP_TRY(launch_group_final(gf, ctx.stream));
        RDF_TRY(pinned_reserve(pin_off + 64 + (size_t)gwords * 8));
        char* pin = ctx.pinned + pin_off;
        HIP_TRY(hipMemcpyAsync(pin, d_flags, 16, hipMemcpyDeviceToHost, ctx.stream));
        HIP_TRY(hipMemcpyAsync(pin + 64, d_gres, (size_t)gwords * 8, hipMemcpyDeviceToHost, ctx.stream));
        HIP_TRY(hipStreamSynchronize(ctx.stream));
        uint32_t flags;
        memcpy(&flags, pin, 4);
        if (flags & 1u) return fail(RDF_DIVIDE_BY_ZERO, "Divide by zero error");
        if (flags & 2u) return fail(RDF_COMPUTE_ERROR, "group id outside [0, %d)", ps.ngroups);
        const uint64_t* w = (const uint64_t*)(pin + 64);
        const int S = ps.ngroups + 1;
        for (int v = 0; v < ps.nvalues; ++v)
            for (int g = 0; g < S; ++g) {
                rdf_group_result& r = ps.gout[(size_t)v * (size_t)S + (size_t)g];
                memset(&r, 0, sizeof r);
                r.dtype = value_dtype[v];
                r.count = (int64_t)(w[2 * ps.nvalues * S + g] - w[(ps.nvalues + v) * S + g]);
                r.is_some = r.count > 0;
                if (is_float(value_dtype[v])) memcpy(&r.sum_f64, &w[v * S + g], 8);
                else r.sum_i64 = (int64_t)w[v * S + g];
            }
        if (ps.grows) for (int g = 0; g < S; ++g) ps.grows[g] = (int64_t)w[2 * ps.nvalues * S + g];
        return RDF_OK;
    }

    if (ps.sink == RDF_SINK_AGG) {
        // fast path: filter(x CMP c) -> aggregates of y, both f64, one chunk, 16-byte-alignable
        FilterAggF64Args fa;
        bool fast = false;
        int cmp = 0;
        if (!use_spec && ctx.opt_fast_filter && nchunks == 1 && ps.nvalues == 1 && ps.filter_root >= 0) {
            const rdf_expr_node& fr = ps.nodes[ps.filter_root];
            const rdf_expr_node& vr = ps.nodes[ps.value_roots[0]];
            if (fr.kind == RDF_NODE_OP && op_is_cmp(fr.op) && vr.kind == RDF_NODE_COLUMN && col_dtype[vr.column] == RDF_F64) {
                const rdf_expr_node& L = ps.nodes[fr.lhs];
                const rdf_expr_node& R = ps.nodes[fr.rhs];
                const rdf_expr_node* coln = nullptr;
                const rdf_expr_node* sc = nullptr;
                cmp = fr.op;
                if (L.kind == RDF_NODE_COLUMN && R.kind == RDF_NODE_SCALAR) { coln = &L; sc = &R; }
                else if (L.kind == RDF_NODE_SCALAR && R.kind == RDF_NODE_COLUMN) {
                    coln = &R; sc = &L;  // c CMP x  ==  x CMP' c
                    cmp = fr.op == RDF_OP_GT ? RDF_OP_LT : fr.op == RDF_OP_GE ? RDF_OP_LE : fr.op == RDF_OP_LT ? RDF_OP_GT : fr.op == RDF_OP_LE ? RDF_OP_GE : fr.op;
                }
                if (coln && col_dtype[coln->column] == RDF_F64 && sc->dtype != RDF_NULLTYPE && (is_numeric(sc->dtype) || sc->dtype == RDF_BOOL)) {
                    const DevChunkCol& x = in_dev[(size_t)coln->column];
                    const DevChunkCol& y = in_dev[(size_t)vr.column];
                    const uintptr_t xa = (uintptr_t)((const double*)x.values + x.offset), ya = (uintptr_t)((const double*)y.values + y.offset);
                    if ((xa & 7) == 0 && (ya & 7) == 0 && (xa & 15) == (ya & 15)) {
                        memset(&fa, 0, sizeof fa);
                        fa.x = (const double*)x.values; fa.x_validity = x.validity; fa.x_offset = x.offset;
                        fa.y = (const double*)y.values; fa.y_validity = y.validity; fa.y_offset = y.offset;
                        fa.n = clen[0];
                        const uint64_t cu = cc.imm_for(*sc, RDF_F64);
                        memcpy(&fa.c, &cu, 8);
                        fa.partials = d_partials;
                        fast = true;
                    }
                }
            }
        }
        if (use_spec) {
            const rdf_status ls = launch_agg_pair(nullptr, nullptr, 0, 0, grid, ps.nvalues, cls, d_partials, d_result, sp.sig.c_str(), &sa);
            if (ls == kRetryInterpreted) return run_program(ps, cols, ncols, nchunks, outs, aggs, len_mismatch_msg, fc);
            RDF_TRY(ls);
        } else if (fast) {
            const int64_t per_block = (int64_t)kBlock * 4 * 2;  // rows per block iteration
            int64_t want = (clen[0] + per_block - 1) / per_block;
            grid = (int)(want < (int64_t)eval_grid_limit() ? want : (int64_t)eval_grid_limit());
            if (grid < 1) grid = 1;
            // partials/result were sized for the eval grid (>= this grid, since its tiles are smaller)
            d_result = d_partials + (size_t)grid * (size_t)ps.nvalues;
            RDF_TRY(launch_agg_pair(nullptr, &fa, cmp, 0, grid, 1, cls, d_partials, d_result));
        } else {
            RDF_TRY(launch_agg_pair(&ea, nullptr, 0, cc.feat(), grid, ps.nvalues, cls, d_partials, d_result));
        }
        // results: flags + aggregates in one D2H
        if (ctx.agg_comm) {
            // rdf_pipeline_dist: the partials never visit this host — all-gathered and folded on the device, one wait for the total
            AggPartial hp[kMaxValues];
            uint32_t flags = 0;
            RDF_TRY(agg_dist_finish(*ctx.agg_comm, d_result, d_flags, ps.nvalues, cls, hp, &flags));
            if (flags & 2u) return fail(RDF_COMPUTE_ERROR, "pipeline_dist: another rank failed before the combine");
            if (flags & 1u) return fail(RDF_DIVIDE_BY_ZERO, "Divide by zero error");
            for (int v = 0; v < ps.nvalues; ++v) fill_agg_result(&aggs[v], value_dtype[v], hp[v]);
            return RDF_OK;
        }
        RDF_TRY(pinned_reserve(pin_off + 64 + sizeof(AggPartial) * kMaxValues));
        char* pin = ctx.pinned + pin_off;
        HIP_TRY(hipMemcpyAsync(pin, d_flags, 16, hipMemcpyDeviceToHost, ctx.stream));
        HIP_TRY(hipMemcpyAsync(pin + 64, d_result, sizeof(AggPartial) * (size_t)ps.nvalues, hipMemcpyDeviceToHost, ctx.stream));
        HIP_TRY(hipStreamSynchronize(ctx.stream));
        dbg.mark("kernel + result");
        uint32_t flags;
        memcpy(&flags, pin, 4);
        if (flags & 1u) return fail(RDF_DIVIDE_BY_ZERO, "Divide by zero error");
        for (int v = 0; v < ps.nvalues; ++v) {
            AggPartial p;
            memcpy(&p, pin + 64 + sizeof(AggPartial) * (size_t)v, sizeof p);
            fill_agg_result(&aggs[v], value_dtype[v], p);
        }
        return RDF_OK;
    }

    // SINK_STORE
    {
        KernelTimer kt;
        if (use_spec) {
            const bool jit = jit_find(sp.sig.c_str()) != nullptr;
            ctx.last_kernel = "spec_kernel<" + sp.sig + ">" + (jit ? " [compiled at run time]" : "");
            const hipError_t le = launch_spec(sp.sig.c_str(), sa, grid, ctx.stream);
            if (le != hipSuccess && jit) { (void)hipGetLastError(); return run_program(ps, cols, ncols, nchunks, outs, aggs, len_mismatch_msg, fc); }   // marked failed: interpreted this time and from now on
            if (le != hipSuccess) return fail(RDF_DEVICE_ERROR, "launch_spec: %s", hipGetErrorString(le));
        }
        else {
            // (the lean kernel adds a tile's NULLs into the low word of the chunk's 64-bit count: chunks of 2^32 rows and more stay on eval_kernel)
            int64_t longest = 0;
            for (int64_t i = 0; i < nchunks; ++i) longest = std::max(longest, clen[(size_t)i]);
            EvalArgs lean;
            if (ctx.opt_interp_lean && cc.feat() == 0 && longest < ((int64_t)1 << 32) && lean_assign(ea, lean, SINK_STORE)) {
                ctx.last_kernel = "eval_kernel<STORE, lean>";
                HIP_TRY(launch_eval_lean(lean, SINK_STORE, grid, ctx.stream, ctx.opt_interp_lean == 2));
            } else { ctx.last_kernel = "eval_kernel<STORE>"; HIP_TRY(launch_eval(ea, SINK_STORE, cc.feat(), grid, ctx.stream)); }
        }
        kt.stop();
    }
    if (frame_store) {   // flags only: lengths are the frame's batch lengths, null counts stay on the device
        RDF_TRY(pinned_reserve(pin_off + 64));
        HIP_TRY(hipMemcpyAsync(ctx.pinned + pin_off, scratch, 16, hipMemcpyDeviceToHost, ctx.stream));
        HIP_TRY(hipStreamSynchronize(ctx.stream));
        uint32_t fl;
        memcpy(&fl, ctx.pinned + pin_off, 4);
        if (fl & 1u) return fail(RDF_DIVIDE_BY_ZERO, "Divide by zero error");
        return RDF_OK;
    }
    RDF_TRY(pinned_reserve(pin_off + 64 + n_nc * 8 + outr.small_bytes + 256));
    char* pin = ctx.pinned + pin_off;
    HIP_TRY(hipMemcpyAsync(pin, scratch, 16 + n_nc * 8, hipMemcpyDeviceToHost, ctx.stream));
    if (mem == RDF_MEM_HOST) RDF_TRY(outr.download(pin_off + ((16 + n_nc * 8 + 255) & ~(size_t)255)));
    else HIP_TRY(hipStreamSynchronize(ctx.stream));
    uint32_t flags;
    memcpy(&flags, pin, 4);
    if (flags & 1u) return fail(RDF_DIVIDE_BY_ZERO, "Divide by zero error");
    for (int v = 0; v < ps.nvalues; ++v)
        for (int64_t c = 0; c < nchunks; ++c) {
            rdf_out& o = outs[(int64_t)v * nchunks + c];
            o.length = clen[(size_t)c];
            memcpy(&o.null_count, pin + 16 + 8 * (size_t)((int64_t)v * nchunks + c), 8);
        }
    return RDF_OK;
}

// one-node-per-op helper for the single-kernel entry points
rdf_expr_node node_col(int c) { rdf_expr_node n; memset(&n, 0, sizeof n); n.kind = RDF_NODE_COLUMN; n.column = c; n.lhs = n.rhs = -1; return n; }
rdf_expr_node node_op(int op, int l, int r, int dtype = 0) { rdf_expr_node n; memset(&n, 0, sizeof n); n.kind = RDF_NODE_OP; n.op = op; n.lhs = l; n.rhs = r; n.dtype = dtype; return n; }

rdf_status pipeline_stream(const ProgramSpec& ps, const rdf_array* cols, int32_t ncols, int64_t nchunks, rdf_agg_result* aggs, const char* msg);   // rdf_capi_stream.inc
int64_t host_input_bytes(const rdf_array* cols, int32_t ncols, int64_t nchunks);
int64_t stream_slab_bytes();
rdf_status pipeline_stream_store(const ProgramSpec& ps, const rdf_array* cols, int32_t ncols, int64_t nchunks, rdf_out* outs, const char* msg);
rdf_status group_pipeline_stream(const ProgramSpec& ps, const rdf_array* cols, int32_t ncols, int64_t nchunks, const char* msg);
rdf_status filter_stream(const rdf_expr_node* nodes, int32_t nnodes, int32_t root, const rdf_array* cols, int32_t ncols, int64_t nchunks, rdf_out* outs);
rdf_status groupby_stream(const rdf_array* keys, const rdf_array* values, int64_t nchunks, int32_t agg, int64_t max_groups,
                          rdf_out* out_keys, rdf_out* out_values, rdf_out* out_counts);

// Every entry point that takes chunk lists comes through here.  Host-resident batches beyond one slab are streamed (slab k + 1
// crosses the link while the kernel runs over slab k, and — for sinks that materialise on the host — slab k - 1's results
// leave on a third stream): aggregates, new columns / masks, the fused grouped aggregation.  Anything else is one run_program.
rdf_status run_program_any(const ProgramSpec& ps, const rdf_array* cols, int ncols, int64_t nchunks, rdf_out* outs, rdf_agg_result* aggs, const char* msg) {
    g_ctx.stream_slabs = 0;
    if (cols && ncols >= 1 && ncols <= kMaxCols && nchunks >= 1 && g_ctx.opt_stream_slab >= 0) {
        bool host = true;
        for (int64_t i = 0; i < (int64_t)ncols * nchunks && host; ++i) host = cols[i].mem == RDF_MEM_HOST && cols[i].length >= 0 && cols[i].offset >= 0 && (cols[i].length == 0 || cols[i].values);
        if (host && host_input_bytes(cols, ncols, nchunks) > stream_slab_bytes()) {
            if (ps.sink == RDF_SINK_AGG && aggs) return pipeline_stream(ps, cols, ncols, nchunks, aggs, msg);
            if (ps.sink == RDF_SINK_GROUP && ps.gout && ps.ngroups >= 1 && ps.nvalues >= 1 && (int64_t)(ps.ngroups + 1) * ps.nvalues <= RDF_MAX_GROUP_SLOTS)
                return group_pipeline_stream(ps, cols, ncols, nchunks, msg);
            if (ps.sink == RDF_SINK_STORE && outs && ps.nvalues == 1 && ps.filter_root < 0 && !ps.frame_outs) {
                bool ohost = true;
                for (int64_t c = 0; c < nchunks && ohost; ++c) ohost = outs[c].mem == RDF_MEM_HOST;
                if (ohost) return pipeline_stream_store(ps, cols, ncols, nchunks, outs, msg);
            }
        }
    }
    return run_program(ps, cols, ncols, nchunks, outs, aggs, msg);
}

rdf_status agg_column(const rdf_array* a, int64_t nchunks, bool as_f64, rdf_agg_result* r) {
    rdf_expr_node nodes[2] = {node_col(0), node_op(RDF_OP_CAST, 0, -1, RDF_F64)};
    ProgramSpec ps;
    memset(&ps, 0, sizeof ps);
    ps.nodes = nodes; ps.nnodes = 2; ps.filter_root = -1; ps.nvalues = 1; ps.value_roots[0] = as_f64 ? 1 : 0; ps.sink = RDF_SINK_AGG;
    return run_program_any(ps, a, 1, nchunks, nullptr, r, "chunk length mismatch");
}

void store_native(void* out, int dt, const rdf_agg_result& r, int which /*0 sum 1 min 2 max*/) {
    if (is_float(dt)) {
        const double v = which == 0 ? r.sum_f64 : which == 1 ? r.min_f64 : r.max_f64;
        if (dt == RDF_F64) *(double*)out = v; else *(float*)out = (float)v;
        return;
    }
    const int64_t v = which == 0 ? r.sum_i64 : which == 1 ? r.min_i64 : r.max_i64;
    switch (dtype_size(dt)) {
        case 1: *(uint8_t*)out = (uint8_t)v; break;
        case 2: *(uint16_t*)out = (uint16_t)v; break;
        case 4: *(uint32_t*)out = (uint32_t)v; break;
        default: *(uint64_t*)out = (uint64_t)v; break;
    }
}

rdf_status agg_entry(const rdf_array* a, int64_t nchunks, void* out_scalar, int32_t* out_is_some, int which, const char* name) {
    if (!out_scalar || !out_is_some) return fail(RDF_INVALID_ARGUMENT, "%s: null output pointer", name);
    // a column always has at least one chunk (ChunkedArray::from_arrays asserts, src/table.rs:25)
    if (nchunks < 1 || !a) return fail(RDF_INVALID_ARGUMENT, "%s: a column has at least one chunk", name);
    if (!is_numeric(a[0].dtype)) return fail(RDF_INVALID_ARGUMENT, "%s: numeric type required", name);
    rdf_agg_result r;
    RDF_TRY(agg_column(a, nchunks, false, &r));
    if (which == 0) { store_native(out_scalar, a[0].dtype, r, 0); *out_is_some = 1; }
    else { *out_is_some = r.is_some; if (r.is_some) store_native(out_scalar, a[0].dtype, r, which); }
    return RDF_OK;
}

}  // namespace

// ================================================================================================
// extern "C" boundary

extern "C" {

const char* rdf_version(void) { return "rdf_mi355x 0.1.0 (gfx950)"; }
const char* rdf_last_error(void) { return g_ctx.err.c_str(); }

rdf_status rdf_device_count(int32_t* count) {
    if (!count) return fail(RDF_INVALID_ARGUMENT, "null pointer");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    *count = e == hipSuccess ? n : 0;
    if (e != hipSuccess) return fail(RDF_DEVICE_ERROR, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return RDF_OK;
}

rdf_status rdf_set_device(int32_t device) {
    Ctx& c = g_ctx;
    if (c.ready && c.device == device) return RDF_OK;
    if (c.ready) {  // drop everything tied to the old device
        (void)hipStreamSynchronize(c.stream);
        for (void* p : c.arena.overflow) (void)hipFree(p);
        c.arena.overflow.clear();
        if (c.arena.base) (void)hipFree(c.arena.base);
        c.arena = Arena();
        for (auto& ev : c.events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
        c.events.clear();
        c.events_used = 0;
        if (c.own_stream) (void)hipStreamDestroy(c.own_stream);
        c.own_stream = c.stream = nullptr;
        c.ready = false;
    }
    HIP_TRY(hipSetDevice(device));
    return ensure_ready();
}

rdf_status rdf_set_stream(void* hip_stream) {
    RDF_TRY(ensure_ready());
    g_ctx.stream = hip_stream ? (hipStream_t)hip_stream : g_ctx.own_stream;
    return RDF_OK;
}

rdf_status rdf_synchronize(void) {
    RDF_TRY(ensure_ready());
    HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    return RDF_OK;
}

rdf_status rdf_dev_alloc(void** ptr, int64_t bytes) {
    if (!ptr || bytes < 0) return fail(RDF_INVALID_ARGUMENT, "bad arguments");
    RDF_TRY(ensure_ready());
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, (size_t)bytes + 64);
    if (e != hipSuccess) return fail(RDF_MEMORY_ERROR, "hipMalloc(%lld): %s", (long long)bytes, hipGetErrorString(e));
    *ptr = p;
    return RDF_OK;
}
rdf_status rdf_dev_free(void* ptr) {
    if (!ptr) return RDF_OK;
    RDF_TRY(ensure_ready());
    HIP_TRY(hipFree(ptr));
    return RDF_OK;
}
rdf_status rdf_copy_h2d(void* dst_dev, const void* src_host, int64_t bytes) {
    RDF_TRY(ensure_ready());
    if (bytes > 0) {
        HIP_TRY(hipMemcpyAsync(dst_dev, src_host, (size_t)bytes, hipMemcpyHostToDevice, g_ctx.stream));
        HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    }
    return RDF_OK;
}
// ---- ingestion: page-locked host memory + uploads that do not block the caller (DataFrame::from_csv / from_arrow,
// src/dataframe.rs:349-407: the reader's buffers are pinned, every record batch's buffers go up on a copy stream while the
// next batch is being parsed, and ONE fence ends the load)
rdf_status rdf_host_alloc(void** ptr, int64_t bytes) {
    if (!ptr || bytes < 0) return fail(RDF_INVALID_ARGUMENT, "host_alloc: bad arguments");
    RDF_TRY(ensure_ready());
    hipError_t e = hipHostMalloc(ptr, (size_t)(bytes > 0 ? bytes : 1), hipHostMallocDefault);
    if (e != hipSuccess) return fail(RDF_MEMORY_ERROR, "hipHostMalloc(%lld) failed: %s", (long long)bytes, hipGetErrorString(e));
    return RDF_OK;
}
rdf_status rdf_host_free(void* ptr) {
    if (ptr) HIP_TRY(hipHostFree(ptr));
    return RDF_OK;
}
rdf_status rdf_host_register(void* ptr, int64_t bytes) {
    if (!ptr || bytes <= 0) return fail(RDF_INVALID_ARGUMENT, "host_register: bad arguments");
    RDF_TRY(ensure_ready());
    hipError_t e = hipHostRegister(ptr, (size_t)bytes, hipHostRegisterDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); g_ctx.err = std::string("hipHostRegister: ") + hipGetErrorString(e); return RDF_MEMORY_ERROR; }   // the caller falls back to blocking copies
    return RDF_OK;
}
rdf_status rdf_host_unregister(void* ptr) {
    if (ptr) HIP_TRY(hipHostUnregister(ptr));
    return RDF_OK;
}
rdf_status rdf_copy_h2d_async(void* dst_dev, const void* src_host, int64_t bytes) {
    RDF_TRY(ensure_ready());
    Ctx& c = g_ctx;
    if (!c.copy_stream) HIP_TRY(hipStreamCreateWithFlags(&c.copy_stream, hipStreamNonBlocking));
    if (bytes > 0) {
        HIP_TRY(hipMemcpyAsync(dst_dev, src_host, (size_t)bytes, hipMemcpyHostToDevice, c.copy_stream));
        c.copy_pending = true;
    }
    return RDF_OK;
}
rdf_status rdf_copy_fence(void) {
    RDF_TRY(ensure_ready());
    Ctx& c = g_ctx;
    if (c.copy_stream && c.copy_pending) HIP_TRY(hipStreamSynchronize(c.copy_stream));
    c.copy_pending = false;
    return RDF_OK;
}

rdf_status rdf_copy_d2h(void* dst_host, const void* src_dev, int64_t bytes) {
    RDF_TRY(ensure_ready());
    if (bytes > 0) {
        HIP_TRY(hipMemcpyAsync(dst_host, src_dev, (size_t)bytes, hipMemcpyDeviceToHost, g_ctx.stream));
        HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    }
    return RDF_OK;
}

// ---------------------------------------------------------------- scalar kernels

rdf_status rdf_binary(int32_t op, const rdf_array* a, const rdf_array* b, int64_t nchunks, rdf_out* out) {
    if (!(op_is_arith(op) || op_is_fbinary(op))) return fail(RDF_INVALID_ARGUMENT, "not a binary op: %d", op);
    if (nchunks < 0 || (nchunks > 0 && (!a || !b || !out))) return fail(RDF_INVALID_ARGUMENT, "bad chunk lists");
    if (nchunks == 0) return RDF_OK;
    for (int64_t c = 0; c < nchunks; ++c)  // compute::add(a, b): per-pair length check comes first
        if (a[c].length != b[c].length) return fail(RDF_COMPUTE_ERROR, "Cannot perform math operation on arrays of different length");
    std::vector<rdf_array> cols((size_t)(2 * nchunks));
    for (int64_t c = 0; c < nchunks; ++c) { cols[(size_t)c] = a[c]; cols[(size_t)(nchunks + c)] = b[c]; }
    rdf_expr_node nodes[3] = {node_col(0), node_col(1), node_op(op, 0, 1)};
    ProgramSpec ps;
    memset(&ps, 0, sizeof ps);
    ps.nodes = nodes; ps.nnodes = 3; ps.filter_root = -1; ps.nvalues = 1; ps.value_roots[0] = 2; ps.sink = RDF_SINK_STORE;
    return run_program_any(ps, cols.data(), 2, nchunks, out, nullptr, "Cannot perform math operation on arrays of different length");
}

rdf_status rdf_unary(int32_t op, const rdf_array* a, int64_t nchunks, rdf_out* out) {
    if (!op_is_unary_math(op)) return fail(RDF_INVALID_ARGUMENT, "not a unary op: %d", op);
    if (nchunks < 0 || (nchunks > 0 && (!a || !out))) return fail(RDF_INVALID_ARGUMENT, "bad chunk lists");
    if (nchunks == 0) return RDF_OK;
    rdf_expr_node nodes[2] = {node_col(0), node_op(op, 0, -1)};
    ProgramSpec ps;
    memset(&ps, 0, sizeof ps);
    ps.nodes = nodes; ps.nnodes = 2; ps.filter_root = -1; ps.nvalues = 1; ps.value_roots[0] = 1; ps.sink = RDF_SINK_STORE;
    return run_program_any(ps, a, 1, nchunks, out, nullptr, "chunk length mismatch");
}

rdf_status rdf_hour(const rdf_array* a, int64_t nchunks, int32_t unit, rdf_out* out) {
    if (unit < RDF_TIME_SECOND || unit > RDF_TIME_DAY) return fail(RDF_INVALID_ARGUMENT, "hour: unknown time unit %d", unit);
    if (nchunks < 0 || (nchunks > 0 && (!a || !out))) return fail(RDF_INVALID_ARGUMENT, "bad chunk lists");
    if (nchunks == 0) return RDF_OK;
    // "hour does not support" anything but the temporal types: their storage is Int32 or Int64
    if (a[0].dtype != RDF_I32 && a[0].dtype != RDF_I64) return fail(RDF_COMPUTE_ERROR, "hour does not support type %d", a[0].dtype);
    rdf_expr_node nodes[3] = {node_col(0), node_op(RDF_OP_HOUR_S + unit, 0, -1), node_op(RDF_OP_CAST, 1, -1, RDF_I32)};
    ProgramSpec ps;
    memset(&ps, 0, sizeof ps);
    ps.nodes = nodes; ps.nnodes = 3; ps.filter_root = -1; ps.nvalues = 1; ps.value_roots[0] = 2; ps.sink = RDF_SINK_STORE;
    ps.casts_always_fit = true;
    return run_program_any(ps, a, 1, nchunks, out, nullptr, "chunk length mismatch");
}

rdf_status rdf_cast(const rdf_array* a, int64_t nchunks, rdf_out* out) {
    if (nchunks < 0 || (nchunks > 0 && (!a || !out))) return fail(RDF_INVALID_ARGUMENT, "bad chunk lists");
    if (nchunks == 0) return RDF_OK;
    rdf_expr_node nodes[2] = {node_col(0), node_op(RDF_OP_CAST, 0, -1, out[0].dtype)};
    ProgramSpec ps;
    memset(&ps, 0, sizeof ps);
    ps.nodes = nodes; ps.nnodes = 2; ps.filter_root = -1; ps.nvalues = 1; ps.value_roots[0] = 1; ps.sink = RDF_SINK_STORE;
    return run_program_any(ps, a, 1, nchunks, out, nullptr, "chunk length mismatch");
}

// ---------------------------------------------------------------- aggregates

rdf_status rdf_sum(const rdf_array* a, int64_t nchunks, void* out_scalar, int32_t* out_is_some) { return agg_entry(a, nchunks, out_scalar, out_is_some, 0, "sum"); }
rdf_status rdf_min(const rdf_array* a, int64_t nchunks, void* out_scalar, int32_t* out_is_some) { return agg_entry(a, nchunks, out_scalar, out_is_some, 1, "min"); }
rdf_status rdf_max(const rdf_array* a, int64_t nchunks, void* out_scalar, int32_t* out_is_some) { return agg_entry(a, nchunks, out_scalar, out_is_some, 2, "max"); }

rdf_status rdf_count(const rdf_array* a, int64_t nchunks, int64_t* out_count, int32_t* out_is_some) {
    if (!out_count || !out_is_some) return fail(RDF_INVALID_ARGUMENT, "count: null output pointer");
    if (nchunks < 0 || (nchunks > 0 && !a)) return fail(RDF_INVALID_ARGUMENT, "count: bad chunk list");
    *out_is_some = 1;  // Some(sum) always (aggregate.rs:79)
    bool known = true;
    int64_t sum = 0;
    for (int64_t c = 0; c < nchunks; ++c) {
        if (a[c].validity == nullptr) sum += a[c].length;
        else if (a[c].null_count >= 0) sum += a[c].length - a[c].null_count;
        else known = false;
    }
    if (known) { *out_count = sum; return RDF_OK; }  // O(#chunks), metadata only — like the reference
    rdf_agg_result r;
    RDF_TRY(agg_column(a, nchunks, false, &r));  // count the validity bits on the device
    *out_count = r.count;
    return RDF_OK;
}

rdf_status rdf_avg(const rdf_array* a, int64_t nchunks, double* out_mean, int32_t* out_is_some) {
    if (!out_mean || !out_is_some) return fail(RDF_INVALID_ARGUMENT, "avg: null output pointer");
    if (nchunks < 0 || (nchunks > 0 && !a)) return fail(RDF_INVALID_ARGUMENT, "avg: bad chunk list");
    if (nchunks == 0) { *out_is_some = 0; return RDF_OK; }
    if (!is_numeric(a[0].dtype)) return fail(RDF_INVALID_ARGUMENT, "avg: numeric type required");
    rdf_agg_result r;
    RDF_TRY(agg_column(a, nchunks, true, &r));  // f64::from(value) then mean, aggregate.rs:47
    *out_is_some = r.is_some;
    if (r.is_some) *out_mean = r.sum_f64 / (double)r.count;
    return RDF_OK;
}

// ---------------------------------------------------------------- expressions

rdf_status rdf_predicate(const rdf_expr_node* nodes, int32_t nnodes, int32_t root, const rdf_array* cols, int32_t ncols,
                         int64_t nchunks, rdf_out* mask) {
    if (!nodes || nnodes <= 0) return fail(RDF_INVALID_ARGUMENT, "empty expression");
    if (nchunks < 0 || (nchunks > 0 && !mask)) return fail(RDF_INVALID_ARGUMENT, "bad chunk lists");
    if (nchunks == 0) return RDF_OK;
    ProgramSpec ps;
    memset(&ps, 0, sizeof ps);
    ps.nodes = nodes; ps.nnodes = nnodes; ps.filter_root = -1; ps.nvalues = 1; ps.value_roots[0] = root; ps.sink = RDF_SINK_STORE;
    for (int64_t c = 0; c < nchunks; ++c)
        if (mask[c].dtype != RDF_BOOL) return fail(RDF_INVALID_ARGUMENT, "predicate root must be boolean");
    return run_program_any(ps, cols, ncols, nchunks, mask, nullptr, "columns of a batch differ in length");
}

rdf_status rdf_pipeline(const rdf_program* prog, const rdf_array* cols, int32_t ncols, int64_t nchunks, rdf_out* outs,
                        rdf_agg_result* aggs) {
    if (!prog || !prog->nodes || prog->nnodes <= 0) return fail(RDF_INVALID_ARGUMENT, "empty program");
    if (prog->nvalues < 1 || prog->nvalues > RDF_MAX_VALUES) return fail(RDF_INVALID_ARGUMENT, "nvalues out of range");
    ProgramSpec ps;
    memset(&ps, 0, sizeof ps);
    ps.nodes = prog->nodes; ps.nnodes = prog->nnodes; ps.filter_root = prog->filter_root; ps.nvalues = prog->nvalues; ps.sink = prog->sink;
    for (int v = 0; v < prog->nvalues; ++v) ps.value_roots[v] = prog->value_roots[v];
    if (ps.sink != RDF_SINK_STORE && ps.sink != RDF_SINK_AGG) return fail(RDF_INVALID_ARGUMENT, "bad sink");
    return run_program_any(ps, cols, ncols, nchunks, outs, aggs, "columns of a batch differ in length");
}

rdf_status rdf_filter_pipeline(const rdf_expr_node* nodes, int32_t nnodes, int32_t root, const rdf_array* cols, int32_t ncols, int64_t nchunks, rdf_out* outs) {
    if (!nodes || nnodes <= 0 || root < 0 || root >= nnodes) return fail(RDF_INVALID_ARGUMENT, "filter_pipeline: empty expression / root out of range");
    if (nchunks < 0 || ncols < 1 || ncols > kMaxFrameCols || (nchunks > 0 && (!cols || !outs))) return fail(RDF_INVALID_ARGUMENT, "filter_pipeline: 1..%d columns, chunk lists and outputs", kMaxFrameCols);
    if (nchunks == 0) return RDF_OK;
    for (int64_t i = 0; i < (int64_t)ncols * nchunks; ++i) {
        const rdf_array& a = cols[i];
        if (a.mem != RDF_MEM_HOST) return fail(RDF_INVALID_ARGUMENT, "filter_pipeline: host-resident batches (RDF_MEM_HOST); a device-resident frame is filtered with rdf_filter_frame");
        if (a.length < 0 || a.offset < 0 || (a.length > 0 && !a.values)) return fail(RDF_INVALID_ARGUMENT, "filter_pipeline: bad chunk %lld", (long long)i);
    }
    return filter_stream(nodes, nnodes, root, cols, ncols, nchunks, outs);
}

rdf_status rdf_stream_stats(int64_t* slabs, int64_t* bytes_staged, int64_t* bytes_direct) {
    if (slabs) *slabs = g_ctx.stream_slabs;
    if (bytes_staged) *bytes_staged = g_ctx.stream_bytes_staged;
    if (bytes_direct) *bytes_direct = g_ctx.stream_bytes_direct;
    return RDF_OK;
}

rdf_status rdf_frame_pin(const rdf_array* cols, int32_t ncols, int64_t nchunks, rdf_frame** out) {
    if (!out) return fail(RDF_INVALID_ARGUMENT, "frame_pin: null output pointer");
    *out = nullptr;
    if (!cols || ncols < 1 || ncols > kMaxFrameCols || nchunks < 1) return fail(RDF_INVALID_ARGUMENT, "frame_pin: 1..%d columns of at least one chunk", kMaxFrameCols);
    int32_t mem = -1;
    RDF_TRY(check_mem(cols, (int64_t)ncols * nchunks, &mem));
    if (mem != RDF_MEM_DEVICE) return fail(RDF_INVALID_ARGUMENT, "frame_pin: device-resident columns only (host buffers are staged per call)");
    RDF_TRY(ensure_ready());
    std::unique_ptr<rdf_frame> f(new rdf_frame());
    f->device = g_ctx.device;
    f->ncols = ncols;
    f->nchunks = nchunks;
    f->clen.assign((size_t)nchunks, 0);
    f->chunk_nullable.assign((size_t)nchunks, 0);
    f->dev.resize((size_t)ncols * (size_t)nchunks);
    for (int k = 0; k < ncols; ++k) {
        const int dt = cols[(int64_t)k * nchunks].dtype;
        if (!(is_numeric(dt) || dt == RDF_BOOL)) return fail(RDF_INVALID_ARGUMENT, "column %d: unsupported dtype %d", k, dt);
        f->col_dtype[k] = dt;
        f->col_nullable[k] = false;
        f->col_aligned16[k] = dt != RDF_BOOL;
        const size_t es = dt == RDF_BOOL ? 1 : (size_t)dtype_size(dt);
        for (int64_t c = 0; c < nchunks; ++c) {
            const rdf_array& a = cols[(int64_t)k * nchunks + c];
            if (a.dtype != dt) return fail(RDF_INVALID_ARGUMENT, "column %d: chunks differ in dtype", k);
            if (k == 0) { f->clen[(size_t)c] = a.length; f->total_rows += a.length; }
            else if (a.length != f->clen[(size_t)c]) return fail(RDF_COMPUTE_ERROR, "columns of a batch differ in length");
            if (a.validity) { f->chunk_nullable[(size_t)c] = 1; f->col_nullable[k] = true; }
            f->dev[(size_t)((int64_t)k * nchunks + c)] = DevChunkCol{a.values, a.validity, a.offset};
            if (a.length > 0 && (((uintptr_t)a.values + (uintptr_t)a.offset * es) & 15) != 0) f->col_aligned16[k] = false;
        }
    }
    for (int k = 0; k < ncols; ++k) {   // zero-copy slices of one buffer (DataFrame::from_csv's batches, a sliced Table): contiguous columns
        const size_t es = f->col_dtype[k] == RDF_BOOL ? 0 : (size_t)dtype_size(f->col_dtype[k]);
        const DevChunkCol& d0 = f->dev[(size_t)((int64_t)k * nchunks)];
        bool contig = es != 0;
        int64_t row = 0;
        for (int64_t c = 0; c < nchunks && contig; ++c) {
            const DevChunkCol& d = f->dev[(size_t)((int64_t)k * nchunks + c)];
            if (f->clen[(size_t)c] > 0) {
                if ((uintptr_t)d.values + (uintptr_t)d.offset * es != (uintptr_t)d0.values + (uintptr_t)(d0.offset + row) * es) contig = false;
                if ((d.validity == nullptr) != (d0.validity == nullptr)) contig = false;
                else if (d.validity && (uintptr_t)d.validity * 8 + (uintptr_t)d.offset != (uintptr_t)d0.validity * 8 + (uintptr_t)(d0.offset + row)) contig = false;
            }
            row += f->clen[(size_t)c];
        }
        f->col_contig[k] = contig;
    }
    f->uniform_len = nchunks > 1 ? f->clen[0] : 0;
    for (int64_t c = 0; c + 1 < nchunks && f->uniform_len > 0; ++c) if (f->clen[(size_t)c] != f->uniform_len) f->uniform_len = 0;
    if (f->uniform_len > 0 && f->clen[(size_t)nchunks - 1] > f->uniform_len) f->uniform_len = 0;
    void* p = nullptr;
    auto undo = [&]() { for (auto& pb : f->pooled) pool_release(pb.first, pb.second); f->pooled.clear(); };
    if (frame_table_alloc(*f, f->dev.size() * sizeof(DevChunkCol) + 64, &p) != RDF_OK) return RDF_MEMORY_ERROR;
    f->d_cols = (DevChunkCol*)p;
    if (frame_table_alloc(*f, f->clen.size() * 8 + 64, &p) != RDF_OK) { undo(); return RDF_MEMORY_ERROR; }
    f->d_clen = (int64_t*)p;
    if (frame_table_upload(f->d_cols, f->dev.data(), f->dev.size() * sizeof(DevChunkCol)) != RDF_OK ||
        frame_table_upload(f->d_clen, f->clen.data(), f->clen.size() * 8) != RDF_OK) { undo(); return RDF_DEVICE_ERROR; }
    *out = f.release();
    return RDF_OK;
}

rdf_status rdf_frame_release(rdf_frame* frame) {
    if (!frame) return RDF_OK;
    if (g_ctx.ready && g_ctx.stream) (void)hipStreamSynchronize(g_ctx.stream);   // kernels of this thread may still read the tables
    for (void* q : frame->allocs) (void)hipFree(q);
    for (auto& pb : frame->pooled) pool_release(pb.first, pb.second);
    for (auto& pr : frame->projections) {
        for (void* q : pr.second->allocs) (void)hipFree(q);
        for (auto& pb : pr.second->pooled) pool_release(pb.first, pb.second);
        delete pr.second;
    }
    delete frame;
    return RDF_OK;
}

rdf_status rdf_pipeline_frame(const rdf_program* prog, rdf_frame* frame, rdf_out* outs, rdf_agg_result* aggs) {
    if (!frame) return fail(RDF_INVALID_ARGUMENT, "pipeline_frame: null frame");
    if (!prog || !prog->nodes || prog->nnodes <= 0) return fail(RDF_INVALID_ARGUMENT, "empty program");
    if (prog->nvalues < 1 || prog->nvalues > RDF_MAX_VALUES) return fail(RDF_INVALID_ARGUMENT, "nvalues out of range");
    RDF_TRY(ensure_ready());
    if (frame->device != g_ctx.device) return fail(RDF_INVALID_ARGUMENT, "pipeline_frame: the frame was pinned on device %d", frame->device);
    ProgramSpec ps;
    memset(&ps, 0, sizeof ps);
    ps.nodes = prog->nodes; ps.nnodes = prog->nnodes; ps.filter_root = prog->filter_root; ps.nvalues = prog->nvalues; ps.sink = prog->sink;
    for (int v = 0; v < prog->nvalues; ++v) ps.value_roots[v] = prog->value_roots[v];
    if (ps.sink != RDF_SINK_STORE && ps.sink != RDF_SINK_AGG) return fail(RDF_INVALID_ARGUMENT, "bad sink");
    return run_program(ps, nullptr, frame->ncols, frame->nchunks, outs, aggs, "columns of a batch differ in length", frame);
}

rdf_status rdf_group_pipeline(const rdf_expr_node* nodes, int32_t nnodes, int32_t filter_root, int32_t group_root, int32_t ngroups,
                              const int32_t* value_roots, int32_t nvalues, const rdf_array* cols, int32_t ncols, int64_t nchunks,
                              rdf_group_result* out, int64_t* group_rows) {
    if (!nodes || nnodes <= 0) return fail(RDF_INVALID_ARGUMENT, "empty program");
    if (!value_roots || nvalues < 1 || nvalues > RDF_MAX_GROUP_VALUES) return fail(RDF_INVALID_ARGUMENT, "nvalues out of range");
    if (group_root < 0 || group_root >= nnodes) return fail(RDF_INVALID_ARGUMENT, "group_root out of range");
    ProgramSpec ps;
    memset(&ps, 0, sizeof ps);
    ps.nodes = nodes; ps.nnodes = nnodes; ps.filter_root = filter_root; ps.nvalues = nvalues; ps.sink = RDF_SINK_GROUP;
    for (int v = 0; v < nvalues; ++v) ps.value_roots[v] = value_roots[v];
    ps.group_root = group_root; ps.ngroups = ngroups; ps.gout = out; ps.grows = group_rows;
    return run_program_any(ps, cols, ncols, nchunks, nullptr, nullptr, "columns of a batch differ in length");
}

rdf_status rdf_group_pipeline_frame(const rdf_expr_node* nodes, int32_t nnodes, int32_t filter_root, int32_t group_root, int32_t ngroups,
                                    const int32_t* value_roots, int32_t nvalues, rdf_frame* frame, rdf_group_result* out,
                                    int64_t* group_rows) {
    if (!frame) return fail(RDF_INVALID_ARGUMENT, "group_pipeline_frame: null frame");
    if (!nodes || nnodes <= 0) return fail(RDF_INVALID_ARGUMENT, "empty program");
    if (!value_roots || nvalues < 1 || nvalues > RDF_MAX_GROUP_VALUES) return fail(RDF_INVALID_ARGUMENT, "nvalues out of range");
    if (group_root < 0 || group_root >= nnodes) return fail(RDF_INVALID_ARGUMENT, "group_root out of range");
    RDF_TRY(ensure_ready());
    if (frame->device != g_ctx.device) return fail(RDF_INVALID_ARGUMENT, "group_pipeline_frame: the frame was pinned on device %d", frame->device);
    ProgramSpec ps;
    memset(&ps, 0, sizeof ps);
    ps.nodes = nodes; ps.nnodes = nnodes; ps.filter_root = filter_root; ps.nvalues = nvalues; ps.sink = RDF_SINK_GROUP;
    for (int v = 0; v < nvalues; ++v) ps.value_roots[v] = value_roots[v];
    ps.group_root = group_root; ps.ngroups = ngroups; ps.gout = out; ps.grows = group_rows;
    return run_program(ps, nullptr, frame->ncols, frame->nchunks, nullptr, nullptr, "columns of a batch differ in length", frame);
}

rdf_status rdf_predicate_frame(const rdf_expr_node* nodes, int32_t nnodes, int32_t root, rdf_frame* frame, rdf_out* mask) {
    if (!frame) return fail(RDF_INVALID_ARGUMENT, "predicate_frame: null frame");
    if (!nodes || nnodes <= 0) return fail(RDF_INVALID_ARGUMENT, "empty expression");
    if (frame->nchunks > 0 && !mask) return fail(RDF_INVALID_ARGUMENT, "bad chunk lists");
    if (frame->nchunks == 0) return RDF_OK;
    RDF_TRY(ensure_ready());
    if (frame->device != g_ctx.device) return fail(RDF_INVALID_ARGUMENT, "predicate_frame: the frame was pinned on device %d", frame->device);
    ProgramSpec ps;
    memset(&ps, 0, sizeof ps);
    ps.nodes = nodes; ps.nnodes = nnodes; ps.filter_root = -1; ps.nvalues = 1; ps.value_roots[0] = root; ps.sink = RDF_SINK_STORE;
    for (int64_t c = 0; c < frame->nchunks; ++c)
        if (mask[c].dtype != RDF_BOOL) return fail(RDF_INVALID_ARGUMENT, "predicate root must be boolean");
    return run_program(ps, nullptr, frame->ncols, frame->nchunks, mask, nullptr, "columns of a batch differ in length", frame);
}

// ---------------------------------------------------------------- filter / take

namespace {

// tile states and ticket counters of one launch_bfilter (arena), the grid: two blocks of 8 waves per CU draw tiles by ticket
rdf_status bfilter_scratch(BFilterArgs& ba) {
    Ctx& ctx = g_ctx;
    const int64_t ntiles = ba.w.t.ntiles;
    const int64_t cap = (int64_t)(eval_grid_limit() / 8) * 2;
    ba.nworkers = (int32_t)std::max<int64_t>(1, std::min<int64_t>(cap, ntiles));
    ba.nclass = std::min(64, ba.nworkers);
    void* ps = nullptr;
    const size_t words = (size_t)ntiles + 8 + 64 * 16;
    RDF_TRY(arena_alloc(sizeof(unsigned long long) * words, &ps));
    HIP_TRY(hipMemsetAsync(ps, 0, sizeof(unsigned long long) * words, ctx.stream));
    ba.tile_state = (unsigned long long*)ps;
    ba.ticket = (unsigned int*)((unsigned long long*)ps + (size_t)ntiles + 8);
    ba.abort_flag = (unsigned int*)((unsigned long long*)ps + (size_t)ntiles + 4);      // (one of the 8 spare words between the two)
    ba.stall_test = ctx.opt_filter_block == 3 ? 1 : 0;
    return RDF_OK;
}
// Long batches, but many of them and none a large share of the frame (a 1e9-row frame in 65 536-row batches): a block draws whole
// batches and keeps the running offset itself — no scanner wave, no tile states (BFilterArgs::short_mode 2).
bool bfilter_owned_ok(int64_t nchunks, int64_t max_len, int64_t total_rows) {
    const int64_t workers = (int64_t)(eval_grid_limit() / 8) * 2;
    if (g_ctx.opt_filter_owned == 2) return g_ctx.opt_filter_block != 3;           // tests: every frame of long batches
    return g_ctx.opt_filter_owned && g_ctx.opt_filter_block != 3 && nchunks >= 8 * workers && max_len <= total_rows / (workers * 16);       // (>= 16 batches per block: the blocks finish within a few per cent of each other; measured level with the scanner form at 65 536-row batches of 1e9 rows, 6 % ahead at 16 384, 9 % ahead with the mask given, behind at 200 000: profiles/r06_filter_owned_batches_ab.jsonl)
}
// after the kernel (stream already synchronised by the caller's result copy): did a wait give up?
rdf_status bfilter_check(const BFilterArgs& ba) {
    Ctx& ctx = g_ctx;
    unsigned int flag = 0;
    HIP_TRY(hipMemcpyAsync(&flag, ba.abort_flag, 4, hipMemcpyDeviceToHost, ctx.stream));
    HIP_TRY(hipStreamSynchronize(ctx.stream));
    if (flag) return fail(RDF_DEVICE_ERROR, "filter: a tile waited %d s for its offset (the prefix of the block-tile kernel made no progress); rdf_set_option(\"filter_block\", 0) takes the wave-tile kernels", kBfWaitSeconds);
    return RDF_OK;
}

struct FilterPrep {
    InputStager in;         // mask chunks first, then the columns
    TableBuilder tb;
    MaskTables mt;
    int tile_rows = kFilterTile;   // kFilterTile or kFilterTileSmall, picked from the mean chunk length
    std::vector<int64_t> clen, tile_start;
    int64_t ntiles = 0;
    int64_t* d_counts = nullptr;  // [ntiles]
    int64_t* d_scan = nullptr;    // [ntiles + 1]
    bool wave = true;             // wave-granular tiles (rdf_filter.hip)
    bool dma_ok = false;          // every column 8 / 4 bytes wide, chunks of >= 1024 rows: the LDS-DMA kernel may take the frame
    FilterWArgs wa;
    size_t o_cols = 0, o_outs = 0;
    size_t pin_off = 0;
};

// Tiles of `tile_rows` rows (never spanning chunks): prefix table, per-tile keep counts, their scan; leaves the per-chunk
// keep totals in `totals`.  Callable again with another tile size (the staged inputs and descriptor tables stay).
rdf_status filter_tiles(FilterPrep& fp, int tile_rows, int64_t nchunks, std::vector<int64_t>& totals) {
    Ctx& ctx = g_ctx;
    fp.tile_rows = tile_rows;
    fp.tile_start.assign((size_t)nchunks + 1, 0);
    for (int64_t c = 0; c < nchunks; ++c) fp.tile_start[(size_t)c + 1] = fp.tile_start[(size_t)c] + (fp.clen[(size_t)c] + tile_rows - 1) / tile_rows;
    fp.ntiles = fp.tile_start[(size_t)nchunks];
    void* pts = nullptr;
    const size_t ts_bytes = sizeof(int64_t) * fp.tile_start.size();
    RDF_TRY(arena_alloc(ts_bytes, &pts));
    RDF_TRY(pinned_reserve(fp.pin_off + ts_bytes + 256));
    memcpy(ctx.pinned + fp.pin_off, fp.tile_start.data(), ts_bytes);
    HIP_TRY(hipMemcpyAsync(pts, ctx.pinned + fp.pin_off, ts_bytes, hipMemcpyHostToDevice, ctx.stream));
    fp.pin_off += (ts_bytes + 255) & ~(size_t)255;
    fp.mt.chunk_tile_start = (const int64_t*)pts;
    fp.mt.ntiles = fp.ntiles;

    void* p = nullptr;
    RDF_TRY(arena_alloc(sizeof(int64_t) * (2 * (size_t)fp.ntiles + 2 + (size_t)scan_scratch_words(fp.ntiles)), &p));
    fp.d_counts = (int64_t*)p;
    fp.d_scan = fp.d_counts + fp.ntiles;
    KernelTimer kt_count;        // (the count + scan pair: what rdf_filter_count's time is, and the first third of a three-pass filter's)
    if (fp.wave) {
        memset(&fp.wa, 0, sizeof fp.wa);
        fp.wa.t = fp.mt;
        fp.wa.prefetch = ctx.opt_filter_gen != 3;   // (3: A/B without the look-ahead)
        {   // chunk lengths that are not multiples of the DMA tile: the instantiation whose end-of-chunk tiles take the DMA path too
            int64_t rows = 0;
            for (int64_t c = 0; c < nchunks; ++c) rows += fp.clen[(size_t)c];
            fp.wa.ends = ctx.opt_filter_ends && fp.tile_rows == kWDmaTile && (double)rows < 0.99 * (double)fp.ntiles * (double)kWDmaTile ? 1 : 0;
        }
        if (nchunks == 1) { fp.wa.mask0 = fp.in.dev[0]; fp.wa.len0 = fp.clen[0]; }
        if (nchunks > 1 && fp.tile_start[(size_t)nchunks - 1] > 0 && nchunks - 1 < ((int64_t)1 << 31))
            fp.wa.tile_inv = (uint64_t)(((unsigned __int128)(uint64_t)(nchunks - 1) << 32) / (unsigned __int128)(uint64_t)fp.tile_start[(size_t)nchunks - 1]);
        HIP_TRY(launch_fcount(fp.wa, fp.tile_rows, fp.d_counts, ctx.stream));
    } else if (nchunks == 1 && fp.tile_rows == kFilterTile && ctx.opt_filter_one) HIP_TRY(launch_mask_count_one(fp.in.dev[0], fp.clen[0], fp.ntiles, fp.d_counts, ctx.stream));
    else HIP_TRY(launch_mask_count(fp.mt, fp.tile_rows, fp.d_counts, ctx.stream));
    HIP_TRY(launch_scan(fp.d_counts, fp.d_scan, fp.ntiles, fp.d_scan + fp.ntiles + 1, ctx.stream));
    kt_count.stop();

    // per-chunk totals = scan[tile_start[c+1]] - scan[tile_start[c]]: fetch the nchunks+1 boundary values
    totals.assign((size_t)nchunks, 0);
    if (nchunks <= 64) {
        RDF_TRY(pinned_reserve(fp.pin_off + 8 * ((size_t)nchunks + 1)));
        int64_t* pin = (int64_t*)(ctx.pinned + fp.pin_off);
        for (int64_t c = 0; c <= nchunks; ++c)
            HIP_TRY(hipMemcpyAsync(pin + c, fp.d_scan + fp.tile_start[(size_t)c], 8, hipMemcpyDeviceToHost, ctx.stream));
        HIP_TRY(hipStreamSynchronize(ctx.stream));
        for (int64_t c = 0; c < nchunks; ++c) totals[(size_t)c] = pin[c + 1] - pin[c];
    } else {
        RDF_TRY(pinned_reserve(fp.pin_off + 8 * ((size_t)fp.ntiles + 1)));
        int64_t* pin = (int64_t*)(ctx.pinned + fp.pin_off);
        HIP_TRY(hipMemcpyAsync(pin, fp.d_scan, 8 * ((size_t)fp.ntiles + 1), hipMemcpyDeviceToHost, ctx.stream));
        HIP_TRY(hipStreamSynchronize(ctx.stream));
        for (int64_t c = 0; c < nchunks; ++c) totals[(size_t)c] = pin[fp.tile_start[(size_t)c + 1]] - pin[fp.tile_start[(size_t)c]];
    }
    return RDF_OK;
}

// Stage mask (+ columns), build the descriptor tables, pick the compaction kernels and their tile size, count + scan.
rdf_status filter_prepare(FilterPrep& fp, const rdf_array* cols, int ncols, const rdf_array* mask, int64_t nchunks,
                          std::vector<int64_t>& totals, bool count = true) {
    Ctx& ctx = g_ctx;
    for (int64_t c = 0; c < nchunks; ++c) fp.in.add(&mask[c]);
    for (int64_t i = 0; i < (int64_t)ncols * nchunks; ++i) fp.in.add(&cols[i]);
    size_t used = 0;
    RDF_TRY(fp.in.finish(fp.pin_off, &used));
    fp.pin_off += (used + 255) & ~(size_t)255;

    fp.clen.resize((size_t)nchunks);
    int64_t rows_total = 0;
    for (int64_t c = 0; c < nchunks; ++c) { fp.clen[(size_t)c] = mask[c].length; rows_total += mask[c].length; }
    const int64_t mean_len = nchunks > 0 ? rows_total / nchunks : 0;
    // first generation (block tiles, one barrier per tile): 1024-row tiles for frames in the reader's batches, 4096 otherwise
    int tile_rows = nchunks > 0 && mean_len <= 2048 ? kFilterTileSmall : kFilterTile;
    if (ctx.opt_filter_tile == kFilterTileSmall || ctx.opt_filter_tile == kFilterTile) tile_rows = ctx.opt_filter_tile;
    // wave-granular kernels (rdf_filter.hip) — the default, with one exception: ONE column of ONE long chunk is still
    // faster on the first-generation block tiles when the LDS-DMA kernel cannot take it (measured per 1e9 rows: 2.34 ms
    // against 2.8 ms for the register-staged wave tiles)
    bool all_wide = ncols > 0;
    for (int k = 0; k < ncols; ++k) { const int es = dtype_size(cols[(int64_t)k * nchunks].dtype); all_wide &= es == 8 || es == 4; }
    fp.dma_ok = all_wide && rows_total >= nchunks * (int64_t)(kWDmaTile * 3 / 4) && ctx.opt_filter_gen == 2;   // most tiles full (the reader's last batch is short)
    fp.wave = ctx.opt_filter_gen >= 2 && !(ctx.opt_filter_gen == 2 && nchunks == 1 && ncols == 1 && !fp.dma_ok);
    if (fp.wave) {
        tile_rows = fp.dma_ok ? kWDmaTile : nchunks > 0 && mean_len <= 256 ? kWTileSmall : kWTile;
        if (!fp.dma_ok && (ctx.opt_filter_tile == kWTileSmall || ctx.opt_filter_tile == kWTile)) tile_rows = ctx.opt_filter_tile;
    }

    const size_t o_mask = fp.tb.reserve(sizeof(DevChunkCol) * (size_t)nchunks);
    const size_t o_len = fp.tb.reserve(sizeof(int64_t) * fp.clen.size());
    fp.o_cols = fp.tb.reserve(sizeof(DevChunkCol) * ((size_t)ncols * (size_t)nchunks + 1));
    fp.o_outs = fp.tb.reserve(sizeof(DevOutChunk) * ((size_t)ncols * (size_t)nchunks + 1));
    RDF_TRY(fp.tb.bind(fp.pin_off));
    memcpy(fp.tb.at<char>(o_mask), fp.in.dev.data(), sizeof(DevChunkCol) * (size_t)nchunks);
    memcpy(fp.tb.at<char>(o_len), fp.clen.data(), sizeof(int64_t) * fp.clen.size());
    if (ncols > 0) memcpy(fp.tb.at<char>(fp.o_cols), fp.in.dev.data() + nchunks, sizeof(DevChunkCol) * (size_t)ncols * (size_t)nchunks);
    RDF_TRY(fp.tb.alloc());
    // the outs table is filled in later (after the counts are known): upload the front part now
    RDF_TRY(fp.tb.upload(fp.pin_off));
    fp.pin_off += (fp.tb.size + 255) & ~(size_t)255;

    fp.mt.mask = fp.tb.dev_at<DevChunkCol>(o_mask);
    fp.mt.chunk_len = fp.tb.dev_at<int64_t>(o_len);
    fp.mt.nchunks = nchunks;
    if (!count) return RDF_OK;          // (the one-pass block kernel: staged inputs and descriptor tables only)
    RDF_TRY(filter_tiles(fp, tile_rows, nchunks, totals));
    if (fp.tile_rows == kWDmaTile) {
        // the LDS-DMA kernel fetches every sector of a tile: a selective filter (< 1/8 of the rows kept) goes to the
        // register-staged wave tiles, which only touch the sectors that hold a kept row
        int64_t kept = 0;
        for (int64_t c = 0; c < nchunks; ++c) kept += totals[(size_t)c];
        if (kept * 8 < rows_total) RDF_TRY(filter_tiles(fp, kWTile, nchunks, totals));
    }
    return RDF_OK;
}

rdf_status filter_validate(const rdf_array* cols, int ncols, const rdf_array* mask, int64_t nchunks, int32_t* mem) {
    if (nchunks < 0 || (nchunks > 0 && !mask)) return fail(RDF_INVALID_ARGUMENT, "bad chunk lists");
    RDF_TRY(check_mem(mask, nchunks, mem));
    RDF_TRY(check_mem(cols, (int64_t)ncols * nchunks, mem));
    for (int64_t c = 0; c < nchunks; ++c) {
        if (mask[c].dtype != RDF_BOOL) return fail(RDF_INVALID_ARGUMENT, "filter mask must be boolean");
        for (int k = 0; k < ncols; ++k) {
            const rdf_array& a = cols[(int64_t)k * nchunks + c];
            if (a.length != mask[c].length) return fail(RDF_COMPUTE_ERROR, "Filter array must have the same length as the data array");
            if (!is_numeric(a.dtype)) return fail(RDF_INVALID_ARGUMENT, "filter: unsupported dtype %d", a.dtype);
        }
    }
    return RDF_OK;
}


// Column::filter (src/table.rs:97-107,213-215) with the mask given, device-resident, in one pass on block tiles: groups of up to
// kMaxFilterCols columns per launch; lengths and null counts come back from the kernel.
rdf_status filter_columns_block(FilterPrep& fp, const rdf_array* cols, int ncols, const rdf_array* mask, int64_t nchunks, rdf_out* outs, int es0, int mode, int64_t max_len = 0) {
    // mode 0: long chunks (block tiles, scanner wave); 1: chunks of at most one wave tile on the wave-tile LDS-DMA kernel; 2: chunks no longer
    // than a block tile on the block kernel's short-batch mode (a chunk on 1 / 2 / 4 / 8 waves of a block)
    const bool short_chunks = mode == 1, block_short = mode == 2;
    Ctx& ctx = g_ctx;
    std::vector<int64_t> unused;
    RDF_TRY(filter_prepare(fp, cols, ncols, mask, nchunks, unused, false));
    const size_t nout = (size_t)ncols * (size_t)nchunks;
    std::vector<DevOutChunk> dev_outs(nout);
    bool nulls = false;
    for (int64_t c = 0; c < nchunks; ++c) nulls |= mask[c].validity != nullptr;
    for (size_t i = 0; i < nout; ++i) {
        dev_outs[i] = DevOutChunk{outs[i].values, cols[i].validity ? outs[i].validity : nullptr};
        nulls |= cols[i].validity != nullptr;
        const int64_t n = std::min<int64_t>(mask[i % (size_t)nchunks].length, outs[i].capacity);           // an upper bound of the rows written: the bitmap words they can touch
        if (dev_outs[i].validity && n > 0) HIP_TRY(hipMemsetAsync(dev_outs[i].validity, 0, (size_t)((n + 63) / 64 * 8), ctx.stream));
    }
    // rows every output of chunk c can take (outputs sized by an earlier rdf_filter_count hold fewer than the chunk has)
    std::vector<int64_t> cap((size_t)nchunks);
    bool tight = false;
    for (int64_t c = 0; c < nchunks; ++c) {
        int64_t m = INT64_MAX;
        for (int k = 0; k < ncols; ++k) m = std::min<int64_t>(m, outs[(int64_t)k * nchunks + c].capacity);
        cap[(size_t)c] = m;
        tight |= m < mask[c].length;
    }
    const int64_t* d_cap = nullptr;
    if (tight) {
        void* pc = nullptr;
        RDF_TRY(arena_alloc(8 * (size_t)nchunks + 64, &pc));
        RDF_TRY(pinned_reserve(fp.pin_off + 8 * (size_t)nchunks + 256));
        memcpy(ctx.pinned + fp.pin_off, cap.data(), 8 * (size_t)nchunks);
        HIP_TRY(hipMemcpyAsync(pc, ctx.pinned + fp.pin_off, 8 * (size_t)nchunks, hipMemcpyHostToDevice, ctx.stream));
        fp.pin_off += (8 * (size_t)nchunks + 255) & ~(size_t)255;
        d_cap = (const int64_t*)pc;
    }
    void* p = nullptr;
    RDF_TRY(arena_alloc(sizeof(int64_t) * (nout + (size_t)nchunks) + 64, &p));
    int64_t* d_nullc = (int64_t*)p;
    int64_t* d_len = d_nullc + nout;
    HIP_TRY(hipMemsetAsync(d_nullc, 0, sizeof(int64_t) * (nout + (size_t)nchunks), ctx.stream));
    RDF_TRY(pinned_reserve(fp.pin_off + sizeof(DevOutChunk) * nout + 256));
    memcpy(ctx.pinned + fp.pin_off, dev_outs.data(), sizeof(DevOutChunk) * nout);
    HIP_TRY(hipMemcpyAsync(fp.tb.dev + fp.o_outs, ctx.pinned + fp.pin_off, sizeof(DevOutChunk) * nout, hipMemcpyHostToDevice, ctx.stream));
    fp.pin_off += (sizeof(DevOutChunk) * nout + 255) & ~(size_t)255;
    std::vector<BFilterArgs> launched;
    {
        std::unique_ptr<KernelTimer> kt;      // (started at the first launch: the tile tables of a million-chunk frame are a millisecond of host work the device would sit out inside the timed region)
        int table_rows = 0;
        const int64_t* d_tile_start = nullptr;
        int64_t ntiles = 0;
        uint64_t tile_inv = 0;
        for (int g = 0; g < ncols; g += kMaxFilterCols) {
            const int nc = ncols - g < kMaxFilterCols ? ncols - g : kMaxFilterCols;
            const int tr = short_chunks ? kWDmaTile : bfilter_tile_rows(es0, nc);
            const int short_shift = block_short ? bfilter_short_shift(es0, nc, max_len) : -1;
            if (block_short) {
                if (short_shift < 0) return fail(RDF_DEVICE_ERROR, "filter: a chunk longer than a block tile on the short-batch path");
                ntiles = (nchunks + (8 >> short_shift) - 1) / (8 >> short_shift);
                d_tile_start = nullptr; tile_inv = 0; table_rows = 0;
            } else if (tr != table_rows) {           // (a last group of one column has longer tiles than the groups before it)
                std::vector<int64_t> ts((size_t)nchunks + 1, 0);
                for (int64_t c = 0; c < nchunks; ++c) ts[(size_t)c + 1] = ts[(size_t)c] + (fp.clen[(size_t)c] + tr - 1) / tr;
                ntiles = ts[(size_t)nchunks];
                tile_inv = 0;
                if (nchunks > 1 && ts[(size_t)nchunks - 1] > 0 && nchunks - 1 < ((int64_t)1 << 31))
                    tile_inv = (uint64_t)(((unsigned __int128)(uint64_t)(nchunks - 1) << 32) / (unsigned __int128)(uint64_t)ts[(size_t)nchunks - 1]);
                void* pts = nullptr;
                const size_t ts_bytes = sizeof(int64_t) * ts.size();
                RDF_TRY(arena_alloc(ts_bytes, &pts));
                RDF_TRY(pinned_reserve(fp.pin_off + ts_bytes + 256));
                memcpy(ctx.pinned + fp.pin_off, ts.data(), ts_bytes);
                HIP_TRY(hipMemcpyAsync(pts, ctx.pinned + fp.pin_off, ts_bytes, hipMemcpyHostToDevice, ctx.stream));
                fp.pin_off += (ts_bytes + 255) & ~(size_t)255;
                d_tile_start = (const int64_t*)pts;
                table_rows = tr;
            }
            BFilterArgs ba;
            memset(&ba, 0, sizeof ba);
            FilterWArgs& wa = ba.w;
            wa.t = fp.mt;
            wa.t.chunk_tile_start = d_tile_start;
            wa.t.ntiles = ntiles;
            wa.tile_inv = tile_inv;
            wa.cols = fp.tb.dev_at<DevChunkCol>(fp.o_cols) + (size_t)g * (size_t)nchunks;
            wa.outs = fp.tb.dev_at<DevOutChunk>(fp.o_outs) + (size_t)g * (size_t)nchunks;
            wa.out_null_counts = d_nullc + (size_t)g * (size_t)nchunks;
            wa.out_cap = d_cap;
            wa.ncols = nc;
            if (nchunks == 1) { wa.mask0 = fp.in.dev[0]; wa.len0 = fp.clen[0]; }
            for (int k = 0; k < nc; ++k) {
                wa.esize[k] = es0;
                if (nchunks == 1) { wa.cols0[k] = fp.in.dev[(size_t)(1 + g + k)]; wa.outs0[k] = dev_outs[(size_t)(g + k)]; }
            }
            if (short_chunks) {
                // the readers' batches: a chunk is ONE wave tile, its kept rows start its output — the wave-tile LDS-DMA kernel without
                // the count and scan passes in front of it (it writes the lengths itself)
                wa.out_len = d_len;
                wa.prefetch = 1;
                if (!kt) kt.reset(new KernelTimer());
                HIP_TRY(launch_fcompact(wa, kWDmaTile, ctx.stream));
                continue;
            }
            ba.nterms = 0;
            ba.out_len = d_len;
            if (block_short) { ba.short_mode = 1; ba.short_shift = short_shift; }
            else {
                int64_t longest = 0, rows = 0;
                for (int64_t c = 0; c < nchunks; ++c) { longest = std::max(longest, fp.clen[(size_t)c]); rows += fp.clen[(size_t)c]; }
                if (bfilter_owned_ok(nchunks, longest, rows)) ba.short_mode = 2;
            }
            RDF_TRY(bfilter_scratch(ba));
            if (!kt) kt.reset(new KernelTimer());
            HIP_TRY(launch_bfilter(ba, es0, nulls, ctx.stream));
            launched.push_back(ba);
        }
        if (kt) kt->stop();
    }
    ctx.last_kernel = short_chunks ? "fcompact_dma_kernel (one pass)" : block_short ? "bfilter_kernel (short batches)" : !launched.empty() && launched[0].short_mode == 2 ? "bfilter_kernel (a block per batch)" : "bfilter_kernel";
    RDF_TRY(pinned_reserve(fp.pin_off + 8 * (nout + (size_t)nchunks) + 256));
    int64_t* pin = (int64_t*)(ctx.pinned + fp.pin_off);
    HIP_TRY(hipMemcpyAsync(pin, d_nullc, 8 * (nout + (size_t)nchunks), hipMemcpyDeviceToHost, ctx.stream));
    HIP_TRY(hipStreamSynchronize(ctx.stream));
    for (const BFilterArgs& b : launched) RDF_TRY(bfilter_check(b));
    const int64_t* len = pin + nout;
    for (int64_t c = 0; c < nchunks; ++c)
        if (len[c] > cap[(size_t)c]) return fail(RDF_MEMORY_ERROR, "output capacity too small");
    for (size_t i = 0; i < nout; ++i) {
        const int64_t n = len[i % (size_t)nchunks];
        outs[i].length = n;
        outs[i].null_count = cols[i].validity ? pin[i] : 0;
        if (!cols[i].validity && outs[i].validity && n > 0) HIP_TRY(hipMemsetAsync(outs[i].validity, 0xFF, (size_t)((n + 7) / 8), ctx.stream));   // no input bitmap: everything kept is valid
    }
    HIP_TRY(hipStreamSynchronize(ctx.stream));
    return RDF_OK;
}
}  // namespace

rdf_status rdf_filter_count(const rdf_array* mask, int64_t nchunks, int64_t* counts) {
    int32_t mem = -1;
    RDF_TRY(filter_validate(nullptr, 0, mask, nchunks, &mem));
    if (nchunks == 0) return RDF_OK;
    if (!counts) return fail(RDF_INVALID_ARGUMENT, "counts is null");
    RDF_TRY(ensure_ready());
    arena_begin();
    FilterPrep fp;
    std::vector<int64_t> totals;
    RDF_TRY(filter_prepare(fp, nullptr, 0, mask, nchunks, totals));
    for (int64_t c = 0; c < nchunks; ++c) counts[c] = totals[(size_t)c];
    return RDF_OK;
}

rdf_status rdf_filter_columns(const rdf_array* cols, int32_t ncols, const rdf_array* mask, int64_t nchunks, rdf_out* outs) {
    if (ncols < 1 || ncols > 256) return fail(RDF_INVALID_ARGUMENT, "filter: 1..256 columns per call");
    int32_t mem = -1;
    RDF_TRY(filter_validate(cols, ncols, mask, nchunks, &mem));
    if (nchunks == 0) return RDF_OK;
    if (!outs) return fail(RDF_INVALID_ARGUMENT, "outs is null");
    RDF_TRY(check_out_mem(outs, (int64_t)ncols * nchunks, mem));
    for (int k = 0; k < ncols; ++k)
        for (int64_t c = 0; c < nchunks; ++c) {
            const rdf_array& a = cols[(int64_t)k * nchunks + c];
            const rdf_out& o = outs[(int64_t)k * nchunks + c];
            if (o.dtype != a.dtype) return fail(RDF_INVALID_ARGUMENT, "filter: output dtype mismatch");
            if (a.dtype != cols[(int64_t)k * nchunks].dtype) return fail(RDF_INVALID_ARGUMENT, "filter: chunks differ in dtype");
            if (a.validity && !o.validity) return fail(RDF_INVALID_ARGUMENT, "output validity buffer required");
        }
    RDF_TRY(ensure_ready());
    Ctx& ctx = g_ctx;
    arena_begin();
    FilterPrep fp;
    std::vector<int64_t> totals;
    // Long device-resident chunks of equally wide columns: ONE pass on block tiles (rdf_bfilter.hip) — no count pass, no scan: a
    // tile's offset comes from the scanner wave.
    {
        int es0 = dtype_size(cols[0].dtype);
        bool roomy = mem == RDF_MEM_DEVICE && (es0 == 8 || es0 == 4) && ctx.opt_filter_block != 0;
        int64_t rows_total = 0;
        for (int64_t c = 0; c < nchunks; ++c) rows_total += mask[c].length;
        for (int k = 0; k < ncols && roomy; ++k) roomy = dtype_size(cols[(int64_t)k * nchunks].dtype) == es0;
        // (outputs smaller than their chunk — a caller that sized them with rdf_filter_count — take the one pass as well: the kernel
        // knows every output's capacity, a chunk that keeps more than it holds is not written and the count it reports fails the call)
        if (roomy && (rows_total + nchunks - 1) / nchunks >= (int64_t)ctx.opt_filter_block_rows) return filter_columns_block(fp, cols, ncols, mask, nchunks, outs, es0, 0);
        // ... chunks no longer than a block tile (the readers' 1024-row batches, chunks of a few thousand rows): the block kernel with a
        // chunk on 1 / 2 / 4 / 8 waves of a block, no prefix between blocks (round 6) — unless most slots would sit half empty
        int64_t max_len = 0;
        for (int64_t c = 0; c < nchunks; ++c) max_len = std::max<int64_t>(max_len, mask[c].length);
        if (roomy && ctx.opt_filter_gen == 2) {
            // (every column group of a call must fit: the groups of eight columns have the shortest tiles)
            // The same choice as rdf_filter_frame's (filter_frame_fused): slots filled to 0.9 -> the short form; chunks that average
            // half a tile and fill their tiles to 0.5 (one column) / 0.65 -> the long forms; slots half full -> the short form.
            const int nc_min = ncols >= 2 ? 2 : 1;
            const int64_t tr = bfilter_tile_rows(es0, nc_min);
            const int sh = ctx.opt_filter_short ? bfilter_short_shift(es0, nc_min, max_len) : -1;
            const double short_fill = sh >= 0 ? (double)rows_total / ((double)nchunks * (double)((tr / 8) << sh)) : 0.0;
            if (sh >= 0 && short_fill >= 0.9) return filter_columns_block(fp, cols, ncols, mask, nchunks, outs, es0, 2, max_len);
            if ((rows_total + nchunks - 1) / nchunks * 2 >= tr) {
                int64_t ntl = 0;
                for (int64_t c = 0; c < nchunks; ++c) ntl += (mask[c].length + tr - 1) / tr;
                if (ntl > 0 && (double)rows_total / ((double)ntl * (double)tr) >= (ncols == 1 ? 0.5 : 0.65)) return filter_columns_block(fp, cols, ncols, mask, nchunks, outs, es0, 0);
            }
            if (sh >= 0 && short_fill >= 0.5) return filter_columns_block(fp, cols, ncols, mask, nchunks, outs, es0, 2, max_len);
        }
        // ... and before that kernel had its short-batch mode: no chunk longer than one wave tile of 1024 rows, most of them full
        if (roomy && max_len <= kWDmaTile && rows_total >= nchunks * (int64_t)(kWDmaTile * 3 / 4) && ctx.opt_filter_gen == 2)
            return filter_columns_block(fp, cols, ncols, mask, nchunks, outs, es0, 1);
    }
    RDF_TRY(filter_prepare(fp, cols, ncols, mask, nchunks, totals));
    for (int k = 0; k < ncols; ++k)
        for (int64_t c = 0; c < nchunks; ++c)
            if (outs[(int64_t)k * nchunks + c].capacity < totals[(size_t)c]) return fail(RDF_MEMORY_ERROR, "output capacity too small");

    // outputs
    const size_t nout = (size_t)ncols * (size_t)nchunks;
    std::vector<DevOutChunk> dev_outs(nout);
    Region outr;
    std::vector<int> vi(nout, -1), bi(nout, -1);
    if (mem == RDF_MEM_HOST) {
        for (int k = 0; k < ncols; ++k)
            for (int64_t c = 0; c < nchunks; ++c) {
                const size_t i = (size_t)((int64_t)k * nchunks + c);
                const int64_t n = totals[(size_t)c];
                if (n == 0) continue;
                vi[i] = outr.add(outs[i].values, (size_t)n * (size_t)dtype_size(outs[i].dtype));
                if (cols[i].validity) bi[i] = outr.add(outs[i].validity, (size_t)((n + 7) / 8));
            }
        RDF_TRY(outr.layout());
        // validity bitmaps are OR-accumulated: zero the whole region once
        HIP_TRY(hipMemsetAsync(outr.dev, 0, outr.total ? outr.total : 256, ctx.stream));
        for (size_t i = 0; i < nout; ++i) {
            dev_outs[i].values = vi[i] >= 0 ? outr.ptr(vi[i]) : nullptr;
            dev_outs[i].validity = bi[i] >= 0 ? (uint8_t*)outr.ptr(bi[i]) : nullptr;
        }
    } else {
        for (size_t i = 0; i < nout; ++i) {
            dev_outs[i] = DevOutChunk{outs[i].values, cols[i].validity ? outs[i].validity : nullptr};
            const int64_t n = totals[(size_t)(i % (size_t)nchunks)];
            if (dev_outs[i].validity && n > 0) HIP_TRY(hipMemsetAsync(dev_outs[i].validity, 0, (size_t)((n + 63) / 64 * 8), ctx.stream));
        }
    }
    // upload the outs table + null counters
    void* p = nullptr;
    RDF_TRY(arena_alloc(sizeof(int64_t) * nout, &p));
    int64_t* d_nullc = (int64_t*)p;
    HIP_TRY(hipMemsetAsync(d_nullc, 0, sizeof(int64_t) * nout, ctx.stream));
    RDF_TRY(pinned_reserve(fp.pin_off + sizeof(DevOutChunk) * nout + 256));
    memcpy(ctx.pinned + fp.pin_off, dev_outs.data(), sizeof(DevOutChunk) * nout);
    HIP_TRY(hipMemcpyAsync(fp.tb.dev + fp.o_outs, ctx.pinned + fp.pin_off, sizeof(DevOutChunk) * nout, hipMemcpyHostToDevice, ctx.stream));
    fp.pin_off += (sizeof(DevOutChunk) * nout + 255) & ~(size_t)255;

    {
        KernelTimer kt;
        for (int g = 0; g < ncols; g += kMaxFilterCols) {  // ranks are recomputed per group of columns (1 bit/row)
            if (fp.wave) {
                FilterWArgs& wa = fp.wa;
                wa.cols = fp.tb.dev_at<DevChunkCol>(fp.o_cols) + (size_t)g * (size_t)nchunks;
                wa.outs = fp.tb.dev_at<DevOutChunk>(fp.o_outs) + (size_t)g * (size_t)nchunks;
                wa.out_null_counts = d_nullc + (size_t)g * (size_t)nchunks;
                wa.tile_scan = fp.d_scan;
                wa.ncols = ncols - g < kMaxFilterCols ? ncols - g : kMaxFilterCols;
                for (int k = 0; k < wa.ncols; ++k) {
                    wa.esize[k] = dtype_size(cols[(int64_t)(g + k) * nchunks].dtype);
                    if (nchunks == 1) { wa.cols0[k] = fp.in.dev[(size_t)(1 + g + k)]; wa.outs0[k] = dev_outs[(size_t)(g + k)]; }
                }
                HIP_TRY(launch_fcompact(wa, fp.tile_rows, ctx.stream));
                continue;
            }
            if (nchunks == 1 && fp.tile_rows == kFilterTile && ctx.opt_filter_one) {   // one long chunk: descriptors in the kernel arguments
                FilterOneArgs oa;
                memset(&oa, 0, sizeof oa);
                oa.mask = fp.in.dev[0];
                oa.clen = fp.clen[0];
                oa.ntiles = fp.ntiles;
                oa.tile_scan = fp.d_scan;
                oa.out_null_counts = d_nullc + (size_t)g;
                oa.ncols = ncols - g < kMaxFilterCols ? ncols - g : kMaxFilterCols;
                for (int k = 0; k < oa.ncols; ++k) {
                    oa.esize[k] = dtype_size(cols[g + k].dtype);
                    oa.cols[k] = fp.in.dev[(size_t)(1 + g + k)];
                    oa.outs[k] = dev_outs[(size_t)(g + k)];
                }
                HIP_TRY(launch_compact_one(oa, ctx.stream));
                continue;
            }
            FilterArgs fa;
            memset(&fa, 0, sizeof fa);
            fa.t = fp.mt;
            fa.cols = fp.tb.dev_at<DevChunkCol>(fp.o_cols) + (size_t)g * (size_t)nchunks;
            fa.outs = fp.tb.dev_at<DevOutChunk>(fp.o_outs) + (size_t)g * (size_t)nchunks;
            fa.out_null_counts = d_nullc + (size_t)g * (size_t)nchunks;
            fa.tile_scan = fp.d_scan;
            fa.ncols = ncols - g < kMaxFilterCols ? ncols - g : kMaxFilterCols;
            for (int k = 0; k < fa.ncols; ++k) fa.esize[k] = dtype_size(cols[(int64_t)(g + k) * nchunks].dtype);
            HIP_TRY(launch_compact(fa, fp.tile_rows, ctx.stream));
        }
        kt.stop();
    }
    ctx.last_kernel = fp.wave ? (fp.tile_rows == kWDmaTile ? "fcompact_dma_kernel" : "fcompact_kernel") : "compact_kernel";
    RDF_TRY(pinned_reserve(fp.pin_off + 8 * nout + 256 + outr.small_bytes + 256));
    int64_t* pin_nc = (int64_t*)(ctx.pinned + fp.pin_off);
    HIP_TRY(hipMemcpyAsync(pin_nc, d_nullc, 8 * nout, hipMemcpyDeviceToHost, ctx.stream));
    if (mem == RDF_MEM_HOST) RDF_TRY(outr.download(fp.pin_off + ((8 * nout + 255) & ~(size_t)255)));
    else HIP_TRY(hipStreamSynchronize(ctx.stream));
    for (size_t i = 0; i < nout; ++i) {
        const int64_t n = totals[i % (size_t)nchunks];
        outs[i].length = n;
        outs[i].null_count = cols[i].validity ? pin_nc[i] : 0;
        if (!cols[i].validity && outs[i].validity && n > 0) {  // no input bitmap: everything kept is valid
            if (mem == RDF_MEM_HOST) memset(outs[i].validity, 0xFF, (size_t)((n + 7) / 8));
            else HIP_TRY(hipMemsetAsync(outs[i].validity, 0xFF, (size_t)((n + 7) / 8), ctx.stream));
        }
    }
    if (mem == RDF_MEM_DEVICE) HIP_TRY(hipStreamSynchronize(ctx.stream));
    return RDF_OK;
}

rdf_status rdf_filter(const rdf_array* col, const rdf_array* mask, int64_t nchunks, rdf_out* out) {
    return rdf_filter_columns(col, 1, mask, nchunks, out);
}

rdf_status rdf_take(const rdf_array* chunks, int64_t nchunks, const rdf_array* indices, rdf_out* out) {
    if (nchunks < 1 || !chunks) return fail(RDF_INVALID_ARGUMENT, "take: a column has at least one chunk");
    if (!indices || !out) return fail(RDF_INVALID_ARGUMENT, "take: null argument");
    if (indices->dtype != RDF_U32 && indices->dtype != RDF_U64) return fail(RDF_INVALID_ARGUMENT, "take: indices must be UInt32/UInt64");
    int32_t mem = -1;
    RDF_TRY(check_mem(chunks, nchunks, &mem));
    RDF_TRY(check_mem(indices, 1, &mem));
    RDF_TRY(check_out_mem(out, 1, mem));
    const int dt = chunks[0].dtype;
    if (!is_numeric(dt) || out->dtype != dt) return fail(RDF_INVALID_ARGUMENT, "take: unsupported or mismatched dtype");
    bool any_validity = indices->validity != nullptr;
    std::vector<int64_t> row_start((size_t)nchunks + 1, 0);
    for (int64_t c = 0; c < nchunks; ++c) {
        if (chunks[c].dtype != dt) return fail(RDF_INVALID_ARGUMENT, "take: chunks differ in dtype");
        row_start[(size_t)c + 1] = row_start[(size_t)c] + chunks[c].length;
        any_validity |= chunks[c].validity != nullptr;
    }
    const int64_t n = indices->length;
    if (out->capacity < n) return fail(RDF_MEMORY_ERROR, "output capacity too small");
    if (any_validity && !out->validity) return fail(RDF_INVALID_ARGUMENT, "output validity buffer required");
    if (n == 0) { out->length = 0; out->null_count = 0; return RDF_OK; }
    RDF_TRY(ensure_ready());
    Ctx& ctx = g_ctx;
    arena_begin();
    size_t pin_off = 0, used = 0;
    InputStager in;
    in.add(indices);
    for (int64_t c = 0; c < nchunks; ++c) in.add(&chunks[c]);
    RDF_TRY(in.finish(pin_off, &used));
    pin_off += (used + 255) & ~(size_t)255;

    TableBuilder tb;
    const size_t o_ch = tb.reserve(sizeof(DevChunkCol) * (size_t)nchunks);
    const size_t o_rs = tb.reserve(sizeof(int64_t) * row_start.size());
    RDF_TRY(tb.bind(pin_off));
    memcpy(tb.at<char>(o_ch), in.dev.data() + 1, sizeof(DevChunkCol) * (size_t)nchunks);
    memcpy(tb.at<char>(o_rs), row_start.data(), sizeof(int64_t) * row_start.size());
    RDF_TRY(tb.alloc());
    RDF_TRY(tb.upload(pin_off));
    pin_off += (tb.size + 255) & ~(size_t)255;

    const size_t es = (size_t)dtype_size(dt);
    Region outr;
    int vi = -1, bi = -1;
    DevOutChunk doc;
    if (mem == RDF_MEM_HOST) {
        vi = outr.add(out->values, (size_t)n * es);
        if (out->validity) bi = outr.add(out->validity, (size_t)((n + 7) / 8));
        RDF_TRY(outr.layout());
        doc.values = outr.ptr(vi);
        doc.validity = bi >= 0 ? (uint8_t*)outr.ptr(bi) : nullptr;
    } else doc = DevOutChunk{out->values, out->validity};

    void* p = nullptr;
    RDF_TRY(arena_alloc(32, &p));
    HIP_TRY(hipMemsetAsync(p, 0, 32, ctx.stream));
    TakeArgs ta;
    memset(&ta, 0, sizeof ta);
    ta.chunks = tb.dev_at<DevChunkCol>(o_ch);
    ta.chunk_row_start = tb.dev_at<int64_t>(o_rs);
    ta.nchunks = nchunks;
    ta.total_rows = row_start[(size_t)nchunks];
    ta.indices = in.dev[0];
    ta.n = n;
    ta.out = doc;
    ta.flags = (uint32_t*)p;
    ta.out_null_count = (int64_t*)((char*)p + 16);
    ta.esize = (int)es;
    ta.idx64 = indices->dtype == RDF_U64;
    {
        KernelTimer kt;
        HIP_TRY(launch_take(ta, ctx.stream));
        kt.stop();
    }
    RDF_TRY(pinned_reserve(pin_off + 256 + outr.small_bytes + 256));
    char* pin = ctx.pinned + pin_off;
    HIP_TRY(hipMemcpyAsync(pin, p, 32, hipMemcpyDeviceToHost, ctx.stream));
    HIP_TRY(hipStreamSynchronize(ctx.stream));
    uint32_t flags;
    memcpy(&flags, pin, 4);
    if (flags & 2u) return fail(RDF_COMPUTE_ERROR, "take index out of bounds (len %lld)", (long long)ta.total_rows);
    if (mem == RDF_MEM_HOST) RDF_TRY(outr.download(pin_off + 256));
    out->length = n;
    memcpy(&out->null_count, pin + 16, 8);
    return RDF_OK;
}

// ---------------------------------------------------------------- ArrayFunctions over List<primitive>

namespace {
rdf_status list_check(const rdf_list_array* l, const char* fn, int64_t* rows, int32_t* mem) {
    if (!l) return fail(RDF_INVALID_ARGUMENT, "%s: null list", fn);
    if (l->offsets.dtype != RDF_I32) return fail(RDF_INVALID_ARGUMENT, "%s: value_offsets must be Int32", fn);
    if (l->offsets.length < 1) return fail(RDF_INVALID_ARGUMENT, "%s: value_offsets hold rows + 1 entries", fn);
    if (!is_numeric(l->values.dtype)) return fail(RDF_INVALID_ARGUMENT, "%s: primitive numeric child values only", fn);
    *rows = l->offsets.length - 1;
    *mem = -1;
    RDF_TRY(check_mem(&l->offsets, 1, mem));
    RDF_TRY(check_mem(&l->values, 1, mem));
    return RDF_OK;
}
uint64_t scalar_bits(const void* value, int dt) {
    uint64_t b = 0;
    memcpy(&b, value, (size_t)dtype_size(dt));
    return h_normalize_int(dt == RDF_F32 || dt == RDF_F64 ? RDF_U64 : dt, b);
}
// the four per-row reductions: contains / position / max / min
rdf_status list_reduce(const rdf_list_array* l, int op, const void* value, rdf_out* out, const char* fn) {
    int64_t n = 0;
    int32_t mem = -1;
    RDF_TRY(list_check(l, fn, &n, &mem));
    if (!out) return fail(RDF_INVALID_ARGUMENT, "%s: null output", fn);
    if ((op == LIST_CONTAINS || op == LIST_POSITION) && !value) return fail(RDF_INVALID_ARGUMENT, "%s: null value pointer", fn);
    const int cdt = l->values.dtype;
    const int odt = op == LIST_CONTAINS ? RDF_BOOL : op == LIST_POSITION ? RDF_I32 : cdt;
    if (out->dtype != odt) return fail(RDF_INVALID_ARGUMENT, "%s: output dtype %d expected", fn, odt);
    RDF_TRY(check_out_mem(out, 1, mem));
    if (out->capacity < n) return fail(RDF_MEMORY_ERROR, "output capacity too small");
    const bool nullable = op != LIST_POSITION && (l->offsets.validity != nullptr || op == LIST_MAX || op == LIST_MIN);
    if (nullable && !out->validity) return fail(RDF_INVALID_ARGUMENT, "output validity buffer required");
    if (n == 0) { out->length = 0; out->null_count = 0; return RDF_OK; }
    RDF_TRY(ensure_ready());
    Ctx& ctx = g_ctx;
    arena_begin();
    size_t pin_off = 0, used = 0;
    // value_offsets (n + 1 entries) and the list validity (n bits) are staged as two arrays: the bitmap is not read past bit n
    rdf_array offs = l->offsets, lv = l->offsets;
    offs.length = n + 1; offs.validity = nullptr; offs.null_count = 0;
    lv.values = l->offsets.validity; lv.validity = nullptr; lv.dtype = RDF_BOOL; lv.length = n; lv.null_count = 0;
    InputStager in;
    in.add(&offs);
    in.add(&l->values);
    if (l->offsets.validity) in.add(&lv);
    RDF_TRY(in.finish(pin_off, &used));
    pin_off += (used + 255) & ~(size_t)255;
    const size_t es = odt == RDF_BOOL ? 0 : (size_t)dtype_size(odt);
    const size_t vbytes = odt == RDF_BOOL ? (size_t)((n + 63) / 64 * 8) : (size_t)n * es, bbytes = (size_t)((n + 63) / 64 * 8);
    Region outr;
    int vi = -1, bi = -1;
    DevOutChunk doc;
    if (mem == RDF_MEM_HOST) {
        vi = outr.add(out->values, odt == RDF_BOOL ? (size_t)((n + 7) / 8) : vbytes);
        if (out->validity) bi = outr.add(out->validity, (size_t)((n + 7) / 8));
        RDF_TRY(outr.layout());
        doc.values = outr.ptr(vi);
        doc.validity = bi >= 0 ? (uint8_t*)outr.ptr(bi) : nullptr;
    } else doc = DevOutChunk{out->values, out->validity};
    void* p = nullptr;
    RDF_TRY(arena_alloc(32, &p));
    HIP_TRY(hipMemsetAsync(p, 0, 32, ctx.stream));
    const int64_t nvals = l->values.length;
    const bool wave_per_row = nvals / n >= 48;   // long lists: a wave per row; short lists: a row per lane
    if (wave_per_row) {   // bitmap outputs are OR-ed in bit by bit
        if (odt == RDF_BOOL) HIP_TRY(hipMemsetAsync(doc.values, 0, mem == RDF_MEM_HOST ? (size_t)((n + 7) / 8) : vbytes, ctx.stream));
        if (doc.validity) HIP_TRY(hipMemsetAsync(doc.validity, 0, mem == RDF_MEM_HOST ? (size_t)((n + 7) / 8) : bbytes, ctx.stream));
    }
    ListArgs la;
    memset(&la, 0, sizeof la);
    la.offsets = in.dev[0];
    if (l->offsets.validity) la.offsets.validity = (const uint8_t*)in.dev[2].values;   // same bit offset as the value_offsets by construction
    la.values = in.dev[1];
    la.n = n;
    la.dtype = cdt;
    la.op = op;
    la.needle = value ? scalar_bits(value, cdt) : 0;
    la.out = doc;
    la.out_null_count = (int64_t*)((char*)p + 16);
    {
        KernelTimer kt;
        ctx.last_kernel = wave_per_row ? "list_wave_kernel" : (op == LIST_CONTAINS || op == LIST_POSITION) ? "list_find_kernel" : "list_extreme_kernel";
        HIP_TRY(launch_list_op(la, wave_per_row, ctx.stream));
        kt.stop();
    }
    RDF_TRY(pinned_reserve(pin_off + 256 + outr.small_bytes + 256));
    char* pin = ctx.pinned + pin_off;
    HIP_TRY(hipMemcpyAsync(pin, p, 32, hipMemcpyDeviceToHost, ctx.stream));
    HIP_TRY(hipStreamSynchronize(ctx.stream));
    if (mem == RDF_MEM_HOST) RDF_TRY(outr.download(pin_off + 256));
    out->length = n;
    memcpy(&out->null_count, pin + 16, 8);
    return RDF_OK;
}
}  // namespace

rdf_status rdf_list_contains(const rdf_list_array* list, const void* value, rdf_out* out) { return list_reduce(list, LIST_CONTAINS, value, out, "array_contains"); }
rdf_status rdf_list_position(const rdf_list_array* list, const void* value, rdf_out* out) { return list_reduce(list, LIST_POSITION, value, out, "array_position"); }
rdf_status rdf_list_max(const rdf_list_array* list, rdf_out* out) { return list_reduce(list, LIST_MAX, nullptr, out, "array_max"); }
rdf_status rdf_list_min(const rdf_list_array* list, rdf_out* out) { return list_reduce(list, LIST_MIN, nullptr, out, "array_min"); }

namespace {
// The list-valued ArrayFunctions that rebuild every row from its own elements (array_remove, array_distinct,
// array_except / intersect / union, array_repeat): count per row -> scan -> write at the scanned offsets.
rdf_status list_rebuild(const rdf_list_array* list, const rdf_list_array* other, int op, uint64_t needle, int32_t count,
                        rdf_out* out_offsets, rdf_out* out_values, const char* fn) {
    int64_t n = 0, nb = 0;
    int32_t mem = -1;
    RDF_TRY(list_check(list, fn, &n, &mem));
    if (!out_offsets || !out_values) return fail(RDF_INVALID_ARGUMENT, "%s: null argument", fn);
    if (other) {
        RDF_TRY(list_check(other, fn, &nb, &mem));
        // array.rs:72-76,116-120,362-366
        if (nb != n) return fail(RDF_COMPUTE_ERROR, "Expected array a and b to have the same length");
        if (other->values.dtype != list->values.dtype) return fail(RDF_INVALID_ARGUMENT, "%s: both lists must have the same child dtype", fn);
    }
    const int cdt = list->values.dtype;
    if (out_offsets->dtype != RDF_I32 || out_values->dtype != cdt) return fail(RDF_INVALID_ARGUMENT, "%s: outputs are (Int32 offsets, child dtype values)", fn);
    RDF_TRY(check_out_mem(out_offsets, 1, mem));
    RDF_TRY(check_out_mem(out_values, 1, mem));
    if (out_offsets->capacity < n + 1) return fail(RDF_MEMORY_ERROR, "output capacity too small (offsets need rows + 1)");
    RDF_TRY(ensure_ready());
    Ctx& ctx = g_ctx;
    arena_begin();
    size_t pin_off = 0, used = 0;
    rdf_array offs = list->offsets, lv = list->offsets, offs_b;
    offs.length = n + 1; offs.validity = nullptr; offs.null_count = 0;
    lv.values = list->offsets.validity; lv.validity = nullptr; lv.dtype = RDF_BOOL; lv.length = n; lv.null_count = 0;
    InputStager in;
    in.add(&offs);
    in.add(&list->values);
    int ib = -1, iv = -1;
    if (other) {
        offs_b = other->offsets;
        offs_b.length = n + 1; offs_b.validity = nullptr; offs_b.null_count = 0;
        ib = 2;
        in.add(&offs_b);
        in.add(&other->values);
    }
    if (list->offsets.validity) { iv = other ? 4 : 2; in.add(&lv); }
    RDF_TRY(in.finish(pin_off, &used));
    pin_off += (used + 255) & ~(size_t)255;
    void *pkept, *pscan, *poff32;
    RDF_TRY(arena_alloc((size_t)(n + 1) * 8, &pkept));
    RDF_TRY(arena_alloc((size_t)(n + 2 + scan_scratch_words(n)) * 8, &pscan));
    RDF_TRY(arena_alloc((size_t)(n + 1) * 4 + 8, &poff32));
    ListArgs la;
    memset(&la, 0, sizeof la);
    la.offsets = in.dev[0];
    if (iv >= 0) la.offsets.validity = (const uint8_t*)in.dev[iv].values;
    la.values = in.dev[1];
    if (other) { la.offsets_b = in.dev[ib]; la.values_b = in.dev[ib + 1]; }
    la.n = n;
    la.dtype = cdt;
    la.op = op;
    la.needle = needle;
    la.count = count;
    la.kept = (int64_t*)pkept;
    const int64_t nelem = list->values.length + (other ? other->values.length : 0);
    const bool wave_per_row = n > 0 && nelem / n >= 48;
    const bool is_remove = op == LIST_REMOVE;
    if (!is_remove && op != LIST_REPEAT) {   // the row-per-wave kernel's tables (rdf_list.hip) and the worklist that feeds it
        void *pta, *ptb = nullptr, *pwork;
        RDF_TRY(arena_alloc((size_t)list->values.length * 8 + 64, &pta));
        if (other) RDF_TRY(arena_alloc((size_t)other->values.length * 8 + 64, &ptb));
        la.tab_a = (uint32_t*)pta;
        la.tab_b = (uint32_t*)ptb;
        if (!wave_per_row) {
            RDF_TRY(arena_alloc((size_t)n * 4 + 64, &pwork));
            la.work = (uint32_t*)pwork + 16;
            la.work_count = (uint32_t*)pwork;
            HIP_TRY(hipMemsetAsync(pwork, 0, 64, ctx.stream));
        }
    } else if (op == LIST_REPEAT && !wave_per_row) {
        void* pwork;
        RDF_TRY(arena_alloc((size_t)n * 4 + 64, &pwork));
        la.work = (uint32_t*)pwork + 16;
        la.work_count = (uint32_t*)pwork;
        HIP_TRY(hipMemsetAsync(pwork, 0, 64, ctx.stream));
    }
    KernelTimer kt;
    ctx.last_kernel = is_remove ? (wave_per_row ? "list_remove_wave_kernel" : "list_remove_kernel") : (wave_per_row ? "list_set_wave_kernel" : "list_set_kernel + list_set_wave_kernel");
    HIP_TRY(is_remove ? launch_list_remove(la, wave_per_row, ctx.stream) : launch_list_set(la, wave_per_row, ctx.stream));
    HIP_TRY(launch_scan((const int64_t*)pkept, (int64_t*)pscan, n, (int64_t*)pscan + n + 1, ctx.stream));
    HIP_TRY(launch_list_offsets((const int64_t*)pscan, n + 1, (int32_t*)poff32, ctx.stream));
    RDF_TRY(pinned_reserve(pin_off + 64));
    HIP_TRY(hipMemcpyAsync(ctx.pinned + pin_off, (int64_t*)pscan + n, 8, hipMemcpyDeviceToHost, ctx.stream));
    HIP_TRY(hipStreamSynchronize(ctx.stream));
    int64_t total = 0;
    memcpy(&total, ctx.pinned + pin_off, 8);
    if (total > INT32_MAX) return fail(RDF_COMPUTE_ERROR, "%s: %lld result elements overflow the Int32 value_offsets", fn, (long long)total);
    if (out_values->capacity < total) return fail(RDF_MEMORY_ERROR, "%s: values capacity too small (need %lld)", fn, (long long)total);
    const size_t es = (size_t)dtype_size(cdt);
    void* dvals = out_values->values;
    if (mem == RDF_MEM_HOST) RDF_TRY(arena_alloc((size_t)total * es + 64, &dvals));
    la.scan = (const int64_t*)pscan;
    la.out = DevOutChunk{dvals, nullptr};
    HIP_TRY(is_remove ? launch_list_remove(la, wave_per_row, ctx.stream) : launch_list_set(la, wave_per_row, ctx.stream));
    kt.stop();
    const hipMemcpyKind kind = mem == RDF_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    HIP_TRY(hipMemcpyAsync(out_offsets->values, poff32, (size_t)(n + 1) * 4, kind, ctx.stream));
    if (mem == RDF_MEM_HOST && total > 0) HIP_TRY(hipMemcpyAsync(out_values->values, dvals, (size_t)total * es, kind, ctx.stream));
    for (rdf_out* o : {out_offsets, out_values})
        if (o->validity) {
            const int64_t len = o == out_offsets ? n + 1 : total;
            if (mem == RDF_MEM_HOST) memset(o->validity, 0xFF, (size_t)((len + 7) / 8));
            else HIP_TRY(hipMemsetAsync(o->validity, 0xFF, (size_t)((len + 7) / 8), ctx.stream));
        }
    HIP_TRY(hipStreamSynchronize(ctx.stream));
    out_offsets->length = n + 1; out_offsets->null_count = 0;
    out_values->length = total; out_values->null_count = 0;
    return RDF_OK;
}
}  // namespace

rdf_status rdf_list_remove(const rdf_list_array* list, const void* value, rdf_out* out_offsets, rdf_out* out_values) {
    if (!value) return fail(RDF_INVALID_ARGUMENT, "array_remove: null argument");
    if (!list || !is_numeric(list->values.dtype)) return list_rebuild(list, nullptr, LIST_REMOVE, 0, 0, out_offsets, out_values, "array_remove");
    return list_rebuild(list, nullptr, LIST_REMOVE, scalar_bits(value, list->values.dtype), 0, out_offsets, out_values, "array_remove");
}
rdf_status rdf_list_distinct(const rdf_list_array* list, rdf_out* out_offsets, rdf_out* out_values) {
    return list_rebuild(list, nullptr, LIST_DISTINCT, 0, 0, out_offsets, out_values, "array_distinct");
}
rdf_status rdf_list_except(const rdf_list_array* a, const rdf_list_array* b, rdf_out* out_offsets, rdf_out* out_values) {
    if (!b) return fail(RDF_INVALID_ARGUMENT, "array_except: null list");
    return list_rebuild(a, b, LIST_EXCEPT, 0, 0, out_offsets, out_values, "array_except");
}
rdf_status rdf_list_intersect(const rdf_list_array* a, const rdf_list_array* b, rdf_out* out_offsets, rdf_out* out_values) {
    if (!b) return fail(RDF_INVALID_ARGUMENT, "array_intersect: null list");
    return list_rebuild(a, b, LIST_INTERSECT, 0, 0, out_offsets, out_values, "array_intersect");
}
rdf_status rdf_list_union(const rdf_list_array* a, const rdf_list_array* b, rdf_out* out_offsets, rdf_out* out_values) {
    if (!b) return fail(RDF_INVALID_ARGUMENT, "array_union: null list");
    return list_rebuild(a, b, LIST_UNION, 0, 0, out_offsets, out_values, "array_union");
}
rdf_status rdf_list_repeat(const rdf_list_array* list, int32_t count, rdf_out* out_offsets, rdf_out* out_values) {
    // `times(count)` sizes a Vec with `count as usize`: a negative count aborts the reference (capacity overflow)
    if (count < 0) return fail(RDF_INVALID_ARGUMENT, "array_repeat: negative count %d", count);
    return list_rebuild(list, nullptr, LIST_REPEAT, 0, count, out_offsets, out_values, "array_repeat");
}

// array_sort = a two-column sort of the child elements by (row number, value), composed from the public entry points
rdf_status rdf_list_sort(const rdf_list_array* list, rdf_out* out_values) {
    int64_t n = 0;
    int32_t mem = -1;
    RDF_TRY(list_check(list, "array_sort", &n, &mem));
    if (!out_values) return fail(RDF_INVALID_ARGUMENT, "array_sort: null output");
    const int cdt = list->values.dtype;
    if (out_values->dtype != cdt) return fail(RDF_INVALID_ARGUMENT, "array_sort: output must have the child dtype");
    RDF_TRY(check_out_mem(out_values, 1, mem));
    RDF_TRY(ensure_ready());
    Ctx& ctx = g_ctx;
    // the slice range [first, last) covered by rows 0 .. n-1
    int32_t ends[2] = {0, 0};
    const int32_t* hoff = (const int32_t*)list->offsets.values + list->offsets.offset;
    if (mem == RDF_MEM_HOST) { ends[0] = hoff[0]; ends[1] = hoff[n]; }
    else { RDF_TRY(rdf_copy_d2h(&ends[0], hoff, 4)); RDF_TRY(rdf_copy_d2h(&ends[1], hoff + n, 4)); }
    const int64_t first = ends[0], total = (int64_t)ends[1] - ends[0];
    if (total < 0 || ends[1] > list->values.length) return fail(RDF_INVALID_ARGUMENT, "array_sort: value_offsets outside the child array");
    if (out_values->capacity < total) return fail(RDF_MEMORY_ERROR, "output capacity too small");
    if (total == 0) { out_values->length = 0; out_values->null_count = 0; return RDF_OK; }
    if (total >= (int64_t)1 << 32) return fail(RDF_INVALID_ARGUMENT, "array_sort: more than 2^32-1 child elements");
    const size_t es = (size_t)dtype_size(cdt);
    struct Tmp { void* p = nullptr; ~Tmp() { if (p) (void)hipFree(p); } } t_off, t_vals, t_rows, t_idx, t_out;
    // device views of the value_offsets and of the child slice
    rdf_array doffs = list->offsets, dvals = list->values;
    if (mem == RDF_MEM_HOST) {
        HIP_TRY(hipMalloc(&t_off.p, (size_t)(n + 1) * 4 + 64));
        HIP_TRY(hipMalloc(&t_vals.p, (size_t)total * es + 64));
        RDF_TRY(rdf_copy_h2d(t_off.p, hoff, (n + 1) * 4));
        RDF_TRY(rdf_copy_h2d(t_vals.p, (const char*)list->values.values + (size_t)(list->values.offset + first) * es, total * (int64_t)es));
        doffs.values = t_off.p; doffs.offset = 0;
        dvals.values = t_vals.p; dvals.offset = 0;
    } else dvals.offset = list->values.offset + first;
    dvals.length = total; dvals.validity = nullptr; dvals.null_count = 0; dvals.mem = RDF_MEM_DEVICE;
    ListArgs la;
    memset(&la, 0, sizeof la);
    la.offsets = DevChunkCol{doffs.values, nullptr, doffs.offset};
    la.n = n;
    rdf_out ov = *out_values;
    if (mem == RDF_MEM_HOST) { HIP_TRY(hipMalloc(&t_out.p, (size_t)total * es + 64)); ov.values = t_out.p; ov.validity = nullptr; ov.mem = RDF_MEM_DEVICE; }
    else ov.validity = nullptr;
    {
        // rows sorted where they lie, keys in LDS (rdf_list.hip: list_sort_lane_kernel / list_sort_block_kernel); a row beyond
        // 4096 elements sends the call through the radix sort below
        arena_begin();
        void* pw = nullptr;
        RDF_TRY(arena_alloc(((size_t)n + 4) * 4 + 64, &pw));
        uint32_t* wc = (uint32_t*)pw;
        HIP_TRY(hipMemsetAsync(wc, 0, 16, ctx.stream));
        la.values = DevChunkCol{dvals.values, nullptr, dvals.offset};
        la.dtype = cdt;
        la.count = (int32_t)first;
        la.out = DevOutChunk{ov.values, nullptr};
        la.work_count = wc;
        la.work = wc + 4;
        KernelTimer kt;
        HIP_TRY(launch_list_sort(la, total / std::max<int64_t>(n, 1) < 24, ctx.stream));
        kt.stop();
        RDF_TRY(pinned_reserve(64));
        HIP_TRY(hipMemcpyAsync(ctx.pinned, wc, 8, hipMemcpyDeviceToHost, ctx.stream));
        HIP_TRY(hipStreamSynchronize(ctx.stream));
        uint32_t hw[2];
        memcpy(hw, ctx.pinned, 8);
        if (!(hw[1] & 1u)) {
            if (mem == RDF_MEM_HOST) RDF_TRY(rdf_copy_d2h(out_values->values, t_out.p, total * (int64_t)es));
            if (out_values->validity) {
                if (mem == RDF_MEM_HOST) memset(out_values->validity, 0xFF, (size_t)((total + 7) / 8));
                else { HIP_TRY(hipMemsetAsync(out_values->validity, 0xFF, (size_t)((total + 7) / 8), ctx.stream)); HIP_TRY(hipStreamSynchronize(ctx.stream)); }
            }
            out_values->length = total;
            out_values->null_count = 0;
            ctx.last_kernel = "list_sort_lane_kernel + list_sort_block_kernel";
            return RDF_OK;
        }
        la.work = nullptr; la.work_count = nullptr; la.count = 0;
    }
    HIP_TRY(hipMalloc(&t_rows.p, (size_t)total * 4 + 64));
    HIP_TRY(hipMalloc(&t_idx.p, (size_t)total * 4 + 64));
    HIP_TRY(launch_list_row_ids(la, (uint32_t*)t_rows.p, (int32_t)first, ctx.stream));
    HIP_TRY(hipStreamSynchronize(ctx.stream));
    rdf_array cols[2];
    cols[0] = rdf_array{t_rows.p, nullptr, 0, total, 0, RDF_U32, RDF_MEM_DEVICE};
    cols[1] = dvals;
    rdf_out oi{t_idx.p, nullptr, total, 0, 0, RDF_U32, RDF_MEM_DEVICE};
    RDF_TRY(rdf_sort_to_indices(cols, 2, 1, nullptr, &oi));
    rdf_array idx{t_idx.p, nullptr, 0, total, 0, RDF_U32, RDF_MEM_DEVICE};
    RDF_TRY(rdf_take(&dvals, 1, &idx, &ov));
    if (mem == RDF_MEM_HOST) RDF_TRY(rdf_copy_d2h(out_values->values, t_out.p, total * (int64_t)es));
    if (out_values->validity) {
        if (mem == RDF_MEM_HOST) memset(out_values->validity, 0xFF, (size_t)((total + 7) / 8));
        else { HIP_TRY(hipMemsetAsync(out_values->validity, 0xFF, (size_t)((total + 7) / 8), ctx.stream)); HIP_TRY(hipStreamSynchronize(ctx.stream)); }
    }
    out_values->length = total;
    out_values->null_count = 0;
    ctx.last_kernel = "list_row_ids_kernel + sort_scatter_kernel + take_kernel";
    return RDF_OK;
}

// ---------------------------------------------------------------- sort

namespace {
// bit_stats protocol of sort_keys_kernel: [0] starts ~0 (min), [1] starts 0 (max)
rdf_status sort_stats_reset(uint64_t* d_stats) {
    HIP_TRY(hipMemsetAsync(d_stats, 0xFF, 8, g_ctx.stream));
    HIP_TRY(hipMemsetAsync(d_stats + 1, 0x00, 8, g_ctx.stream));
    return RDF_OK;
}
// -> bias (smallest non-null key) and the number of low bytes of (key - bias) that can differ between two rows
rdf_status sort_key_range(const uint64_t* d_stats, size_t pin_off, uint64_t* bias, int* nbytes, uint64_t* kmax_out = nullptr) {
    Ctx& ctx = g_ctx;
    RDF_TRY(pinned_reserve(pin_off + 64));
    HIP_TRY(hipMemcpyAsync(ctx.pinned + pin_off, d_stats, 16, hipMemcpyDeviceToHost, ctx.stream));
    HIP_TRY(hipStreamSynchronize(ctx.stream));
    uint64_t st[2];
    memcpy(st, ctx.pinned + pin_off, 16);
    *bias = 0; *nbytes = 0;
    if (kmax_out) *kmax_out = 0;
    if (st[0] > st[1]) { *bias = 1; return RDF_OK; }   // no non-null key at all: [bias, kmax] = [1, 0] is the empty range
    *bias = st[0];
    if (kmax_out) *kmax_out = st[1];
    for (uint64_t range = st[1] - st[0]; range; range >>= 8) ++*nbytes;
    return RDF_OK;
}
}  // namespace

namespace {

// The digit passes of one sort column over (keys, idx) ping-pong buffers, second generation (rdf_sort.hip): every digit's
// histogram from ONE read of the keys, then one read + one write of the pairs per digit.  `need` low bytes of (key - bias) vary.
constexpr size_t kOsClassTicketBytes = (size_t)16 * 64 * 128;
struct OsScratch { unsigned long long* state = nullptr; unsigned long long* tickets = nullptr; unsigned int* ctick = nullptr; int64_t* hist = nullptr; int64_t ntiles = 0; int seq = 0;
                   int64_t* bh0 = nullptr; int64_t* bh1 = nullptr; int64_t sgrid = 0; };
rdf_status os_scratch_alloc(int64_t n, OsScratch& o) {
    o.ntiles = (n + os_tile_items() - 1) / os_tile_items();
    void* p = nullptr;
    if (g_ctx.opt_sort_gen == 2) {   // static ranges: per-block digit counts and their scan
        o.sgrid = sr_grid(o.ntiles);
        RDF_TRY(arena_alloc((size_t)(256 * o.sgrid + 1) * 8, &p));
        o.bh0 = (int64_t*)p;
        RDF_TRY(arena_alloc((size_t)(256 * o.sgrid + 1 + scan_scratch_words(256 * o.sgrid)) * 8, &p));
        o.bh1 = (int64_t*)p;
        return RDF_OK;
    }
    RDF_TRY(arena_alloc((size_t)(o.ntiles + kOsStatePadTiles) * 256 * 8 + 64, &p));
    o.state = (unsigned long long*)p;
    HIP_TRY(hipMemsetAsync(p, 0, (size_t)(o.ntiles + kOsStatePadTiles) * 256 * 8, g_ctx.stream));
    RDF_TRY(arena_alloc(16 * 8 + 9 * 256 * 8, &p));
    o.tickets = (unsigned long long*)p;
    o.hist = (int64_t*)((char*)p + 16 * 8);
    o.seq = 0;
    // os_scatter3_kernel: 64 ticket counters per pass, 128 bytes apart, for the up to 16 passes of one column
    RDF_TRY(arena_alloc(kOsClassTicketBytes, &p));
    o.ctick = (unsigned int*)p;
    HIP_TRY(hipMemsetAsync(p, 0, kOsClassTicketBytes, g_ctx.stream));
    return RDF_OK;
}
rdf_status os_column_passes(OsScratch& o, uint64_t* const keys[2], uint32_t* const idxb[2], const uint8_t* nullflags, int64_t n, uint64_t bias, int need,
                            bool null_pass, int& kcur, int& icur, const uint32_t*& idx_cur, uint64_t range = ~0ull /* max key - bias, when known */,
                            int f64_keys = 0 /* 1: the keys are the bits of doubles (2: stored inverted, descending); -1: of 4-byte floats */) {
    Ctx& ctx = g_ctx;
    if (need == 0 && !null_pass) return RDF_OK;
    if (ctx.opt_sort_gen == 2) {
        for (int p = 0; p < need + (null_pass ? 1 : 0); ++p) {
            const bool np = p == need;
            OsPassArgs pa;
            memset(&pa, 0, sizeof pa);
            pa.keys_in = keys[kcur]; pa.idx_in = idx_cur; pa.keys_out = keys[kcur ^ 1]; pa.idx_out = idxb[icur ^ 1];
            pa.nullflags = np ? nullflags : nullptr;
            pa.n = n; pa.ntiles = o.ntiles; pa.bias = bias; pa.shift = 8 * p;
            HIP_TRY(launch_sr_hist(pa, o.bh0, ctx.stream));
            HIP_TRY(launch_scan(o.bh0, o.bh1, 256 * o.sgrid, o.bh1 + 256 * o.sgrid + 1, ctx.stream));
            HIP_TRY(launch_sr_scatter(pa, o.bh1, ctx.stream));
            kcur ^= 1; icur ^= 1; idx_cur = idxb[icur];
        }
        return RDF_OK;
    }
    OsBucket fb;
    memset(&fb, 0, sizeof fb);
    auto one_pass = [&](int hist_row, int shift, int mask, bool np, int& launched, int64_t rows) -> rdf_status {
        if (o.seq >= 16000) { HIP_TRY(hipMemsetAsync(o.state, 0, (size_t)o.ntiles * 256 * 8, ctx.stream)); o.seq = 0; }
        OsPassArgs pa;
        memset(&pa, 0, sizeof pa);
        pa.keys_in = keys[kcur]; pa.idx_in = idx_cur; pa.keys_out = keys[kcur ^ 1]; pa.idx_out = idxb[icur ^ 1];
        pa.nullflags = np ? nullflags : nullptr;
        pa.state = o.state; pa.ticket = o.tickets + launched; pa.class_tickets = ctx.opt_sort_pipe ? o.ctick + (size_t)launched * 64 * 32 : nullptr; pa.bases = o.hist + hist_row * 256;
        pa.n = rows; pa.ntiles = (rows + os_tile_items() - 1) / os_tile_items(); pa.bias = bias; pa.shift = shift; pa.mask = mask; pa.seq = ++o.seq;
        pa.super_tiles = ctx.opt_sort_pipe ? 1 : ctx.opt_sort_super_force ? ctx.opt_sort_super_force : os_super_tiles(pa.ntiles, ctx.opt_sort_super);
        if (!np) pa.fb = fb;
        static const bool dbg3 = getenv("RDF_DEBUG_SORT") != nullptr;
        if (dbg3) { pa.debug = o.tickets + 8; HIP_TRY(hipMemsetAsync(pa.debug, 0, 48, ctx.stream)); }
        HIP_TRY(launch_os_scatter(pa, ctx.stream));
        if (dbg3) {
            unsigned long long h[6];
            HIP_TRY(hipMemcpyAsync(h, pa.debug, 48, hipMemcpyDeviceToHost, ctx.stream));
            HIP_TRY(hipStreamSynchronize(ctx.stream));
            fprintf(stderr, "[rdf] digit pass (shift %d, %lld tiles): waiting for offsets %.2f us per tile of %.2f us; scanner rounds %llu, idle %llu, partial %llu\n", shift, (long long)pa.ntiles,
                    h[5] ? h[0] * 0.01 / h[5] : 0.0, h[5] ? h[1] * 0.01 / h[5] : 0.0, h[2], h[3], h[4]);
        }
        ++launched;
        kcur ^= 1; icur ^= 1; idx_cur = idxb[icur];
        return RDF_OK;
    };
    // Keys that vary in 25 bits or more: stable passes over the TOP bits only (as many as leave buckets of ~1000 rows), then every
    // bucket is sorted on its remaining bits by one block in LDS — 2 (3 from ~2.7e8 rows) passes + one read and write of the
    // pairs instead of 4 .. 8 passes.  NULL keys (they all carry one key value: one huge bucket) are moved behind the others
    // FIRST by the stable NULLs-last pass and stay there, in order, while the rows in front of them are sorted.  A bucket above
    // kOsLocalMax rows (keys crowded on few top-bit patterns) sends the column through the byte passes below, which sort any order.
    int sig = 8 * need;                       // bits of (key - bias) that vary
    if (range != ~0ull) { sig = 0; for (uint64_t r = range; r; r >>= 1) ++sig; sig = std::min(sig, 8 * need); }
    // Top bits: as few passes of <= 8 bits as leave buckets the LDS finish takes, then as many bits as those passes carry for
    // free — integer keys: 2 passes (16 bits) up to ~1.2e8 rows (1e8 rows: 1526 rows per bucket on average, sorted as 2048),
    // a third pass beyond with ~500 rows per bucket; value buckets of doubles: ~500 per bucket always (a bell-shaped column's
    // densest bucket holds 4-5 x the average; measured at 5e7 rows: 800 per bucket sorts uniform doubles in 3.20 instead of
    // 3.35 ms but sends normally distributed ones to the byte passes, 5.5 instead of 4.3 ms)
    const int per_bucket = f64_keys > 0 ? 512 : 1800;
    int B = 12;
    bool sampled = false;                     // the bucket bits of a column of doubles were planned from a sample
    while ((n >> B) > 512 && B < 24) ++B;
    if (f64_keys <= 0) {
        int np = 1;
        while (np < 3 && (n >> (8 * np)) > per_bucket) ++np;
        B = std::max(12, std::min(B, 8 * np));
        if ((n >> B) > per_bucket) B = 24;      // (no pass count fits: the condition below fails)
    }
    if (f64_keys > 0 && range != ~0ull) {
        // doubles: sign and exponent crowd the key bits' top patterns, so the buckets are cut in VALUE space (OsBucket)
        auto value_of = [&](uint64_t stored) { const uint64_t ord = f64_keys == 2 ? ~stored : stored; const uint64_t b = (ord >> 63) ? (ord ^ 0x8000000000000000ull) : ~ord; double x; memcpy(&x, &b, 8); return x; };
        const double x0 = value_of(bias), x1 = value_of(bias + range);
        double lo = -HUGE_VAL, hi = HUGE_VAL;
        if (x0 == x0 && x1 == x1) { lo = std::min(x0, x1); hi = std::max(x0, x1); }          // (a NaN at either end of the key range: no range)
        // The buckets are cut from the range the ROWS lie in and as their values are distributed, not from [min, max] in equal
        // widths: a sample of <= 8192 keys gives (a) that range — extended by half its width on either side (the tails of 1e9
        // draws of a bell curve reach 1.6 x as far as those of 8192); keys outside fall into 1/32 of the buckets at either end,
        // laid out geometrically (OsBucket::tail): far outliers, infinities, NaNs and the thin ends of heavy-tailed columns —, (b) the share of every
        // one of 256 equal-width segments of it, which gets that share of the buckets (OsSeg: a piecewise-linear, monotone map
        // that fills the buckets of bell-shaped and heavy-tailed columns as evenly as those of uniform ones), and (c) warnings
        // that no map helps: a key met twice (a value held by thousands of rows), a region that keeps concentrating however
        // far one zooms in (x^6 of an exponential), infinities / NaNs by the thousand — those columns take the byte passes at
        // once instead of finding out at the bucket-size check after the passes.
        if (ctx.opt_sort_msd && ctx.opt_sort_sample && n >= 65536 && n < ((int64_t)1 << 32)) {
            const int S = (int)std::min<int64_t>(8192, n);
            void* ps = nullptr;
            RDF_TRY(arena_alloc((size_t)S * 8, &ps));
            HIP_TRY(launch_os_sample(keys[kcur], n, S, (uint64_t*)ps, ctx.stream));
            std::vector<uint64_t> hs((size_t)S);
            HIP_TRY(hipMemcpyAsync(hs.data(), ps, (size_t)S * 8, hipMemcpyDeviceToHost, ctx.stream));
            HIP_TRY(hipStreamSynchronize(ctx.stream));
            std::vector<double> xs;
            xs.reserve((size_t)S);
            int outside = 0;
            for (uint64_t k : hs) {
                if (k < bias || k - bias > range) continue;                        // a NULL row's placeholder key
                const double x = value_of(k);
                if (std::isfinite(x)) xs.push_back(x); else ++outside;
            }
            OsPlan plan;
            os_plan_f64(xs.data(), xs.size(), outside, n, lo, hi, f64_keys == 2 ? 1 : 0, plan);       // rdf_sort_map.h
            sampled = plan.sampled;
            if (sampled) {
                fb = plan.fb;
                B = fb.bits;
                if (!fb.flat) {
                    void* pseg = nullptr;
                    RDF_TRY(arena_alloc(sizeof(OsSeg) * plan.segs.size(), &pseg));
                    HIP_TRY(hipMemcpyAsync(pseg, plan.segs.data(), sizeof(OsSeg) * plan.segs.size(), hipMemcpyHostToDevice, ctx.stream));
                    HIP_TRY(hipStreamSynchronize(ctx.stream));                     // (plan is a local)
                    fb.seg = (const OsSeg*)pseg;
                }
            }
            static const bool dbg = getenv("RDF_DEBUG_SORT") != nullptr;
            if (dbg) fprintf(stderr, "[rdf] sort: sample of %zu finite keys (%d not finite, a value at most %d times) in [%g, %g] of [%g, %g], fullest segment %.1f x uneven inside -> %s, %d bucket bits\n",
                             xs.size(), outside, plan.most_equal, plan.smin, plan.smax, lo, hi, plan.within,
                             sampled ? (fb.flat ? "evenly spread: one linear map between the tails" : "buckets by segment shares") : "byte passes", plan.bits);
        } else {
            const double scale = (double)((int64_t)1 << B) / (hi - lo);
            if (std::isfinite(lo) && std::isfinite(hi) && hi > lo && std::isfinite(scale)) { fb.lo = lo; fb.scale = scale; fb.bits = B; fb.flip = f64_keys == 2; }
        }
    }
    // (f64_keys < 0: the bit patterns of 4-byte floats — sign and exponent crowd their top digit like a double's, the top-digit check below
    // would send them to the byte passes after a histogram and a host round trip, 0.17 ms per 5e7 keys; four byte passes are what a
    // value-bucket map would have to beat, and it would not: DESIGN.md 7.3)
    if (ctx.opt_sort_msd && f64_keys >= 0 && n >= 65536 && n < ((int64_t)1 << 32) && ((n >> B) <= per_bucket || (sampled && fb.bits)) && (sig + 7) / 8 >= (B + 7) / 8 + 2 && (fb.bits || sig - B <= 52)) {
        const int R = fb.bits ? 0 : sig - B;
        const int npass = (B + 7) / 8;
        const int nbuckets = 1 << B;
        int launched = 0;
        int64_t nv = n;                       // rows in front of the NULL keys
        int ks = -1, is = -1;                 // the buffers the NULL rows were left in
        HIP_TRY(hipMemsetAsync(o.tickets, 0, 16 * 8 + 9 * 256 * 8, ctx.stream));
    HIP_TRY(hipMemsetAsync(o.ctick, 0, kOsClassTicketBytes, ctx.stream));
        OsHistArgs ha;
        memset(&ha, 0, sizeof ha);
        ha.bias = bias; ha.hist = o.hist; ha.generic = 1; ha.fb = fb;
        if (null_pass) {
            ha.keys = keys[kcur]; ha.nullflags = nullflags; ha.n = n; ha.npass = 0;
            HIP_TRY(launch_os_hist(ha, ctx.stream));
            RDF_TRY(one_pass(8, 0, 255, true, launched, n));
            ks = kcur; is = icur;
            int64_t front = 0;
            HIP_TRY(hipMemcpyAsync(&front, o.hist + 8 * 256 + 1, 8, hipMemcpyDeviceToHost, ctx.stream));   // where the NULL run starts
            HIP_TRY(hipStreamSynchronize(ctx.stream));
            nv = front;
            HIP_TRY(hipMemsetAsync(o.hist, 0, 9 * 256 * 8, ctx.stream));
        }
        if (nv > 0) {
            ha.keys = keys[kcur]; ha.nullflags = nullptr; ha.n = nv; ha.npass = npass;
            int sh = R;
            for (int p = 0; p < npass; ++p) {
                const int w = p == 0 ? B - 8 * (npass - 1) : 8;
                ha.shift[p] = sh; ha.mask[p] = (1 << w) - 1;
                sh += w;
            }
            HIP_TRY(launch_os_hist(ha, ctx.stream));
            static const bool dbg = getenv("RDF_DEBUG_SORT") != nullptr;
            if (!(sampled && fb.bits)) {   // the top digit's counts tell crowded keys (floats of one sign and a few exponents, ...) before any pass is spent on
                // them: a top-digit value held by more rows than its share of the buckets could take at kOsLocalMax rows each
                // (a map planned from a sample of the keys has answered that already: no host round trip for it; a plan that
                // misjudged the column is caught by the largest bucket below, two passes later)
                int64_t top[257];
                HIP_TRY(hipMemcpyAsync(top, o.hist + (npass - 1) * 256, 256 * 8, hipMemcpyDeviceToHost, ctx.stream));
                HIP_TRY(hipStreamSynchronize(ctx.stream));
                top[256] = nv;
                const int wtop = npass == 1 ? B : 8;
                int64_t most = 0;
                for (int i = 0; i < (1 << wtop); ++i) most = std::max(most, top[i + 1 == (1 << wtop) ? 256 : i + 1] - top[i]);
                if (most > ((int64_t)kOsLocalMax << (B - wtop))) {
                    if (dbg) fprintf(stderr, "[rdf] sort: %lld of %lld rows share their top %d bits -> byte passes\n", (long long)most, (long long)nv, wtop);
                    goto byte_passes;
                }
            }
            for (int p = 0; p < npass; ++p) RDF_TRY(one_pass(p, ha.shift[p], ha.mask[p], false, launched, nv));
            void* pb = nullptr;
            RDF_TRY(arena_alloc((size_t)(nbuckets + 2) * 4 + 64, &pb));
            uint32_t* bstart = (uint32_t*)pb;
            unsigned int* dmax = (unsigned int*)(bstart + nbuckets + 1);
            HIP_TRY(hipMemsetAsync(dmax, 0, 4, ctx.stream));
            HIP_TRY(launch_os_bounds(keys[kcur], nv, bias, R, nbuckets, fb, bstart, dmax, ctx.stream));
            unsigned int maxlen = 0;
            HIP_TRY(hipMemcpyAsync(&maxlen, dmax, 4, hipMemcpyDeviceToHost, ctx.stream));
            HIP_TRY(hipStreamSynchronize(ctx.stream));
            if (dbg) fprintf(stderr, "[rdf] sort: %lld rows (+ %lld NULL keys), %d top bits of %d in %d passes, %d buckets, largest %u rows -> %s\n", (long long)nv, (long long)(n - nv), B, sig, npass, nbuckets, maxlen, maxlen <= (unsigned)kOsLocalMax ? "LDS finish" : "byte passes");
            if (maxlen <= (unsigned)kOsLocalMax) {
                OsLocalArgs la;
                memset(&la, 0, sizeof la);
                la.keys_in = keys[kcur]; la.idx_in = idx_cur; la.keys_out = keys[kcur ^ 1]; la.idx_out = idxb[icur ^ 1];
                la.bstart = bstart; la.bias = bias; la.rbits = R; la.nbuckets = nbuckets; la.wide = fb.bits ? 1 : 0;
                la.lds_items = 256;
                while (la.lds_items < (int)maxlen) la.lds_items <<= 1;
                HIP_TRY(launch_os_local(la, ctx.stream));
                ctx.sort_used_local = true;
                kcur ^= 1; icur ^= 1; idx_cur = idxb[icur];
                if (nv < n) {   // the NULL rows follow the sorted ones into the buffers that now hold the order
                    if (ks != kcur) HIP_TRY(hipMemcpyAsync(keys[kcur] + nv, keys[ks] + nv, (size_t)(n - nv) * 8, hipMemcpyDeviceToDevice, ctx.stream));
                    if (is != icur) HIP_TRY(hipMemcpyAsync(idxb[icur] + nv, idxb[is] + nv, (size_t)(n - nv) * 4, hipMemcpyDeviceToDevice, ctx.stream));
                }
                return RDF_OK;
            }
            // (the byte passes below start from whatever order the rows are in now, NULLs-last pass included)
        } else return RDF_OK;                 // every key is NULL: the NULLs-last pass was the whole sort
    }
byte_passes:
    fb.bits = 0;
    HIP_TRY(hipMemsetAsync(o.tickets, 0, 16 * 8 + 9 * 256 * 8, ctx.stream));
    HIP_TRY(hipMemsetAsync(o.ctick, 0, kOsClassTicketBytes, ctx.stream));
    OsHistArgs ha;
    memset(&ha, 0, sizeof ha);
    ha.keys = keys[kcur]; ha.nullflags = null_pass ? nullflags : nullptr; ha.n = n; ha.bias = bias; ha.npass = need; ha.hist = o.hist;
    HIP_TRY(launch_os_hist(ha, ctx.stream));
    int launched = 0;
    for (int p = 0; p < need + (null_pass ? 1 : 0); ++p) {
        const bool np = p == need;
        if (o.seq >= 16000) { HIP_TRY(hipMemsetAsync(o.state, 0, (size_t)o.ntiles * 256 * 8, ctx.stream)); o.seq = 0; }
        OsPassArgs pa;
        memset(&pa, 0, sizeof pa);
        pa.keys_in = keys[kcur]; pa.idx_in = idx_cur; pa.keys_out = keys[kcur ^ 1]; pa.idx_out = idxb[icur ^ 1];
        pa.nullflags = np ? nullflags : nullptr;
        pa.state = o.state; pa.ticket = o.tickets + launched; pa.class_tickets = ctx.opt_sort_pipe ? o.ctick + (size_t)launched * 64 * 32 : nullptr; pa.bases = o.hist + (np ? 8 : p) * 256;
        pa.n = n; pa.ntiles = o.ntiles; pa.bias = bias; pa.shift = 8 * p; pa.seq = ++o.seq;
        pa.super_tiles = ctx.opt_sort_pipe ? 1 : ctx.opt_sort_super_force ? ctx.opt_sort_super_force : os_super_tiles(pa.ntiles, ctx.opt_sort_super);
        static const bool dbg = getenv("RDF_DEBUG_SORT") != nullptr;
        if (dbg) { pa.debug = o.tickets + 8; }
        HIP_TRY(launch_os_scatter(pa, ctx.stream));
        if (dbg) {
            unsigned long long h[6];
            HIP_TRY(hipMemcpyAsync(h, pa.debug, 48, hipMemcpyDeviceToHost, ctx.stream));
            HIP_TRY(hipStreamSynchronize(ctx.stream));
            HIP_TRY(hipMemsetAsync(pa.debug, 0, 48, ctx.stream));
            const double t = (double)std::max<unsigned long long>(h[5], 1);
            fprintf(stderr, "[rdf] os_scatter pass %d: per tile cycles (s_memtime, 100 MHz): ticket %.0f load+rank %.0f barrier %.0f look-back %.0f sort+write %.0f (%llu tiles)\n", p, h[0] / t, h[1] / t, h[2] / t, h[3] / t, h[4] / t, h[5]);
        }
        ++launched;
        kcur ^= 1; icur ^= 1; idx_cur = idxb[icur];
    }
    return RDF_OK;
}

// The radix passes of DataFrame::sort over device-resident descriptor tables: d_chunks[k * nchunks + c] = chunk c of sort
// column k (column 0 most significant), d_row_start = prefix of the batch lengths.  *idx_out = the sorted row order (u32,
// arena memory, valid until the next arena_begin).  Shared by rdf_sort_to_indices and rdf_sort_frame.
rdf_status sort_core(const DevChunkCol* d_chunks, const int64_t* d_row_start, int64_t nchunks, int64_t n, int ncols, const int* dts,
                     const bool* nullable, const rdf_sort_options* opts, size_t pin_off, const uint32_t** idx_out) {
    Ctx& ctx = g_ctx;
    const int64_t ntiles = (n + kSortTile - 1) / kSortTile;
    void *pk0, *pk1, *pi0, *pi1, *pnf, *ph0, *ph1;
    RDF_TRY(arena_alloc((size_t)n * 8, &pk0));
    RDF_TRY(arena_alloc((size_t)n * 8, &pk1));
    RDF_TRY(arena_alloc((size_t)n * 4, &pi0));
    RDF_TRY(arena_alloc((size_t)n * 4, &pi1));
    RDF_TRY(arena_alloc((size_t)n, &pnf));
    const int64_t sgrid = sort_grid(ntiles);
    RDF_TRY(arena_alloc((size_t)(256 * sgrid + 1) * 8, &ph0));
    RDF_TRY(arena_alloc((size_t)(256 * sgrid + 1 + scan_scratch_words(256 * sgrid)) * 8, &ph1));
    void* pstats;
    RDF_TRY(arena_alloc(64, &pstats));
    uint64_t* d_stats = (uint64_t*)pstats;
    uint64_t* keys[2] = {(uint64_t*)pk0, (uint64_t*)pk1};
    uint32_t* idxb[2] = {(uint32_t*)pi0, (uint32_t*)pi1};
    int kcur = 0;               // keys[kcur] holds the current keys
    const uint32_t* idx_cur = nullptr;  // nullptr = identity order
    int icur = 1;               // idxb[icur ^ 1] receives the next order

    OsScratch os;
    ctx.sort_used_local = false;
    const bool gen2 = ctx.opt_sort_gen >= 2;   // rdf_sort.hip
    if (gen2) RDF_TRY(os_scratch_alloc(n, os));
    KernelTimer kt;
    for (int k = ncols - 1; k >= 0; --k) {  // LSD over the sort columns: least significant criterion first
        const int dt = dts[k];
        const bool has_nulls = nullable[k];
        SortKeyArgs ka;
        memset(&ka, 0, sizeof ka);
        ka.chunks = d_chunks + (size_t)k * (size_t)nchunks;
        ka.chunk_row_start = d_row_start;
        ka.nchunks = nchunks;
        ka.n = n;
        ka.idx = idx_cur;
        ka.keys = keys[kcur];
        ka.nullflags = has_nulls ? (uint8_t*)pnf : nullptr;
        ka.dtype = dt;
        ka.descending = opts ? opts[k].descending : 0;
        ka.bit_stats = d_stats;
        RDF_TRY(sort_stats_reset(d_stats));
        HIP_TRY(launch_sort_keys(ka, ctx.stream));
        // radix passes cover only the bytes of (max key - min key): a 32-bit range stored as i64 takes 4 passes, a dictionary code 1
        uint64_t bias = 0;
        int need = 0;
        uint64_t kmax = 0;
        RDF_TRY(sort_key_range(d_stats, pin_off, &bias, &need, &kmax));
        if (k == 0 && idx_cur == nullptr && need == 0 && !has_nulls) need = 1;   // all keys equal: one pass still writes the identity order
        if (gen2) { RDF_TRY(os_column_passes(os, keys, idxb, (const uint8_t*)pnf, n, bias, std::min(need, dtype_size(dt)), has_nulls, kcur, icur, idx_cur, kmax >= bias ? kmax - bias : ~0ull, dt == RDF_F64 ? (ka.descending ? 2 : 1) : dt == RDF_F32 ? -1 : 0)); continue; }
        const int npass = dtype_size(dt) + (has_nulls ? 1 : 0);
        for (int p = 0; p < npass; ++p) {
            if (p < dtype_size(dt) && p >= need) continue;
            SortPassArgs pa;
            memset(&pa, 0, sizeof pa);
            pa.bias = bias;
            pa.keys_in = keys[kcur];
            pa.idx_in = idx_cur;
            pa.keys_out = keys[kcur ^ 1];
            pa.idx_out = idxb[icur ^ 1];
            pa.nullflags = p == dtype_size(dt) ? (const uint8_t*)pnf : nullptr;  // the nulls-last pass
            pa.hist = (int64_t*)ph0;
            pa.n = n;
            pa.ntiles = ntiles;
            pa.shift = 8 * p;
            HIP_TRY(launch_sort_hist(pa, ctx.stream));
            HIP_TRY(launch_scan((const int64_t*)ph0, (int64_t*)ph1, 256 * sgrid, (int64_t*)ph1 + 256 * sgrid + 1, ctx.stream));
            pa.hist = (int64_t*)ph1;
            HIP_TRY(launch_sort_scatter(pa, ctx.stream));
            kcur ^= 1;
            icur ^= 1;
            idx_cur = idxb[icur];
        }
    }
    kt.stop();
    ctx.last_kernel = ctx.sort_used_local ? "os_scatter_kernel+os_local_kernel" : "sort_scatter_kernel";
    *idx_out = idx_cur;
    return RDF_OK;
}
}  // namespace

rdf_status rdf_sort_to_indices(const rdf_array* cols, int32_t ncols, int64_t nchunks, const rdf_sort_options* opts,
                               rdf_out* out_indices) {
    if (ncols < 1 || !cols) return fail(RDF_COMPUTE_ERROR, "Sort criteria cannot be empty");  // src/dataframe.rs:195-199
    if (nchunks < 1 || !out_indices) return fail(RDF_INVALID_ARGUMENT, "sort: bad arguments");
    int32_t mem = -1;
    RDF_TRY(check_mem(cols, (int64_t)ncols * nchunks, &mem));
    RDF_TRY(check_out_mem(out_indices, 1, mem));
    if (out_indices->dtype != RDF_U32) return fail(RDF_INVALID_ARGUMENT, "sort: indices are UInt32");
    std::vector<int64_t> row_start((size_t)nchunks + 1, 0);
    for (int64_t c = 0; c < nchunks; ++c) row_start[(size_t)c + 1] = row_start[(size_t)c] + cols[c].length;
    const int64_t n = row_start[(size_t)nchunks];
    for (int k = 0; k < ncols; ++k)
        for (int64_t c = 0; c < nchunks; ++c) {
            const rdf_array& a = cols[(int64_t)k * nchunks + c];
            if (!is_numeric(a.dtype) || a.dtype != cols[(int64_t)k * nchunks].dtype) return fail(RDF_INVALID_ARGUMENT, "sort: numeric columns of one dtype per column");
            if (a.length != cols[c].length) return fail(RDF_COMPUTE_ERROR, "sort: columns of a batch differ in length");
        }
    if (n >= (int64_t)1 << 32) return fail(RDF_INVALID_ARGUMENT, "sort: UInt32 indices cap a column at 2^32-1 rows (src/table.rs:218)");
    if (out_indices->capacity < n) return fail(RDF_MEMORY_ERROR, "output capacity too small");
    if (n == 0) { out_indices->length = 0; out_indices->null_count = 0; return RDF_OK; }
    RDF_TRY(ensure_ready());
    Ctx& ctx = g_ctx;
    arena_begin();
    size_t pin_off = 0, used = 0;
    InputStager in;
    for (int64_t i = 0; i < (int64_t)ncols * nchunks; ++i) in.add(&cols[i]);
    RDF_TRY(in.finish(pin_off, &used));
    pin_off += (used + 255) & ~(size_t)255;
    TableBuilder tb;
    const size_t o_ch = tb.reserve(sizeof(DevChunkCol) * in.dev.size());
    const size_t o_rs = tb.reserve(sizeof(int64_t) * row_start.size());
    RDF_TRY(tb.bind(pin_off));
    memcpy(tb.at<char>(o_ch), in.dev.data(), sizeof(DevChunkCol) * in.dev.size());
    memcpy(tb.at<char>(o_rs), row_start.data(), sizeof(int64_t) * row_start.size());
    RDF_TRY(tb.alloc());
    RDF_TRY(tb.upload(pin_off));
    pin_off += (tb.size + 255) & ~(size_t)255;

    int dts[kMaxFrameCols];
    bool nullable[kMaxFrameCols];
    if (ncols > kMaxFrameCols) return fail(RDF_INVALID_ARGUMENT, "sort: at most %d sort columns", kMaxFrameCols);
    for (int k = 0; k < ncols; ++k) {
        dts[k] = cols[(int64_t)k * nchunks].dtype;
        nullable[k] = false;
        for (int64_t c = 0; c < nchunks; ++c) nullable[k] |= cols[(int64_t)k * nchunks + c].validity != nullptr;
    }
    const uint32_t* idx_cur = nullptr;
    RDF_TRY(sort_core(tb.dev_at<DevChunkCol>(o_ch), tb.dev_at<int64_t>(o_rs), nchunks, n, ncols, dts, nullable, opts, pin_off, &idx_cur));
    if (mem == RDF_MEM_HOST) {
        HIP_TRY(hipMemcpyAsync(out_indices->values, idx_cur, (size_t)n * 4, hipMemcpyDeviceToHost, ctx.stream));
        if (out_indices->validity) memset(out_indices->validity, 0xFF, (size_t)((n + 7) / 8));
    } else {
        HIP_TRY(hipMemcpyAsync(out_indices->values, idx_cur, (size_t)n * 4, hipMemcpyDeviceToDevice, ctx.stream));
        if (out_indices->validity) HIP_TRY(hipMemsetAsync(out_indices->validity, 0xFF, (size_t)((n + 7) / 8), ctx.stream));
    }
    HIP_TRY(hipStreamSynchronize(ctx.stream));
    out_indices->length = n;
    out_indices->null_count = 0;
    return RDF_OK;
}

// ---------------------------------------------------------------- join

namespace {
// Radix-sort (key bits, row) pairs already produced by sort_keys_kernel: `width` key bytes, then the nulls-last pass.
struct SortBuffers { uint64_t* keys[2]; uint32_t* idx[2]; void* nullflags; void* hist0; void* hist1; int64_t sgrid, ntiles; };
rdf_status sort_buffers_alloc(int64_t n, SortBuffers& b) {
    b.ntiles = (n + kSortTile - 1) / kSortTile;
    b.sgrid = sort_grid(b.ntiles);
    void* p;
    for (int i = 0; i < 2; ++i) { RDF_TRY(arena_alloc((size_t)n * 8 + 8, &p)); b.keys[i] = (uint64_t*)p; }
    for (int i = 0; i < 2; ++i) { RDF_TRY(arena_alloc((size_t)n * 4 + 8, &p)); b.idx[i] = (uint32_t*)p; }
    RDF_TRY(arena_alloc((size_t)n + 8, &b.nullflags));
    RDF_TRY(arena_alloc((size_t)(256 * b.sgrid + 1) * 8, &b.hist0));
    RDF_TRY(arena_alloc((size_t)(256 * b.sgrid + 1 + scan_scratch_words(256 * b.sgrid)) * 8, &b.hist1));
    return RDF_OK;
}
// keys[0] holds the unsorted key bits (identity order).  On return keys[*kcur] / idx[*icur] are sorted.
rdf_status radix_sort_rows(const SortBuffers& b, int64_t n, int width, bool has_nulls, int* kcur_out, int* icur_out, const uint64_t* d_stats, size_t pin_off,
                           uint64_t* kmin_out, uint64_t* kmax_out) {
    Ctx& ctx = g_ctx;
    int kcur = 0, icur = 1;
    const uint32_t* idx_cur = nullptr;
    const int npass = width + (has_nulls ? 1 : 0);
    uint64_t bias = 0;
    int need = 0;
    uint64_t kmax = 0;
    RDF_TRY(sort_key_range(d_stats, pin_off, &bias, &need, &kmax));
    if (kmax_out) *kmax_out = kmax;
    *kmin_out = bias;
    if (need == 0 && !has_nulls) need = 1;   // all keys equal: one pass still writes the identity order
    if (ctx.opt_sort_gen >= 2) {
        OsScratch os;
        RDF_TRY(os_scratch_alloc(n, os));
        RDF_TRY(os_column_passes(os, b.keys, b.idx, (const uint8_t*)b.nullflags, n, bias, std::min(need, width), has_nulls, kcur, icur, idx_cur, kmax >= bias ? kmax - bias : ~0ull));
        *kcur_out = kcur;
        *icur_out = icur;
        return RDF_OK;
    }
    for (int p = 0; p < npass; ++p) {
        if (p < width && p >= need) continue;
        SortPassArgs pa;
        memset(&pa, 0, sizeof pa);
        pa.bias = bias;
        pa.keys_in = b.keys[kcur];
        pa.idx_in = idx_cur;
        pa.keys_out = b.keys[kcur ^ 1];
        pa.idx_out = b.idx[icur ^ 1];
        pa.nullflags = p == width ? (const uint8_t*)b.nullflags : nullptr;
        pa.hist = (int64_t*)b.hist0;
        pa.n = n;
        pa.ntiles = b.ntiles;
        pa.shift = 8 * p;
        HIP_TRY(launch_sort_hist(pa, ctx.stream));
        HIP_TRY(launch_scan((const int64_t*)b.hist0, (int64_t*)b.hist1, 256 * b.sgrid, (int64_t*)b.hist1 + 256 * b.sgrid + 1, ctx.stream));
        pa.hist = (int64_t*)b.hist1;
        HIP_TRY(launch_sort_scatter(pa, ctx.stream));
        kcur ^= 1;
        icur ^= 1;
        idx_cur = b.idx[icur];
    }
    *kcur_out = kcur;
    *icur_out = icur;
    return RDF_OK;
}
}  // namespace

rdf_status rdf_equijoin_indices(const rdf_array* left_keys, int64_t left_nchunks, const rdf_array* right_keys, int64_t right_nchunks,
                                int32_t join_type, rdf_out* out_left, rdf_out* out_right, int64_t* out_rows) {
    return rdf_equijoin_indices_multi(left_keys, left_nchunks, right_keys, right_nchunks, 1, join_type, out_left, out_right, out_rows);
}

rdf_status rdf_equijoin_indices_multi(const rdf_array* left_keys, int64_t left_nchunks, const rdf_array* right_keys, int64_t right_nchunks,
                                      int32_t nkeys, int32_t join_type, rdf_out* out_left, rdf_out* out_right, int64_t* out_rows) {
    if (!left_keys || !right_keys || left_nchunks < 1 || right_nchunks < 1 || !out_rows) return fail(RDF_INVALID_ARGUMENT, "join: bad arguments");
    if (nkeys < 1 || nkeys > 4) return fail(RDF_INVALID_ARGUMENT, "join: 1 to 4 key columns per side");
    if (join_type < RDF_JOIN_LEFT || join_type > RDF_JOIN_FULL) return fail(RDF_INVALID_ARGUMENT, "join: bad join type");
    if ((out_left == nullptr) != (out_right == nullptr)) return fail(RDF_INVALID_ARGUMENT, "join: give both outputs or neither (count only)");
    int32_t mem = -1;
    RDF_TRY(check_mem(left_keys, (int64_t)nkeys * left_nchunks, &mem));
    RDF_TRY(check_mem(right_keys, (int64_t)nkeys * right_nchunks, &mem));
    if (out_left) { RDF_TRY(check_out_mem(out_left, 1, mem)); RDF_TRY(check_out_mem(out_right, 1, mem)); }
    int kdt[4] = {0, 0, 0, 0};
    int64_t nleft = 0, nright = 0;
    bool lnulls = false, rnulls = false;
    for (int k = 0; k < nkeys; ++k) {   // key pair k: left_keys[k * left_nchunks + c] against right_keys[k * right_nchunks + c]
        kdt[k] = left_keys[(int64_t)k * left_nchunks].dtype;
        if (!is_numeric(kdt[k])) return fail(RDF_INVALID_ARGUMENT, "join: numeric key columns only");
        int64_t nl = 0, nr = 0;
        for (int64_t c = 0; c < left_nchunks; ++c) { const rdf_array& a = left_keys[(int64_t)k * left_nchunks + c]; if (a.dtype != kdt[k]) return fail(RDF_INVALID_ARGUMENT, "join: key columns must share one dtype (cast first)"); nl += a.length; lnulls |= a.validity != nullptr; }
        for (int64_t c = 0; c < right_nchunks; ++c) { const rdf_array& a = right_keys[(int64_t)k * right_nchunks + c]; if (a.dtype != kdt[k]) return fail(RDF_INVALID_ARGUMENT, "join: key columns must share one dtype (cast first)"); nr += a.length; rnulls |= a.validity != nullptr; }
        if (k == 0) { nleft = nl; nright = nr; }
        else if (nl != nleft || nr != nright) return fail(RDF_COMPUTE_ERROR, "join: key columns of one side differ in length");
    }
    const int dt = kdt[0];
    if (nleft >= (int64_t)1 << 32 || nright >= (int64_t)1 << 32) return fail(RDF_INVALID_ARGUMENT, "join: UInt32 indices cap a side at 2^32-1 rows");
    if (out_left && (out_left->dtype != RDF_U32 || out_right->dtype != RDF_U32)) return fail(RDF_INVALID_ARGUMENT, "join: indices are UInt32");
    const bool swap = join_type == RDF_JOIN_RIGHT;  // probe = right, build = left
    const rdf_array* pk = swap ? right_keys : left_keys;
    const rdf_array* bk = swap ? left_keys : right_keys;
    const int64_t pnc = swap ? right_nchunks : left_nchunks, bnc = swap ? left_nchunks : right_nchunks;
    const int64_t np = swap ? nright : nleft, nb = swap ? nleft : nright;
    const bool pnulls = swap ? rnulls : lnulls, bnulls = swap ? lnulls : rnulls;
    const bool outer = join_type != RDF_JOIN_INNER;
    const bool full = join_type == RDF_JOIN_FULL;
    if (np == 0 && (!full || nb == 0)) {
        *out_rows = 0;
        if (out_left) { out_left->length = out_right->length = 0; out_left->null_count = out_right->null_count = 0; }
        return RDF_OK;
    }
    RDF_TRY(ensure_ready());
    Ctx& ctx = g_ctx;
    arena_begin();
    size_t pin_off = 0, used = 0;
    InputStager in;   // staged order: probe key 0 chunks, probe key 1 chunks, ..., then the build side the same way
    for (int64_t c = 0; c < (int64_t)nkeys * pnc; ++c) in.add(&pk[c]);
    for (int64_t c = 0; c < (int64_t)nkeys * bnc; ++c) in.add(&bk[c]);
    RDF_TRY(in.finish(pin_off, &used));
    pin_off += (used + 255) & ~(size_t)255;
    std::vector<int64_t> prs((size_t)pnc + 1, 0), brs((size_t)bnc + 1, 0);
    for (int64_t c = 0; c < pnc; ++c) prs[(size_t)c + 1] = prs[(size_t)c] + pk[c].length;
    for (int64_t c = 0; c < bnc; ++c) brs[(size_t)c + 1] = brs[(size_t)c] + bk[c].length;
    TableBuilder tb;
    const size_t o_ch = tb.reserve(sizeof(DevChunkCol) * in.dev.size());
    const size_t o_prs = tb.reserve(sizeof(int64_t) * prs.size());
    const size_t o_brs = tb.reserve(sizeof(int64_t) * brs.size());
    RDF_TRY(tb.bind(pin_off));
    memcpy(tb.at<char>(o_ch), in.dev.data(), sizeof(DevChunkCol) * in.dev.size());
    memcpy(tb.at<char>(o_prs), prs.data(), sizeof(int64_t) * prs.size());
    memcpy(tb.at<char>(o_brs), brs.data(), sizeof(int64_t) * brs.size());
    RDF_TRY(tb.alloc());
    RDF_TRY(tb.upload(pin_off));
    pin_off += (tb.size + 255) & ~(size_t)255;

    // build side: key bits + null flags, sorted (NULL keys last)
    SortBuffers sb;
    RDF_TRY(sort_buffers_alloc(nb > 0 ? nb : 1, sb));
    void* p;
    RDF_TRY(arena_alloc(64, &p));
    unsigned long long* d_cnt = (unsigned long long*)p;  // [0] build nulls, [1] append cursor
    HIP_TRY(hipMemsetAsync(d_cnt, 0, 64, ctx.stream));
    KernelTimer kt;
    int kcur = 0, icur = 0;
    uint64_t bkmin = 1, bkmax = 0;   // key range of the non-NULL build keys ([1, 0] = none)
    // Round 5: one key column -> the build side is sorted by key * golden ratio and the table of its distinct keys is laid out by a
    // scan (JoinPlaceArgs) — join_table_kernel's compare-and-swap per distinct key on a random line was 8.6 of the 18.9 ms of a
    // 1e8 x 1e8 join.  Equal keys stay adjacent and in row order (the sort is stable), so the pairs come out as before.
    const bool hashed = nkeys == 1 && ctx.opt_join_table == 2 && nb > 0 && nb < ((int64_t)1 << 31) - 1;
    void* pstats;
    RDF_TRY(arena_alloc(128, &pstats));
    uint64_t* d_bstats = (uint64_t*)pstats;
    uint64_t* d_pstats = d_bstats + 4;
    uint64_t* d_ostats = d_bstats + 8;     // (hashed build) the build keys' own [min, max]
    const uint64_t* pbits[4] = {nullptr, nullptr, nullptr, nullptr};
    const uint64_t* bbits[4] = {nullptr, nullptr, nullptr, nullptr};
    // One column per side: the order-preserving key bits ARE the join key.  Several: the key is a 64-bit hash of the
    // tuple's key bits (candidates are verified column by column in the probe kernels), NULL in any column = NULL key.
    auto side_keys = [&](bool build, int64_t n, int64_t nch, size_t chunk0, const int64_t* d_row_start, bool has_nulls, uint64_t* out_keys,
                         uint8_t* nullflags, uint64_t* d_stats, const uint64_t** bits_out) -> rdf_status {
        if (nkeys > 1 && has_nulls) HIP_TRY(hipMemsetAsync(nullflags, 0, (size_t)n, ctx.stream));
        JoinCombineArgs ca;
        memset(&ca, 0, sizeof ca);
        for (int k = 0; k < nkeys; ++k) {
            SortKeyArgs ka;
            memset(&ka, 0, sizeof ka);
            ka.chunks = tb.dev_at<DevChunkCol>(o_ch) + chunk0 + (size_t)k * (size_t)nch;
            ka.chunk_row_start = d_row_start;
            ka.nchunks = nch;
            ka.n = n;
            ka.nullflags = has_nulls ? nullflags : nullptr;
            ka.dtype = kdt[k];
            if (nkeys == 1) { ka.keys = out_keys; ka.bit_stats = d_stats; if (build && hashed) { ka.hash_mul = 0x9E3779B97F4A7C15ull; ka.raw_stats = d_ostats; } }   // (the key range stays what the probe's early-out compares against; the sort sees the hashed keys' range)
            else {
                void* pb;
                RDF_TRY(arena_alloc((size_t)n * 8 + 8, &pb));
                ka.keys = (uint64_t*)pb;
                ka.null_or = 1;
                bits_out[k] = ca.bits[k] = (const uint64_t*)pb;
            }
            HIP_TRY(launch_sort_keys(ka, ctx.stream));
        }
        if (nkeys > 1) {
            ca.nkeys = nkeys; ca.n = n; ca.nullflags = has_nulls ? nullflags : nullptr; ca.out = out_keys; ca.bit_stats = d_stats;
            HIP_TRY(launch_join_combine(ca, ctx.stream));
        }
        return RDF_OK;
    };
    RDF_TRY(sort_stats_reset(d_bstats));
    RDF_TRY(sort_stats_reset(d_pstats));
    RDF_TRY(sort_stats_reset(d_ostats));
    if (nb > 0) {
        RDF_TRY(side_keys(true, nb, bnc, (size_t)nkeys * (size_t)pnc, tb.dev_at<int64_t>(o_brs), bnulls, sb.keys[0], (uint8_t*)sb.nullflags, d_bstats, bbits));
        if (bnulls) HIP_TRY(launch_count_bytes((const uint8_t*)sb.nullflags, nb, d_cnt, ctx.stream));
        RDF_TRY(radix_sort_rows(sb, nb, nkeys == 1 && !hashed ? dtype_size(dt) : 8, bnulls, &kcur, &icur, d_bstats, pin_off, &bkmin, &bkmax));
    }
    // probe side: key bits + null flags in row order
    void *ppk, *ppn, *pcounts, *poffs;
    RDF_TRY(arena_alloc((size_t)(np > 0 ? np : 1) * 8, &ppk));
    RDF_TRY(arena_alloc((size_t)(np > 0 ? np : 1) + 8, &ppn));
    RDF_TRY(arena_alloc((size_t)(np + 1) * 8, &pcounts));
    RDF_TRY(arena_alloc((size_t)(np + 2 + scan_scratch_words(np)) * 8, &poffs));
    if (np > 0) RDF_TRY(side_keys(false, np, pnc, 0, tb.dev_at<int64_t>(o_prs), pnulls, (uint64_t*)ppk, (uint8_t*)ppn, d_pstats, pbits));
    RDF_TRY(pinned_reserve(pin_off + 256));
    HIP_TRY(hipMemcpyAsync(ctx.pinned + pin_off, d_cnt, 8, hipMemcpyDeviceToHost, ctx.stream));
    if (hashed) HIP_TRY(hipMemcpyAsync(ctx.pinned + pin_off + 8, d_ostats, 16, hipMemcpyDeviceToHost, ctx.stream));
    HIP_TRY(hipStreamSynchronize(ctx.stream));
    unsigned long long bnull_count = 0;
    memcpy(&bnull_count, ctx.pinned + pin_off, 8);
    if (hashed) {
        uint64_t st[2];
        memcpy(st, ctx.pinned + pin_off + 8, 16);
        if (st[0] > st[1]) { bkmin = 1; bkmax = 0; } else { bkmin = st[0]; bkmax = st[1]; }
    }
    const int64_t nrv = nb - (int64_t)bnull_count;

    void* pmatched = nullptr;
    if (full) { RDF_TRY(arena_alloc((size_t)((nb + 31) / 32 + 1) * 4, &pmatched)); HIP_TRY(hipMemsetAsync(pmatched, 0, (size_t)((nb + 31) / 32 + 1) * 4, ctx.stream)); }
    // bucket index over the sorted build keys: about one bucket per build row, on the top bits of (key - min)
    int64_t nbuckets = 1;
    while (nbuckets < nrv && nbuckets < ((int64_t)1 << 24)) nbuckets <<= 1;
    int bshift = 0;
    if (bkmax >= bkmin) while (((bkmax - bkmin) >> bshift) >= (uint64_t)nbuckets) ++bshift;
    void *pbuckets, *pfirst;
    RDF_TRY(arena_alloc((size_t)nbuckets * 8, &pbuckets));
    RDF_TRY(arena_alloc((size_t)(np > 0 ? np : 1) * 4, &pfirst));
    HIP_TRY(hipMemsetAsync(pbuckets, 0, (size_t)nbuckets * 8, ctx.stream));
    // one key column: a table of the distinct build keys at load <= 0.5 instead — one random transaction per probe row, where the
    // bucket index costs three (bucket, sorted keys, build row)
    const bool use_table = nkeys == 1 && nrv > 0 && nb < ((int64_t)1 << 31) - 1 && ctx.opt_join_table;
    void* ptable = nullptr;
    int tbits = 10;
    if (use_table && hashed) {
        // the table holds the DISTINCT build keys: no more of them than rows, and no more than the key range spans — a build side of
        // 1e8 rows over a few dozen dictionary codes gets a table of 1024 slots, not 2e8 whose empty stretches a few lanes would
        // have to write one slot after another (the placement fills the gaps in front of its keys itself)
        int64_t nd = nrv;
        if (bkmax >= bkmin && bkmax - bkmin < (uint64_t)nrv) nd = (int64_t)(bkmax - bkmin) + 1;
        else if (nrv >= 65536) {
            // ... and a few keys spread over a WIDE range (hashed identifiers, 37 of them over 3e6 rows): the sorted build side is
            // read once more to COUNT its distinct keys (equal keys are neighbours; 8 B/row at the read rate, 0.1 ms per 1e8 rows,
            // and a wait) — sized from the rows, the few lanes in front of such keys wrote millions of empty slots one by one
            HIP_TRY(hipMemsetAsync(d_cnt + 3, 0, 8, ctx.stream));
            HIP_TRY(launch_join_distinct(sb.keys[kcur], nrv, (unsigned long long*)(d_cnt + 3), ctx.stream));
            unsigned long long distinct = 0;
            HIP_TRY(hipMemcpyAsync(ctx.pinned + pin_off, d_cnt + 3, 8, hipMemcpyDeviceToHost, ctx.stream));
            HIP_TRY(hipStreamSynchronize(ctx.stream));
            memcpy(&distinct, ctx.pinned + pin_off, 8);
            HIP_TRY(hipMemsetAsync(d_cnt + 3, 0, 8, ctx.stream));          // (the placement's overflow flag lives in the same word)
            if (distinct > 0 && (int64_t)distinct < nd) nd = (int64_t)distinct;
        }
        while (((int64_t)1 << tbits) < 2 * nd) ++tbits;
        const int64_t cap = ((int64_t)1 << tbits) + (1 << 16);      // no wrap-around: the last home slot's cluster runs into the margin
        RDF_TRY(arena_alloc((size_t)cap * 16 + 64, &ptable));      // (not cleared: the placement writes every slot, the empty ones included)
        JoinPlaceArgs pa;
        memset(&pa, 0, sizeof pa);
        pa.rkeys = sb.keys[kcur]; pa.ridx = sb.idx[icur]; pa.nrv = nrv; pa.table = (uint64_t*)ptable; pa.cap = cap; pa.tshift = 64 - tbits;
        pa.ntiles = (nrv + kJoinPlaceTile - 1) / kJoinPlaceTile;
        void* ptiles;
        RDF_TRY(arena_alloc((size_t)pa.ntiles * 16 + 16, &ptiles));
        pa.tiles = (int64_t*)ptiles;
        pa.flags = d_cnt + 3;
        HIP_TRY(launch_join_place(pa, ctx.stream));
        // The table is not cleared: a placement that overflowed (a cluster ran past the margin) leaves the slots behind the last
        // key it placed unwritten, and a probe would walk them.  Look at the flag BEFORE anything probes (one 8-byte copy and a wait:
        // microseconds against the milliseconds of a join this size); the round-3 table takes the call if it is set.
        unsigned long long overflow = 0;
        HIP_TRY(hipMemcpyAsync(ctx.pinned + pin_off, d_cnt + 3, 8, hipMemcpyDeviceToHost, ctx.stream));
        HIP_TRY(hipStreamSynchronize(ctx.stream));
        memcpy(&overflow, ctx.pinned + pin_off, 8);
        if (overflow) {
            kt.stop();
            ctx.opt_join_table = 1;
            const rdf_status st = rdf_equijoin_indices_multi(left_keys, left_nchunks, right_keys, right_nchunks, nkeys, join_type, out_left, out_right, out_rows);
            ctx.opt_join_table = 2;
            return st;
        }
    } else if (use_table) {
        while (((int64_t)1 << tbits) < 2 * nrv) ++tbits;
        RDF_TRY(arena_alloc(((size_t)16 << tbits) + 64, &ptable));
        HIP_TRY(hipMemsetAsync(ptable, 0, (size_t)16 << tbits, ctx.stream));
        JoinTableArgs ta;
        memset(&ta, 0, sizeof ta);
        ta.rkeys = sb.keys[kcur]; ta.ridx = sb.idx[icur]; ta.nrv = nrv; ta.table = (uint64_t*)ptable; ta.tmask = ((uint64_t)1 << tbits) - 1; ta.tshift = 64 - tbits;
        HIP_TRY(launch_join_table(ta, ctx.stream));
    } else if (nrv > 0) {
        JoinBucketArgs ba;
        memset(&ba, 0, sizeof ba);
        ba.rkeys = sb.keys[kcur]; ba.nrv = nrv; ba.buckets = (uint32_t*)pbuckets; ba.kmin = bkmin; ba.bucket_shift = bshift;
        HIP_TRY(launch_join_buckets(ba, ctx.stream));
    }
    JoinProbeArgs ja;
    memset(&ja, 0, sizeof ja);
    ja.buckets = (const uint32_t*)pbuckets;
    ja.kmin = bkmin; ja.kmax = bkmax; ja.bucket_shift = bshift;
    ja.first = (uint32_t*)pfirst;
    if (use_table) { ja.table = (const uint64_t*)ptable; ja.tmask = hashed ? ~0ull : ((uint64_t)1 << tbits) - 1; ja.tshift = 64 - tbits; ja.hashed = hashed ? 1 : 0; }
    ja.nkeys = nkeys;
    for (int k = 0; k < 4; ++k) { ja.pbits[k] = pbits[k]; ja.bbits[k] = bbits[k]; }
    ja.lkeys = (const uint64_t*)ppk;
    ja.lnull = pnulls ? (const uint8_t*)ppn : nullptr;
    ja.rkeys = sb.keys[kcur];
    ja.ridx = sb.idx[icur];
    ja.nl = np;
    ja.nrv = nrv;
    ja.outer = outer;
    ja.counts = (int64_t*)pcounts;
    ja.matched = (uint32_t*)pmatched;
    ja.unmatched = d_cnt + 2;
    HIP_TRY(launch_join_count(ja, ctx.stream));
    HIP_TRY(launch_scan((const int64_t*)pcounts, (int64_t*)poffs, np, (int64_t*)poffs + np + 1, ctx.stream));
    JoinAppendArgs aa;
    memset(&aa, 0, sizeof aa);
    if (full) {
        aa.ridx = sb.idx[icur];
        aa.matched = (const uint32_t*)pmatched;
        aa.nr = nb;
        aa.nrv = nrv;
        aa.cursor = d_cnt + 1;
        aa.count_only = 1;
        HIP_TRY(launch_join_append(aa, ctx.stream));
    }
    HIP_TRY(hipMemcpyAsync(ctx.pinned + pin_off, (int64_t*)poffs + np, 8, hipMemcpyDeviceToHost, ctx.stream));
    HIP_TRY(hipMemcpyAsync(ctx.pinned + pin_off + 8, d_cnt + 1, 24, hipMemcpyDeviceToHost, ctx.stream));
    HIP_TRY(hipStreamSynchronize(ctx.stream));
    int64_t probe_rows = 0;
    unsigned long long appended = 0, unmatched = 0, place_overflow = 0;
    memcpy(&probe_rows, ctx.pinned + pin_off, 8);
    memcpy(&appended, ctx.pinned + pin_off + 8, 8);
    memcpy(&unmatched, ctx.pinned + pin_off + 16, 8);
    memcpy(&place_overflow, ctx.pinned + pin_off + 24, 8);
    if (hashed && place_overflow) {     // a cluster ran past the table's margin (keys that crowd on the last slots): the round-3 table takes the call
        kt.stop();
        ctx.opt_join_table = 1;
        const rdf_status st = rdf_equijoin_indices_multi(left_keys, left_nchunks, right_keys, right_nchunks, nkeys, join_type, out_left, out_right, out_rows);
        ctx.opt_join_table = 2;
        return st;
    }
    const int64_t total = probe_rows + (int64_t)appended;
    *out_rows = total;
    if (!out_left) { kt.stop(); return RDF_OK; }
    if (out_left->capacity < total || out_right->capacity < total) return fail(RDF_MEMORY_ERROR, "join: output capacity too small (need %lld rows)", (long long)total);
    rdf_out* out_probe = swap ? out_right : out_left;
    rdf_out* out_build = swap ? out_left : out_right;
    if ((outer && !out_build->validity) || (full && !out_probe->validity)) return fail(RDF_INVALID_ARGUMENT, "output validity buffer required");
    // device outputs
    const size_t words = (size_t)((total + 31) / 32 + 2);
    void *dop, *dob, *dvp, *dvb;
    RDF_TRY(arena_alloc((size_t)(total + 1) * 4, &dop));
    RDF_TRY(arena_alloc((size_t)(total + 1) * 4, &dob));
    RDF_TRY(arena_alloc(words * 4, &dvp));
    RDF_TRY(arena_alloc(words * 4, &dvb));
    HIP_TRY(hipMemsetAsync(dvp, 0xFF, words * 4, ctx.stream));
    HIP_TRY(hipMemsetAsync(dvb, 0xFF, words * 4, ctx.stream));
    ja.offsets = (const int64_t*)poffs;
    ja.out_probe = (uint32_t*)dop;
    ja.out_build = (uint32_t*)dob;
    ja.out_build_validity = (uint32_t*)dvb;
    ja.matched = nullptr;
    HIP_TRY(launch_join_write(ja, ctx.stream));
    if (full) {
        unsigned long long start = (unsigned long long)probe_rows;
        HIP_TRY(hipMemcpyAsync(d_cnt + 1, &start, 8, hipMemcpyHostToDevice, ctx.stream));
        HIP_TRY(hipStreamSynchronize(ctx.stream));
        aa.count_only = 0;
        aa.out_probe = (uint32_t*)dop;
        aa.out_build = (uint32_t*)dob;
        aa.out_probe_validity = (uint32_t*)dvp;
        HIP_TRY(launch_join_append(aa, ctx.stream));
    }
    kt.stop();
    ctx.last_kernel = "join_write_kernel";
    const hipMemcpyKind kind = mem == RDF_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    if (total > 0) {
        HIP_TRY(hipMemcpyAsync(out_probe->values, dop, (size_t)total * 4, kind, ctx.stream));
        HIP_TRY(hipMemcpyAsync(out_build->values, dob, (size_t)total * 4, kind, ctx.stream));
        if (out_probe->validity) HIP_TRY(hipMemcpyAsync(out_probe->validity, dvp, (size_t)((total + 7) / 8), kind, ctx.stream));
        if (out_build->validity) HIP_TRY(hipMemcpyAsync(out_build->validity, dvb, (size_t)((total + 7) / 8), kind, ctx.stream));
    }
    // null counts: build-side NULLs = probe rows without a partner; probe-side NULLs = appended rows
    HIP_TRY(hipStreamSynchronize(ctx.stream));
    out_probe->length = out_build->length = total;
    out_probe->null_count = (int64_t)appended;
    out_build->null_count = (int64_t)unmatched;
    return RDF_OK;
}

// ---------------------------------------------------------------- group-by

// Tail of both partitioned GROUP BY paths: read back {special sums, counts, flags, cursor}, append the two special
// groups (the key whose hash is the LDS free marker; the NULL key) and copy the dense results to the caller.
static rdf_status groupby_finish_partitioned(void* pspec, void* d_keys, void* d_sums, void* d_counts, int kdt, int64_t max_groups, int32_t mem,
                                             rdf_out* out_keys, rdf_out* out_sums, rdf_out* out_counts, size_t pin_off, bool single_pass) {
    Ctx& ctx = g_ctx;
    struct { void* out_keys; void* out_sums; void* out_counts; } ga = {d_keys, d_sums, d_counts};
        // specials + cursor + flags
        RDF_TRY(pinned_reserve(pin_off + 256));
        HIP_TRY(hipMemcpyAsync(ctx.pinned + pin_off, pspec, 128, hipMemcpyDeviceToHost, ctx.stream));
        HIP_TRY(hipStreamSynchronize(ctx.stream));
        unsigned long long hs[4];
        unsigned int hf[8];
        memcpy(hs, ctx.pinned + pin_off, 32);
        memcpy(hf, ctx.pinned + pin_off + 32, 32);
        const int64_t ng_main = hf[2];
        if ((hf[4] & 4u) || ng_main > max_groups) return fail(RDF_MEMORY_ERROR, "groupby: more than max_groups (%lld) distinct keys", (long long)max_groups);
        // append the two special groups on the host side of the copy
        const size_t kes2 = (size_t)dtype_size(kdt);
        int64_t ng = ng_main;
        int64_t null_idx = -1;
        auto put = [&](uint64_t key, unsigned long long sum, unsigned long long cnt) -> rdf_status {
            HIP_TRY(hipMemcpyAsync((char*)ga.out_keys + (size_t)ng * kes2, &key, kes2, hipMemcpyHostToDevice, ctx.stream));
            HIP_TRY(hipMemcpyAsync((char*)ga.out_sums + (size_t)ng * 8, &sum, 8, hipMemcpyHostToDevice, ctx.stream));
            HIP_TRY(hipMemcpyAsync((char*)ga.out_counts + (size_t)ng * 8, &cnt, 8, hipMemcpyHostToDevice, ctx.stream));
            HIP_TRY(hipStreamSynchronize(ctx.stream));
            ++ng;
            return RDF_OK;
        };
        if (hf[0]) {  // the key whose hash equals the LDS free marker: unmix on the host
            uint64_t z = ~0ull;
            if (single_pass) { z ^= z >> 32; z *= 0xF1DE83E19937733Dull; z ^= z >> 32; }   // inverse of gb_hash
            else { z ^= z >> 31; z ^= z >> 62; z *= 0x319642b2d24d8ec3ull; z ^= z >> 27; z ^= z >> 54; z *= 0x96de1b173f119089ull; z ^= z >> 30; z ^= z >> 60; }   // inverse of mix64
            RDF_TRY(put(z, hs[0], hs[2]));
        }
        if (hf[1]) { null_idx = ng; RDF_TRY(put(0, hs[1], hs[3])); }
        if (ng > out_keys->capacity || ng > out_sums->capacity || ng > out_counts->capacity)
            return fail(RDF_MEMORY_ERROR, "groupby: more than max_groups (%lld) distinct keys", (long long)max_groups);
        const hipMemcpyKind kind = mem == RDF_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
        if (ng > 0) {
            HIP_TRY(hipMemcpyAsync(out_keys->values, ga.out_keys, (size_t)ng * kes2, kind, ctx.stream));
            HIP_TRY(hipMemcpyAsync(out_sums->values, ga.out_sums, (size_t)ng * 8, kind, ctx.stream));
            HIP_TRY(hipMemcpyAsync(out_counts->values, ga.out_counts, (size_t)ng * 8, kind, ctx.stream));
        }
        rdf_out* outs3[3] = {out_keys, out_sums, out_counts};
        for (rdf_out* o : outs3)
            if (o->validity && ng > 0) {
                if (mem == RDF_MEM_HOST) memset(o->validity, 0xFF, (size_t)((ng + 7) / 8));
                else HIP_TRY(hipMemsetAsync(o->validity, 0xFF, (size_t)((ng + 7) / 8), ctx.stream));
            }
        if (null_idx >= 0) {
            const uint8_t byte = (uint8_t)~(1u << (null_idx & 7));
            if (mem == RDF_MEM_HOST) out_keys->validity[null_idx >> 3] &= byte;
            else HIP_TRY(hipMemcpyAsync(out_keys->validity + (null_idx >> 3), &byte, 1, hipMemcpyHostToDevice, ctx.stream));
        }
        HIP_TRY(hipStreamSynchronize(ctx.stream));
        out_keys->length = out_sums->length = out_counts->length = ng;
        out_keys->null_count = null_idx >= 0 ? 1 : 0;
        out_sums->null_count = out_counts->null_count = 0;
    return RDF_OK;
}

}  // extern "C"
namespace {
// The first-generation hash GROUP BY (sum / count of one key column): histogram -> scatter (with the combining variant for
// skewed keys) -> aggregate, the radix-sort variant and the single HBM table.  Since round 2 the default is rdf_groupby_agg
// (rdf_capi_groupby.inc); this stays as its fallback for heavily skewed keys and as the A/B baseline (gb_partition = 1 / 2).
rdf_status legacy_groupby_sum(const rdf_array* keys, const rdf_array* values, int64_t nchunks, int64_t max_groups,
                              rdf_out* out_keys, rdf_out* out_sums, rdf_out* out_counts) {
    if (nchunks < 1 || !keys) return fail(RDF_INVALID_ARGUMENT, "groupby: a column has at least one chunk");
    if (!out_keys || !out_sums || !out_counts) return fail(RDF_INVALID_ARGUMENT, "groupby: null output");
    if (max_groups < 1) return fail(RDF_INVALID_ARGUMENT, "groupby: max_groups must be positive");
    int32_t mem = -1;
    RDF_TRY(check_mem(keys, nchunks, &mem));
    if (values) RDF_TRY(check_mem(values, nchunks, &mem));
    RDF_TRY(check_out_mem(out_keys, 1, mem));
    RDF_TRY(check_out_mem(out_sums, 1, mem));
    RDF_TRY(check_out_mem(out_counts, 1, mem));
    const int kdt = keys[0].dtype;
    if (!(kdt >= RDF_I8 && kdt <= RDF_U64)) return fail(RDF_INVALID_ARGUMENT, "groupby: integer key column required");
    const int vdt = values ? values[0].dtype : -1;
    if (values && !is_numeric(vdt)) return fail(RDF_INVALID_ARGUMENT, "groupby: numeric value column required");
    const int sdt = values && is_float(vdt) ? RDF_F64 : RDF_I64;
    if (out_keys->dtype != kdt || out_sums->dtype != sdt || out_counts->dtype != RDF_I64)
        return fail(RDF_INVALID_ARGUMENT, "groupby: outputs must be (key dtype, %s, Int64)", sdt == RDF_F64 ? "Float64" : "Int64");
    bool null_keys = false;
    std::vector<int64_t> clen((size_t)nchunks), tile_start((size_t)nchunks + 1, 0);
    for (int64_t c = 0; c < nchunks; ++c) {
        if (keys[c].dtype != kdt || (values && values[c].dtype != vdt)) return fail(RDF_INVALID_ARGUMENT, "groupby: chunks differ in dtype");
        if (values && values[c].length != keys[c].length) return fail(RDF_COMPUTE_ERROR, "groupby: key and value chunks differ in length");
        null_keys |= keys[c].validity != nullptr;
        clen[(size_t)c] = keys[c].length;
        tile_start[(size_t)c + 1] = tile_start[(size_t)c] + (keys[c].length + kEvalTile - 1) / kEvalTile;
    }
    if (null_keys && !out_keys->validity) return fail(RDF_INVALID_ARGUMENT, "output validity buffer required");
    int64_t nrows_total = 0;
    for (int64_t c = 0; c < nchunks; ++c) nrows_total += keys[c].length;
    const int64_t cap_needed = std::min<int64_t>(max_groups + 2, nrows_total + 2);    // the contract rdf_groupby_agg states
    if (out_keys->capacity < cap_needed || out_sums->capacity < cap_needed || out_counts->capacity < cap_needed)
        return fail(RDF_MEMORY_ERROR, "output capacity too small (need max_groups + 2)");
    RDF_TRY(ensure_ready());
    Ctx& ctx = g_ctx;
    arena_begin();
    size_t pin_off = 0, used = 0;
    InputStager in;
    for (int64_t c = 0; c < nchunks; ++c) in.add(&keys[c]);
    if (values) for (int64_t c = 0; c < nchunks; ++c) in.add(&values[c]);
    RDF_TRY(in.finish(pin_off, &used));
    pin_off += (used + 255) & ~(size_t)255;
    TableBuilder tb;
    const size_t o_k = tb.reserve(sizeof(DevChunkCol) * (size_t)nchunks);
    const size_t o_v = tb.reserve(sizeof(DevChunkCol) * (size_t)nchunks);
    const size_t o_ts = tb.reserve(sizeof(int64_t) * tile_start.size());
    const size_t o_len = tb.reserve(sizeof(int64_t) * clen.size());
    std::vector<int64_t> row_start((size_t)nchunks + 1, 0);
    for (int64_t c = 0; c < nchunks; ++c) row_start[(size_t)c + 1] = row_start[(size_t)c] + clen[(size_t)c];
    const size_t o_rs = tb.reserve(sizeof(int64_t) * row_start.size());
    RDF_TRY(tb.bind(pin_off));
    memcpy(tb.at<char>(o_k), in.dev.data(), sizeof(DevChunkCol) * (size_t)nchunks);
    if (values) memcpy(tb.at<char>(o_v), in.dev.data() + nchunks, sizeof(DevChunkCol) * (size_t)nchunks);
    memcpy(tb.at<char>(o_ts), tile_start.data(), sizeof(int64_t) * tile_start.size());
    memcpy(tb.at<char>(o_len), clen.data(), sizeof(int64_t) * clen.size());
    memcpy(tb.at<char>(o_rs), row_start.data(), sizeof(int64_t) * row_start.size());
    RDF_TRY(tb.alloc());
    RDF_TRY(tb.upload(pin_off));
    pin_off += (tb.size + 255) & ~(size_t)255;

    // High cardinality: radix-partition on the hashed key, aggregate each partition in LDS (no HBM atomics per row).
    bool value_nulls = false;
    if (values) for (int64_t c = 0; c < nchunks; ++c) value_nulls |= values[c].validity != nullptr;
    const int64_t nrows = row_start[(size_t)nchunks];
    if (ctx.opt_gb_partition && max_groups > 1024 && max_groups <= kGbMaxGroups && nrows > 0 && ctx.opt_gb_partition != 2) {
        // single scatter pass on 9 hash bits, then one LDS table per partition
        constexpr int P = 1 << kGbPartBits;
        const int64_t ntiles = tile_start[(size_t)nchunks];
        const int64_t nsuper = (ntiles + kGbSuper / kEvalTile - 1) / (kGbSuper / kEvalTile);
        int nb = (int)std::min<int64_t>(nsuper, (int64_t)eval_grid_limit() / 4);   // 2 resident blocks of 512 threads per CU
        if (nb < 1) nb = 1;
        void *precs, *hist0, *hist1, *ptmp, *pspec;
        RDF_TRY(arena_alloc((size_t)nrows * 16 + 64, &precs));
        RDF_TRY(arena_alloc((size_t)((int64_t)P * nb + 1) * 8, &hist0));
        RDF_TRY(arena_alloc((size_t)((int64_t)P * nb + 1 + scan_scratch_words((int64_t)P * nb)) * 8, &hist1));
        const int64_t cap_out = max_groups + 2;
        RDF_TRY(arena_alloc((size_t)cap_out * 24 + 64, &ptmp));
        RDF_TRY(arena_alloc(128, &pspec));
        HIP_TRY(hipMemsetAsync(pspec, 0, 128, ctx.stream));
        unsigned long long* sp_sums = (unsigned long long*)pspec;         // [2]
        unsigned long long* sp_counts = sp_sums + 2;                      // [2]
        unsigned int* sp_flag = (unsigned int*)(sp_counts + 2);           // [2]
        unsigned int* d_cursor = sp_flag + 2;
        uint32_t* d_flags2 = (uint32_t*)(sp_flag + 4);
        GbPartArgs pa;
        memset(&pa, 0, sizeof pa);
        pa.keys = tb.dev_at<DevChunkCol>(o_k);
        pa.values = tb.dev_at<DevChunkCol>(o_v);
        pa.chunk_tile_start = tb.dev_at<int64_t>(o_ts);
        pa.chunk_len = tb.dev_at<int64_t>(o_len);
        pa.nchunks = nchunks;
        pa.ntiles = ntiles;
        if (nchunks == 1) { pa.key0 = in.dev[0]; if (values) pa.val0 = in.dev[1]; pa.len0 = clen[0]; }
        pa.ablate_stores = (ctx.opt_gb_debug == 2 || ctx.opt_gb_debug == 7 || ctx.opt_gb_debug == 8) ? ctx.opt_gb_debug : 0;
        pa.key_dtype = kdt;
        pa.value_dtype = vdt;
        pa.hist = (int64_t*)hist0;
        pa.recs = (uint64_t*)precs;
        pa.special_sums = sp_sums;
        pa.special_counts = sp_counts;
        pa.special = sp_flag;
        KernelTimer kt;
        HIP_TRY(launch_gb_hist(pa, nb, ctx.stream));
        HIP_TRY(launch_scan((const int64_t*)hist0, (int64_t*)hist1, (int64_t)P * nb, (int64_t*)hist1 + (int64_t)P * nb + 1, ctx.stream));
        pa.hist = (int64_t*)hist1;
        // skewed key distribution (one partition far above the average)?  Then equal keys are combined inside each
        // super-tile before they are scattered, so a hot key does not serialise one block's LDS atomics.
        unsigned int* d_skew = d_cursor + 1;
        HIP_TRY(launch_gb_skew((const int64_t*)hist1, nb, d_skew, ctx.stream));
        RDF_TRY(pinned_reserve(pin_off + 64));
        HIP_TRY(hipMemcpyAsync(ctx.pinned + pin_off, d_skew, 4, hipMemcpyDeviceToHost, ctx.stream));
        HIP_TRY(hipStreamSynchronize(ctx.stream));
        unsigned int skew = 0;
        memcpy(&skew, ctx.pinned + pin_off, 4);
        if (ctx.opt_gb_debug == 3) skew = 1;   // tests: force the combining variant
        void* pemit = nullptr;
        if (skew) { RDF_TRY(arena_alloc((size_t)((int64_t)P * nb + 1) * 8, &pemit)); pa.emitted = (int64_t*)pemit; }
        HIP_TRY(launch_gb_scatter(pa, nb, skew != 0, ctx.stream));
        GbAggArgs ga;
        memset(&ga, 0, sizeof ga);
        ga.recs = (const uint64_t*)precs;
        ga.scan = (const int64_t*)hist1;
        ga.nblocks = nb;
        ga.emitted = (const int64_t*)pemit;
        ga.is_f64 = sdt == RDF_F64;
        ga.has_values = values != nullptr;
        ga.key_dtype = kdt;
        char* tmp = (char*)ptmp;
        ga.out_keys = tmp;
        ga.out_sums = tmp + (size_t)cap_out * 8;
        ga.out_counts = (int64_t*)(tmp + (size_t)cap_out * 16);
        ga.cursor = d_cursor;
        ga.flags = d_flags2;
        ga.max_out = max_groups;
        ga.ablate_lds = (ctx.opt_gb_debug == 1 || (ctx.opt_gb_debug >= 4 && ctx.opt_gb_debug <= 6)) ? ctx.opt_gb_debug : 0;
        HIP_TRY(launch_gb_aggregate(ga, ctx.stream));
        kt.stop();
        ctx.last_kernel = skew ? "gb_aggregate_kernel(combined)" : "gb_aggregate_kernel";
        RDF_TRY(groupby_finish_partitioned(pspec, ga.out_keys, ga.out_sums, ga.out_counts, kdt, max_groups, mem, out_keys, out_sums, out_counts, pin_off, true));
        return RDF_OK;
    }
    if (ctx.opt_gb_partition && max_groups > 1024 && !value_nulls && nrows > 0) {
        const int npass = max_groups > 131072 ? 2 : 1;
        const int64_t stiles = (nrows + kSortTile - 1) / kSortTile;
        const int64_t sgrid = sort_grid(stiles);
        void *ph[4], *hist0, *hist1, *ptmp, *pspec;
        for (int i = 0; i < 4; ++i) RDF_TRY(arena_alloc((size_t)nrows * 8, &ph[i]));
        RDF_TRY(arena_alloc((size_t)(256 * sgrid + 1) * 8, &hist0));
        RDF_TRY(arena_alloc((size_t)(256 * sgrid + 1 + scan_scratch_words(256 * sgrid)) * 8, &hist1));
        const int64_t cap_out = max_groups + 2;
        RDF_TRY(arena_alloc((size_t)cap_out * 24 + 64, &ptmp));
        RDF_TRY(arena_alloc(128, &pspec));
        HIP_TRY(hipMemsetAsync(pspec, 0, 128, ctx.stream));
        unsigned long long* sp_sums = (unsigned long long*)pspec;         // [2]
        unsigned long long* sp_counts = sp_sums + 2;                      // [2]
        unsigned int* sp_flag = (unsigned int*)(sp_counts + 2);           // [2]
        unsigned int* d_cursor = sp_flag + 2;
        uint32_t* d_flags2 = (uint32_t*)(sp_flag + 4);
        GroupPrepArgs pa;
        memset(&pa, 0, sizeof pa);
        pa.keys = tb.dev_at<DevChunkCol>(o_k);
        pa.values = tb.dev_at<DevChunkCol>(o_v);
        pa.chunk_tile_start = tb.dev_at<int64_t>(o_ts);
        pa.chunk_len = tb.dev_at<int64_t>(o_len);
        pa.chunk_row_start = tb.dev_at<int64_t>(o_rs);
        pa.nchunks = nchunks;
        pa.ntiles = tile_start[(size_t)nchunks];
        pa.key_dtype = kdt;
        pa.value_dtype = vdt;
        pa.hkeys = (uint64_t*)ph[0];
        pa.vals = (uint64_t*)ph[1];
        pa.special_sums = sp_sums;
        pa.special_counts = sp_counts;
        pa.special = sp_flag;
        KernelTimer kt;
        HIP_TRY(launch_groupby_prepare(pa, ctx.stream));
        int cur = 0;  // (ph[cur], ph[cur + 1]) hold the current streams
        for (int p = 0; p < npass; ++p) {
            SortPassArgs sa;
            memset(&sa, 0, sizeof sa);
            sa.keys_in = (const uint64_t*)ph[cur];
            sa.pay_in = (const uint64_t*)ph[cur + 1];
            sa.keys_out = (uint64_t*)ph[cur ^ 2];
            sa.pay_out = (uint64_t*)ph[(cur ^ 2) + 1];
            sa.hist = (int64_t*)hist0;
            sa.n = nrows;
            sa.ntiles = stiles;
            sa.shift = 64 - 8 * npass + 8 * p;
            HIP_TRY(launch_sort_hist64(sa, ctx.stream));
            HIP_TRY(launch_scan((const int64_t*)hist0, (int64_t*)hist1, 256 * sgrid, (int64_t*)hist1 + 256 * sgrid + 1, ctx.stream));
            sa.hist = (int64_t*)hist1;
            HIP_TRY(launch_sort_scatter64(sa, ctx.stream));
            cur ^= 2;
        }
        GroupAggArgs ga;
        memset(&ga, 0, sizeof ga);
        ga.hkeys = (const uint64_t*)ph[cur];
        ga.vals = (const uint64_t*)ph[cur + 1];
        ga.n = nrows;
        ga.part_bits = 8 * npass;
        ga.is_f64 = sdt == RDF_F64;
        ga.has_values = values != nullptr;
        ga.key_dtype = kdt;
        char* tmp = (char*)ptmp;
        ga.out_keys = tmp;
        ga.out_sums = tmp + (size_t)cap_out * 8;
        ga.out_counts = (int64_t*)(tmp + (size_t)cap_out * 16);
        ga.cursor = d_cursor;
        ga.flags = d_flags2;
        ga.max_out = max_groups;
        HIP_TRY(launch_groupby_partitions(ga, ctx.stream));
        kt.stop();
        ctx.last_kernel = "groupby_partitions_kernel";
        RDF_TRY(groupby_finish_partitioned(pspec, ga.out_keys, ga.out_sums, ga.out_counts, kdt, max_groups, mem, out_keys, out_sums, out_counts, pin_off, false));
        return RDF_OK;
    }

    int64_t capacity = 1024;
    while (capacity < 2 * max_groups) capacity <<= 1;
    void* p = nullptr;
    const size_t tab_bytes = sizeof(uint64_t) * (size_t)(3 * capacity + 4) + 64;
    RDF_TRY(arena_alloc(tab_bytes, &p));
    GroupTable t;
    t.keys = (unsigned long long*)p;
    t.sums = t.keys + capacity;
    t.counts = t.sums + capacity + 2;
    t.special = (unsigned int*)(t.counts + capacity + 2);
    t.ngroups = t.special + 2;
    t.flags = (uint32_t*)(t.special + 4);
    unsigned int* cursor = t.special + 6;
    t.capacity = capacity;
    // keys <- i64::MIN pattern (0x80 00 .. per 8 bytes): fill via a 64-bit pattern memset
    // every slot key <- i64::MIN (the free marker): the fill kernel with a span of 1 writes the constant
    HIP_TRY(launch_fill_i64((int64_t*)t.keys, capacity, 0, 0, 0, INT64_MIN, INT64_MIN + 1, ctx.stream));
    HIP_TRY(hipMemsetAsync(t.sums, 0, sizeof(uint64_t) * (size_t)(2 * capacity + 4) + 64, ctx.stream));

    GroupByArgs ga;
    memset(&ga, 0, sizeof ga);
    ga.keys = tb.dev_at<DevChunkCol>(o_k);
    ga.values = tb.dev_at<DevChunkCol>(o_v);
    ga.chunk_tile_start = tb.dev_at<int64_t>(o_ts);
    ga.chunk_len = tb.dev_at<int64_t>(o_len);
    ga.nchunks = nchunks;
    ga.ntiles = tile_start[(size_t)nchunks];
    ga.key_dtype = kdt;
    ga.value_dtype = vdt;
    ga.t = t;
    ga.max_groups = max_groups;
    {
        KernelTimer kt;
        HIP_TRY(launch_groupby_build(ga, ctx.stream));
        kt.stop();
    }
    // group count + flags
    RDF_TRY(pinned_reserve(pin_off + 64));
    unsigned int pin[8];  // copied out: a later pinned_reserve may move the staging buffer
    HIP_TRY(hipMemcpyAsync(ctx.pinned + pin_off, t.special, 32, hipMemcpyDeviceToHost, ctx.stream));
    HIP_TRY(hipStreamSynchronize(ctx.stream));
    memcpy(pin, ctx.pinned + pin_off, 32);
    const int64_t ngroups = (int64_t)pin[2] + (pin[0] ? 1 : 0) + (pin[1] ? 1 : 0);
    if ((pin[4] & 4u) || (int64_t)pin[2] > max_groups) return fail(RDF_MEMORY_ERROR, "groupby: more than max_groups (%lld) distinct keys", (long long)max_groups);
    pin_off += 256;

    // outputs
    const size_t kes = (size_t)dtype_size(kdt);
    Region outr;
    int ik = -1, ikv = -1, is = -1, ic = -1;
    GroupEmitArgs ea;
    memset(&ea, 0, sizeof ea);
    ea.t = t;
    ea.cursor = cursor;
    ea.key_dtype = kdt;
    if (mem == RDF_MEM_HOST) {
        ik = outr.add(out_keys->values, (size_t)ngroups * kes);
        if (out_keys->validity) ikv = outr.add(out_keys->validity, (size_t)((ngroups + 7) / 8));
        is = outr.add(out_sums->values, (size_t)ngroups * 8);
        ic = outr.add(out_counts->values, (size_t)ngroups * 8);
        RDF_TRY(outr.layout());
        ea.out_keys = outr.ptr(ik);
        ea.out_keys_validity = ikv >= 0 ? (uint8_t*)outr.ptr(ikv) : nullptr;
        ea.out_sums = outr.ptr(is);
        ea.out_counts = (int64_t*)outr.ptr(ic);
    } else {
        ea.out_keys = out_keys->values;
        ea.out_keys_validity = out_keys->validity;
        ea.out_sums = out_sums->values;
        ea.out_counts = (int64_t*)out_counts->values;
    }
    if (ea.out_keys_validity) HIP_TRY(hipMemsetAsync(ea.out_keys_validity, 0, (size_t)((ngroups + 63) / 64 * 8), ctx.stream));
    if (ngroups > 0) HIP_TRY(launch_groupby_emit(ea, ctx.stream));
    if (mem == RDF_MEM_HOST) {
        RDF_TRY(pinned_reserve(pin_off + outr.small_bytes + 256));
        RDF_TRY(outr.download(pin_off));
    } else HIP_TRY(hipStreamSynchronize(ctx.stream));
    out_keys->length = out_sums->length = out_counts->length = ngroups;
    out_keys->null_count = pin[1] ? 1 : 0;
    out_sums->null_count = out_counts->null_count = 0;
    if (out_sums->validity && ngroups > 0) {
        if (mem == RDF_MEM_HOST) memset(out_sums->validity, 0xFF, (size_t)((ngroups + 7) / 8));
        else HIP_TRY(hipMemsetAsync(out_sums->validity, 0xFF, (size_t)((ngroups + 7) / 8), ctx.stream));
    }
    if (out_counts->validity && ngroups > 0) {
        if (mem == RDF_MEM_HOST) memset(out_counts->validity, 0xFF, (size_t)((ngroups + 7) / 8));
        else HIP_TRY(hipMemsetAsync(out_counts->validity, 0xFF, (size_t)((ngroups + 7) / 8), ctx.stream));
    }
    if (mem == RDF_MEM_DEVICE) HIP_TRY(hipStreamSynchronize(ctx.stream));
    return RDF_OK;
}
}  // namespace

#include "rdf_capi_groupby.inc"
#include "rdf_capi_frame.inc"
#include "rdf_capi_stream.inc"
#include "rdf_capi_comm.inc"

extern "C" {

// ---------------------------------------------------------------- synthetic data / timing

rdf_status rdf_fill_uniform_f64(double* dev_ptr, int64_t n, uint64_t seed, uint64_t column_id, int64_t first_row, double lo, double hi) {
    RDF_TRY(ensure_ready());
    if (n > 0) HIP_TRY(launch_fill_f64(dev_ptr, n, seed, column_id, first_row, lo, hi, g_ctx.stream));
    HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    return RDF_OK;
}
rdf_status rdf_fill_uniform_i64(int64_t* dev_ptr, int64_t n, uint64_t seed, uint64_t column_id, int64_t first_row, int64_t lo, int64_t hi) {
    if (hi <= lo) return fail(RDF_INVALID_ARGUMENT, "fill_uniform_i64: hi must exceed lo");
    RDF_TRY(ensure_ready());
    if (n > 0) HIP_TRY(launch_fill_i64(dev_ptr, n, seed, column_id, first_row, lo, hi, g_ctx.stream));
    HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    return RDF_OK;
}
rdf_status rdf_fill_validity(uint8_t* dev_ptr, int64_t nbits, uint64_t seed, uint64_t column_id, int64_t first_row, double null_fraction) {
    RDF_TRY(ensure_ready());
    if (nbits > 0) HIP_TRY(launch_fill_validity(dev_ptr, nbits, seed, column_id, first_row, null_fraction, g_ctx.stream));
    HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    return RDF_OK;
}

rdf_status rdf_set_option(const char* name, int64_t value) {
    if (!name) return fail(RDF_INVALID_ARGUMENT, "null option name");
    if (strcmp(name, "spec") == 0) g_ctx.opt_spec = value != 0;
    else if (strcmp(name, "fast_filter") == 0) g_ctx.opt_fast_filter = value != 0;
    else if (strcmp(name, "vec_bitmap") == 0) g_ctx.opt_vec_bitmap = value != 0;
    else if (strcmp(name, "gb_partition") == 0) g_ctx.opt_gb_partition = (int)value;
    else if (strcmp(name, "gb_debug") == 0) g_ctx.opt_gb_debug = (int)value;
    else if (strcmp(name, "filter_tile") == 0) g_ctx.opt_filter_tile = (int)value;
    else if (strcmp(name, "filter_one") == 0) g_ctx.opt_filter_one = value != 0;
    else if (strcmp(name, "take_rows") == 0) g_ctx.opt_take_rows = (int)value;
    else if (strcmp(name, "sort_gen") == 0) g_ctx.opt_sort_gen = (int)value;
    else if (strcmp(name, "sort_msd") == 0) g_ctx.opt_sort_msd = (int)value;
    else if (strcmp(name, "sort_pipe") == 0) g_ctx.opt_sort_pipe = value != 0;
    else if (strcmp(name, "sort_sample") == 0) g_ctx.opt_sort_sample = (int)value;
    else if (strcmp(name, "stream_table_kernel") == 0) g_ctx.opt_stream_table_kernel = value != 0;
    else if (strcmp(name, "join_table") == 0) g_ctx.opt_join_table = value < 0 || value > 2 ? 2 : (int)value;
    else if (strcmp(name, "jit") == 0) g_ctx.opt_jit = (int)value;
    else if (strcmp(name, "gb_skew_plan") == 0) g_ctx.opt_gb_skew_plan = (int)value;
    else if (strcmp(name, "gb_compact") == 0) g_ctx.opt_gb_compact = (int)value;
    else if (strcmp(name, "gb_bucket") == 0) g_ctx.opt_gb_bucket = (int)value;
    else if (strcmp(name, "gb_hot") == 0) g_ctx.opt_gb_hot = (int)value;
    else if (strcmp(name, "spec_blocks_per_cu") == 0) g_ctx.opt_spec_blocks = (int)value;
    else if (strcmp(name, "spec_tile_rot") == 0) g_ctx.opt_spec_tile_rot = value < 0 ? -1 : (int)value;
    else if (strcmp(name, "spec_xcd_swz") == 0) g_ctx.opt_spec_xcd_swz = value < 0 ? -1 : value != 0;
    else if (strcmp(name, "spec_grid_adj") == 0) g_ctx.opt_spec_grid_adj = (int)value;
    else if (strcmp(name, "gspec_blocks_per_cu") == 0) g_ctx.opt_gspec_blocks = (int)value;
    else if (strcmp(name, "filter_gen") == 0) g_ctx.opt_filter_gen = (int)value;
    else if (strcmp(name, "filter_fused") == 0) g_ctx.opt_filter_fused = (int)value;
    else if (strcmp(name, "filter_block") == 0) g_ctx.opt_filter_block = value == 3 ? 3 : value != 0;      // (3: tests — the scanner wave stays idle, every wait must time out)
    else if (strcmp(name, "interp_lean") == 0) g_ctx.opt_interp_lean = value == 2 ? 2 : value != 0;   // (2: the lean kernel with one tile per trip of its step loop — the A/B of its two-tile form)
    else if (strcmp(name, "sort_super") == 0) {
        if (value > 100) g_ctx.opt_sort_super_force = value > 164 ? 64 : (int)value - 100;      // (tests: K tiles per ticket on inputs of any size)
        else { g_ctx.opt_sort_super_force = 0; g_ctx.opt_sort_super = value < 1 ? 1 : value > 64 ? 64 : (int)value; }
    }
    else if (strcmp(name, "filter_owned") == 0) g_ctx.opt_filter_owned = value == 2 ? 2 : value != 0;      // (2: tests — whatever the number and lengths of the batches)
    else if (strcmp(name, "filter_ends") == 0) g_ctx.opt_filter_ends = value != 0;
    else if (strcmp(name, "filter_mixed") == 0) g_ctx.opt_filter_mixed = value == 2 ? 2 : value != 0;
    else if (strcmp(name, "filter_short") == 0) g_ctx.opt_filter_short = value != 0;
    else if (strcmp(name, "filter_block_rows") == 0) g_ctx.opt_filter_block_rows = value < 1 ? 1 : (int)value;
    else if (strcmp(name, "filter_lookback") == 0) g_ctx.opt_filter_lookback = value == 1 ? 1 : value == 2 ? 2 : 3;
    else if (strcmp(name, "comm_max_bytes") == 0) g_ctx.opt_comm_max_bytes = value;
    else if (strcmp(name, "stream_slab_bytes") == 0) g_ctx.opt_stream_slab = value;
    else return fail(RDF_INVALID_ARGUMENT, "unknown option %s", name);
    return RDF_OK;
}
int32_t rdf_spec_catalog_size(void) { return spec_catalog_size(); }
const char* rdf_jit_status(void) {
    thread_local std::string line;
    line = jit_status();
    return line.c_str();
}
const char* rdf_last_kernel(void) { return g_ctx.last_kernel.c_str(); }

// What bare streaming kernels reach on this device (rdf_probe.hip): the denominators measurements are held against.
rdf_status rdf_probe_stream(int32_t kind, const void* a, void* b, void* c, int64_t bytes, int32_t reps, double* best_gbps, char* shape, int32_t shape_len) {
    if (kind < 0 || kind > 2 || !a || (kind >= 1 && !b) || (kind == 2 && !c) || bytes < 16 || !best_gbps) return fail(RDF_INVALID_ARGUMENT, "probe_stream: kind 0..2 with its device buffers, bytes >= 16");
    RDF_TRY(ensure_ready());
    Ctx& ctx = g_ctx;
    arena_begin();
    void* sink = nullptr;
    RDF_TRY(arena_alloc(64, &sink));
    const int64_t nvec = bytes / 16;
    const double moved = (double)nvec * 16.0 * (kind == 0 ? 1.0 : kind == 1 ? 2.0 : 3.0);
    const int ncu = std::max(1, eval_grid_limit() / 8);
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    double best = 0.0;
    int best_u = 0, best_b = 0;
    if (reps < 1) reps = 3;
    rdf_status st = RDF_OK;
    for (int u : {1, 4})
        for (int per_cu : {4, 8, 16}) {
            const int64_t want = (nvec + (int64_t)kBlock * u - 1) / ((int64_t)kBlock * u);
            const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)ncu * per_cu));
            for (int r = 0; r <= reps && st == RDF_OK; ++r) {          // (the first launch of a shape is not counted)
                hipError_t e = hipEventRecord(e0, ctx.stream);
                if (e == hipSuccess) e = launch_probe(kind, u, grid, a, b, c, nvec, (uint32_t*)sink, ctx.stream);
                if (e == hipSuccess) e = hipEventRecord(e1, ctx.stream);
                if (e == hipSuccess) e = hipEventSynchronize(e1);
                float ms = 0;
                if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
                if (e != hipSuccess) { st = fail(RDF_DEVICE_ERROR, "probe_stream: %s", hipGetErrorString(e)); break; }
                const double gbps = ms > 0 ? moved / (ms * 1e-3) / 1e9 : 0.0;
                if (r > 0 && gbps > best) { best = gbps; best_u = u; best_b = per_cu; }
            }
        }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    RDF_TRY(st);
    *best_gbps = best;
    if (shape && shape_len > 0)
        snprintf(shape, (size_t)shape_len, "%s, %d x 16-byte vectors per lane and iteration, %d blocks of 256 threads per CU, nontemporal, best of %d launches",
                 kind == 0 ? "read-only (xor fold)" : kind == 1 ? "copy" : "two reads + one write", best_u, best_b, reps);
    return RDF_OK;
}

rdf_status rdf_kernel_timing_reset(int32_t enable) {
    RDF_TRY(ensure_ready());
    HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    g_ctx.timing = enable != 0;
    g_ctx.events_used = 0;
    return RDF_OK;
}
rdf_status rdf_kernel_timing_get(double* total_ms, int64_t* launches) {
    if (!total_ms || !launches) return fail(RDF_INVALID_ARGUMENT, "null pointer");
    RDF_TRY(ensure_ready());
    HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    double tot = 0.0;
    for (size_t i = 0; i < g_ctx.events_used; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, g_ctx.events[i].first, g_ctx.events[i].second));
        tot += ms;
    }
    *total_ms = tot;
    *launches = (int64_t)g_ctx.events_used;
    return RDF_OK;
}

}  // extern "C"
