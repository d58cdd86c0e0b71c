// rdf_device.h — structures shared by the HIP kernels (rdf_kernels.hip) and the host side of the
// C ABI (rdf_capi.cpp).  Not part of the public interface.
#pragma once
#include "rdf_sort_map.h"
#include <stdint.h>
#include <hip/hip_runtime.h>

#include <string>
#include "../../include/rdf_mi355x.h"

namespace rdfk {

constexpr int kBlock = 256;          // 4 wavefronts of 64
constexpr int kVPT = 4;              // rows per thread per tile in the fused evaluator
constexpr int kEvalTile = kBlock * kVPT;   // 1024 rows: one reference RecordBatch (src/dataframe.rs:352)
constexpr int kFilterTile = 4096;    // rows per compaction tile (64 mask words, 16 per wave): measured best of 2048 / 4096 / 8192
constexpr int kFilterTileSmall = 1024;   // the tile for frames in small RecordBatches (mean chunk length <= 2048 rows)
constexpr int kMaxCode = 56;         // accumulator-machine instructions per program
constexpr int kMaxCols = 8;          // columns referenced by one program
constexpr int kMaxFrameCols = 64;    // columns of a pinned frame (programs run over frames of <= kMaxCols columns)
constexpr int kPreCols = 4;          // columns preloaded into registers per tile
constexpr int kMaxTmp = 4;           // LDS spill slots for bushy expression trees
constexpr int kMaxValues = RDF_MAX_VALUES;
constexpr int kMaxGroupValues = RDF_MAX_GROUP_VALUES;   // value expressions of the grouped sink
constexpr int kMaxFilterCols = 8;    // columns per compaction launch

// One (column, chunk): an Arrow array resident in HBM.
struct DevChunkCol {
    const void*    values;
    const uint8_t* validity;  // nullptr = all valid
    int64_t        offset;    // elements / bits
};
struct DevOutChunk {
    void*    values;
    uint8_t* validity;        // nullptr = not requested
};

// Accumulator-machine bytecode.  The host compiles an rdf_expr_node tree (Sethi-Ullman order) into
// this; the opcode stream is wave-uniform, so the interpreter's branches are scalar branches.
enum : uint8_t { BC_LOAD = 0, BC_STORE_TMP = 1, BC_BIN = 2, BC_UN = 3, BC_CAST = 4, BC_EMIT = 5, BC_FILTER = 6, BC_GROUP = 7 };
enum : uint8_t { SRC_NONE = 0, SRC_COL = 1, SRC_IMM = 2, SRC_TMP = 3 };

struct Instr {
    uint8_t  bc;
    uint8_t  op;         // rdf_op for BC_BIN / BC_UN
    uint8_t  dtype;      // domain the op computes in / CAST target / EMIT value dtype
    uint8_t  src_kind;
    uint8_t  src_dtype;  // dtype of the operand as stored (converted to `dtype` on fetch); CAST: source dtype
    uint8_t  swapped;    // BC_BIN: acc = operand OP acc
    uint16_t src;        // column index / tmp slot / EMIT: value index
    uint64_t imm;        // SRC_IMM payload already in the `dtype` domain
};
static_assert(sizeof(Instr) == 16, "Instr must stay 16 bytes");
// eval_lean_kernel (rdf_eval_lean.hip) dispatches a step on a handler index the host wrote into Instr::swapped bits 1..7
// (lean_assign, rdf_capi.cpp; bit 0 stays the swap flag, which is all eval_kernel reads).  LH_NONE: the step has no handler.
enum : int {
    LH_NONE = 0, LH_LOAD, LH_STORE_TMP, LH_FILTER, LH_EMIT, LH_NOT, LH_CAST_I2F, LH_CAST_U2F,
    LH_F_GT, LH_F_GE, LH_F_EQ, LH_F_NE, LH_F_LT, LH_F_LE, LH_F_ADD, LH_F_SUB, LH_F_RSUB, LH_F_MUL, LH_F_DIV, LH_F_RDIV,
    LH_I_ADD, LH_I_SUB, LH_I_RSUB, LH_I_MUL, LH_AND, LH_OR
};

// Per-block partial aggregate of one value expression ({sum,min,max,count}, AggregateFunctions).
struct AggPartial {
    uint64_t sum, mn, mx;
    int64_t  cnt;
};
enum : int32_t { CLS_F64 = 0, CLS_SIGNED = 1, CLS_UNSIGNED = 2 };

enum : int32_t { SINK_STORE = 0, SINK_AGG = 1, SINK_GROUP = 2 };

struct EvalArgs {
    // chunk tables (device memory) — or the inline copies below when nchunks == 1
    const DevChunkCol* cols;             // [ncols * nchunks]
    const int64_t*     chunk_tile_start; // [nchunks + 1]
    const int64_t*     chunk_len;        // [nchunks]
    DevOutChunk*       outs;             // [nvalues * nchunks] (SINK_STORE)
    int64_t*           out_null_counts;  // [nvalues * nchunks] (SINK_STORE)
    AggPartial*        partials;         // [gridDim.x * nvalues] (SINK_AGG)
    uint64_t*          group_partials;   // [gridDim.x * group_words] (SINK_GROUP), see group_words()
    uint32_t*          flags;            // bit 0: divide by zero at a valid slot; bit 1: group id out of range
    int64_t            nchunks, ntiles;
    DevChunkCol        inline_cols[kMaxCols];
    DevOutChunk        inline_outs[kMaxValues];
    int64_t            inline_len;
    int32_t            ncols, nvalues, ncode, ntmp;
    int32_t            col_dtype[kMaxCols];
    int32_t            value_cls[kMaxGroupValues];
    int32_t            ngroups, group_replicas;   // SINK_GROUP: ids in [0, ngroups) + the NULL group; LDS copies of the table
    Instr              code[kMaxCode];
};

// SINK_GROUP accumulator table, in 64-bit words, S = ngroups + 1 slots per row:
//   [v * S + g] sum of value v in group g   |   [(nvalues + v) * S + g] its count   |   [2 * nvalues * S + g] rows of group g
inline __host__ __device__ int group_words(int ngroups, int nvalues) { return (ngroups + 1) * (2 * nvalues + 1); }

// Ahead-of-time specialised grouped sink (rdf_gspec.hip): tiles of kEvalTile rows, same chunk tables as EvalArgs.
constexpr int kGSpecCols = 8;
constexpr int kGSpecImm = 8;
struct GSpecArgs {
    const DevChunkCol* cols_tab;         // [program column * nchunks + chunk] (nchunks > 1)
    const int64_t*     chunk_tile_start; // [nchunks + 1]
    const int64_t*     chunk_len;        // [nchunks]
    DevChunkCol        cols[kGSpecCols]; // nchunks == 1: canonical column k
    int32_t            col_map[kGSpecCols];  // canonical column -> program column
    int64_t            nchunks, ntiles, n;
    uint64_t           imm[kGSpecImm];
    uint64_t*          group_partials;   // [gridDim.x * group_words(ngroups, nvalues)]
    uint32_t*          flags;
    int32_t            ngroups, nvalues, vec_bitmap;
    int32_t            xcd_swz, tile_rot, pad1;   // tile walk as in SpecArgs: XCD x takes the x-th contiguous eighth of every row of gridDim.x tiles; rows rotated by tile_rot blocks
};

struct GroupFinalArgs {
    const uint64_t* partials;   // [nblocks * words]
    uint64_t*       result;     // [words]
    int32_t         nblocks, words, ngroups, nvalues;
    int32_t         value_cls[kMaxGroupValues];
};

struct AggFinalArgs {
    const AggPartial* partials;  // [nblocks * nvalues]
    AggPartial*       result;    // [nvalues]
    int32_t           nblocks, nvalues;
    int32_t           value_cls[kMaxValues];
};

// Specialised headline kernel: filter(x CMP c) -> {sum,min,max,count}(y), f64, one chunk.
struct FilterAggF64Args {
    const double*  x;   const uint8_t* x_validity;   int64_t x_offset;
    const double*  y;   const uint8_t* y_validity;   int64_t y_offset;   // y == x when same column
    int64_t        n;
    double         c;
    AggPartial*    partials;   // [gridDim.x]
};

// Specialised straight-line kernels (rdf_spec_kernel.hip.h): up to kSpecCols columns and kSpecImm literals (the catalogs' programs
// read at most 4 of each; a program compiled at run time may use them all).
constexpr int kSpecCols = 8;
constexpr int kSpecImm = 8;
struct SpecArgs {
    DevChunkCol        cols[kSpecCols];  // nchunks == 1: inline descriptors
    DevOutChunk        out;              // nchunks == 1, SINK_STORE
    const DevChunkCol* cols_tab;         // nchunks > 1: [NC * nchunks], canonical column order
    const DevOutChunk* outs_tab;         // nchunks > 1, SINK_STORE: [nchunks]
    const int64_t*     chunk_tile_start; // nchunks > 1: [nchunks + 1], tiles of spec_rows_per_tile rows (one wave iteration)
    const int64_t*     chunk_len;        // nchunks > 1: [nchunks]
    int64_t            nchunks, ntiles;
    int64_t            n;                // nchunks == 1: rows
    uint64_t           tile_inv;         // floor(((nchunks - 1) << 32) / chunk_tile_start[nchunks - 1]): the tile -> chunk guess is a multiply (0: search)
    uint64_t           imm[kSpecImm];
    int64_t*           out_null_count;   // SINK_STORE: [nchunks]
    AggPartial*        partials;         // SINK_AGG: [gridDim.x * nvalues]
    uint32_t*          flags;
    int32_t            vec_bitmap;       // 1: bitmap words through the vector memory path (default); 0: scalar loads (A/B)
    int32_t            rt[8];            // shape-specialised kernels: operator of runtime-op node k (rdf_op | swap << 8)
    int32_t            alias[kSpecCols]; // canonical column k repeats column alias[k] < k (-1: its own column): registers are copied, not reloaded
    int64_t            tile_rot;         // tile walk: wave p of row i takes tile i * S + (p + i * tile_rot) mod S (S = waves of the grid); 0 <= tile_rot < S, a multiple of the waves per block
    int32_t            xcd_swz, pad0;    // 1: XCD x (= block index mod 8) walks the x-th contiguous eighth of every row of S tiles
};

struct MaskTables {
    const DevChunkCol* mask;             // [nchunks]; values = bit-packed booleans
    const int64_t*     chunk_tile_start; // [nchunks + 1], tiles of kFilterTile rows
    const int64_t*     chunk_len;        // [nchunks]
    int64_t            nchunks, ntiles;
};
struct FilterArgs {
    MaskTables         t;
    const DevChunkCol* cols;             // [ncols * nchunks]
    DevOutChunk*       outs;             // [ncols * nchunks]
    int64_t*           out_null_counts;  // [ncols * nchunks]
    const int64_t*     tile_scan;        // [ntiles + 1] exclusive scan of per-tile keep counts
    int32_t            ncols;
    int32_t            esize[kMaxFilterCols];
};

// Compaction of a frame held in ONE chunk (the long-column case): every descriptor travels in the kernel arguments, so no
// table is read between the data loads of a tile.
struct FilterOneArgs {
    DevChunkCol        mask;
    int64_t            clen, ntiles;
    const int64_t*     tile_scan;        // [ntiles + 1]
    int64_t*           out_null_counts;  // [ncols]
    int32_t            ncols;
    int32_t            esize[kMaxFilterCols];
    DevChunkCol        cols[kMaxFilterCols];
    DevOutChunk        outs[kMaxFilterCols];
};
hipError_t launch_compact_one(const FilterOneArgs& a, hipStream_t s);   // kFilterTile-row tiles

// Wave-granular compaction (rdf_filter.hip, the default): a tile is one wave's kWTile (or kWTileSmall) rows of one chunk.
constexpr int kWTile = 512;          // 8 mask words per wave: half a reference RecordBatch (16 words cost 192 VGPRs: 2 waves per SIMD)
constexpr int kWDmaTile = 1024;      // LDS-DMA compaction: 16 mask words per wave = one reference RecordBatch
constexpr int kWTileSmall = 256;     // frames whose chunks are shorter still (mean chunk length <= 256 rows)
struct FilterWArgs {
    MaskTables         t;                // tiles of kWTile / kWTileSmall rows
    DevChunkCol        mask0;            // nchunks == 1: descriptors inline
    int64_t            len0;
    const DevChunkCol* cols;             // [ncols * nchunks]
    const DevOutChunk* outs;             // [ncols * nchunks]
    int64_t*           out_null_counts;  // [ncols * nchunks]
    const int64_t*     tile_scan;        // [ntiles + 1] exclusive scan of the per-tile keep counts
    uint64_t           tile_inv;         // see find_chunk_tile_inv
    int32_t            ncols, prefetch;  // prefetch: look the next tile up under the current tile's loads (chunked frames)
    int32_t            ends, pad_;       // ends (fcompact_dma_kernel): 1 = tiles at the END of a chunk take the LDS-DMA path too (chunk lengths that are not multiples of the tile)
    int32_t            esize[kMaxFilterCols];
    DevChunkCol        cols0[kMaxFilterCols];   // nchunks == 1
    DevOutChunk        outs0[kMaxFilterCols];
    const int64_t*     out_cap;          // (one-pass paths) [nchunks] rows every output of chunk c can hold, or nullptr = as many as the chunk has: a chunk that keeps more is NOT written (its count still is: the host reports the capacity error)
    int64_t*           out_len;          // fcompact_dma_kernel with tile_scan == nullptr (every chunk is ONE tile: its kept rows start its output): [nchunks] kept rows per chunk (pre-zeroed)
};
// DataFrame::filter in ONE pass (rdf_filter_frame, predicates of the form `col CMP literal [AND|OR col CMP literal]`): the
// predicate is evaluated on the tile the compaction has just brought into LDS — no mask is written, counted or read back —
// and every batch of the output starts where the input batch's mask would (64-row rounded positions), so a tile needs no
// prefix beyond its own batch: none at all for the readers' 1024-row batches, a decoupled look-back between the tiles of a
// longer batch.  (rdf_filter.hip: ffilter_dma_kernel)
struct FusedTerm { int32_t col, op, dtype, pad; double lit; };   // frame column CMP (double)literal, compared in f64 (src/expression.rs:844-845)
struct FusedFilterArgs {
    FilterWArgs         w;            // tile tables, column / output descriptors, null counters (mask / tile_scan unused)
    FusedTerm           term[2];
    int32_t             nterms, combine;     // combine: RDF_OP_AND / RDF_OP_OR of the two terms
    int32_t             lookback, ends;      // lookback: some batch spans several tiles: tiles are taken by ticket, prefixes by look-back; ends: 1 = the instantiation whose tiles at the END of a batch take the LDS-DMA path too (batch lengths that are not multiples of the tile)
    int64_t*            out_len;             // [nchunks] kept rows per batch (pre-zeroed)
    unsigned long long* tile_state;          // (lookback; pre-zeroed) [ntiles] tile states, status << 62 | rows; 8 spare words; [ntiles] super-tile states
    unsigned int*       ticket;              // (lookback; pre-zeroed) 64 counters, 128 bytes apart
};
hipError_t launch_ffilter(const FusedFilterArgs& a, hipStream_t s);
// The same operators on LONG batches (rdf_bfilter.hip): a tile is a BLOCK's rows held in registers, every tile publishes its row
// count one iteration before it asks for its offset, and one wave of the grid (block 0) turns counts into prefixes — no walk over
// the window of tiles in flight.  All columns of a launch are equally wide (8 or 4 bytes).
// `column CMP literal` as the block kernel evaluates it: float columns compare (double)x with the literal; for integer columns the
// host has turned the comparison into the interval of x on which `(double)x CMP literal` holds (x -> (double)x is monotone)
struct BfTerm {
    int32_t  col, kind, op, inv;      // kind 0: float compare `op` against lit; 1: keep <=> lo <= (x ^ bias) <= hi as unsigned, inverted if inv
    double   lit;
    uint64_t lo, hi, bias;
};
struct BFilterArgs {
    FilterWArgs         w;            // tables over tiles of bfilter_tile_rows() rows; t.mask / mask0 used when nterms == 0; tile_scan unused
    BfTerm              bterm[2];
    int32_t             nterms, combine;     // nterms == 0: the kept rows are given by a mask (Column::filter); else `term[0] [combine term[1]]`
    int32_t             nclass, nworkers;    // ticket counters in use (min(64, nworkers)), worker blocks (the grid is nworkers + 1: block 0 scans)
    int64_t*            out_len;             // [nchunks] kept rows per batch (pre-zeroed), or nullptr
    unsigned long long* tile_state;          // (pre-zeroed) [ntiles]: 0 -> count -> prefix
    unsigned int*       ticket;              // (pre-zeroed) 64 counters, 128 bytes apart
    int32_t             stall_test, pad;     // tests: 1 = the scanner does nothing (every wait must give up and the call must fail, not hang)
    int32_t             keep_out, force_multi;   // keep_out 1 (predicate form): the kept rows are ALSO written, in row order, into the mask chunks w.t.mask / w.mask0 describe (bit offsets 0, 8-byte aligned: the frame's own mask) — a frame of two column widths is compacted by two launches, the second one by that mask; force_multi 1: the several-column instantiation (its tile geometry) whatever ncols
    int32_t             short_mode, short_shift;   // short_mode 1: no batch is longer than a tile — a batch takes 1 << short_shift waves of one block, w.t.ntiles counts tiles of 8 >> short_shift BATCHES (chunk_tile_start / tile_inv / tile_state unused), block 0 works like the others; 2: the long form's tiles and tables, but a block draws whole batches (ticket = batch) and adds up the rows in front of a tile itself (tile_state unused, no scanner block)
    unsigned int*       abort_flag;          // (pre-zeroed) set by a wait that saw no progress for kBfWaitSeconds: every waiter then leaves, the host reports a device error — a stuck prefix must cost a call, not the GPU
};
constexpr int kBfWaitSeconds = 4;
int bfilter_tile_rows(int esize, int ncols);
int bfilter_short_shift(int esize, int ncols, int64_t max_len);      // -1: some batch is longer than a tile
hipError_t launch_bfilter(const BFilterArgs& a, int esize, bool nulls, hipStream_t s);
hipError_t launch_fcount(const FilterWArgs& a, int tile_rows, int64_t* tile_counts, hipStream_t s);
hipError_t launch_fcompact(const FilterWArgs& a, int tile_rows, hipStream_t s);
hipError_t launch_mask_count_one(const DevChunkCol& mask, int64_t clen, int64_t ntiles, int64_t* tile_counts, hipStream_t s);

// Stable LSD radix sort of (key, row index) pairs, 8 bits per pass (DataFrame::sort -> lexsort_to_indices).
constexpr int kSortItems = 8;                       // items per thread per tile
constexpr int kSortTile = kBlock * kSortItems;      // 2048 items per tile
struct SortKeyArgs {                                 // key[i] = order-preserving transform of column[idx[i]]
    const DevChunkCol* chunks;                       // [nchunks]
    const int64_t*     chunk_row_start;              // [nchunks + 1]
    int64_t            nchunks, n;
    const uint32_t*    idx;                          // current order (nullptr = identity)
    uint64_t*          keys;                         // out
    uint8_t*           nullflags;                    // out (per ORIGINAL row), nullptr when the column has no bitmap
    uint64_t*          bit_stats;                    // [2] in/out: min and max over the non-null keys written ([0] starts ~0, [1] starts 0)
    int32_t            dtype, descending;
    int32_t            null_or, pad;                 // multi-column join keys: nullflags[row] |= isnull (the buffer starts zeroed)
    uint64_t           hash_mul;                     // != 0 (the join's hash-ordered build side): keys[i] = key bits * hash_mul, bit_stats over THOSE,
    uint64_t*          raw_stats;                    //   and [min, max] of the key bits themselves here
};
struct SortPassArgs {
    const uint64_t* keys_in;  const uint32_t* idx_in;   // idx_in nullptr = identity (first pass)
    uint64_t*       keys_out; uint32_t*       idx_out;
    const uint64_t* pay_in;   uint64_t*       pay_out;  // 64-bit payload variant (group-by partitioning); idx_* unused then
    const uint8_t*  nullflags;                       // digit source of the nulls-last pass (indexed by row), else nullptr
    int64_t*        hist;                            // [256 * sort_grid] digit-major per-block counts, then their exclusive scan
    int64_t         n, ntiles;
    int32_t         shift;                           // bit offset of this pass's digit in (key - bias)
    uint64_t        bias;                            // smallest key of the column: digits are taken from key - bias, so a narrow key RANGE needs few passes
};

// Second-generation radix passes (rdf_sort.hip): all digit histograms in one read, then one read + one write per digit with
// decoupled look-back between tiles.
constexpr int kOsStatePadTiles = 768;                // zeroed tile states behind the last tile: what os_scatter3_kernel's scanners may read past the end (two windows of 256 tiles and a run)
constexpr int kOsItems = 16;                         // items per thread per tile: 4096-item tiles
// f64 keys: the top bits of the key bits are sign and exponent — doubles of one magnitude crowd a few of their patterns —, so the
// most-significant-first passes take their digits from the VALUE bucket floor((x - lo) * scale) instead (monotone in the key bits:
// rounding is monotone; NaNs go below / above everything as their bits order them), bits > 0 turns it on
// (OsSeg, OsBucket: rdf_sort_map.h — the bucket map of a Float64 sort column, shared with the host planner and its CPU test)
struct OsHistArgs {
    const uint64_t* keys;                            // [n] current order
    const uint8_t*  nullflags;                       // [n] by original row, or nullptr: its 1s are counted for the nulls-last pass
    int64_t         n;
    uint64_t        bias;
    int32_t         npass, generic;                  // digits 0..npass-1 of (key - bias): bytes, or (generic) the fields shift[p], mask[p]
    int64_t*        hist;                            // [9 * 256] zeroed; on return the exclusive scan of every pass's counts (row 8: the nulls-last pass)
    int32_t         shift[8], mask[8];
    OsBucket        fb;
};
struct OsPassArgs {
    const uint64_t* keys_in;  const uint32_t* idx_in;   // idx_in nullptr = identity
    uint64_t*       keys_out; uint32_t*       idx_out;
    const uint8_t*  nullflags;                       // the nulls-last pass: digit = nullflags[row], else nullptr
    unsigned long long* state;                       // [ntiles * 256] tile states {seq | flag | value}, zeroed once per sort
    unsigned long long* ticket;                      // next tile of this pass (zeroed)
    const int64_t*  bases;                           // [256] global start of each digit's run in this pass
    int64_t         n, ntiles;
    uint64_t        bias;
    int32_t         shift, seq;                      // seq: 1, 2, ... one per pass launched on `state`
    int32_t         mask, pad;                       // digit = ((key - bias) >> shift) & mask; 0 = 255
    OsBucket        fb;                              // ... or (value bucket >> shift) & mask
    unsigned long long* debug;                       // RDF_DEBUG: [6] cycle sums of the phases (ticket, load + rank, barrier, look-back, sort + write), tiles
    unsigned int*   class_tickets;                   // os_scatter3_kernel (round 6): 64 ticket counters of this pass, 128 bytes apart (zeroed); nullptr = os_scatter_kernel
    int32_t         nclass, super_tiles;             // nclass: counters in use (set by the launcher); super_tiles: K > 1 = os_scatter4_kernel (round 6): a ticket is K consecutive tiles, `state` holds one row of 256 words per super-tile
};
// Most-significant-digits-first finish (keys that vary in more than 32 bits): after stable passes over the TOP bits of the keys
// the rows lie in buckets of <= kOsLocalMax rows that share those bits; every bucket is then sorted on the remaining `rbits`
// low bits by one block in LDS — one read + one write of the pairs instead of one per remaining byte.
constexpr int kOsLocalMax = 4096;
struct OsLocalArgs {
    const uint64_t* keys_in;  const uint32_t* idx_in;   // idx_in nullptr = identity
    uint64_t*       keys_out; uint32_t*       idx_out;
    const uint32_t* bstart;                          // [nbuckets + 1] first row of every bucket
    uint64_t        bias;
    int32_t         rbits, nbuckets;                 // bucket = (key - bias) >> rbits
    uint32_t        len_lo, len_hi;                  // this launch sorts the buckets with len_lo < rows <= len_hi (two launches when a few
                                                     // buckets are long: the LDS a block asks for sets how many blocks a CU holds)
    int32_t         lds_items, wide;                 // LDS the launch carries: the next power of two >= the largest bucket; wide: value
                                                     // buckets — the rows of a bucket share no key bits, the whole key is compared
};
hipError_t launch_os_bounds(const uint64_t* keys, int64_t n, uint64_t bias, int rbits, int nbuckets, const OsBucket& fb, uint32_t* bstart, unsigned int* maxlen, hipStream_t s);
hipError_t launch_os_local(const OsLocalArgs& a, hipStream_t s);
hipError_t launch_os_sample(const uint64_t* keys, int64_t n, int nsamp, uint64_t* out, hipStream_t s);   // nsamp keys from jittered strides
hipError_t launch_os_hist(const OsHistArgs& a, hipStream_t s);
hipError_t launch_os_scatter(const OsPassArgs& a, hipStream_t s);
int os_tile_items();
int os_super_tiles(int64_t ntiles, int max_k);
int sr_grid(int64_t ntiles);
hipError_t launch_sr_hist(const OsPassArgs& a, int64_t* hist, hipStream_t s);      // hist: [256 * sr_grid] digit-major per-block counts
hipError_t launch_sr_scatter(const OsPassArgs& a, const int64_t* hist, hipStream_t s);   // hist: their exclusive scan

// Equi-join indices (calc_equijoin_indices, src/functions/join.rs:19-137): sort the build side by key, binary-search
// every probe row, count -> scan -> write.
struct JoinProbeArgs {
    const uint64_t* lkeys;     // [nl] order-preserving key bits of the probe side
    const uint8_t*  lnull;     // [nl] 1 = NULL key (never matches), or nullptr
    const uint64_t* rkeys;     // [nrv] sorted key bits of the build side (NULL keys excluded: they sort last)
    const uint32_t* ridx;      // [nr] build-side row of each sorted position
    int64_t         nl, nrv;
    int32_t         outer;     // probe rows without a match are emitted with a NULL build index
    int64_t*        counts;    // [nl] out (count phase)
    const int64_t*  offsets;   // [nl + 1] exclusive scan of counts (write phase)
    uint32_t*       out_probe; uint32_t* out_build;
    uint32_t*       out_build_validity;   // bitmap words preset to ones; cleared where the build side is NULL
    uint32_t*       matched;   // [ceil(nrv / 32)] sorted build positions that found a partner (FULL), or nullptr
    unsigned long long* unmatched;  // count phase: probe rows without a partner
    // bucket index over the sorted build keys: bucket b = (key - kmin) >> bucket_shift holds the sorted positions
    // [buckets[2b], buckets[2b + 1]) — a probe is one table read plus a search among the (usually 1-2) keys of its bucket
    const uint32_t* buckets;   // [2 * nbuckets], zero = empty
    uint64_t        kmin, kmax;
    int32_t         bucket_shift, pad;
    uint32_t*       first;     // [nl] count phase out / write phase in: first sorted build position of the row's partners, ~0 = none
    // multi-column keys: lkeys / rkeys are 64-bit hashes of the key tuple; a candidate pair is verified on every column's key bits
    int32_t         nkeys, pad2;
    const uint64_t* pbits[4];  // [nkeys][nl] key bits of the probe side, row order
    const uint64_t* bbits[4];  // [nkeys][nr] key bits of the build side, ROW order (indexed through ridx)
    // one key column, fewer than 2^31 build rows: an open-addressing table over the DISTINCT build keys, 16 bytes per slot:
    // {key bits, first sorted position << 32 | info}, info = the key's build rows (>= 2) or 1 << 31 | the build row of a key that
    // occurs once, 0 = empty slot.  A probe is ONE random memory transaction: the count phase leaves the build row itself in
    // `first` (bit 31 set) for such keys and the write phase never touches ridx.
    const uint64_t* table;     // [2 * (tmask + 1)] or nullptr: the bucket index above
    uint64_t        tmask;
    int32_t         tshift;    // slot = (key * golden ratio) >> tshift
    int32_t         hashed;    // the table holds key * golden ratio (a bijection) in slot order, placed by a scan: no wrap-around (tmask = ~0)
};
struct JoinTableArgs { const uint64_t* rkeys; const uint32_t* ridx; int64_t nrv; uint64_t* table; uint64_t tmask; int32_t tshift, pad; };
// Round 5: the same table WITHOUT atomics.  The build side is sorted by h = key * golden ratio (odd multiplier: a bijection on 64 bits),
// so the distinct keys arrive in slot order and linear probing's layout is a scan: slot(r) = max(home(r), slot(r - 1) + 1) = r +
// max over q <= r of (home(q) - q) for the r-th distinct key.  Phase 0: per tile of 2048 sorted rows {distinct keys, max(home - local
// index)}; phase 1 (one block): their exclusive prefixes under (cA, mA) + (cB, mB) = (cA + cB, max(mA, mB - cA)); phase 2: every tile
// places its keys — near-sequential 16-byte stores instead of 1e8 compare-and-swaps on random lines.  No wrap-around: the table has
// `cap` slots >= 2^tbits + a margin, a slot beyond it raises flags[0] and the host falls back to join_table_kernel.
struct JoinPlaceArgs { const uint64_t* rkeys; const uint32_t* ridx; int64_t nrv; uint64_t* table; int64_t cap; int64_t* tiles; int64_t ntiles;
                       unsigned long long* flags; int32_t tshift, phase; };
struct JoinCombineArgs { const uint64_t* bits[4]; int32_t nkeys, pad; int64_t n; const uint8_t* nullflags; uint64_t* out; uint64_t* bit_stats; };
struct JoinBucketArgs { const uint64_t* rkeys; int64_t nrv; uint32_t* buckets; uint64_t kmin; int32_t bucket_shift; };
struct JoinAppendArgs {        // FULL: build rows nobody matched (and NULL-key build rows) with a NULL probe index
    const uint32_t* ridx; const uint32_t* matched;
    int64_t         nr, nrv;
    unsigned long long* cursor; // running output row (starts at the probe phase's total)
    uint32_t*       out_probe; uint32_t* out_build; uint32_t* out_probe_validity;
    int32_t         count_only;
};

// Partitioned GROUP BY (high cardinality): keys are replaced by an invertible 64-bit mix, the (hashed key, value)
// pairs are radix-partitioned on the top hash bits with the sort kernels, and every partition is aggregated
// in an LDS table and emitted directly.
struct GroupPrepArgs {
    const DevChunkCol* keys;             // [nchunks]
    const DevChunkCol* values;           // [nchunks]
    const int64_t*     chunk_tile_start; // [nchunks + 1], tiles of kEvalTile rows
    const int64_t*     chunk_len;
    const int64_t*     chunk_row_start;  // [nchunks + 1]
    int64_t            nchunks, ntiles;
    int32_t            key_dtype, value_dtype;
    uint64_t*          hkeys;            // [n] out: mix64(key)
    uint64_t*          vals;             // [n] out: value bits (f64 bits / wrapping i64), 0 when there is no value column
    unsigned long long* special_sums;    // [2]: rows whose hashed key equals the LDS free marker / rows with a NULL key
    unsigned long long* special_counts;  // [2]
    unsigned int*      special;          // [2] group exists
};
struct GroupAggArgs {
    const uint64_t* hkeys; const uint64_t* vals;
    int64_t         n;
    int32_t         part_bits;           // partitions = 2^part_bits, partition id = hkey >> (64 - part_bits)
    int32_t         is_f64, has_values, key_dtype;
    void*           out_keys; void* out_sums; int64_t* out_counts;
    unsigned int*   cursor;              // output cursor
    uint32_t*       flags;               // bit 2: an LDS table overflowed (max_groups was too small for the data)
    int64_t         max_out;
};

// Single-pass partitioned GROUP BY (the default for 1024 < max_groups <= kGbMaxGroups): rows are scattered ONCE on the
// top kGbPartBits bits of mix64(key) as 16-byte (hashed key, value bits) records — per-block LDS staging turns the
// scatter into runs of consecutive records — and every partition is aggregated in an LDS table.
// HBM traffic: 8 (histogram) + 16 + 16 (scatter) + 16 (aggregate) = 56 B/row.
constexpr int kGbBlock = 512;                       // threads per block of the three kernels
constexpr int kGbRows = 8;                          // rows per thread per iteration
constexpr int kGbSuper = kGbBlock * kGbRows;        // 4096 rows = 4 tiles of kEvalTile rows per block iteration
constexpr int kGbPartBits = 9;                      // 512 partitions
constexpr int kGbSlots = 3989;                      // LDS table slots per partition: prime (double hashing), 20 B each -> 2 blocks per CU
constexpr int64_t kGbMaxGroups = (int64_t)(1 << kGbPartBits) * 2600;   // keeps the expected load of a table under 0.65
struct GbPartArgs {
    const DevChunkCol* keys;             // [nchunks]
    const DevChunkCol* values;           // [nchunks]
    const int64_t*     chunk_tile_start; // [nchunks + 1], tiles of kEvalTile rows
    const int64_t*     chunk_len;
    int64_t            nchunks, ntiles;
    DevChunkCol        key0, val0;       // nchunks == 1: the chunk's descriptors inline (kernel arguments = scalar registers)
    int64_t            len0;
    int32_t            key_dtype, value_dtype;
    int32_t            ablate_stores, pad;   // bench ablation (rdf_set_option("gb_debug", 2)): run the scatter without its global stores
    int64_t*           hist;             // histogram kernel: out counts [digit * gridDim.x + block]; scatter: their exclusive scan
    uint64_t*          recs;             // scatter out: [2 * rows] records, see rdf_kernels.hip
    int64_t*           emitted;          // combining scatter variant out: [digit * gridDim.x + block] records really written
    unsigned long long* special_sums;    // [2]: rows whose hashed key equals the LDS free marker / rows with a NULL key
    unsigned long long* special_counts;  // [2]
    unsigned int*      special;          // [2] group exists
};
struct GbAggArgs {
    const uint64_t* recs;
    const int64_t*  scan;                // [ (1 << kGbPartBits) * nblocks + 1 ] exclusive scan of the histogram
    const int64_t*  emitted;             // combined (skewed) inputs: real record count of every (partition, block) range, else nullptr
    int64_t         nblocks;             // blocks of the histogram / scatter kernels
    int32_t         is_f64, has_values, key_dtype;
    int32_t         ablate_lds;          // bench ablation (rdf_set_option("gb_debug", 1)): stream the records without the LDS table work
    void*           out_keys; void* out_sums; int64_t* out_counts;
    unsigned int*   cursor;
    uint32_t*       flags;               // bit 2: an LDS table overflowed / more than max_out groups
    int64_t         max_out;
};

// Hash GROUP BY key -> {sum(value), count(value)}: open addressing, linear probing, 64-bit keys.
struct GroupTable {
    unsigned long long* keys;   // [capacity] slot keys, kGroupEmpty = free
    unsigned long long* sums;   // [capacity + 2] f64 bits or wrapping i64; +0/+1 = the sentinel-key and null-key groups
    unsigned long long* counts; // [capacity + 2]
    unsigned int*       special;   // [2] 1 if the sentinel-key / null-key group exists
    unsigned int*       ngroups;   // distinct keys inserted in the table proper
    uint32_t*           flags;     // bit 2: table overflow
    int64_t             capacity;  // power of two
};
constexpr unsigned long long kGroupEmpty = 0x8000000000000000ull;  // i64::MIN doubles as the free marker
struct GroupByArgs {
    const DevChunkCol* keys;             // [nchunks]
    const DevChunkCol* values;           // [nchunks] (values pointer null = count rows only)
    const int64_t*     chunk_tile_start; // [nchunks + 1], tiles of kEvalTile rows
    const int64_t*     chunk_len;
    int64_t            nchunks, ntiles;
    int32_t            key_dtype, value_dtype;  // value_dtype < 0: no values
    GroupTable         t;
    int64_t            max_groups;
};
struct GroupEmitArgs {
    GroupTable t;
    void*      out_keys;  uint8_t* out_keys_validity;
    void*      out_sums;  int64_t* out_counts;
    unsigned int* cursor;   // running output index
    int32_t    key_dtype;
};

// ---- second-generation GROUP BY (rdf_groupby.hip): sum / min / max per group, no histogram pass ----
enum : int32_t { AGG_SUM = 0, AGG_MIN = 1, AGG_MAX = 2 };
// MIN / MAX accumulate order-preserving 64-bit images of the values (ord_bits): unsigned atomic min / max then serve every
// value class; the identity (~0 for MIN, 0 for MAX) is what a NULL or NaN value contributes.
constexpr int kG2PartBits = 8;                      // 256 partitions: carry (32 KB) + staging (32 KB) leave room for TWO scatter blocks per CU
constexpr int kG2Block = 512;                       // scatter block: 8 waves; the two blocks of a CU overlap each other's load / LDS / store phases
#ifndef RDF_G2_ROWS
#define RDF_G2_ROWS 4
#endif
constexpr int kG2Rows = RDF_G2_ROWS;                // rows per thread per super-tile (4: two blocks per CU; 2: LDS and registers for three)
constexpr int kG2BlocksPerCU = kG2Rows == 2 ? 3 : 2;
constexpr int kG2MaxBlocks = 512;      // scatter blocks of one call: what the aggregate pass's region list holds in LDS (two per CU of the MI355X's 256; a part with more CUs runs the same 512)
constexpr int kG2Super = kG2Block * kG2Rows;        // 2048 rows = 2 tiles of kEvalTile rows
constexpr int kG2Line = 8;                          // records per 128-byte line: the only unit ever written
// 12-byte records (keys inside a window of 2^39): a unit of 16 records = one 128-byte line of values + 64 bytes of key words
// (value not NULL : 1 | hash remainder : 31), the two halves in areas of their own; 4-byte records when only rows are counted
constexpr int kG2LineC = 16;
constexpr uint64_t kG2cMul = 0x4F1BBD2385ull, kG2cInv = 0x4B1AF15D4Dull;   // the compact hash: multiplication mod 2^39 and its inverse (rdf_groupby.hip)
constexpr int kG2Slots = 7919;                      // aggregation: LDS table slots per partition (prime; 20 B each = 158 KB, one 1024-thread block per CU)
constexpr int kG2AggBlock = 1024;
constexpr int64_t kG2MaxGroups = (int64_t)(1 << kG2PartBits) * 5100;   // expected table load <= 0.65
constexpr int kG2StreamGroups = 2048;               // up to here one LDS table per block holds every group (gb2_stream_kernel)
struct Gb2Args {
    const DevChunkCol* keys;             // [nchunks]
    const DevChunkCol* values;           // [nchunks]
    const int64_t*     chunk_tile_start; // [nchunks + 1], tiles of kEvalTile rows
    const int64_t*     chunk_len;
    int64_t            nchunks, ntiles;
    DevChunkCol        key0, val0;       // nchunks == 1: descriptors inline
    int64_t            len0;
    int32_t            key_dtype, value_dtype;   // value_dtype < 0: count rows
    int32_t            op, vcls;         // AGG_*, CLS_* of the accumulator
    // scatter output: region (partition p, block b) = lines [(p * nb + b) * cap_lines, +nlines[p * nb + b])
    uint64_t*          recs;
    uint32_t*          nlines;           // [P * nb] lines written (compact records: RECORDS written — their key words have no dead marker)
    int64_t            cap_lines;
    unsigned long long* special_sums;    // [2]: the key whose hash is the LDS free marker / the NULL key
    unsigned long long* special_counts;  // [2]
    unsigned int*      special;          // [2] group exists
    uint32_t*          flags;            // bit 2: more groups than promised; bit 4: a region overflowed (skewed keys)
    // stream kernel: the global table the block tables are merged into
    GroupTable         t;
    int32_t            replicas, sub_slots;   // LDS table = replicas sub-tables of sub_slots slots (lane % replicas picks one)
    int32_t            table_slots, pad2;     // slots of the block's LDS table (20 B each)
    int32_t            ablate, fast;          // fast: 8-byte keys / values, no bitmaps, 16-byte aligned chunks (host-checked); bench ablations of the scatter (rdf_set_option("gb_debug", 21..24)): results invalid
    // skewed keys (capacity plan from the skew probe's histogram): partition p's regions start at line part_off[p] and hold
    // part_cap[p] lines each (region (p, b) = part_off[p] + b * part_cap[p]); nullptr: every region holds cap_lines lines
    const uint32_t*    part_off;
    const uint32_t*    part_cap;
    // compact records: key word j of line l at recs_k[l * 16 + j], value at recs[l * 16 + j]; the key's hash is taken of
    // (key - key_base) in 39 bits — a key outside [key_base, key_base + 2^39) sets flag bit 6 and the host re-runs with 16-byte records
    uint32_t*          recs_k;
    uint64_t           key_base;
    int32_t            compact;
    // heavy hitters (Zipf-like keys): the up to ~100 keys that hold most of a skewed input (found by the probe: classes of the hash's
    // top 16 bits far above the mean, then the keys inside them).  hot_mode 2 (gb2_stream_kernel): only their rows, folded in LDS
    // per block and merged into the global table `t`; hot_mode 1 (gb2_scatter_kernel): every other row; 0: no split
    int32_t            hot_mode;
    const uint32_t*    hot_bitmap;       // [2048] class bitmap, then [256] x 64-bit slots: the hashed hot keys, open addressing from (hash >> 24) & 255, ~0 = free
};
struct Gb2Work { int32_t p, b0, b1, multi; };   // aggregate work item: regions [b0, b1) of partition p; multi: the partition is cut into several items
struct Gb2AggArgs {
    const uint64_t* recs;
    const uint32_t* nlines;              // [P * nb]
    int64_t         nb, cap_lines;
    int32_t         op, vcls, has_values, key_dtype;
    void*           out_keys; uint64_t* out_acc; int64_t* out_counts;   // dense, raw accumulators (gb2_finish_kernel converts)
    unsigned int*   cursor;
    uint32_t*       flags;
    int64_t         max_out;
    // capacity plan (skewed keys): per-partition region layout, and a work list that cuts the big partitions into several
    // items — their groups meet in the global table `t` (HBM atomics), the others are emitted straight from LDS as always
    const uint32_t* part_off;
    const uint32_t* part_cap;
    const Gb2Work*  work;
    int32_t         nwork, compact;
    GroupTable      t;
    const uint32_t* recs_k;              // compact records (see Gb2Args)
    uint64_t        key_base;
};
// (key, accumulator, count) triples -> global table: the merge of partial groups (multi-GPU exchange) and the path for
// more groups than LDS tables hold
struct Gb2MergeArgs {
    const uint64_t* keys; const uint64_t* acc; const int64_t* counts;   // [n] raw 64-bit keys, accumulators in table form (ord bits for MIN / MAX)
    const uint8_t*  key_validity;        // bit i = 0: row i belongs to the NULL-key group (nullptr: none)
    int64_t         n;
    int32_t         op, vcls;
    GroupTable      t;
};
struct Gb2FinishArgs {                    // raw accumulators -> output values (+ validity for MIN / MAX of all-NULL groups)
    uint64_t*       acc; const int64_t* counts;
    uint8_t*        validity;            // may be nullptr
    int64_t         n;
    int32_t         op, vcls, native_in; // native_in: acc holds native bits to be turned INTO table form (the merge's input side)
    unsigned long long* nulls;           // out: groups whose value is NULL
};
// multi-GPU exchange of partial groups: rows -> owner = hash(key) % world, packed [n][3] words grouped by owner
struct GxPackArgs {
    const uint64_t* keys; const uint64_t* acc; const int64_t* counts;
    int64_t         n;
    int32_t         world, phase;        // phase 0: count rows per owner; 1: scatter to offsets
    unsigned long long* owner_counts;    // [world]
    unsigned long long* cursors;         // [world] running write positions (start = exclusive scan of owner_counts)
    uint64_t*       packed;              // [n * words]
    int32_t         words, pad;          // 3: (key, partial, count) triples; 2: (key, value) rows (counts == nullptr)
};
// several key columns <-> one packed 64-bit key (range-compressed fields; code 0 of a nullable field = NULL)
constexpr int kMaxKeyCols = 4;
struct KeyPackArgs {
    const DevChunkCol* cols;             // [nkeys * nchunks]
    const int64_t*     chunk_row_start;  // [nchunks + 1]
    int64_t            nchunks, n;
    int32_t            nkeys;
    int32_t            dtype[kMaxKeyCols], shift[kMaxKeyCols], nullable[kMaxKeyCols];
    uint64_t           bias[kMaxKeyCols], mask[kMaxKeyCols];   // field = ((order-preserving bits - bias) + nullable) << shift
    const uint64_t*    dict[kMaxKeyCols];   // non-null: the column's sorted distinct order-preserving images; field = (rank + nullable) << shift
    int64_t            dict_n[kMaxKeyCols];
    uint64_t*          packed;           // pack: out [n]; unpack: in [n]
    void*              out_values[kMaxKeyCols];   // unpack: dense outputs
    uint8_t*           out_validity[kMaxKeyCols];
    unsigned long long* out_nulls;       // unpack: [nkeys] null counts
};
hipError_t launch_gb2_stream(const Gb2Args& a, int grid, hipStream_t s);
hipError_t launch_gb2_scatter(const Gb2Args& a, int grid, hipStream_t s);
hipError_t launch_gb2_skew_probe(const Gb2Args& a, int64_t tile_step, unsigned int* hist, unsigned long long* minmax /* [2], may be nullptr */, unsigned int* hbins /* [65536] by the hash's top 16 bits, may be nullptr */,
                                 unsigned long long* ktab /* [2 * 4096] (hashed key, samples) of the keys in the classes of a.hot_bitmap, may be nullptr */, hipStream_t s);
hipError_t launch_gb2_aggregate(const Gb2AggArgs& a, hipStream_t s);
hipError_t launch_gb2_merge(const Gb2MergeArgs& a, hipStream_t s);
hipError_t launch_gb2_table_rows(const Gb2Args& a, hipStream_t s);
hipError_t launch_gb2_emit(const GroupEmitArgs& a, hipStream_t s);
hipError_t launch_gb2_finish(const Gb2FinishArgs& a, hipStream_t s);
hipError_t launch_gb2_fill(uint64_t* p, int64_t n, uint64_t v, hipStream_t s);
hipError_t launch_gx_pack(const GxPackArgs& a, hipStream_t s);
hipError_t launch_gx_scan(const unsigned long long* counts, unsigned long long* cursors, int world, hipStream_t s);
hipError_t launch_gx_unpack(const uint64_t* packed, int64_t n, uint64_t* keys, uint64_t* acc, int64_t* counts, int words, hipStream_t s);
hipError_t launch_key_pack(const KeyPackArgs& a, hipStream_t s);
hipError_t launch_key_unpack(const KeyPackArgs& a, hipStream_t s);
size_t gb2_scatter_lds_bytes(bool compact);

struct TakeArgs {
    const DevChunkCol* chunks;           // [nchunks]
    const int64_t*     chunk_row_start;  // [nchunks + 1]
    int64_t            nchunks, total_rows;
    DevChunkCol        indices;          // u32 or u64
    int64_t            n;
    DevOutChunk        out;
    int64_t*           out_null_count;
    uint32_t*          flags;            // bit 1: index out of bounds
    int32_t            esize, idx64;
};

// ---- frame-level operators (rdf_frameops.hip) ----
struct FrameTabArgs {                      // descriptors of up to kMaxFilterCols columns whose chunk c starts pos[c] elements into its buffer
    const int64_t* pos;                    // [nchunks]
    int64_t        nchunks;
    int32_t        ncols, pad;
    char*          values[kMaxFilterCols];
    uint8_t*       validity[kMaxFilterCols];   // nullptr: the column carries no bitmap
    int32_t        esize[kMaxFilterCols];
    DevOutChunk*   outs;                   // [ncols * nchunks] or nullptr
    DevChunkCol*   cols;                   // [ncols * nchunks] or nullptr
};
struct TakeColsArgs {                      // one gather pass over up to kMaxFilterCols columns of a frame
    const DevChunkCol* cols_tab;           // [ncols * nchunks] (nchunks > 1)
    const int64_t*     chunk_row_start;    // [nchunks + 1]
    int64_t            nchunks, total_rows;
    int64_t            uniform_len;        // > 0: every chunk but the last holds this many rows (the readers' batches): lookup by arithmetic
    DevChunkCol        indices;            // u32 or u64
    int64_t            n;
    int64_t*           out_null_counts;    // [ncols]
    uint32_t*          flags;              // bit 1: index out of bounds
    int32_t            ncols, idx64;
    int32_t            need_lookup, pad;        // some column of this launch needs the row -> (batch, element) lookup
    int32_t            esize[kMaxFilterCols];
    int32_t            col_nullable[kMaxFilterCols];
    int32_t            contig[kMaxFilterCols];  // the column is one run of memory (one chunk, or consecutive slices of one buffer): cols0[k], element = row
    DevChunkCol        cols0[kMaxFilterCols];
    DevOutChunk        outs[kMaxFilterCols];
};
constexpr int kMaxRowSlots = 16;            // 8-byte slots of a row record (<= one 128-byte line)
struct TakeRowsArgs {                      // take through interleaved row records (rows_pack_kernel -> rows_gather_kernel)
    const DevChunkCol* cols_tab;           // [ncols * nchunks]
    const int64_t*     chunk_row_start;
    int64_t            nchunks, total_rows, uniform_len;
    DevChunkCol        indices;
    int64_t            n;
    uint64_t*          recs;               // [total_rows * nslot]
    int64_t*           out_null_counts;    // [ncols]
    uint32_t*          flags;
    int32_t            ncols, idx64, need_lookup, nslot;
    int32_t            flag_slot, pad;     // slot of the validity flags (bit k = column k valid) or -1
    int32_t            esize[kMaxRowSlots];
    int32_t            contig[kMaxRowSlots];
    DevChunkCol        cols0[kMaxRowSlots];
    DevOutChunk        outs[kMaxRowSlots];
};
hipError_t launch_take_rows(const TakeRowsArgs& a, hipStream_t s);
hipError_t launch_idx_locality(const DevChunkCol& indices, int64_t n, bool idx64, unsigned int* near, hipStream_t s);   // *near (zeroed) += sampled neighbours (of 4096) less than 64 rows apart
hipError_t launch_frame_totals(const int64_t* tile_scan, const int64_t* chunk_tile_start, int64_t nchunks, int64_t* out_len, int64_t* padded, hipStream_t s);
hipError_t launch_frame_tables(const FrameTabArgs& a, hipStream_t s);
hipError_t launch_frame_mask_tables(const int64_t* pos, int64_t nchunks, uint8_t* values, uint8_t* validity, DevOutChunk* outs, DevChunkCol* cols, hipStream_t s);
hipError_t launch_frame_pad(const int64_t* len, int64_t n, int64_t* padded, hipStream_t s);
hipError_t launch_frame_mask_count(const uint64_t* mask, const int64_t* pos, const int64_t* chunk_tile_start, const int64_t* chunk_len, int64_t nchunks,
                                   int64_t ntiles, int tile_rows, uint64_t tile_inv, int64_t* counts, hipStream_t s);
hipError_t launch_take_cols(const TakeColsArgs& a, hipStream_t s);

// ArrayFunctions over List<primitive> (rdf_list.hip)
enum : int32_t { LIST_CONTAINS = 0, LIST_POSITION = 1, LIST_MAX = 2, LIST_MIN = 3,
                 LIST_REMOVE = 4, LIST_DISTINCT = 5, LIST_EXCEPT = 6, LIST_INTERSECT = 7, LIST_UNION = 8, LIST_REPEAT = 9 };
struct ListArgs {
    DevChunkCol offsets;     // int32 value_offsets [n + 1]; its validity bits are the LIST rows' validity
    DevChunkCol values;      // child values
    int64_t     n;           // list rows
    int32_t     dtype, op;   // child dtype, LIST_*
    uint64_t    needle;      // contains / position: the value, as raw bits of the child dtype
    DevOutChunk out;
    int64_t*    out_null_count;
    int64_t*    kept;        // array_remove pass 1 out: kept elements per row
    const int64_t* scan;     // array_remove pass 2 in: their exclusive scan (nullptr selects pass 1)
    DevChunkCol offsets_b;   // second list of except / intersect / union (its validity is not looked at, array.rs:82,126,372)
    DevChunkCol values_b;
    int32_t     count;       // array_repeat
    uint32_t*   tab_a;       // set functions, one row per wave: value -> first index tables, 2 slots per child element
    uint32_t*   tab_b;
    uint32_t*   work;        // rows left to the row-per-wave kernel by the row-per-lane one (nullptr: all rows)
    uint32_t*   work_count;
};
hipError_t launch_list_op(const ListArgs& a, bool wave_per_row, hipStream_t s);
hipError_t launch_list_row_ids(const ListArgs& a, uint32_t* row_ids, int32_t first, hipStream_t s);
hipError_t launch_list_remove(const ListArgs& a, bool wave_per_row, hipStream_t s);
hipError_t launch_list_set(const ListArgs& a, bool wave_per_row, hipStream_t s);   // distinct / except / intersect / union / repeat
hipError_t launch_list_sort(const ListArgs& a, bool lane_first, hipStream_t s);   // rows sorted in LDS; a.count = child index of the first element, a.work_count[1] = flag "a row too long"
hipError_t launch_list_offsets(const int64_t* scan, int64_t n1, int32_t* out, hipStream_t s);

// ---- launch wrappers (defined in rdf_kernels.hip) ----
int  eval_grid_limit();   // persistent grid size for streaming kernels
hipError_t launch_probe(int kind, int u, int grid, const void* a, void* b, void* c, int64_t nvec, uint32_t* sink, hipStream_t s);   // rdf_probe.hip: bare streams
hipError_t launch_eval_lean(const EvalArgs& a, int sink, int grid, hipStream_t s, bool one_tile = false);   // SINK_AGG / SINK_STORE programs whose every step has a lean handler (rdf_eval_lean.hip)
hipError_t launch_eval(const EvalArgs& a, int sink, int feat, int grid, hipStream_t s);  // feat: 0 basic, 1 +int div, 2 +libm
bool gspec_available(const char* sig);
hipError_t launch_gspec(const char* sig, const GSpecArgs& a, int grid, hipStream_t s);
int gspec_catalog_size();
hipError_t launch_group_final(const GroupFinalArgs& a, hipStream_t s);
hipError_t launch_agg_final(const AggFinalArgs& a, hipStream_t s);
bool spec_available(const char* sig);
int  spec_rows_per_tile(const char* sig);
int  spec_catalog_size();
hipError_t launch_spec(const char* sig, const SpecArgs& a, int grid, hipStream_t s);
// rdf_jit.cpp: spec_kernel<Prog> instantiated at run time (hiprtc) for an exact-program signature the catalog does not hold
struct JitKernel { void* fn; int rows_per_tile; int nvalues; };   // nvalues: grouped programs ("G..." signatures, gspec_kernel)
const JitKernel* jit_find(const char* sig);          // ready on the current device (compiled earlier, or read from the cache directory), or nullptr
// ... compiling it if need be: wait = false starts the compiler on a helper thread and returns nullptr (the interpreter answers
// this call, the kernel takes over when it is ready); wait = true waits for it.  nullptr afterwards: not possible (remembered)
const JitKernel* jit_spec_kernel(const char* sig, bool wait);
void jit_mark_failed(const char* sig);               // a launch failed: this (device, signature) is interpreted from now on
std::string jit_status();                            // one line: compiler / sources / cache directory found or not, counts
hipError_t jit_launch(const JitKernel& k, const SpecArgs& a, int grid, hipStream_t s);
hipError_t jit_launch_grouped(const JitKernel& k, const GSpecArgs& a, int grid, hipStream_t s);
int jit_compiled_count();
void jit_shutdown();                                 // kills compilers in flight and joins every helper thread (atexit / library unload; idempotent)
hipError_t launch_filter_agg_f64(const FilterAggF64Args& a, int cmp_op, int grid, hipStream_t s);
hipError_t launch_mask_count(const MaskTables& t, int tile_rows, int64_t* tile_counts, hipStream_t s);   // tile_rows: kFilterTile or kFilterTileSmall
hipError_t launch_scan(const int64_t* counts, int64_t* scan, int64_t n, int64_t* scratch, hipStream_t s);
int64_t scan_scratch_words(int64_t n);
hipError_t launch_compact(const FilterArgs& a, int tile_rows, hipStream_t s);
hipError_t launch_take(const TakeArgs& a, hipStream_t s);
int  sort_grid(int64_t ntiles);
hipError_t launch_sort_keys(const SortKeyArgs& a, hipStream_t s);
hipError_t launch_sort_hist(const SortPassArgs& a, hipStream_t s);
hipError_t launch_sort_scatter(const SortPassArgs& a, hipStream_t s);
hipError_t launch_join_buckets(const JoinBucketArgs& a, hipStream_t s);
hipError_t launch_copy_small(const void* src_pinned, void* dst_dev, size_t bytes, hipStream_t s);   // src: page-locked, device-mapped host memory
hipError_t launch_join_table(const JoinTableArgs& a, hipStream_t s);
hipError_t launch_join_place(JoinPlaceArgs a, hipStream_t s);     // the three phases
hipError_t launch_join_distinct(const uint64_t* sorted_keys, int64_t n, unsigned long long* out, hipStream_t s);     // distinct values of a sorted array, added to *out
constexpr int kJoinPlaceTile = 2048;
hipError_t launch_join_combine(const JoinCombineArgs& a, hipStream_t s);
hipError_t launch_join_count(const JoinProbeArgs& a, hipStream_t s);
hipError_t launch_join_write(const JoinProbeArgs& a, hipStream_t s);
hipError_t launch_join_append(const JoinAppendArgs& a, hipStream_t s);
hipError_t launch_count_bytes(const uint8_t* p, int64_t n, unsigned long long* out, hipStream_t s);
hipError_t launch_sort_hist64(const SortPassArgs& a, hipStream_t s);
hipError_t launch_sort_scatter64(const SortPassArgs& a, hipStream_t s);
hipError_t launch_gb_hist(const GbPartArgs& a, int grid, hipStream_t s);
hipError_t launch_gb_skew(const int64_t* scan, int64_t nblocks, unsigned int* flag, hipStream_t s);
hipError_t launch_gb_scatter(const GbPartArgs& a, int grid, bool dedup, hipStream_t s);
hipError_t launch_gb_aggregate(const GbAggArgs& a, hipStream_t s);
hipError_t launch_groupby_prepare(const GroupPrepArgs& a, hipStream_t s);
hipError_t launch_groupby_partitions(const GroupAggArgs& a, hipStream_t s);
hipError_t launch_groupby_build(const GroupByArgs& a, hipStream_t s);
hipError_t launch_groupby_emit(const GroupEmitArgs& a, hipStream_t s);
hipError_t launch_fill_f64(double* p, int64_t n, uint64_t seed, uint64_t col, int64_t first_row, double lo, double hi, hipStream_t s);
hipError_t launch_fill_i64(int64_t* p, int64_t n, uint64_t seed, uint64_t col, int64_t first_row, int64_t lo, int64_t hi, hipStream_t s);
hipError_t launch_fill_validity(uint8_t* p, int64_t nbits, uint64_t seed, uint64_t col, int64_t first_row, double null_fraction, hipStream_t s);

}  // namespace rdfk
