/*
 * rdf_mi355x.h — C ABI of librdf_mi355x.so, the MI355X (gfx950) engine for rust-dataframe's
 * Arrow compute hot path.
 *
 * The reference (nevi-me/rust-dataframe, Rust) has no FFI of its own; the seams this ABI replaces are
 * the Rust signatures named next to each entry point (paths relative to the reference root).  A Rust
 * shim binds these with `extern "C"` and passes raw Arrow buffer pointers
 * (`array.data().buffers()[0].raw_data()`, `null_buffer()`, `offset()`, `len()`, `null_count()`);
 * see INTEGRATION.md.
 *
 * Conventions (mirroring the reference):
 *   - inputs are borrowed and never written; outputs are caller-allocated and owned by the caller;
 *   - a column is a list of `nchunks` arrays (ChunkedArray, src/table.rs:13-18); chunk i of every
 *     column of a frame is RecordBatch i (src/dataframe.rs:128-163);
 *   - errors are values (rdf_status mirrors DataFrameError, src/error.rs:6-15); nothing aborts;
 *   - every entry point is re-entrant; state (stream, arena, last error) is per calling thread;
 *   - `mem` says where the buffers live: RDF_MEM_HOST (Arrow buffers in host RAM: staged to HBM,
 *     computed there, results copied back) or RDF_MEM_DEVICE (already resident in HBM: kernels run
 *     in place, nothing crosses PCIe).  All arrays of one call must share one `mem`.
 *   - there is NO CPU fallback: without a usable gfx950 device every compute entry point returns
 *     RDF_DEVICE_ERROR.
 *
 * Buffer rules (Arrow columnar format): values little-endian natives, element `offset` is the first
 * logical element; validity is LSB-first, 1 = valid, NULL = all valid, and shares `offset`.
 * RDF_BOOL arrays are bit-packed in `values` like a validity bitmap.  Bitmap buffers must be
 * readable up to the next 8-byte boundary (Arrow allocates in 64-byte multiples).  Output buffers
 * with mem == RDF_MEM_DEVICE must have room for `capacity` elements rounded up to a multiple of 64
 * (values and bitmap alike).
 */
#ifndef RDF_MI355X_H
#define RDF_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* DataFrameError (src/error.rs:6-15) / ArrowError as raised on the path. */
typedef enum {
    RDF_OK = 0,
    RDF_COMPUTE_ERROR = 1,    /* ArrowError::ComputeError / DataFrameError::ComputeError */
    RDF_DIVIDE_BY_ZERO = 2,   /* ArrowError::DivideByZero / DataFrameError::DivideByZero */
    RDF_INVALID_ARGUMENT = 3, /* ArrowError::InvalidArgumentError, the reference's panic!("Unsupported operation") arms */
    RDF_MEMORY_ERROR = 4,     /* DataFrameError::MemoryError */
    RDF_DEVICE_ERROR = 5      /* no counterpart: HIP runtime failure / no gfx950 device */
} rdf_status;

/* arrow::datatypes::DataType subset that reaches the path (src/evaluation.rs:107-293). */
typedef enum {
    RDF_I8 = 0, RDF_I16 = 1, RDF_I32 = 2, RDF_I64 = 3,
    RDF_U8 = 4, RDF_U16 = 5, RDF_U32 = 6, RDF_U64 = 7,
    RDF_F32 = 8, RDF_F64 = 9, RDF_BOOL = 10,
    RDF_NULLTYPE = 11 /* only as the dtype of a Scalar::Null literal (src/expression.rs:719) */
} rdf_dtype;

typedef enum { RDF_MEM_HOST = 0, RDF_MEM_DEVICE = 1 } rdf_mem;

/* One Arrow array (chunk), borrowed and read-only: PrimitiveArray<T> / BooleanArray. */
typedef struct {
    const void*    values;
    const uint8_t* validity;   /* NULL = no nulls */
    int64_t        offset;     /* in elements (and bits) */
    int64_t        length;
    int64_t        null_count; /* -1 = unknown */
    int32_t        dtype;      /* rdf_dtype */
    int32_t        mem;        /* rdf_mem */
} rdf_array;

/* One output chunk.  Caller allocates `values` (capacity elements) and, when nulls can result,
 * `validity`; callee fills them and sets length / null_count.  Output offset is always 0. */
typedef struct {
    void*    values;
    uint8_t* validity;   /* may be NULL when no input carries a validity bitmap */
    int64_t  capacity;   /* in elements */
    int64_t  length;     /* set by callee */
    int64_t  null_count; /* set by callee */
    int32_t  dtype;
    int32_t  mem;
} rdf_out;

typedef enum {
    /* arrow::compute::{add,subtract,multiply,divide} via ScalarFunctions (src/functions/scalar.rs:16-103) */
    RDF_OP_ADD = 1, RDF_OP_SUB = 2, RDF_OP_MUL = 3, RDF_OP_DIV = 4,
    /* math_op users (src/functions/scalar.rs:148,274,291): atan2(a,b), hypot(a,b), log of a in base b */
    RDF_OP_ATAN2 = 5, RDF_OP_HYPOT = 6, RDF_OP_LOG = 7,
    /* scalar_op users (src/functions/scalar.rs:106-452) */
    RDF_OP_ABS = 8, RDF_OP_ACOS = 9, RDF_OP_ASIN = 10, RDF_OP_ATAN = 11, RDF_OP_CBRT = 12,
    RDF_OP_CEIL = 13, RDF_OP_COS = 14, RDF_OP_COSH = 15, RDF_OP_DEGREES = 16, RDF_OP_EXP = 17,
    RDF_OP_EXPM1 = 18, RDF_OP_FLOOR = 19, RDF_OP_LOG10 = 20, RDF_OP_LOG2 = 21, RDF_OP_RADIANS = 22,
    RDF_OP_ROUND = 23, RDF_OP_SIN = 24, RDF_OP_SINH = 25, RDF_OP_SQRT = 26, RDF_OP_TAN = 27,
    RDF_OP_TANH = 28,
    /* arrow::compute::cast (src/evaluation.rs:296-315) */
    RDF_OP_CAST = 29,
    /* BooleanFilter (src/expression.rs:752-763): comparisons are evaluated in f64 (:844-845) */
    RDF_OP_GT = 30, RDF_OP_GE = 31, RDF_OP_EQ = 32, RDF_OP_NE = 33, RDF_OP_LT = 34, RDF_OP_LE = 35,
    RDF_OP_NOT = 36, RDF_OP_AND = 37, RDF_OP_OR = 38,
    /* arrow::compute::hour via ScalarFunctions::hour (src/functions/scalar.rs:267-273), one opcode per time unit of
     * the temporal input (Time32 s/ms, Time64 us/ns, Date64 ms, Timestamp s/ms/us/ns; Date32 counts days: hour 0).
     * Operand Int32 or Int64 (the temporal types' storage), result of the operand's type, 0..23. */
    RDF_OP_HOUR_S = 39, RDF_OP_HOUR_MS = 40, RDF_OP_HOUR_US = 41, RDF_OP_HOUR_NS = 42, RDF_OP_HOUR_DAY = 43,
    /* ScalarFunction::{Cotangent, Secant, Cosecant} (src/expression.rs:670-672): plannable names the reference never
     * evaluates (its plan builder panics, :487-489).  Unary math like the others: 1 / tan(x), 1 / cos(x), 1 / sin(x) in
     * the operand's float type (IEEE division: cot(0) = +inf). */
    RDF_OP_COT = 44, RDF_OP_SEC = 45, RDF_OP_CSC = 46
} rdf_op;

/* time units of the temporal arrays handed to rdf_hour */
typedef enum { RDF_TIME_SECOND = 0, RDF_TIME_MILLISECOND = 1, RDF_TIME_MICROSECOND = 2, RDF_TIME_NANOSECOND = 3, RDF_TIME_DAY = 4 } rdf_time_unit;

/* ------------------------------------------------------------------ library / device plumbing */

const char* rdf_version(void);
/* Thread-local message of the last failing call -> DataFrameError::ComputeError(String). */
const char* rdf_last_error(void);
rdf_status  rdf_device_count(int32_t* count);
rdf_status  rdf_set_device(int32_t device);
/* Use the caller's hipStream_t for this thread (NULL = the library's own stream). */
rdf_status  rdf_set_stream(void* hip_stream);
rdf_status  rdf_synchronize(void);
rdf_status  rdf_dev_alloc(void** ptr, int64_t bytes);
rdf_status  rdf_dev_free(void* ptr);
rdf_status  rdf_copy_h2d(void* dst_dev, const void* src_host, int64_t bytes);
rdf_status  rdf_copy_d2h(void* dst_host, const void* src_dev, int64_t bytes);
/* Ingestion (DataFrame::from_csv / from_arrow, src/dataframe.rs:349-407): page-locked host buffers and uploads that do not
 * block the reader.  rdf_host_alloc gives a pinned buffer (a CSV parser writes its typed column into it, an IPC file is read
 * into it); rdf_host_register pins memory the caller already holds (a file image) — RDF_MEMORY_ERROR when the platform
 * refuses, the caller then uses rdf_copy_h2d.  rdf_copy_h2d_async queues the upload on the thread's copy stream and returns:
 * the source must be pinned and stay untouched until rdf_copy_fence, which waits for every queued upload of the thread.
 * Kernels launched after the fence see the data. */
rdf_status  rdf_host_alloc(void** ptr, int64_t bytes);
rdf_status  rdf_host_free(void* ptr);
rdf_status  rdf_host_register(void* ptr, int64_t bytes);
rdf_status  rdf_host_unregister(void* ptr);
rdf_status  rdf_copy_h2d_async(void* dst_dev, const void* src_host_pinned, int64_t bytes);
rdf_status  rdf_copy_fence(void);

/* ------------------------------------------------------------------ scalar kernels */

/* ScalarFunctions::{add,subtract,multiply,divide,par_multiply} (src/functions/scalar.rs:16-103) and
 * ::{atan2,hypot,log} (:148,:274,:291).  a[i] op b[i] per chunk pair; validity = AND; chunk length
 * mismatch -> RDF_COMPUTE_ERROR; DIV with a zero divisor at a valid slot -> RDF_DIVIDE_BY_ZERO;
 * integers wrap.  a, b, out: nchunks entries each, one dtype. */
rdf_status rdf_binary(int32_t op, const rdf_array* a, const rdf_array* b, int64_t nchunks, rdf_out* out);

/* ScalarFunctions::{abs,acos,...,tanh} through scalar_op (src/functions/scalar.rs:525-540):
 * out[i] = f(a[i]) where valid, null elsewhere. */
rdf_status rdf_unary(int32_t op, const rdf_array* a, int64_t nchunks, rdf_out* out);

/* Function::Cast arm (src/evaluation.rs:296-315): arrow::compute::cast per chunk to out[i].dtype.  The arrow crate of
 * the reference's era casts numeric arrays element by element through num::cast::cast and appends NULL where that
 * returns None: an integer that the target type cannot represent (-1 -> UInt8, 300 -> UInt8), NaN / an out-of-range
 * float on the way to an integer (floats truncate toward zero when the truncated value fits); int -> float, float -> float
 * and numeric <-> Boolean always succeed; input NULLs stay NULL.  Since a narrowing / sign-changing / float -> integer cast
 * can produce NULLs, its outputs need a validity buffer even when the input has none.  The same rule holds for
 * RDF_OP_CAST inside fused programs (the NULLs it produces are skipped by aggregates, dropped by filters). */
rdf_status rdf_cast(const rdf_array* a, int64_t nchunks, rdf_out* out);

/* ScalarFunctions::hour (src/functions/scalar.rs:267-273) = arrow::compute::hour per chunk: the hour of day of a
 * Time32 / Time64 / Date32 / Date64 / Timestamp array, passed as its Int32 / Int64 storage plus its `unit`
 * (rdf_time_unit; timestamps carry no time zone here, like the reference's).  out: RDF_I32 per chunk, NULL where
 * the input is NULL.  hour = floor_mod(floor_div(value, units per second), 86400) / 3600 — chrono's
 * NaiveDateTime::from_timestamp / NaiveTime::from_num_seconds_from_midnight for every value they accept; values
 * the reference panics on (a time of day outside [0, 86400 s), a negative sub-second remainder) follow the same
 * formula instead. */
rdf_status rdf_hour(const rdf_array* a, int64_t nchunks, int32_t unit, rdf_out* out);

/* ------------------------------------------------------------------ aggregate kernels */

/* AggregateFunctions::sum (src/functions/aggregate.rs:82-93): nulls skipped, empty/all-null -> 0,
 * always Some.  out_scalar has the array's native type (integers wrap). */
rdf_status rdf_sum(const rdf_array* a, int64_t nchunks, void* out_scalar, int32_t* out_is_some);
/* AggregateFunctions::min / max (:12-31) with the evident intent (min is min; floats accepted;
 * None when every slot is null or there are no chunks).  See DESIGN.md "divergences". */
rdf_status rdf_min(const rdf_array* a, int64_t nchunks, void* out_scalar, int32_t* out_is_some);
rdf_status rdf_max(const rdf_array* a, int64_t nchunks, void* out_scalar, int32_t* out_is_some);
/* AggregateFunctions::count (:70-80): sum(len - null_count) as i64; counts validity bits when
 * null_count is unknown (-1). */
rdf_status rdf_count(const rdf_array* a, int64_t nchunks, int64_t* out_count, int32_t* out_is_some);
/* AggregateFunctions::avg (:32-65): mean of the valid values as f64, None when there are none. */
rdf_status rdf_avg(const rdf_array* a, int64_t nchunks, double* out_mean, int32_t* out_is_some);

/* ------------------------------------------------------------------ expressions */

typedef enum { RDF_NODE_COLUMN = 0, RDF_NODE_SCALAR = 1, RDF_NODE_OP = 2 } rdf_node_kind;

/* One node of an expression tree stored as an array, children before parents.  It restates
 * BooleanFilter / BooleanInput / Scalar (src/expression.rs:718-763) and, for value expressions,
 * a run of Calculation steps (src/expression.rs:410-500) folded into one tree. */
typedef struct {
    int32_t kind;    /* rdf_node_kind */
    int32_t op;      /* rdf_op when kind == RDF_NODE_OP */
    int32_t dtype;   /* SCALAR: literal type (RDF_NULLTYPE for Scalar::Null); OP CAST: target type */
    int32_t lhs;     /* child node index or -1 */
    int32_t rhs;     /* child node index or -1 */
    int32_t column;  /* COLUMN: index into the cols argument */
    double  f64;     /* SCALAR of RDF_F32 / RDF_F64 */
    int64_t i64;     /* SCALAR of integer / RDF_BOOL type */
} rdf_expr_node;

/* BooleanFilter::eval_to_array for every RecordBatch (src/expression.rs:766-861 as driven by
 * DataFrame::evaluate_boolean_filter, src/dataframe.rs:612-624).  cols is laid out
 * cols[c * nchunks + i] = chunk i of column c.  mask: nchunks RDF_BOOL outputs (values + validity
 * bitmaps); the value bit of a null slot is 0.  The root must be boolean-typed. */
rdf_status rdf_predicate(const rdf_expr_node* nodes, int32_t nnodes, int32_t root,
                         const rdf_array* cols, int32_t ncols, int64_t nchunks, rdf_out* mask);

/* ------------------------------------------------------------------ filter / take */

/* Rows Column::filter would keep per chunk (two-phase filter: size the outputs with this). */
rdf_status rdf_filter_count(const rdf_array* mask, int64_t nchunks, int64_t* counts);
/* Column::filter -> ChunkedArray::filter -> arrow::compute::filter per chunk pair
 * (src/table.rs:97-107,213-215): keeps rows whose mask bit is set (and valid), order and chunk
 * boundaries preserved, validity carried.  mask[i].length must equal col[i].length. */
rdf_status rdf_filter(const rdf_array* col, const rdf_array* mask, int64_t nchunks, rdf_out* out);
/* DataFrame::filter's per-column loop (src/dataframe.rs:183-187) as ONE pass: ranks are computed
 * once per tile and every column is compacted with them.  cols/outs laid out [c * nchunks + i];
 * 1..256 columns per call. */
rdf_status rdf_filter_columns(const rdf_array* cols, int32_t ncols, const rdf_array* mask,
                              int64_t nchunks, rdf_out* outs);
/* Column::take (src/table.rs:218-241): gather over the virtual concatenation of the chunks
 * (Column::to_array, :180-182, without the concat copy).  indices: ONE RDF_U32 (drop-in) or RDF_U64
 * array; null index -> null; out of range -> RDF_COMPUTE_ERROR.  out: ONE chunk (B4 in SURVEY.md). */
rdf_status rdf_take(const rdf_array* chunks, int64_t nchunks, const rdf_array* indices, rdf_out* out);

/* ------------------------------------------------------------------ sort */

/* SortCriteria (src/expression.rs:305-318) as arrow's SortOptions: the reference always passes
 * nulls_first = false (src/dataframe.rs:208), so nulls sort last whatever this field says. */
typedef struct { int32_t descending; int32_t nulls_first; } rdf_sort_options;

/* DataFrame::sort -> arrow::compute::lexsort_to_indices (src/dataframe.rs:194-214): the row order given
 * by the sort columns, column 0 most significant; ties keep ascending row order (stable); floats in
 * IEEE total order (NaN after +inf).  cols[c * nchunks + i]; indices refer to the concatenation of the
 * chunks (what Column::take consumes).  out_indices: ONE RDF_U32 array of all rows. */
rdf_status rdf_sort_to_indices(const rdf_array* cols, int32_t ncols, int64_t nchunks, const rdf_sort_options* opts,
                               rdf_out* out_indices);

/* ------------------------------------------------------------------ join */

typedef enum { RDF_JOIN_LEFT = 0, RDF_JOIN_RIGHT = 1, RDF_JOIN_INNER = 2, RDF_JOIN_FULL = 3 } rdf_join_type;  /* JoinType, src/expression.rs:339-345 */

/* calc_equijoin_indices (src/functions/join.rs:19-137) for ONE numeric key column per side (same dtype: the
 * caller casts first, as join.rs:17-18 says): the (left row, right row) pairs of the equi-join, as two
 * UInt32 index arrays with NULL where a side has no partner — exactly what DataFrame::join feeds to
 * Column::take (src/dataframe.rs:705-711).  NULL keys never match; LEFT/RIGHT/FULL keep them with a
 * NULL partner.  FULL is a true full outer join (the reference's FullJoin arm drops unmatched non-NULL
 * rows: not copied).  Pair order: probe rows ascending, partners ascending, then (FULL) the unmatched
 * build rows in unspecified order — the reference's order is HashMap iteration order.
 * *out_rows = rows of the result; with out_left == out_right == NULL only the count is computed;
 * capacity too small -> RDF_MEMORY_ERROR with *out_rows set. */
rdf_status rdf_equijoin_indices(const rdf_array* left_keys, int64_t left_nchunks, const rdf_array* right_keys,
                                int64_t right_nchunks, int32_t join_type, rdf_out* out_left, rdf_out* out_right,
                                int64_t* out_rows);
/* The same for 1..4 key columns per side (JoinCriteria.criteria holds a Vec of column pairs, src/expression.rs:332-337;
 * build_hash_inputs hashes the tuple, src/functions/join.rs:139-235): left_keys[k * left_nchunks + i] pairs with
 * right_keys[k * right_nchunks + i]; pair k shares one dtype; a row with a NULL in any key column never matches.
 * Rows are matched through a 64-bit hash of the tuple and every candidate pair is verified column by column. */
rdf_status rdf_equijoin_indices_multi(const rdf_array* left_keys, int64_t left_nchunks, const rdf_array* right_keys,
                                      int64_t right_nchunks, int32_t nkeys, int32_t join_type, rdf_out* out_left,
                                      rdf_out* out_right, int64_t* out_rows);

/* ------------------------------------------------------------------ group-by */

/* Transformation::GroupAggregate(groups, [Sum, Count]) for ONE integer key column — planned by
 * Dataset::try_aggregate (src/expression.rs:114-221) but not executed by the reference
 * (src/evaluation.rs:73 panics), so the semantics are SQL's: NULL keys form one group, NULL values
 * are skipped, a group's count is its number of non-null values (of rows when values == NULL).
 * Outputs are ONE chunk each, in unspecified group order: keys (key dtype), sums (Float64 for float
 * values, wrapping Int64 otherwise), counts (Int64); capacity >= max_groups + 2.  More than
 * max_groups distinct keys -> RDF_MEMORY_ERROR.  f64 sums are accumulated with hardware atomics:
 * the rounding order is not deterministic (within 1e-6 relative of any sequential order). */
rdf_status rdf_groupby_sum(const rdf_array* keys, const rdf_array* values, int64_t nchunks, int64_t max_groups,
                           rdf_out* out_keys, rdf_out* out_sums, rdf_out* out_counts);   /* = rdf_groupby_agg(keys, 1, values, ..., RDF_AGG_SUM, ...) */

/* AggregateFunction (src/expression.rs:696-711).  Avg is Sum / Count on the caller's side (AggregateFunctions::avg,
 * src/functions/aggregate.rs:32-65). */
typedef enum { RDF_AGG_SUM = 0, RDF_AGG_MIN = 1, RDF_AGG_MAX = 2, RDF_AGG_COUNT = 3 } rdf_agg_fn;

#define RDF_MAX_GROUP_KEYS 4

/* Transformation::GroupAggregate(groups, [aggregation]) for 1..RDF_MAX_GROUP_KEYS integer grouping columns and ONE
 * aggregation of one value column (Dataset::try_aggregate, src/expression.rs:114-221, plans a list of them: one call
 * each; the reference never executes the step, src/evaluation.rs:73 panics — SQL semantics, parity unpinned by the
 * reference).  keys[k * nchunks + i] = chunk i of grouping column k; a NULL in a grouping column is a group value of
 * its own; NULL values are skipped; counts[g] = non-NULL values of group g (rows, for RDF_AGG_COUNT / values == NULL).
 * out_keys: nkeys one-chunk outputs (key dtypes; validity required for a nullable grouping column);
 * out_values: Float64 for float values, UInt64 for MIN / MAX of UInt64 values, Int64 otherwise (sums wrap);
 *   MIN / MAX of a group without a non-NULL value is NULL (validity required when the value column is nullable);
 *   NaN never wins a MIN / MAX unless every value of the group is NaN (the column aggregates' rule, rdf_min / rdf_max);
 * out_counts: Int64.  Capacities >= min(max_groups, rows) + 2; group order unspecified.  More than max_groups distinct key
 * tuples -> RDF_MEMORY_ERROR.  Several grouping columns are packed into one 64-bit key after range compression
 * (bits(max - min) per column, + 1 code for NULL); when the ranges together pass 64 bits (several sparse columns) the
 * widest columns are dictionary-coded instead (rank among the column's distinct values, found by a count-only GROUP BY
 * of that column).  Tuples needing more than 64 bits either way -> RDF_INVALID_ARGUMENT.
 * f64 sums are accumulated with hardware atomics: the rounding order is not deterministic (<= 1e-6 relative). */
rdf_status rdf_groupby_agg(const rdf_array* keys, int32_t nkeys, const rdf_array* values, int64_t nchunks, int32_t agg,
                           int64_t max_groups, rdf_out* out_keys, rdf_out* out_values, rdf_out* out_counts);

/* Merge of partial groups: (key, partial aggregate, count) triples -> one row per key, partials combined by `agg`
 * (sums added, minima / maxima compared, counts added).  What a rank does with the partial groups it receives in the
 * multi-GPU GROUP BY (replaces the panic! at src/evaluation.rs:73 for RecordBatches sharded over GPUs).
 * One chunk each: keys Int64 / UInt64, partial Float64 / Int64 / UInt64 (may be NULL for counts only), counts Int64;
 * a partial with count 0 (or a NULL one) contributes nothing but its key.  Outputs as rdf_groupby_agg. */
rdf_status rdf_groupby_merge(const rdf_array* keys, const rdf_array* partial, const rdf_array* counts, int32_t agg,
                             int64_t max_groups, rdf_out* out_keys, rdf_out* out_values, rdf_out* out_counts);

/* The exchange itself, device-resident: rows (key, partial, count) of this rank's local groups are bucketed by owning
 * rank, owner = ((key * 0x9E3779B97F4A7C15) >> 33) % world, into `packed_dev` — n rows of 3 x 64-bit words, the rows of
 * owner 0 first — and owner_counts[r] (host) receives the number of rows for rank r: exactly the send buffer and split
 * sizes of an all_to_all(v) (RCCL over xGMI on the GPU box).  _unpack splits a received buffer back into three columns
 * for rdf_groupby_merge.  RDF_MEM_DEVICE only; 8-byte key and partial dtypes; no NULL keys. */
rdf_status rdf_group_exchange_pack(const rdf_array* keys, const rdf_array* partial, const rdf_array* counts, int32_t world,
                                   void* packed_dev, int64_t* owner_counts);
rdf_status rdf_group_exchange_unpack(const void* packed_dev, int64_t n, rdf_out* keys, rdf_out* partial, rdf_out* counts);
/* The row-shuffle fallback of the same exchange: when a rank's rows hold about as many groups as rows, pre-aggregating
 * them gains nothing — the rows themselves are bucketed by owner (16 bytes each: key, value; no NULLs), exchanged the
 * same way, and every rank runs rdf_groupby_agg over the rows it received.  packed_dev: 16 * rows bytes. */
rdf_status rdf_row_exchange_pack(const rdf_array* keys, const rdf_array* values, int32_t world, void* packed_dev, int64_t* owner_counts);
rdf_status rdf_row_exchange_unpack(const void* packed_dev, int64_t n, rdf_out* keys, rdf_out* values);

/* ------------------------------------------------------------------ ArrayFunctions over List<primitive> columns */

/* A ListArray with primitive numeric children as the reference's ArrayFunctions see it (src/functions/array.rs):
 * row i is the slice values[value_offset(i) .. value_offset(i + 1)) of the child array.  `offsets` is the Int32
 * value_offsets buffer described as an RDF_I32 array of rows + 1 elements whose validity / offset / null_count
 * fields describe the LIST rows (bit i = list i is not NULL); `values` is the child array.  Child validity is
 * ignored, exactly as the reference's value_slice() ignores it. */
typedef struct {
    rdf_array offsets;
    rdf_array values;
} rdf_list_array;

/* ArrayFunctions::array_contains (array.rs:15-37): NULL list -> NULL, else whether the slice holds `value`
 * (a native scalar of the child dtype; float equality is IEEE: NaN equals nothing).  out: RDF_BOOL, rows elements. */
rdf_status rdf_list_contains(const rdf_list_array* list, const void* value, rdf_out* out);
/* ArrayFunctions::array_position (array.rs:233-260): 1-based position of the first occurrence, 0 when absent or
 * when the list is NULL (never NULL).  out: RDF_I32. */
rdf_status rdf_list_position(const rdf_list_array* list, const void* value, rdf_out* out);
/* ArrayFunctions::array_max / array_min (array.rs:182-231): per-row extremum, NULL for a NULL list.  An EMPTY
 * list gives NULL (the reference unwraps None and panics); floats are accepted, NaN loses against any number
 * (the column aggregates' rule).  out: child dtype. */
rdf_status rdf_list_max(const rdf_list_array* list, rdf_out* out);
rdf_status rdf_list_min(const rdf_list_array* list, rdf_out* out);
/* ArrayFunctions::array_remove (array.rs:262-292): every element equal to `value` dropped, order kept; a NULL
 * list becomes an empty (valid) list like the reference's ListBuilder::append(true).  out_offsets: RDF_I32 with
 * rows + 1 elements (starting at 0), out_values: child dtype, capacity >= the slice total. */
rdf_status rdf_list_remove(const rdf_list_array* list, const void* value, rdf_out* out_offsets, rdf_out* out_values);
/* ArrayFunctions::array_sort (array.rs:320-354): every row's slice sorted ascending (floats in IEEE total order);
 * the value_offsets do not change.  out_values: the child values of rows 0 .. rows-1 re-ordered, i.e.
 * values[value_offset(0) .. value_offset(rows)). */
rdf_status rdf_list_sort(const rdf_list_array* list, rdf_out* out_values);
/* The set-valued ArrayFunctions, each row rebuilt with the array_tool crate's Vec algebra (Cargo.toml:19) under the
 * element type's `==` (NaN equals nothing).  Outputs as for rdf_list_remove; a NULL row of `list` / `a` becomes an
 * empty valid list, `b`'s validity is not looked at (array.rs:82,126,372).
 *   array_distinct  (array.rs:39-65):   `unique()`      first occurrences, in order.  The reference never closes the
 *                                        row of a non-NULL list (no `b.append(true)`, a private, untested function);
 *                                        this entry point returns the evident per-row result.
 *   array_except    (array.rs:66-109):  `a.uniq(b)`     unique(a) without the members of b
 *   array_intersect (array.rs:110-153): `a.intersect(b)` unique(a) restricted to the members of b
 *   array_union     (array.rs:356-399): `a.union(b)`    unique(a ++ b)
 *   array_repeat    (array.rs:294-326): `times(count)`  the row's slice `count` times over (count >= 0)
 * a and b must have the same number of rows (RDF_COMPUTE_ERROR "Expected array a and b to have the same length")
 * and the same child dtype; a result with more than 2^31-1 elements is RDF_COMPUTE_ERROR (Int32 value_offsets).
 * out_values capacity: distinct / except / intersect <= elements of a, union <= a + b, repeat = a * count. */
rdf_status rdf_list_distinct(const rdf_list_array* list, rdf_out* out_offsets, rdf_out* out_values);
rdf_status rdf_list_except(const rdf_list_array* a, const rdf_list_array* b, rdf_out* out_offsets, rdf_out* out_values);
rdf_status rdf_list_intersect(const rdf_list_array* a, const rdf_list_array* b, rdf_out* out_offsets, rdf_out* out_values);
rdf_status rdf_list_union(const rdf_list_array* a, const rdf_list_array* b, rdf_out* out_offsets, rdf_out* out_values);
rdf_status rdf_list_repeat(const rdf_list_array* list, int32_t count, rdf_out* out_offsets, rdf_out* out_values);

/* ------------------------------------------------------------------ fused batch loop */

typedef enum {
    RDF_SINK_STORE = 0, /* materialise every value expression as a new column (Evaluate::calculate) */
    RDF_SINK_AGG = 1    /* fold every value expression into {sum,min,max,count} (AggregateFunctions) */
} rdf_sink;

#define RDF_MAX_VALUES 4

/* A maximal run of Calculate / Filter / aggregate steps of Evaluate::evaluate
 * (src/evaluation.rs:66-96) fused into one pass per batch. */
typedef struct {
    const rdf_expr_node* nodes;
    int32_t nnodes;
    int32_t filter_root;                 /* BooleanFilter root or -1: rows where it is false/null are dropped */
    int32_t nvalues;                     /* 1..RDF_MAX_VALUES */
    int32_t value_roots[RDF_MAX_VALUES];
    int32_t sink;                        /* rdf_sink */
} rdf_program;

/* AggregateFunctions results for one value expression.  sum/min/max are returned in the value's
 * class: f64 for F32/F64 values, i64 bit pattern (wrapped to the value's width) otherwise. */
typedef struct {
    double  sum_f64, min_f64, max_f64;
    int64_t sum_i64, min_i64, max_i64;
    int64_t count;    /* valid rows that passed the filter */
    int32_t is_some;  /* count > 0 */
    int32_t dtype;    /* value dtype */
} rdf_agg_result;

/* Host-resident batches (RDF_MEM_HOST) beyond one slab (256 MiB; rdf_set_option("stream_slab_bytes", n), -1 = never) are STREAMED
 * by every entry point that takes chunk lists — rdf_pipeline (both sinks), rdf_binary / rdf_unary / rdf_cast / rdf_hour /
 * rdf_predicate, the column aggregates, rdf_group_pipeline, rdf_groupby_agg (one Int64 / UInt64 key column), rdf_filter_pipeline:
 * slab k + 1 crosses the link while the kernel runs over slab k, results that are columns leave on a third stream, aggregates
 * are folded in slab order.  Float SUMS of a streamed call are therefore folded per slab (f64): their low bits depend on where the
 * slabs are cut (stream_slab_bytes, the batch lengths) and — with the run-time compiler working in the background, jit = 1 —
 * on which slab was the first to run compiled; every such result is within the 1e-6 relative the path promises for f64 sums, but
 * two calls need not agree bit for bit.  Integer results, extrema, counts, columns, masks and filtered rows do not depend on it.
 *
 * cols[c * nchunks + i].  SINK_STORE: outs[v * nchunks + i] receives value v of batch i
 * (filter_root must be -1: filter then store is rdf_predicate + rdf_filter_columns).
 * SINK_AGG: aggs[v] receives the aggregates, outs may be NULL. */
rdf_status rdf_pipeline(const rdf_program* prog, const rdf_array* cols, int32_t ncols, int64_t nchunks,
                        rdf_out* outs, rdf_agg_result* aggs);
/* DataFrame::filter(&BooleanFilter) (src/dataframe.rs:178-189) over HOST-resident RecordBatches in ONE call: the predicate
 * (BooleanFilter::eval_to_array, src/expression.rs:766-861) is evaluated and every column compacted on the device, slab by
 * slab — slab k + 1 crosses the link while slab k is filtered and slab k - 1's kept rows travel back on a third stream —,
 * so a frame of any size (larger than free HBM included) is filtered at the link's rate with two slabs of HBM.  Column
 * index c of the expression = column c of `cols`; cols / outs laid out [c * nchunks + i], RDF_MEM_HOST; outs[c * nchunks + i]
 * needs the capacity of its input batch (and a validity buffer where the column carries one) and receives the kept rows of
 * batch i (ChunkedArray::filter keeps batch boundaries, src/table.rs:97-107).  What rdf_predicate + rdf_filter_columns do in
 * two calls with the mask making a round trip through the host; a device-resident frame is filtered with rdf_filter_frame. */
rdf_status rdf_filter_pipeline(const rdf_expr_node* nodes, int32_t nnodes, int32_t root, const rdf_array* cols, int32_t ncols,
                               int64_t nchunks, rdf_out* outs);
/* Host-resident frames are STREAMED: when the RDF_MEM_HOST arrays of a SINK_AGG call hold more than one slab of bytes
 * (rdf_set_option("stream_slab_bytes", ...), default 256 MiB) the batch list is cut into slabs of whole batches (a longer batch
 * on 64-row boundaries), slab k + 1 crosses the link on the copy stream while the fused kernel runs over slab k, and the
 * slabs' partial aggregates are folded in slab order — the batch loop of Evaluate::evaluate (src/evaluation.rs:66-96) over
 * what a reader produced (DataFrame::from_csv / from_arrow, src/dataframe.rs:349-407), with two slabs of HBM in use whatever
 * the frame's size.  Column buffers in page-locked memory (rdf_host_alloc / rdf_host_register) of at least 256 KiB leave with
 * one asynchronous copy each; pageable memory and the readers' small batches are packed into a page-locked staging buffer by
 * a few host threads first.  rdf_stream_stats: slabs (0 = the last rdf_pipeline call of this thread was not streamed), bytes
 * that went through the staging buffer, bytes copied straight out of the caller's page-locked memory. */
rdf_status rdf_stream_stats(int64_t* slabs, int64_t* bytes_staged, int64_t* bytes_direct);

/* A frame handle for repeated calls over the same device-resident columns.  The reference's DataFrame holds its
 * RecordBatches for its lifetime (src/dataframe.rs:30-48) and every query walks them again; with the reader's 1024-row
 * batches (src/dataframe.rs:352) a 1e9-row frame is a million chunks, and validating / translating a million rdf_array
 * descriptors per call (14 ms) costs ten times the kernel (1.2 ms).  rdf_frame_pin does that work once: dtypes and batch
 * lengths checked, descriptors and tile tables kept in HBM.  cols[c * nchunks + i], RDF_MEM_DEVICE only; the buffers
 * must stay alive and unchanged until rdf_frame_release.  rdf_pipeline_frame = rdf_pipeline over the pinned columns
 * (column index c of the program = column c of the pin call); outputs and results exactly as rdf_pipeline.  A handle
 * belongs to the device it was pinned on and is used by one thread at a time. */
typedef struct rdf_frame rdf_frame;
rdf_status rdf_frame_pin(const rdf_array* cols, int32_t ncols, int64_t nchunks, rdf_frame** out);
rdf_status rdf_frame_release(rdf_frame* frame);
rdf_status rdf_pipeline_frame(const rdf_program* prog, rdf_frame* frame, rdf_out* outs, rdf_agg_result* aggs);

/* ------------------------------------------------------------------ frame-level operators
 *
 * DataFrame::filter / take / sort and GroupAggregate over a pinned frame, returning a NEW frame whose buffers the library
 * owns (rdf_frame_release gives them back).  Nothing per RecordBatch crosses to the host: descriptor tables are built by
 * kernels, so a frame held in the readers' 1024-row batches (src/dataframe.rs:352: 976 563 batches for 1e9 rows) costs what
 * its kernels cost — the list-taking entry points above pay O(batches) host work per call (walking rdf_array / rdf_out lists).
 * Frames returned here are ordinary frames: rdf_pipeline_frame, rdf_filter_frame, ... run over them (<= 8 columns for the
 * program-evaluating entry points), rdf_frame_column exports their descriptors.  Every batch of an owned frame starts on a
 * 64-row boundary of its column buffer (values 16-byte, bitmaps 8-byte aligned); null counts are reported as unknown (-1). */

/* columns, batches, rows of a frame */
rdf_status rdf_frame_info(rdf_frame* frame, int32_t* ncols, int64_t* nchunks, int64_t* rows);
/* The rdf_array descriptors of one column (nchunks entries, RDF_MEM_DEVICE, borrowed from the frame): the bridge back to
 * the list-taking entry points and to arrow::array::ArrayData on the Rust side.  O(batches) host work, once per frame. */
rdf_status rdf_frame_column(rdf_frame* frame, int32_t col, rdf_array* chunks);
/* DataFrame::filter(&BooleanFilter) (src/dataframe.rs:178-189): the predicate is evaluated over every batch
 * (BooleanFilter::eval_to_array, src/expression.rs:766-861) and EVERY column of the frame is compacted with it in one pass
 * (Column::filter per column in the reference); batch boundaries are kept (ChunkedArray::filter, src/table.rs:97-107), a
 * batch may become empty.  `root` must be boolean-typed; column index c of the expression = column c of the frame.  The frame may
 * have up to 64 columns; the predicate reads at most 8 of them. */
rdf_status rdf_filter_frame(rdf_frame* frame, const rdf_expr_node* nodes, int32_t nnodes, int32_t root, rdf_frame** out);
/* DataFrame::take's per-column loop (src/dataframe.rs:216-222; DataFrame::join's, :705-711) as ONE gather pass: the index
 * list is read once, each row is resolved to (batch, element) once, and the gathers of all columns of a row are in flight
 * together.  cols[c * nchunks + i]; indices: ONE RDF_U32 / RDF_U64 array; outs: ncols one-chunk outputs.  Semantics per
 * column exactly rdf_take's (null index -> null, out of range -> RDF_COMPUTE_ERROR). */
rdf_status rdf_take_columns(const rdf_array* cols, int32_t ncols, int64_t nchunks, const rdf_array* indices, rdf_out* outs);
/* The same over a pinned frame -> a one-batch frame of indices->length rows (indices: host or device memory). */
rdf_status rdf_take_frame(rdf_frame* frame, const rdf_array* indices, rdf_frame** out);
/* DataFrame::sort (src/dataframe.rs:194-222): lexsort_to_indices over columns sort_cols[0..nsort) of the frame (criterion 0
 * most significant, rdf_sort_to_indices' rules), then the take of every column by that order in one pass.  out_indices
 * (optional, RDF_U32, device memory) receives the row order; out (optional) the sorted one-batch frame. */
rdf_status rdf_sort_frame(rdf_frame* frame, const int32_t* sort_cols, int32_t nsort, const rdf_sort_options* opts,
                          rdf_out* out_indices, rdf_frame** out);
/* rdf_groupby_agg over columns of a frame: grouping columns key_cols[0..nkeys), ONE aggregation of column value_col
 * (ignored for RDF_AGG_COUNT).  -> a one-batch frame of nkeys + 2 columns: the group keys, the aggregate (rdf_groupby_agg's
 * output type), the counts (Int64).  One grouping column runs on the frame's own tables; several are packed first. */
rdf_status rdf_groupby_agg_frame(rdf_frame* frame, const int32_t* key_cols, int32_t nkeys, int32_t value_col, int32_t agg,
                                 int64_t max_groups, rdf_frame** out);

/* ------------------------------------------------------------------ fused grouped aggregation */

#define RDF_MAX_GROUP_VALUES 8
#define RDF_MAX_GROUP_SLOTS  1024   /* (ngroups + 1) * nvalues must not exceed this */

/* sum/count of one value expression inside one group.  sum in the value's class: f64 for F32/F64
 * values (accumulated in f64), wrapping i64 otherwise (widened, not re-wrapped to the value's width). */
typedef struct {
    double  sum_f64;
    int64_t sum_i64;
    int64_t count;    /* rows of the group that passed the filter and whose value is not NULL */
    int32_t is_some;  /* count > 0 */
    int32_t dtype;    /* value dtype */
} rdf_group_result;

/* Transformation::GroupAggregate(groups, [Sum|Average|Count ...]) after a run of Calculate/Filter steps,
 * for a SMALL DENSE group domain (TPC-H Q1's returnflag x linestatus; BASELINE.json config C5), fused into
 * one pass per batch: rows are dropped by `filter_root` (or kept when -1), `group_root` is an
 * integer-valued expression giving each row's group id in [0, ngroups) (e.g. flag * 2 + status over
 * dictionary codes; a NULL id lands in the extra group `ngroups`), and every value expression is summed
 * and counted per group: out[v * (ngroups + 1) + g].  group_rows[g] (may be NULL) = rows in group g after
 * the filter, i.e. SQL count(*).  An id outside [0, ngroups) at a row that passed the filter ->
 * RDF_COMPUTE_ERROR.  Averages are sum / count on the caller's side (AggregateFunctions::avg,
 * src/functions/aggregate.rs:32-65).  The reference plans this step (Dataset::try_aggregate,
 * src/expression.rs:114-221) but does not execute it (src/evaluation.rs:73 panics): semantics are SQL's,
 * parity unpinned by the reference.  f64 sums: accumulation order is not deterministic (<= 1e-6 rel).
 * Large or sparse key domains: rdf_groupby_sum. */
rdf_status rdf_group_pipeline(const rdf_expr_node* nodes, int32_t nnodes, int32_t filter_root, int32_t group_root,
                              int32_t ngroups, const int32_t* value_roots, int32_t nvalues,
                              const rdf_array* cols, int32_t ncols, int64_t nchunks,
                              rdf_group_result* out, int64_t* group_rows);
/* rdf_group_pipeline / rdf_predicate over the pinned columns (same results, same errors). */
rdf_status rdf_group_pipeline_frame(const rdf_expr_node* nodes, int32_t nnodes, int32_t filter_root, int32_t group_root, int32_t ngroups,
                                    const int32_t* value_roots, int32_t nvalues, rdf_frame* frame, rdf_group_result* out,
                                    int64_t* group_rows);
rdf_status rdf_predicate_frame(const rdf_expr_node* nodes, int32_t nnodes, int32_t root, rdf_frame* frame, rdf_out* mask);

/* ------------------------------------------------------------------ multi-GPU: the exchange behind the boundary
 *
 * Where the reference panics (Transformation::GroupAggregate, src/evaluation.rs:73) and everywhere its batch loop is
 * per-chunk independent (src/evaluation.rs:66-96, src/functions/aggregate.rs:88-90), RecordBatches shard over the GPUs of a
 * node by contiguous row ranges.  A communicator is ONE RANK's end of the group: rank r runs on one GPU and is
 * driven by one host thread (one process per GPU, or one thread per GPU inside a single process — the shape a single-process
 * library like the reference needs).  Two transports, same entry points:
 *   RDF_COMM_RCCL  librccl.so (loaded on first use, never linked): ncclAllGather for the split sizes and the partial
 *                  aggregates, grouped ncclSend / ncclRecv over xGMI for the all-to-all(v) of partial groups / rows, on the
 *                  communicator's own stream, ordered against the thread's compute stream by events;
 *   RDF_COMM_PEER  single process only, no RCCL: every rank pulls its share out of the other ranks' send buffers with
 *                  hipMemcpyPeerAsync (xGMI DMA between the devices of one process).  `devices` may name one GPU several
 *                  times (ranks sharing a GPU: how the N > 1 logic is tested on a one-GPU box).
 * Every rank must make the same sequence of collective calls (rdf_comm_barrier / _allgather, rdf_agg_combine,
 * rdf_group_combine, rdf_groupby_agg_dist, rdf_groupby_agg_frame_dist) with matching arguments; an error one rank meets
 * before its exchange is carried to every rank by the exchange itself (all of them return an error, nobody waits for a
 * peer that left).  A communicator belongs to the device it was created for and is used by one thread at a time. */
typedef struct rdf_comm rdf_comm;
#define RDF_COMM_ID_BYTES 128
#define RDF_COMM_MAX_RANKS 64
typedef enum { RDF_COMM_RCCL = 0, RDF_COMM_PEER = 1 } rdf_comm_kind;

/* ncclGetUniqueId: rank 0 makes one and hands it to every rank through whatever channel the host has (a file, a socket,
 * MPI, torch's store); every rank then calls rdf_comm_init_rank on the device it selected with rdf_set_device. */
rdf_status rdf_comm_unique_id(uint8_t id[RDF_COMM_ID_BYTES]);
rdf_status rdf_comm_init_rank(int32_t world, int32_t rank, const uint8_t id[RDF_COMM_ID_BYTES], rdf_comm** out);
/* Single process: all `ndev` communicators at once (ncclCommInitAll / the peer-copy transport); out[i] is then used by the
 * thread that called rdf_set_device(devices[i]). */
rdf_status rdf_comm_init_all(int32_t ndev, const int32_t* devices, int32_t kind, rdf_comm** out);
rdf_status rdf_comm_destroy(rdf_comm* comm);
/* world size, rank, device, transport, RCCL's version code (0 for RDF_COMM_PEER); any pointer may be NULL */
rdf_status rdf_comm_info(rdf_comm* comm, int32_t* world, int32_t* rank, int32_t* device, int32_t* kind, int32_t* rccl_version);
rdf_status rdf_comm_barrier(rdf_comm* comm);
/* `bytes` of host memory from every rank, in rank order, into all_host (world * bytes); bytes <= 1 MiB. */
rdf_status rdf_comm_allgather(rdf_comm* comm, const void* mine_host, int64_t bytes, void* all_host);

/* AggregateFunctions over row-sharded batches: every rank passes the rdf_agg_result of its shard (rdf_pipeline's SINK_AGG
 * output) and gets the aggregates of the whole column back — the partials are all-gathered (world x 72 bytes per value) and
 * folded in rank order on every rank, exactly AggregateFunctions' own left fold over chunks (src/functions/aggregate.rs:82-93)
 * with ranks in place of chunks: identical bits on every rank; integer sums wrap to the value's width; NaN never displaces a
 * number in min / max. */
rdf_status rdf_agg_combine(rdf_comm* comm, rdf_agg_result* aggs, int32_t nvalues);
/* rdf_pipeline (an aggregating program, RDF_SINK_AGG) + rdf_agg_combine in ONE call over this rank's device-resident shard:
 * the shard's partial {sum, min, max, count} never visit the host — they are all-gathered (RCCL, on the communicator's stream,
 * ordered behind the kernel by an event) and folded on the device by the same fixed-order tree on every rank, and the total is
 * the first thing the host reads: one wait per step instead of two (the per-step host round trip of the pair of calls is
 * latency, the same at any number of GPUs).  An error a rank's kernel raises (a zero divisor) is returned on every rank.
 * The peer transport, and host-resident batches (which may stream), run the two calls this replaces.
 * rdf_pipeline_frame_dist: the same over a pinned frame. */
rdf_status rdf_pipeline_dist(rdf_comm* comm, const rdf_program* prog, const rdf_array* cols, int32_t ncols, int64_t nchunks,
                             rdf_agg_result* aggs);
rdf_status rdf_pipeline_frame_dist(rdf_comm* comm, const rdf_program* prog, rdf_frame* frame, rdf_agg_result* aggs);
/* The same for rdf_group_pipeline's output (small dense group domain, TPC-H Q1): out[nvalues * (ngroups + 1)] and
 * group_rows[ngroups + 1] (may be NULL) are replaced by the totals over all ranks. */
rdf_status rdf_group_combine(rdf_comm* comm, rdf_group_result* out, int64_t* group_rows, int32_t ngroups, int32_t nvalues);

typedef enum { RDF_EXCHANGE_AUTO = 0, RDF_EXCHANGE_GROUPS = 1, RDF_EXCHANGE_ROWS = 2 } rdf_exchange_mode;
/* what the last exchange of a rank moved (bench lines, tests) */
typedef struct {
    int32_t exchange;            /* RDF_EXCHANGE_GROUPS or RDF_EXCHANGE_ROWS: what travelled */
    int32_t rounds;              /* grouped send / recv rounds (every rank runs the same number) */
    int64_t local_groups;        /* partial groups of this rank before the exchange (rows, for RDF_EXCHANGE_ROWS) */
    int64_t rows_sent, rows_sent_remote, rows_received;   /* records to all owners / to other ranks / from all ranks */
    int64_t bytes_sent, bytes_sent_remote, bytes_received;
    double  exchange_ms;         /* device time from the start of the pack to the end of the unpack (hipEvents) */
} rdf_exchange_stats;

/* Transformation::GroupAggregate over row-sharded batches (replaces the panic! at src/evaluation.rs:73 for N GPUs; SURVEY.md
 * 8e): this rank's shard is aggregated locally (rdf_groupby_agg), its partial groups are bucketed by owner on the device
 * (owner = ((key * 0x9E3779B97F4A7C15) >> 33) % world, rdf_group_exchange_pack's rule), the (key, partial, count) triples
 * travel to their owners, and every rank merges what it receives (rdf_groupby_merge: sums added, minima / maxima compared,
 * counts added).  The outputs hold the groups THIS RANK OWNS — the union over the ranks is the result, no key twice.
 * RDF_EXCHANGE_ROWS shuffles the shard's rows instead (16 bytes each) and aggregates them once, at their owner: what pays
 * when pre-aggregation cannot shrink a shard (RDF_EXCHANGE_AUTO: when 2 * max_groups * world >= the rows of all ranks,
 * decided from an all-gather of the shard sizes so that every rank takes the same path; needs one chunk per column, no NULLs).
 * ONE grouping column of Int64 / UInt64 without NULLs (cast narrower keys first), device-resident inputs and outputs
 * (RDF_MEM_DEVICE); values / agg / max_groups / output types and capacities as rdf_groupby_agg (max_groups bounds the groups
 * of the whole result; capacity >= min(max_groups, rows of all ranks) + 2 is always enough).  stats may be NULL. */
rdf_status rdf_groupby_agg_dist(rdf_comm* comm, const rdf_array* keys, const rdf_array* values, int64_t nchunks, int32_t agg,
                                int64_t max_groups, int32_t exchange, rdf_out* out_keys, rdf_out* out_values, rdf_out* out_counts,
                                rdf_exchange_stats* stats);
/* The same over a pinned frame: grouping column key_col, ONE aggregation of column value_col -> a one-batch frame of the
 * groups this rank owns: (key, aggregate, count) as rdf_groupby_agg_frame's. */
rdf_status rdf_groupby_agg_frame_dist(rdf_comm* comm, rdf_frame* frame, int32_t key_col, int32_t value_col, int32_t agg,
                                      int64_t max_groups, int32_t exchange, rdf_frame** out, rdf_exchange_stats* stats);

/* ------------------------------------------------------------------ synthetic data (bench/tests) */

/* x[row] = lo + (hi-lo) * u(seed, column_id, first_row + row), u in [0,1) from a counter-based
 * SplitMix64 hash; identical on every rank/device and in the CPU oracle.  Device memory only. */
rdf_status rdf_fill_uniform_f64(double* dev_ptr, int64_t n, uint64_t seed, uint64_t column_id,
                                int64_t first_row, double lo, double hi);
/* x[row] = lo + (hash mod span), span = hi - lo (> 0). */
rdf_status rdf_fill_uniform_i64(int64_t* dev_ptr, int64_t n, uint64_t seed, uint64_t column_id,
                                int64_t first_row, int64_t lo, int64_t hi);
/* validity bit = (hash(seed, column_id, row) mod 2^32) >= null_fraction * 2^32; nbits bits written. */
rdf_status rdf_fill_validity(uint8_t* dev_ptr, int64_t nbits, uint64_t seed, uint64_t column_id,
                             int64_t first_row, double null_fraction);

/* Per-thread tunables, for tests and ablations: "spec" (1 = use the ahead-of-time specialised kernels
 * when the program shape is in the catalogs, default; 0 = always the general evaluator), "fast_filter",
 * "vec_bitmap" (accepted and ignored since round 5: the specialised kernels read bitmap words on the scalar unit only), "gb_partition" (hash GROUP BY: 3 = second
 * generation, default: LDS-table stream <= 2048 groups, line-aligned scatter + LDS tables <= 1.3 M, else one table in HBM;
 * 4 = its scatter path whatever max_groups says; 1 = first-generation histogram + scatter; 2 = radix-sort partitioning;
 * 0 = one table in HBM),
 * "gb_debug" (1 / 2: ablations of the aggregate / scatter pass, results invalid; 3: force the skew variant),
 * "jit" (a program shape no catalog holds: 1 = the specialised kernel template is compiled for it at run time — `hipcc` as a child
 * process on a helper thread, about half a second; THIS call and the ones until it is ready are answered by the general evaluator,
 * a code object found in the cache directory is loaded at once —, default; 2 = the call waits for the compiler; 0 = always the
 * general evaluator),
 * "gb_compact" (scatter path, keys inside a window of 2^39: 1 = 4-byte records when only rows are counted, default; 2 = also
 * 12-byte records for sum / min / max; 0 = 16-byte records always), "gb_skew_plan" (1 = per-partition capacity plan when the probe finds a few heavy
 * partitions, default; 0 = the first-generation combining path instead; 2 = always),
 * "filter_tile" (0: compaction tile from the mean chunk length; 1024 / 4096 force one), "filter_one" (one-chunk
 * compaction kernel with kernel-argument descriptors, default on), "take_rows" (rdf_take_frame / rdf_sort_frame: 1 = gather
 * interleaved row records when the index list is long and the frame wide, default; 0 = always column by column; 2 = always records),
 * "gb_hot" (skewed keys on the scatter path: 1 = the few dozen hash classes that hold the heavy hitters get a pass of their own — folded in
 * LDS per block — and the scatter takes the other rows, default; 0 = capacity plan / first-generation combining path),
 * "gb_bucket" (partition tables of the scatter path's aggregate pass: 0 = one key per probe for keys packed into at most 4 x max_groups
 * values — the multiplicative hash never collides there —, four keys per 32-byte bucket otherwise, default; 1 / 4 force one),
 * "filter_fused" (rdf_filter_frame with a `column CMP literal [AND | OR column CMP literal]` predicate over 4- / 8-byte columns: 1 = the
 * predicate runs inside the compaction kernel, one pass, default; 2 = the same (it forced the kernel on long batches while those took
 * three passes); 0 = predicate -> mask, count, compact),
 * "filter_block" (rdf_filter_frame's one-pass form and rdf_filter / rdf_filter_columns over device-resident chunks, on frames of equally wide 4- / 8-byte columns in LONG batches: 1 = block tiles held in registers, every tile's row count
 * published one iteration before its offset is asked for, offsets from one scanner wave, default; 0 = the wave-tile kernels),
 * "filter_block_rows" (the mean batch length from which that kernel is taken, default 8192),
 * "filter_mixed" (rdf_filter_frame's one-pass form over frames of 8- AND 4-byte columns: the block kernel twice — the columns of the predicate's
 * width first, which also writes the kept rows into the frame's own mask, then the other width's columns by that mask; measured behind the
 * wave-tile kernel, which takes any widths: 0 = never, default; 1 = where that kernel's 1024-row tiles come out partial; 2 = wherever the
 * block kernel's forms apply),
 * "filter_ends" (the wave-tile one-pass kernel on frames whose batch lengths are not multiples of its 1024-row tile: 1 = the tiles at the end
 * of a batch take the LDS-DMA path too, default; 0 = they load row by row),
 * "filter_short" (the same operators on frames whose batches are no longer than one block tile — the readers' 1024-row RecordBatches,
 * chunks of a few thousand rows: 1 = the block kernel with a batch on 1 / 2 / 4 / 8 waves of one block and no prefix between blocks,
 * default; 0 = the wave-tile kernels; "filter_block" 0 switches both forms off),
 * "interp_lean" (interpreted programs — no specialised kernel, no compiler — over at most 4 columns of 8-byte types whose every step is a
 * comparison, f64 / 64-bit integer arithmetic, a Boolean connective, an integer -> f64 cast or the filter, aggregated or stored: 1 = the
 * interpreter's branch-free kernel with host-assigned step handlers (eval_lean_kernel, 1.4-2.2 x the general kernel on aggregates), default;
 * 2 = the same with one tile per trip of its step loop (the A/B of its two-tile form); 0 = eval_kernel),
 * "filter_lookback" (the wave-tile kernel on batches longer than a tile: 3 = one tile per 64 finds the rows in front of them all, from tile
 * counts and older totals, default; 2 = from totals only; 1 = every tile walks the totals and batches beyond 1 048 576 rows take the three passes),
 * "join_table" (equi-join on one key column: 2 = the build side sorted by hash, the table of its distinct keys laid out by a scan,
 * default; 1 = sorted by key, table slots claimed by compare-and-swap; 0 = a bucket index over the sorted build keys),
 * "sort_msd" (keys that vary in 25 bits or more: 1 = passes over the top bits, buckets finished in LDS, default; 0 = one pass per byte),
 * "sort_pipe" (the digit passes: 0 = decoupled look-back between the tiles, default; 1 = a tile's digit counts go out one iteration before
 * its offsets are asked for and scanner blocks turn counts into offsets — round 6's experiment, measured slower, kept for A/B),
 * "sort_sample" (Float64 sort keys: 1 = the value buckets of those passes are planned from a sample of the keys — the range the rows lie
 * in without far outliers / infinities / NaNs, as many bucket bits as the densest region needs —, default; 0 = [min, max], ~500 rows per bucket),
 * "stream_slab_bytes" (rdf_pipeline over host memory: bytes per slab of the streamed batch loop, 0 = 256 MiB, -1 = never stream),
 * "comm_max_bytes" (most bytes one ncclSend / peer copy of the group-by exchange moves, default 256 MiB: larger shares go in
 * several rounds; every rank of a communicator must use the same value). */
rdf_status rdf_set_option(const char* name, int64_t value);
/* Number of program shapes with a specialised kernel. */
int32_t    rdf_spec_catalog_size(void);
/* One line about the run-time compiler of shapes outside the catalogs: whether `hipcc` and the kernel sources were found (and
 * where), the code-object cache directory ($RDF_JIT_CACHE, else $XDG_CACHE_HOME/rdf_mi355x/jit, else ~/.cache/rdf_mi355x/jit;
 * RDF_JIT_CACHE=off disables it), and how many kernels were compiled / read from the cache / failed / are being compiled.
 * Without a compiler such programs run on the interpreter (same results; 1.4-2.5 x slower on its lean kernel's class, 3-20 x otherwise) unless the cache holds them. */
const char* rdf_jit_status(void);
/* Name of the dominant kernel the last rdf_pipeline-family call of this thread launched. */
const char* rdf_last_kernel(void);

/* Average duration (ms) and launch count of the dominant kernel launched by this thread since the
 * last reset, from hipEvents recorded on the stream the kernels run on (bench.py's roofline leg). */
rdf_status rdf_kernel_timing_reset(int32_t enable);
rdf_status rdf_kernel_timing_get(double* total_ms, int64_t* launches);

/* Measurement helper, not an operator: what BARE streaming kernels reach on this device, GB/s of bytes moved — kind 0: read `bytes`
 * of a (xor fold, the cheapest consumer), 1: copy a -> b, 2: a, b -> c (two reads, one write).  Several loop shapes and grids are
 * timed with HIP events on the library's stream, `reps` launches each after one warm-up; the best is returned with its description
 * in `shape` (may be NULL).  These are the denominators bench.py's `roofline.peak_measured` and DESIGN.md's memory model use; the
 * reference has no counterpart (its benches time kernels on the CPU, src/functions/scalar.rs:621-671). */
rdf_status rdf_probe_stream(int32_t kind, const void* a, void* b, void* c, int64_t bytes, int32_t reps, double* best_gbps, char* shape, int32_t shape_len);

#ifdef __cplusplus
}
#endif
#endif /* RDF_MI355X_H */
