//! rdf_shim.rs — the binding a rust-dataframe maintainer would add (e.g. as `src/gpu/mod.rs`) to run the Arrow compute
//! hot path on an MI355X through `librdf_mi355x.so` (C ABI: `include/rdf_mi355x.h`).
//!
//! SOURCE ONLY: the build image has no Rust toolchain, so this file is neither compiled nor tested here.  The same ABI is
//! exercised end to end from C (`tests/test_abi.py` compiles the header with gcc and checks every export), from C++
//! (`include/rdf_frame.hpp`, `tests/cpp/`) and from Python / ctypes (`rust_dataframe_amd/_abi.py`).  Written against the
//! reference's pinned-era `arrow` crate (no `arrow::ffi`): raw buffer pointers are passed.
//!
//! Layout: (1) `sys` — the declarations, one per export of the header; (2) views over Arrow arrays and output buffers;
//! (3) replacement bodies with the reference's own signatures for the functions of SURVEY.md section 8b.
#![allow(non_camel_case_types, dead_code)]

pub mod sys {
    use std::os::raw::{c_char, c_void};

    // rdf_dtype / rdf_mem / rdf_status (include/rdf_mi355x.h)
    pub const RDF_I8: i32 = 0;  pub const RDF_I16: i32 = 1; pub const RDF_I32: i32 = 2;  pub const RDF_I64: i32 = 3;
    pub const RDF_U8: i32 = 4;  pub const RDF_U16: i32 = 5; pub const RDF_U32: i32 = 6;  pub const RDF_U64: i32 = 7;
    pub const RDF_F32: i32 = 8; pub const RDF_F64: i32 = 9; pub const RDF_BOOL: i32 = 10; pub const RDF_NULLTYPE: i32 = 11;
    pub const RDF_MEM_HOST: i32 = 0; pub const RDF_MEM_DEVICE: i32 = 1;
    pub const RDF_OK: i32 = 0; pub const RDF_COMPUTE_ERROR: i32 = 1; pub const RDF_DIVIDE_BY_ZERO: i32 = 2;
    pub const RDF_INVALID_ARGUMENT: i32 = 3; pub const RDF_MEMORY_ERROR: i32 = 4; pub const RDF_DEVICE_ERROR: i32 = 5;
    // rdf_op
    pub const RDF_OP_ADD: i32 = 1; pub const RDF_OP_SUB: i32 = 2; pub const RDF_OP_MUL: i32 = 3; pub const RDF_OP_DIV: i32 = 4;
    pub const RDF_OP_ATAN2: i32 = 5; pub const RDF_OP_HYPOT: i32 = 6; pub const RDF_OP_LOG: i32 = 7;
    pub const RDF_OP_ABS: i32 = 8; pub const RDF_OP_ACOS: i32 = 9; pub const RDF_OP_ASIN: i32 = 10; pub const RDF_OP_ATAN: i32 = 11;
    pub const RDF_OP_CBRT: i32 = 12; pub const RDF_OP_CEIL: i32 = 13; pub const RDF_OP_COS: i32 = 14; pub const RDF_OP_COSH: i32 = 15;
    pub const RDF_OP_DEGREES: i32 = 16; pub const RDF_OP_EXP: i32 = 17; pub const RDF_OP_EXPM1: i32 = 18; pub const RDF_OP_FLOOR: i32 = 19;
    pub const RDF_OP_LOG10: i32 = 20; pub const RDF_OP_LOG2: i32 = 21; pub const RDF_OP_RADIANS: i32 = 22; pub const RDF_OP_ROUND: i32 = 23;
    pub const RDF_OP_SIN: i32 = 24; pub const RDF_OP_SINH: i32 = 25; pub const RDF_OP_SQRT: i32 = 26; pub const RDF_OP_TAN: i32 = 27;
    pub const RDF_OP_TANH: i32 = 28; pub const RDF_OP_CAST: i32 = 29;
    pub const RDF_OP_GT: i32 = 30; pub const RDF_OP_GE: i32 = 31; pub const RDF_OP_EQ: i32 = 32; pub const RDF_OP_NE: i32 = 33;
    pub const RDF_OP_LT: i32 = 34; pub const RDF_OP_LE: i32 = 35; pub const RDF_OP_NOT: i32 = 36; pub const RDF_OP_AND: i32 = 37; pub const RDF_OP_OR: i32 = 38;
    pub const RDF_OP_HOUR_S: i32 = 39; pub const RDF_OP_HOUR_MS: i32 = 40; pub const RDF_OP_HOUR_US: i32 = 41; pub const RDF_OP_HOUR_NS: i32 = 42; pub const RDF_OP_HOUR_DAY: i32 = 43;
    pub const RDF_OP_COT: i32 = 44; pub const RDF_OP_SEC: i32 = 45; pub const RDF_OP_CSC: i32 = 46;
    pub const RDF_NODE_COLUMN: i32 = 0; pub const RDF_NODE_SCALAR: i32 = 1; pub const RDF_NODE_OP: i32 = 2;
    pub const RDF_SINK_STORE: i32 = 0; pub const RDF_SINK_AGG: i32 = 1;
    pub const RDF_JOIN_LEFT: i32 = 0; pub const RDF_JOIN_RIGHT: i32 = 1; pub const RDF_JOIN_INNER: i32 = 2; pub const RDF_JOIN_FULL: i32 = 3;
    pub const RDF_AGG_SUM: i32 = 0; pub const RDF_AGG_MIN: i32 = 1; pub const RDF_AGG_MAX: i32 = 2; pub const RDF_AGG_COUNT: i32 = 3;
    pub const RDF_MAX_VALUES: usize = 4; pub const RDF_MAX_GROUP_KEYS: usize = 4;

    #[repr(C)] #[derive(Clone, Copy)]
    pub struct rdf_array { pub values: *const c_void, pub validity: *const u8, pub offset: i64, pub length: i64,
                           pub null_count: i64, pub dtype: i32, pub mem: i32 }
    #[repr(C)] #[derive(Clone, Copy)]
    pub struct rdf_out { pub values: *mut c_void, pub validity: *mut u8, pub capacity: i64, pub length: i64,
                         pub null_count: i64, pub dtype: i32, pub mem: i32 }
    #[repr(C)] #[derive(Clone, Copy)]
    pub struct rdf_expr_node { pub kind: i32, pub op: i32, pub dtype: i32, pub lhs: i32, pub rhs: i32, pub column: i32,
                               pub f64_: f64, pub i64_: i64 }
    #[repr(C)] pub struct rdf_program { pub nodes: *const rdf_expr_node, pub nnodes: i32, pub filter_root: i32, pub nvalues: i32,
                                        pub value_roots: [i32; RDF_MAX_VALUES], pub sink: i32 }
    #[repr(C)] #[derive(Clone, Copy, Default)]
    pub struct rdf_agg_result { pub sum_f64: f64, pub min_f64: f64, pub max_f64: f64, pub sum_i64: i64, pub min_i64: i64,
                                pub max_i64: i64, pub count: i64, pub is_some: i32, pub dtype: i32 }
    #[repr(C)] #[derive(Clone, Copy, Default)]
    pub struct rdf_group_result { pub sum_f64: f64, pub sum_i64: i64, pub count: i64, pub is_some: i32, pub dtype: i32 }
    #[repr(C)] #[derive(Clone, Copy)] pub struct rdf_sort_options { pub descending: i32, pub nulls_first: i32 }
    #[repr(C)] #[derive(Clone, Copy)] pub struct rdf_list_array { pub offsets: rdf_array, pub values: rdf_array }
    #[repr(C)] pub struct rdf_frame { _opaque: [u8; 0] }
    #[repr(C)] pub struct rdf_comm { _opaque: [u8; 0] }
    #[repr(C)] #[derive(Clone, Copy, Default)]
    pub struct rdf_exchange_stats { pub exchange: i32, pub rounds: i32, pub local_groups: i64, pub rows_sent: i64, pub rows_sent_remote: i64,
                                    pub rows_received: i64, pub bytes_sent: i64, pub bytes_sent_remote: i64, pub bytes_received: i64, pub exchange_ms: f64 }
    pub const RDF_COMM_RCCL: i32 = 0; pub const RDF_COMM_PEER: i32 = 1;
    pub const RDF_EXCHANGE_AUTO: i32 = 0; pub const RDF_EXCHANGE_GROUPS: i32 = 1; pub const RDF_EXCHANGE_ROWS: i32 = 2;

    #[link(name = "rdf_mi355x")]
    extern "C" {
        // plumbing
        pub fn rdf_version() -> *const c_char;
        pub fn rdf_last_error() -> *const c_char;
        pub fn rdf_device_count(count: *mut i32) -> i32;
        pub fn rdf_set_device(device: i32) -> i32;
        pub fn rdf_set_stream(hip_stream: *mut c_void) -> i32;
        pub fn rdf_synchronize() -> i32;
        pub fn rdf_dev_alloc(ptr: *mut *mut c_void, bytes: i64) -> i32;
        pub fn rdf_dev_free(ptr: *mut c_void) -> i32;
        pub fn rdf_copy_h2d(dst_dev: *mut c_void, src_host: *const c_void, bytes: i64) -> i32;
        pub fn rdf_copy_d2h(dst_host: *mut c_void, src_dev: *const c_void, bytes: i64) -> i32;
        // ScalarFunctions (src/functions/scalar.rs)
        pub fn rdf_binary(op: i32, a: *const rdf_array, b: *const rdf_array, nchunks: i64, out: *mut rdf_out) -> i32;
        pub fn rdf_unary(op: i32, a: *const rdf_array, nchunks: i64, out: *mut rdf_out) -> i32;
        pub fn rdf_cast(a: *const rdf_array, nchunks: i64, out: *mut rdf_out) -> i32;
        pub fn rdf_hour(a: *const rdf_array, nchunks: i64, unit: i32, out: *mut rdf_out) -> i32;
        // AggregateFunctions (src/functions/aggregate.rs)
        pub fn rdf_sum(a: *const rdf_array, nchunks: i64, out_scalar: *mut c_void, out_is_some: *mut i32) -> i32;
        pub fn rdf_min(a: *const rdf_array, nchunks: i64, out_scalar: *mut c_void, out_is_some: *mut i32) -> i32;
        pub fn rdf_max(a: *const rdf_array, nchunks: i64, out_scalar: *mut c_void, out_is_some: *mut i32) -> i32;
        pub fn rdf_count(a: *const rdf_array, nchunks: i64, out_count: *mut i64, out_is_some: *mut i32) -> i32;
        pub fn rdf_avg(a: *const rdf_array, nchunks: i64, out_mean: *mut f64, out_is_some: *mut i32) -> i32;
        // BooleanFilter / filter / take (src/expression.rs:766-861, src/table.rs:97-107,213-241)
        pub fn rdf_predicate(nodes: *const rdf_expr_node, nnodes: i32, root: i32, cols: *const rdf_array, ncols: i32,
                             nchunks: i64, mask: *mut rdf_out) -> i32;
        pub fn rdf_filter_count(mask: *const rdf_array, nchunks: i64, counts: *mut i64) -> i32;
        pub fn rdf_filter(col: *const rdf_array, mask: *const rdf_array, nchunks: i64, out: *mut rdf_out) -> i32;
        pub fn rdf_filter_columns(cols: *const rdf_array, ncols: i32, mask: *const rdf_array, nchunks: i64, outs: *mut rdf_out) -> i32;
        // DataFrame::filter over host-resident batches in one streamed call (src/dataframe.rs:178-189)
        pub fn rdf_filter_pipeline(nodes: *const rdf_expr_node, nnodes: i32, root: i32, cols: *const rdf_array, ncols: i32,
                                   nchunks: i64, outs: *mut rdf_out) -> i32;
        pub fn rdf_take(chunks: *const rdf_array, nchunks: i64, indices: *const rdf_array, out: *mut rdf_out) -> i32;
        // sort / join (src/dataframe.rs:194-222, src/functions/join.rs:19-137)
        pub fn rdf_sort_to_indices(cols: *const rdf_array, ncols: i32, nchunks: i64, opts: *const rdf_sort_options,
                                   out_indices: *mut rdf_out) -> i32;
        pub fn rdf_equijoin_indices(left_keys: *const rdf_array, left_nchunks: i64, right_keys: *const rdf_array, right_nchunks: i64,
                                    join_type: i32, out_left: *mut rdf_out, out_right: *mut rdf_out, out_rows: *mut i64) -> i32;
        pub fn rdf_equijoin_indices_multi(left_keys: *const rdf_array, left_nchunks: i64, right_keys: *const rdf_array,
                                          right_nchunks: i64, nkeys: i32, join_type: i32, out_left: *mut rdf_out,
                                          out_right: *mut rdf_out, out_rows: *mut i64) -> i32;
        // Transformation::GroupAggregate (planned by Dataset::try_aggregate, src/expression.rs:114-221; evaluation.rs:73 panics)
        pub fn rdf_groupby_sum(keys: *const rdf_array, values: *const rdf_array, nchunks: i64, max_groups: i64,
                               out_keys: *mut rdf_out, out_sums: *mut rdf_out, out_counts: *mut rdf_out) -> i32;
        pub fn rdf_groupby_agg(keys: *const rdf_array, nkeys: i32, values: *const rdf_array, nchunks: i64, agg: i32, max_groups: i64,
                               out_keys: *mut rdf_out, out_values: *mut rdf_out, out_counts: *mut rdf_out) -> i32;
        pub fn rdf_groupby_merge(keys: *const rdf_array, partial: *const rdf_array, counts: *const rdf_array, agg: i32, max_groups: i64,
                                 out_keys: *mut rdf_out, out_values: *mut rdf_out, out_counts: *mut rdf_out) -> i32;
        pub fn rdf_group_exchange_pack(keys: *const rdf_array, partial: *const rdf_array, counts: *const rdf_array, world: i32,
                                       packed_dev: *mut c_void, owner_counts: *mut i64) -> i32;
        pub fn rdf_group_exchange_unpack(packed_dev: *const c_void, n: i64, keys: *mut rdf_out, partial: *mut rdf_out,
                                         counts: *mut rdf_out) -> i32;
        pub fn rdf_row_exchange_pack(keys: *const rdf_array, values: *const rdf_array, world: i32, packed_dev: *mut c_void,
                                     owner_counts: *mut i64) -> i32;
        pub fn rdf_row_exchange_unpack(packed_dev: *const c_void, n: i64, keys: *mut rdf_out, values: *mut rdf_out) -> i32;
        // several GPUs: the communicator and every collective of the path live behind the boundary —
        // the library loads RCCL itself; Evaluate::evaluate's GroupAggregate arm (src/evaluation.rs:73) calls rdf_groupby_agg_dist
        pub fn rdf_comm_unique_id(id: *mut u8) -> i32;                                   // 128 bytes (ncclUniqueId)
        pub fn rdf_comm_init_rank(world: i32, rank: i32, id: *const u8, out: *mut *mut rdf_comm) -> i32;
        pub fn rdf_comm_init_all(ndev: i32, devices: *const i32, kind: i32, out: *mut *mut rdf_comm) -> i32;
        pub fn rdf_comm_destroy(comm: *mut rdf_comm) -> i32;
        pub fn rdf_comm_info(comm: *mut rdf_comm, world: *mut i32, rank: *mut i32, device: *mut i32, kind: *mut i32, rccl_version: *mut i32) -> i32;
        pub fn rdf_comm_barrier(comm: *mut rdf_comm) -> i32;
        pub fn rdf_comm_allgather(comm: *mut rdf_comm, mine_host: *const c_void, bytes: i64, all_host: *mut c_void) -> i32;
        pub fn rdf_agg_combine(comm: *mut rdf_comm, aggs: *mut rdf_agg_result, nvalues: i32) -> i32;
        pub fn rdf_pipeline_dist(comm: *mut rdf_comm, prog: *const rdf_program, cols: *const rdf_array, ncols: i32, nchunks: i64,
                                 aggs: *mut rdf_agg_result) -> i32;
        pub fn rdf_pipeline_frame_dist(comm: *mut rdf_comm, prog: *const rdf_program, frame: *mut rdf_frame, aggs: *mut rdf_agg_result) -> i32;
        pub fn rdf_group_combine(comm: *mut rdf_comm, out: *mut rdf_group_result, group_rows: *mut i64, ngroups: i32, nvalues: i32) -> i32;
        pub fn rdf_groupby_agg_dist(comm: *mut rdf_comm, keys: *const rdf_array, values: *const rdf_array, nchunks: i64, agg: i32,
                                    max_groups: i64, exchange: i32, out_keys: *mut rdf_out, out_values: *mut rdf_out,
                                    out_counts: *mut rdf_out, stats: *mut rdf_exchange_stats) -> i32;
        pub fn rdf_groupby_agg_frame_dist(comm: *mut rdf_comm, frame: *mut rdf_frame, key_col: i32, value_col: i32, agg: i32,
                                          max_groups: i64, exchange: i32, out: *mut *mut rdf_frame, stats: *mut rdf_exchange_stats) -> i32;
        pub fn rdf_group_pipeline(nodes: *const rdf_expr_node, nnodes: i32, filter_root: i32, group_root: i32, ngroups: i32,
                                  value_roots: *const i32, nvalues: i32, cols: *const rdf_array, ncols: i32, nchunks: i64,
                                  out: *mut rdf_group_result, group_rows: *mut i64) -> i32;
        // ArrayFunctions over List<primitive> (src/functions/array.rs:15-399)
        pub fn rdf_list_contains(list: *const rdf_list_array, value: *const c_void, out: *mut rdf_out) -> i32;
        pub fn rdf_list_position(list: *const rdf_list_array, value: *const c_void, out: *mut rdf_out) -> i32;
        pub fn rdf_list_max(list: *const rdf_list_array, out: *mut rdf_out) -> i32;
        pub fn rdf_list_min(list: *const rdf_list_array, out: *mut rdf_out) -> i32;
        pub fn rdf_list_remove(list: *const rdf_list_array, value: *const c_void, out_offsets: *mut rdf_out, out_values: *mut rdf_out) -> i32;
        pub fn rdf_list_sort(list: *const rdf_list_array, out_values: *mut rdf_out) -> i32;
        pub fn rdf_list_distinct(list: *const rdf_list_array, out_offsets: *mut rdf_out, out_values: *mut rdf_out) -> i32;
        pub fn rdf_list_except(a: *const rdf_list_array, b: *const rdf_list_array, out_offsets: *mut rdf_out, out_values: *mut rdf_out) -> i32;
        pub fn rdf_list_intersect(a: *const rdf_list_array, b: *const rdf_list_array, out_offsets: *mut rdf_out, out_values: *mut rdf_out) -> i32;
        pub fn rdf_list_union(a: *const rdf_list_array, b: *const rdf_list_array, out_offsets: *mut rdf_out, out_values: *mut rdf_out) -> i32;
        pub fn rdf_list_repeat(list: *const rdf_list_array, count: i32, out_offsets: *mut rdf_out, out_values: *mut rdf_out) -> i32;
        // the fused batch loop (src/evaluation.rs:66-96); host-resident frames above one slab are streamed (rdf_stream_stats says how)
        pub fn rdf_jit_status() -> *const c_char;          // the run-time compiler: found or not, cache directory, counts
        pub fn rdf_stream_stats(slabs: *mut i64, bytes_staged: *mut i64, bytes_direct: *mut i64) -> i32;
        pub fn rdf_pipeline(prog: *const rdf_program, cols: *const rdf_array, ncols: i32, nchunks: i64, outs: *mut rdf_out,
                            aggs: *mut rdf_agg_result) -> i32;
        // a frame pinned for repeated queries: descriptors validated once, tables kept in HBM (device-resident columns)
        pub fn rdf_frame_pin(cols: *const rdf_array, ncols: i32, nchunks: i64, out: *mut *mut rdf_frame) -> i32;
        pub fn rdf_frame_release(frame: *mut rdf_frame) -> i32;
        pub fn rdf_pipeline_frame(prog: *const rdf_program, frame: *mut rdf_frame, outs: *mut rdf_out, aggs: *mut rdf_agg_result) -> i32;
        pub fn rdf_group_pipeline_frame(nodes: *const rdf_expr_node, nnodes: i32, filter_root: i32, group_root: i32, ngroups: i32,
                                        value_roots: *const i32, nvalues: i32, frame: *mut rdf_frame, out: *mut rdf_group_result,
                                        group_rows: *mut i64) -> i32;
        pub fn rdf_predicate_frame(nodes: *const rdf_expr_node, nnodes: i32, root: i32, frame: *mut rdf_frame, mask: *mut rdf_out) -> i32;
        // frame in, frame out: DataFrame::filter / take / sort (src/dataframe.rs:178-222) and GroupAggregate without walking the batch list
        pub fn rdf_frame_info(frame: *mut rdf_frame, ncols: *mut i32, nchunks: *mut i64, rows: *mut i64) -> i32;
        pub fn rdf_frame_column(frame: *mut rdf_frame, col: i32, chunks: *mut rdf_array) -> i32;
        pub fn rdf_filter_frame(frame: *mut rdf_frame, nodes: *const rdf_expr_node, nnodes: i32, root: i32, out: *mut *mut rdf_frame) -> i32;
        pub fn rdf_take_columns(cols: *const rdf_array, ncols: i32, nchunks: i64, indices: *const rdf_array, outs: *mut rdf_out) -> i32;
        pub fn rdf_take_frame(frame: *mut rdf_frame, indices: *const rdf_array, out: *mut *mut rdf_frame) -> i32;
        pub fn rdf_sort_frame(frame: *mut rdf_frame, sort_cols: *const i32, nsort: i32, opts: *const rdf_sort_options,
                              out_indices: *mut rdf_out, out: *mut *mut rdf_frame) -> i32;
        pub fn rdf_groupby_agg_frame(frame: *mut rdf_frame, key_cols: *const i32, nkeys: i32, value_col: i32, agg: i32,
                                     max_groups: i64, out: *mut *mut rdf_frame) -> i32;
        // ingestion: pinned reader buffers, uploads that do not block the reader, one fence per load (src/dataframe.rs:349-407)
        pub fn rdf_host_alloc(ptr: *mut *mut c_void, bytes: i64) -> i32;
        pub fn rdf_host_free(ptr: *mut c_void) -> i32;
        pub fn rdf_host_register(ptr: *mut c_void, bytes: i64) -> i32;
        pub fn rdf_host_unregister(ptr: *mut c_void) -> i32;
        pub fn rdf_copy_h2d_async(dst_dev: *mut c_void, src_host_pinned: *const c_void, bytes: i64) -> i32;
        pub fn rdf_copy_fence() -> i32;
        // synthetic data, switches, introspection (bench / tests)
        pub fn rdf_fill_uniform_f64(dev_ptr: *mut f64, n: i64, seed: u64, column_id: u64, first_row: i64, lo: f64, hi: f64) -> i32;
        pub fn rdf_fill_uniform_i64(dev_ptr: *mut i64, n: i64, seed: u64, column_id: u64, first_row: i64, lo: i64, hi: i64) -> i32;
        pub fn rdf_fill_validity(dev_ptr: *mut u8, nbits: i64, seed: u64, column_id: u64, first_row: i64, null_fraction: f64) -> i32;
        pub fn rdf_set_option(name: *const c_char, value: i64) -> i32;
        pub fn rdf_spec_catalog_size() -> i32;
        pub fn rdf_last_kernel() -> *const c_char;
        pub fn rdf_kernel_timing_reset(enable: i32) -> i32;
        pub fn rdf_kernel_timing_get(total_ms: *mut f64, launches: *mut i64) -> i32;
        pub fn rdf_probe_stream(kind: i32, a: *const c_void, b: *mut c_void, c: *mut c_void, bytes: i64, reps: i32, best_gbps: *mut f64, shape: *mut c_char, shape_len: i32) -> i32;
    }
}

// ------------------------------------------------------------------------------------------------
// (2) views over Arrow arrays, output buffers, error mapping

use self::sys::*;
use arrow::array::{Array, ArrayData, ArrayRef, BooleanArray, ListArray, PrimitiveArray, UInt32Array};
use arrow::buffer::MutableBuffer;
use arrow::datatypes::{ArrowNumericType, ArrowPrimitiveType, DataType};
use arrow::error::ArrowError;
use std::ffi::CStr;
use std::os::raw::c_void;

/// DataType -> rdf_dtype for the types the hot path computes on (src/evaluation.rs:107-293).
pub fn rdf_dtype(dt: &DataType) -> i32 {
    match dt {
        DataType::Int8 => RDF_I8, DataType::Int16 => RDF_I16, DataType::Int32 => RDF_I32, DataType::Int64 => RDF_I64,
        DataType::UInt8 => RDF_U8, DataType::UInt16 => RDF_U16, DataType::UInt32 => RDF_U32, DataType::UInt64 => RDF_U64,
        DataType::Float32 => RDF_F32, DataType::Float64 => RDF_F64, DataType::Boolean => RDF_BOOL,
        // temporal arrays pass as their storage (rdf_hour takes the unit separately)
        DataType::Date32(_) | DataType::Time32(_) => RDF_I32,
        DataType::Date64(_) | DataType::Time64(_) | DataType::Timestamp(_, _) => RDF_I64,
        _ => RDF_NULLTYPE,
    }
}

/// A chunk as the library sees it: pointers into the array's own buffers, nothing copied.
pub fn view(a: &dyn Array) -> rdf_array {
    let d = a.data();
    rdf_array {
        values: d.buffers()[0].raw_data() as *const c_void,
        validity: d.null_buffer().map_or(std::ptr::null(), |b| b.raw_data()),
        offset: d.offset() as i64,
        length: d.len() as i64,
        null_count: d.null_count() as i64,
        dtype: rdf_dtype(d.data_type()),
        mem: RDF_MEM_HOST,
    }
}
pub fn views<T: ArrowPrimitiveType>(chunks: &[&PrimitiveArray<T>]) -> Vec<rdf_array> { chunks.iter().map(|c| view(*c)).collect() }

/// Caller-allocated output chunk (values + optional validity), 64-byte padded like every Arrow buffer.
pub struct OutBuf { values: MutableBuffer, validity: Option<MutableBuffer>, dtype: DataType, capacity: usize }
impl OutBuf {
    pub fn new(dtype: DataType, capacity: usize, nullable: bool) -> Self {
        let width = match rdf_dtype(&dtype) { RDF_BOOL => 0, RDF_I8 | RDF_U8 => 1, RDF_I16 | RDF_U16 => 2, RDF_I32 | RDF_U32 | RDF_F32 => 4, _ => 8 };
        let vbytes = if width == 0 { (capacity + 7) / 8 + 8 } else { capacity * width + 16 };
        let mut values = MutableBuffer::new(vbytes);
        values.resize(vbytes).unwrap();
        let validity = if nullable { let mut b = MutableBuffer::new((capacity + 7) / 8 + 8); b.resize((capacity + 7) / 8 + 8).unwrap(); Some(b) } else { None };
        OutBuf { values, validity, dtype, capacity }
    }
    pub fn as_out(&mut self) -> rdf_out {
        rdf_out { values: self.values.raw_data_mut() as *mut c_void,
                  validity: self.validity.as_mut().map_or(std::ptr::null_mut(), |b| b.raw_data_mut()),
                  capacity: self.capacity as i64, length: 0, null_count: 0, dtype: rdf_dtype(&self.dtype), mem: RDF_MEM_HOST }
    }
    /// Wrap what the library wrote (`length`, `null_count` come back in the rdf_out).
    pub fn finish(self, out: &rdf_out) -> ArrayRef {
        let mut b = ArrayData::builder(self.dtype).len(out.length as usize).null_count(out.null_count as usize).add_buffer(self.values.freeze());
        if let Some(v) = self.validity { b = b.null_bit_buffer(v.freeze()); }
        arrow::array::make_array(b.build())
    }
}

fn last_error() -> String { unsafe { CStr::from_ptr(rdf_last_error()).to_string_lossy().into_owned() } }
/// rdf_status -> ArrowError / DataFrameError (src/error.rs:6-21).
pub fn status(code: i32) -> Result<(), ArrowError> {
    match code {
        RDF_OK => Ok(()),
        RDF_DIVIDE_BY_ZERO => Err(ArrowError::DivideByZero),
        RDF_INVALID_ARGUMENT => Err(ArrowError::InvalidArgumentError(last_error())),
        RDF_MEMORY_ERROR => Err(ArrowError::MemoryError(last_error())),
        _ => Err(ArrowError::ComputeError(last_error())),
    }
}

// ------------------------------------------------------------------------------------------------
// (3) replacement bodies, same signatures as the reference

/// ScalarFunctions::add (src/functions/scalar.rs:16-34); subtract / multiply / divide / par_multiply alike.
pub fn add<T: ArrowNumericType>(left: Vec<&PrimitiveArray<T>>, right: Vec<&PrimitiveArray<T>>) -> Result<Vec<ArrayRef>, ArrowError> {
    binary_op(RDF_OP_ADD, &left, &right)
}
fn binary_op<T: ArrowNumericType>(op: i32, left: &[&PrimitiveArray<T>], right: &[&PrimitiveArray<T>]) -> Result<Vec<ArrayRef>, ArrowError> {
    let (a, b) = (views(left), views(right));
    let mut bufs: Vec<OutBuf> = left.iter().zip(right.iter())
        .map(|(l, r)| OutBuf::new(T::get_data_type(), l.len(), l.null_count() + r.null_count() > 0)).collect();
    let mut outs: Vec<rdf_out> = bufs.iter_mut().map(|b| b.as_out()).collect();
    status(unsafe { rdf_binary(op, a.as_ptr(), b.as_ptr(), a.len() as i64, outs.as_mut_ptr()) })?;
    Ok(bufs.into_iter().zip(outs.iter()).map(|(b, o)| b.finish(o)).collect())
}

/// ScalarFunctions::sin ... tanh, abs (src/functions/scalar.rs:106-452): `scalar_op(array, |x| x.sin())` per chunk.
pub fn unary_op<T: ArrowNumericType>(op: i32, array: Vec<&PrimitiveArray<T>>) -> Result<Vec<ArrayRef>, ArrowError> {
    let a = views(&array);
    let mut bufs: Vec<OutBuf> = array.iter().map(|x| OutBuf::new(T::get_data_type(), x.len(), x.null_count() > 0)).collect();
    let mut outs: Vec<rdf_out> = bufs.iter_mut().map(|b| b.as_out()).collect();
    status(unsafe { rdf_unary(op, a.as_ptr(), a.len() as i64, outs.as_mut_ptr()) })?;
    Ok(bufs.into_iter().zip(outs.iter()).map(|(b, o)| b.finish(o)).collect())
}

/// AggregateFunctions::sum (src/functions/aggregate.rs:82-93); min / max / count / avg alike.
pub fn sum<T: ArrowNumericType>(chunks: &[&PrimitiveArray<T>]) -> Result<Option<T::Native>, ArrowError> where T::Native: Default {
    let a = views(chunks);
    let (mut value, mut is_some) = (T::Native::default(), 0i32);
    status(unsafe { rdf_sum(a.as_ptr(), a.len() as i64, &mut value as *mut T::Native as *mut c_void, &mut is_some) })?;
    Ok(if is_some != 0 { Some(value) } else { None })
}

/// ChunkedArray::filter / Column::filter (src/table.rs:97-107, 213-215): chunk boundaries are kept.
/// NOTE: this per-call form marshals one descriptor per chunk on the host: on a column held in the readers' 1024-row batches
/// (1e9 rows = a million chunks) the call is host-bound (1 s of marshalling around a 2.5 ms kernel).  A DataFrame-level caller
/// (`DataFrame::filter`, `::take`, `::sort`) should pin the frame once and use `GpuFrame::filter / take / sort` below, which cost
/// what their kernels cost whatever the batch size; these two bodies are for one-off calls on a few large chunks.
pub fn filter_chunks<T: ArrowNumericType>(chunks: &[&PrimitiveArray<T>], mask: &[&BooleanArray]) -> Result<Vec<ArrayRef>, ArrowError> {
    let (c, m) = (views(chunks), mask.iter().map(|x| view(*x)).collect::<Vec<_>>());
    let mut counts = vec![0i64; m.len()];
    status(unsafe { rdf_filter_count(m.as_ptr(), m.len() as i64, counts.as_mut_ptr()) })?;
    let mut bufs: Vec<OutBuf> = chunks.iter().zip(counts.iter()).map(|(x, n)| OutBuf::new(T::get_data_type(), *n as usize, x.null_count() > 0)).collect();
    let mut outs: Vec<rdf_out> = bufs.iter_mut().map(|b| b.as_out()).collect();
    status(unsafe { rdf_filter(c.as_ptr(), m.as_ptr(), c.len() as i64, outs.as_mut_ptr()) })?;
    Ok(bufs.into_iter().zip(outs.iter()).map(|(b, o)| b.finish(o)).collect())
}

/// Column::take (src/table.rs:218-241): no `to_array()` concatenation copy; ONE output chunk.
pub fn take_chunks<T: ArrowNumericType>(chunks: &[&PrimitiveArray<T>], indices: &UInt32Array) -> Result<ArrayRef, ArrowError> {
    let (c, i) = (views(chunks), view(indices));
    let nullable = indices.null_count() > 0 || chunks.iter().any(|x| x.null_count() > 0);
    let mut buf = OutBuf::new(T::get_data_type(), indices.len(), nullable);
    let mut out = buf.as_out();
    status(unsafe { rdf_take(c.as_ptr(), c.len() as i64, &i, &mut out) })?;
    Ok(buf.finish(&out))
}

/// One rank's shard of a DataFrame sharded over the GPUs of a node by row ranges: what Evaluate::evaluate's
/// GroupAggregate arm (src/evaluation.rs:73, a panic in the reference) and AggregateFunctions run on when N > 1.
pub struct ShardedFrame { pub local: GpuFrame, comm: *mut rdf_comm }
impl ShardedFrame {
    /// `id`: rdf_comm_unique_id of rank 0, handed over by the host (file, socket, ...); the calling thread drives `local`'s device.
    pub fn new(local: GpuFrame, world: i32, rank: i32, id: &[u8; 128]) -> Result<ShardedFrame, ArrowError> {
        let mut comm: *mut rdf_comm = std::ptr::null_mut();
        status(unsafe { rdf_comm_init_rank(world, rank, id.as_ptr(), &mut comm) })?;
        Ok(ShardedFrame { local, comm })
    }
    /// GroupAggregate(group column, [agg(value column)]) over all shards -> the groups THIS rank owns (key, aggregate, count).
    pub fn group_aggregate(&self, key_col: i32, value_col: i32, agg: i32, max_groups: i64) -> Result<(GpuFrame, rdf_exchange_stats), ArrowError> {
        let (mut out, mut st) = (std::ptr::null_mut(), rdf_exchange_stats::default());
        status(unsafe { rdf_groupby_agg_frame_dist(self.comm, self.local.handle, key_col, value_col, agg, max_groups, RDF_EXCHANGE_AUTO, &mut out, &mut st) })?;
        Ok((GpuFrame { handle: out }, st))
    }
    /// AggregateFunctions::{sum, min, max, count} of a fused program over the whole sharded column: partials of the shard in,
    /// totals out, identical on every rank (rank-order fold).
    pub fn aggregate(&self, prog: &rdf_program, nvalues: usize) -> Result<Vec<rdf_agg_result>, ArrowError> {
        let mut aggs = vec![rdf_agg_result::default(); RDF_MAX_VALUES];
        // (kernel, all-gather of the partials and their fold stay on the device: one host wait per call)
        status(unsafe { rdf_pipeline_frame_dist(self.comm, prog, self.local.handle, aggs.as_mut_ptr()) })?;
        aggs.truncate(nvalues);
        Ok(aggs)
    }
}
impl Drop for ShardedFrame { fn drop(&mut self) { unsafe { rdf_comm_destroy(self.comm); } } }

/// A DataFrame resident in HBM: the handle is pinned once and every operator returns a new handle (released in Drop).
pub struct GpuFrame { handle: *mut rdf_frame }
impl Drop for GpuFrame { fn drop(&mut self) { unsafe { rdf_frame_release(self.handle); } } }
impl GpuFrame {
    /// DataFrame::filter (src/dataframe.rs:178-189): predicate over every batch, every column compacted in one pass.
    pub fn filter(&self, nodes: &[rdf_expr_node], root: i32) -> Result<GpuFrame, ArrowError> {
        let mut out: *mut rdf_frame = std::ptr::null_mut();
        status(unsafe { rdf_filter_frame(self.handle, nodes.as_ptr(), nodes.len() as i32, root, &mut out) })?;
        Ok(GpuFrame { handle: out })
    }
    /// DataFrame::sort (src/dataframe.rs:194-222): lexsort_to_indices + Column::take of every column.
    pub fn sort(&self, sort_cols: &[i32], opts: &[rdf_sort_options]) -> Result<GpuFrame, ArrowError> {
        let mut out: *mut rdf_frame = std::ptr::null_mut();
        status(unsafe { rdf_sort_frame(self.handle, sort_cols.as_ptr(), sort_cols.len() as i32, opts.as_ptr(), std::ptr::null_mut(), &mut out) })?;
        Ok(GpuFrame { handle: out })
    }
    /// DataFrame::take (src/dataframe.rs:216-222).
    pub fn take(&self, indices: &UInt32Array) -> Result<GpuFrame, ArrowError> {
        let (i, mut out) = (view(indices), std::ptr::null_mut());
        status(unsafe { rdf_take_frame(self.handle, &i, &mut out) })?;
        Ok(GpuFrame { handle: out })
    }
    /// (columns, batches, rows); rdf_frame_column(handle, c, views) then gives the device views of a column.
    pub fn shape(&self) -> (i32, i64, i64) {
        let (mut c, mut b, mut r) = (0i32, 0i64, 0i64);
        unsafe { rdf_frame_info(self.handle, &mut c, &mut b, &mut r) };
        (c, b, r)
    }
}

/// Transformation::GroupAggregate(groups, [aggregation]) — `Evaluate::evaluate` panics here today (src/evaluation.rs:73).
/// One aggregation over 1..4 integer grouping columns of any cardinality; keys[k][chunk].
pub fn group_aggregate(keys: &[Vec<rdf_array>], key_types: &[DataType], values: Option<&[rdf_array]>, agg: i32, value_out: DataType,
                       max_groups: usize) -> Result<(Vec<ArrayRef>, ArrayRef, ArrayRef), ArrowError> {
    let nchunks = keys[0].len();
    let flat: Vec<rdf_array> = keys.iter().flat_map(|k| k.iter().cloned()).collect();
    let cap = max_groups + 2;
    let mut kbufs: Vec<OutBuf> = key_types.iter().map(|t| OutBuf::new(t.clone(), cap, true)).collect();
    let mut kouts: Vec<rdf_out> = kbufs.iter_mut().map(|b| b.as_out()).collect();
    let (mut vbuf, mut cbuf) = (OutBuf::new(value_out, cap, true), OutBuf::new(DataType::Int64, cap, false));
    let (mut vout, mut cout) = (vbuf.as_out(), cbuf.as_out());
    status(unsafe { rdf_groupby_agg(flat.as_ptr(), keys.len() as i32, values.map_or(std::ptr::null(), |v| v.as_ptr()), nchunks as i64,
                                    agg, max_groups as i64, kouts.as_mut_ptr(), &mut vout, &mut cout) })?;
    Ok((kbufs.into_iter().zip(kouts.iter()).map(|(b, o)| b.finish(o)).collect(), vbuf.finish(&vout), cbuf.finish(&cout)))
}

/// ArrayFunctions::array_contains (src/functions/array.rs:15-37): a ListArray travels as two views.
pub fn list_view<T: ArrowNumericType>(array: &ListArray) -> rdf_list_array {
    let data = array.data();
    let child = array.values();
    let child = child.as_any().downcast_ref::<PrimitiveArray<T>>().unwrap();
    rdf_list_array {
        offsets: rdf_array { values: data.buffers()[0].raw_data() as *const c_void,            // i32 value_offsets, len + 1
                             validity: data.null_buffer().map_or(std::ptr::null(), |b| b.raw_data()),
                             offset: data.offset() as i64, length: array.len() as i64 + 1, null_count: -1, dtype: RDF_I32, mem: RDF_MEM_HOST },
        values: view(child),
    }
}
pub fn array_contains<T: ArrowNumericType>(array: &ListArray, val: T::Native) -> Result<ArrayRef, ArrowError> {
    let l = list_view::<T>(array);
    let mut buf = OutBuf::new(DataType::Boolean, array.len(), true);
    let mut out = buf.as_out();
    status(unsafe { rdf_list_contains(&l, &val as *const T::Native as *const c_void, &mut out) })?;
    Ok(buf.finish(&out))
}

// Evaluate::evaluate (src/evaluation.rs:66-96): fuse each maximal run of Calculate / Filter steps into ONE rdf_pipeline call —
// flatten the Calculations into [rdf_expr_node] (Column -> COLUMN index, ScalarFunction -> OP, Cast -> OP CAST with the
// target dtype), SINK_STORE for a projection, SINK_AGG when the run ends in an aggregate; `include/rdf_frame.hpp`
// (`rdf::Evaluate`) is the C++ statement of exactly that lowering.
