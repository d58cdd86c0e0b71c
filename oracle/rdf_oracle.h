/*
 * rdf_oracle.h — CPU oracle for the rust-dataframe hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (librdf_mi355x.so) never links, imports or calls it.
 *
 * It is a plain-C restatement of the reference's algorithm for the path (single-threaded, chunk list
 * in -> chunk list out, one fully materialised array per plan step), sharing only the struct/enum
 * declarations of include/rdf_mi355x.h so tests can hand the same descriptors to both sides.
 * Every function cites the reference lines it follows.  All buffers are host memory.
 *
 * Pinning status: pinned against the reference's own known answers where its tests hold any
 * (tests/test_oracle_golden.py: abs/acos/cos, count, avg, add on the CSV fixture, sort+take); the
 * arithmetic the reference delegates to the un-vendored `arrow` crate (git branch
 * rust-parquet-arrow-writer, un-pinned, Cargo.toml:9) is restated from the Arrow columnar
 * semantics and cross-checked against pyarrow.  For filter, sum, min, max, divide and comparisons
 * the reference holds no test vector: PARITY UNPINNED for those.
 */
#ifndef RDF_ORACLE_H
#define RDF_ORACLE_H

#include "../include/rdf_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

const char* ora_last_error(void);

rdf_status ora_binary(int32_t op, const rdf_array* a, const rdf_array* b, int64_t nchunks, rdf_out* out);
rdf_status ora_unary(int32_t op, const rdf_array* a, int64_t nchunks, rdf_out* out);
rdf_status ora_cast(const rdf_array* a, int64_t nchunks, rdf_out* out);
rdf_status ora_hour(const rdf_array* a, int64_t nchunks, int32_t unit, rdf_out* out);

rdf_status ora_sum(const rdf_array* a, int64_t nchunks, void* out_scalar, int32_t* out_is_some);
rdf_status ora_min(const rdf_array* a, int64_t nchunks, void* out_scalar, int32_t* out_is_some);
rdf_status ora_max(const rdf_array* a, int64_t nchunks, void* out_scalar, int32_t* out_is_some);
rdf_status ora_count(const rdf_array* a, int64_t nchunks, int64_t* out_count, int32_t* out_is_some);
rdf_status ora_avg(const rdf_array* a, int64_t nchunks, double* out_mean, int32_t* out_is_some);

rdf_status ora_predicate(const rdf_expr_node* nodes, int32_t nnodes, int32_t root,
                         const rdf_array* cols, int32_t ncols, int64_t nchunks, rdf_out* mask);

rdf_status ora_filter_count(const rdf_array* mask, int64_t nchunks, int64_t* counts);
rdf_status ora_filter(const rdf_array* col, const rdf_array* mask, int64_t nchunks, rdf_out* out);
rdf_status ora_filter_columns(const rdf_array* cols, int32_t ncols, const rdf_array* mask,
                              int64_t nchunks, rdf_out* outs);
rdf_status ora_take(const rdf_array* chunks, int64_t nchunks, const rdf_array* indices, rdf_out* out);

rdf_status ora_sort_to_indices(const rdf_array* cols, int32_t ncols, int64_t nchunks, const rdf_sort_options* opts,
                               rdf_out* out_indices);

rdf_status ora_equijoin_indices(const rdf_array* left_keys, int64_t left_nchunks, const rdf_array* right_keys, int64_t right_nchunks,
                                int32_t join_type, rdf_out* out_left, rdf_out* out_right, int64_t* out_rows);

rdf_status ora_equijoin_indices_multi(const rdf_array* left_keys, int64_t left_nchunks, const rdf_array* right_keys, int64_t right_nchunks,
                                      int32_t nkeys, int32_t join_type, rdf_out* out_left, rdf_out* out_right, int64_t* out_rows);

rdf_status ora_groupby_sum(const rdf_array* keys, const rdf_array* values, int64_t nchunks, int64_t max_groups,
                           rdf_out* out_keys, rdf_out* out_sums, rdf_out* out_counts);

rdf_status ora_groupby_agg(const rdf_array* keys, int32_t nkeys, const rdf_array* values, int64_t nchunks, int32_t agg, int64_t max_groups,
                           rdf_out* out_keys, rdf_out* out_values, rdf_out* out_counts);
rdf_status ora_groupby_merge(const rdf_array* keys, const rdf_array* partial, const rdf_array* counts, int32_t agg, int64_t max_groups,
                             rdf_out* out_keys, rdf_out* out_values, rdf_out* out_counts);

rdf_status ora_pipeline(const rdf_program* prog, const rdf_array* cols, int32_t ncols, int64_t nchunks,
                        rdf_out* outs, rdf_agg_result* aggs);

rdf_status ora_group_pipeline(const rdf_expr_node* nodes, int32_t nnodes, int32_t filter_root, int32_t group_root, int32_t ngroups,
                              const int32_t* value_roots, int32_t nvalues, const rdf_array* cols, int32_t ncols, int64_t nchunks,
                              rdf_group_result* out, int64_t* group_rows);

rdf_status ora_list_contains(const rdf_list_array* list, const void* value, rdf_out* out);
rdf_status ora_list_position(const rdf_list_array* list, const void* value, rdf_out* out);
rdf_status ora_list_max(const rdf_list_array* list, rdf_out* out);
rdf_status ora_list_min(const rdf_list_array* list, rdf_out* out);
rdf_status ora_list_remove(const rdf_list_array* list, const void* value, rdf_out* out_offsets, rdf_out* out_values);
rdf_status ora_list_sort(const rdf_list_array* list, rdf_out* out_values);
rdf_status ora_list_distinct(const rdf_list_array* list, rdf_out* out_offsets, rdf_out* out_values);
rdf_status ora_list_except(const rdf_list_array* a, const rdf_list_array* b, rdf_out* out_offsets, rdf_out* out_values);
rdf_status ora_list_intersect(const rdf_list_array* a, const rdf_list_array* b, rdf_out* out_offsets, rdf_out* out_values);
rdf_status ora_list_union(const rdf_list_array* a, const rdf_list_array* b, rdf_out* out_offsets, rdf_out* out_values);
rdf_status ora_list_repeat(const rdf_list_array* list, int32_t count, rdf_out* out_offsets, rdf_out* out_values);

rdf_status ora_fill_uniform_f64(double* ptr, int64_t n, uint64_t seed, uint64_t column_id,
                                int64_t first_row, double lo, double hi);
rdf_status ora_fill_uniform_i64(int64_t* ptr, int64_t n, uint64_t seed, uint64_t column_id,
                                int64_t first_row, int64_t lo, int64_t hi);
rdf_status ora_fill_validity(uint8_t* ptr, int64_t nbits, uint64_t seed, uint64_t column_id,
                             int64_t first_row, double null_fraction);

#ifdef __cplusplus
}
#endif
#endif
