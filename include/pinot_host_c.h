/* pinot_host_c.h -- C entry points of libpinot_host.so, the C++ mirror of the reference's HOST side of this path
 * (pinot_amd/csrc/host/pinot_host.h: dictionaries, predicate evaluators, QueryContext, GpuPlanMaker, operators, results blocks,
 * combine, the v1 / v3 segment-directory loader) plus the writers that produce columns in Pinot's on-disk layouts.
 *
 * This is NOT the drop-in boundary -- that is include/pinot_gpu.h, which a JNI shim binds.  These functions exist so
 * that the mirror can be driven from tests and tools without a JVM: SQL text in, JSON results out.  Status codes: 0 ok,
 * 1 QueryException (bad query / bad segment), 2 UnsupportedOperationException (plan-time fallback to the CPU plan), 3 other.
 * Strings returned as char* are malloc'ed: release them with ph_free.  ph_last_error() is per thread. */
#ifndef PINOT_HOST_C_H
#define PINOT_HOST_C_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* ph_last_error(void);
void ph_free(char* p);

/* ---- ImmutableSegment built from caller-owned buffers (IndexSegment / DataSource, pinot-segment-spi) ---- */
void* ph_segment_create(const char* name, int32_t num_docs);
int32_t ph_segment_add_int_column(void* seg, const char* name, int32_t has_dictionary, int32_t bits, int32_t cardinality, const void* fwd,
                                  uint64_t fwd_size, const void* dict, uint64_t dict_size, const void* inv, uint64_t inv_size);
/* data_type: pg_data_type (INT, LONG, FLOAT, DOUBLE); dictionary columns pass the big-endian fixed-width .dict buffer */
int32_t ph_segment_add_numeric_column(void* seg, const char* name, int32_t data_type, int32_t has_dictionary, int32_t bits, int32_t cardinality,
                                      const void* fwd, uint64_t fwd_size, const void* dict, uint64_t dict_size, const void* inv, uint64_t inv_size);
/* values: the dictionary's strings in dictId order, each NUL-terminated */
int32_t ph_segment_add_string_column(void* seg, const char* name, int32_t bits, int32_t cardinality, const void* fwd, uint64_t fwd_size,
                                     const char* values, const void* inv, uint64_t inv_size);
/* the <column>.bitmap.nullvalue file of a column added before (DataSource.getNullValueVector) */
int32_t ph_segment_set_null_vector(void* seg, const char* column, const void* data, uint64_t size);
int32_t ph_segment_load(void* seg, int32_t device);                 /* pg_segment_open; device -1: stay on the host (CPU tests) */
void ph_segment_destroy(void* seg);                                 /* IndexSegment.destroy() */

/* ---- ImmutableSegmentLoader.load(indexDir): v1 file-per-index or v3 columns.psf + index_map ---- */
void* ph_segment_load_directory(const char* index_dir, int32_t device, int32_t* status);
char* ph_segment_describe(void* segment, int32_t* status);          /* JSON: columns, types, indexes, what was not offloaded */

/* ---- GpuPlanMaker (InstancePlanMakerImplV2.makeSegmentPlanNode) + combine ---- */
int32_t ph_plan_maker_init(int32_t device, int32_t time_kernels);
/* gpu.devices ("0-7", "0,2,4", "0-3,6") parsed, and the device each of `count` segments of the given sizes goes to when opened in that
 * order (least resident bytes first; an entry -(device << 48 | bytes) gives bytes back): JSON {"devices": [...], "placement": [...]}.
 * No device is touched. */
char* ph_plan_maker_placement(const char* devices_text, const int64_t* segment_bytes, int32_t count, int32_t* status);
char* ph_parse_sql(const char* sql, int32_t* status);               /* QueryContextConverterUtils.getQueryContext for the SQL subset */
char* ph_lower_predicate(const char* sql_predicate, const void* dict, int32_t cardinality, int32_t* status);   /* PredicateEvaluatorProvider */
/* the physical filter operator tree of the WHERE clause (FilterPlanNode + FilterOperatorUtils: leaf operator per predicate, MatchAll / Empty folding,
 * AND children by priority) as text; the segment need not be loaded on a device */
char* ph_explain_filter(void* segment, const char* sql, int32_t* status);
/* RangePredicateEvaluatorFactory.newDictionaryBasedEvaluator over an INT dictionary; bounds as strings, "*" = unbounded */
char* ph_lower_range_predicate(const void* dict, int32_t cardinality, const char* lower, int32_t lower_inclusive, const char* upper, int32_t upper_inclusive, int32_t* status);
/* the raw-value range evaluator of an INT (0) / LONG (1) column: inclusive [rawLower, rawUpper] */
char* ph_lower_raw_range_predicate(int32_t data_type, const char* lower, int32_t lower_inclusive, const char* upper, int32_t upper_inclusive, int32_t* status);
char* ph_execute_sql(void** segments, int32_t num_segments, const char* sql, int32_t max_execution_threads, int32_t* status);

/* ---- DataTable V4 (DataTableImplV4.toBytes of the intermediate results: what the server sends the broker; host/datatable_v4.cpp) ---- */
/* malloc-ed buffer of *out_size bytes (ph_free), or NULL with *status set */
uint8_t* ph_execute_sql_datatable(void** segments, int32_t num_segments, const char* sql, int32_t max_execution_threads, int64_t* out_size, int32_t* status);
/* the writer alone, over a results block given as flat arrays (CPU tests): see host/c_api.cpp for the array layouts */
uint8_t* ph_datatable_v4_build(int32_t is_group_by, int32_t num_functions, const int32_t* function_types, const char* const* function_columns, int32_t num_keys,
                               const char* const* key_names, const int32_t* key_types, int64_t num_rows, const int64_t* key_longs, const double* key_doubles,
                               const char* const* key_strings, const int64_t* counts, const double* sums, const double* mins, const double* maxs,
                               const uint8_t* is_null, const int64_t* stats, int32_t null_handling, int32_t limit_reached, int32_t segments_processed,
                               int32_t segments_matched, int64_t* out_size, int32_t* status);

/* ---- the group-by table behind the combine operator and the broker's reducer (IndexedTable / TableResizer / GroupByUtils; host/indexed_table.cpp) ---- */
/* GroupByCombineOperator + GroupByDataTableReducer over group-by blocks given as flat arrays (block b = the next block_rows[b] rows); `sql` supplies
 * aggregations, GROUP BY, ORDER BY, LIMIT and the trim options.  JSON: {"combined": block, "reduced": rows, "table": sizes}; ph_free it. */
char* ph_group_by_combine(const char* sql, int32_t num_blocks, const int64_t* block_rows, const int32_t* key_types, const int64_t* key_longs,
                          const double* key_doubles, const char* const* key_strings, const uint8_t* key_is_null, const int64_t* counts, const double* sums,
                          const double* mins, const double* maxs, const uint8_t* is_null, int32_t* status);
int32_t ph_group_by_table_capacity(int32_t limit, int32_t min_num_groups);          /* GroupByUtils.getTableCapacity */
int32_t ph_group_by_trim_threshold(int32_t trim_size, int32_t trim_threshold);      /* GroupByUtils.getIndexedTableTrimThreshold */

/* ---- writers in the reference's layouts (FixedBitSVForwardIndexWriter, SegmentDictionaryCreator, FixedByteChunkForwardIndexWriter v2,
 *      BitmapInvertedIndexWriter, RoaringBitmap portable serialization) and the synthetic-column generator of the benchmarks ---- */
int32_t ph_num_bits_per_value(int32_t max_value);
int64_t ph_fixedbit_size(int64_t num_docs, int32_t bits);
void ph_fixedbit_pack(const int32_t* dict_ids, int64_t num_docs, int32_t bits, uint8_t* out, int32_t threads);
void ph_generate_packed_uniform(uint64_t seed, int64_t num_docs, int32_t cardinality, int32_t bits, uint8_t* out, int32_t threads);
void ph_generate_uniform(uint64_t seed, int64_t start, int64_t count, int32_t cardinality, int32_t* out);
void ph_dict_write_int(const int32_t* sorted_values, int32_t length, uint8_t* out);
void ph_dict_write_fixed(const void* sorted_values, int32_t length, int32_t entry_size, uint8_t* out);
int64_t ph_raw_size_v2(int32_t num_docs, int32_t num_docs_per_chunk);
void ph_raw_write_int_v2(const int32_t* values, int32_t num_docs, int32_t num_docs_per_chunk, uint8_t* out);
int64_t ph_raw_size_fixed_v2(int32_t num_docs, int32_t num_docs_per_chunk, int32_t entry_size);
void ph_raw_write_fixed_v2(const void* values, int32_t num_docs, int32_t num_docs_per_chunk, int32_t entry_size, uint8_t* out);
int64_t ph_roaring_serialize(const int32_t* sorted_doc_ids, int64_t n, int32_t run_optimize, uint8_t* out);     /* out NULL: size only */
int64_t ph_inverted_build(const int32_t* dict_ids, int32_t num_docs, int32_t cardinality, int32_t run_optimize, uint8_t* out);

#ifdef __cplusplus
}
#endif
#endif
