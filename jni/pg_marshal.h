/* Flat-array marshalling between a managed caller (the JNI shim, jni/pinot_gpu_jni.c) and the C ABI of include/pinot_gpu.h.
 *
 * A JVM hands primitive arrays across JNI, not C structs: this file turns those arrays into a pg_query / pg_segment_desc and a pg_result
 * back into arrays.  It is plain C with no JNI types, so that the same code the JNI functions call is built and tested here, where no
 * JDK exists (tests/test_marshal.py drives it through ctypes: the built pg_query is compared field by field with the one the tests
 * build directly, and on the GPU the goldens run through it).
 *
 * Array layouts (the Java side, java/org/apache/pinot/gpu/GpuQueryLowering.java, writes exactly these):
 *   filter_nodes   int32[3 * num_nodes]     {op, predicate, num_children} per pg_filter_node, postfix order
 *   pred_ints      int32[4 * num_preds]     {kind, column, eval, exclusive} per pg_predicate
 *   pred_longs     int64[2 * num_preds]     {lo, hi}
 *   set_offsets    int32[num_preds + 1]     predicate p owns set_words[set_offsets[p] .. set_offsets[p + 1]) (DICT_SET; empty otherwise)
 *   set_words      uint32[]                 bit d of a predicate's words = dictId d matches
 *   aggregations   int32[2 * num_aggs]      {function, column}
 *   group_by       int32[num_group_by]
 * Result arrays (sized by the caller from pgm_result_rows / num_aggregations):
 *   header         int64[PGM_HEADER_LEN]    see the PGM_H_* indexes
 *   group_ids      int32[rows]              raw group ids (group-by only; row numbers when the raw key is beyond an int: PGM_H_GROUP_KEY_KIND)
 *   group_keys     int32[rows * num_group_by]  the dictIds of every group's key, group-by column order (what identifies a group whichever
 *                                           holder the key space calls for: pg_result.group_key_dict_ids)
 *   counts         int64[rows * num_aggs]   row-major by row, like pg_result.group_aggregations
 *   sums / mins / maxs  double[rows * num_aggs];  sums_i64 int64[...], sum_exact int32[...]
 */
#ifndef PG_MARSHAL_H
#define PG_MARSHAL_H

#include <stdint.h>

#include "../include/pinot_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Arities of the flat arrays above and the slots of the result array: ONE definition, used by pg_marshal.c, pinot_gpu_jni.c and -- under the
 * same names -- by java/org/apache/pinot/gpu/PinotGpuNative.java (tests/test_java_constants.py compares the two languages). */
enum {
  PGM_FILTER_NODE_INTS = 3, PGM_PRED_INTS = 4, PGM_PRED_LONGS = 2, PGM_AGG_INTS = 2, PGM_COLUMN_INTS = 6, PGM_COLUMN_BUFFERS = 8,
  PGM_RESULT_ARRAYS = 9
};
/* One query of a batch call (PinotGpuNative.executeBatch): Object[PGM_QUERY_ARRAYS] = the seven flat arrays above in this order, then
 * int[PGM_Q_LIMIT_FLAGS_LEN] {numGroupsLimit, flags}. */
enum {
  PGM_Q_FILTER_NODES = 0, PGM_Q_PRED_INTS = 1, PGM_Q_PRED_LONGS = 2, PGM_Q_SET_OFFSETS = 3, PGM_Q_SET_WORDS = 4, PGM_Q_AGGREGATIONS = 5,
  PGM_Q_GROUP_BY = 6, PGM_Q_LIMIT_FLAGS = 7, PGM_QUERY_ARRAYS = 8, PGM_Q_LIMIT_FLAGS_LEN = 2
};
enum {
  PGM_R_HEADER = 0, PGM_R_GROUP_IDS = 1, PGM_R_COUNTS = 2, PGM_R_SUMS = 3, PGM_R_SUMS_I64 = 4, PGM_R_SUM_EXACT = 5, PGM_R_MINS = 6,
  PGM_R_MAXS = 7, PGM_R_GROUP_KEYS = 8
};

typedef struct pgm_query pgm_query;      /* owns the pg_query and every array it points into */

/* NULL (and pgm_last_error) on inconsistent arguments: negative counts, offsets that do not ascend or leave set_words. */
pgm_query* pgm_query_build(const int32_t* filter_nodes, int32_t num_nodes, const int32_t* pred_ints, const int64_t* pred_longs,
                           int32_t num_preds, const int32_t* set_offsets, const uint32_t* set_words, int32_t num_set_words,
                           const int32_t* aggregations, int32_t num_aggs, const int32_t* group_by, int32_t num_group_by,
                           int32_t num_groups_limit, int32_t flags);
const pg_query* pgm_query_get(const pgm_query* q);
void pgm_query_free(pgm_query* q);

typedef struct pgm_segment pgm_segment;  /* owns the pg_segment_desc, its column array and the name strings */

/* Column c: names[c]; col_ints[6 * c ..] = {stored_type, fwd_encoding, bits_per_value, cardinality, has_dictionary, reserved};
 * col_buffers[8 * c ..] = {fwd address, fwd size, dict address, dict size, inverted address, inverted size, null-vector address, size}
 * (addresses of the caller's mapped index buffers, 0 / 0 where an index does not exist). */
pgm_segment* pgm_segment_build(const char* name, int64_t crc, int32_t device_id, int32_t num_docs, int32_t num_columns, const char* const* names,
                               const int32_t* col_ints, const int64_t* col_buffers);
const pg_segment_desc* pgm_segment_get(const pgm_segment* s);
void pgm_segment_free(pgm_segment* s);

enum {
  PGM_H_NUM_DOCS_SCANNED = 0, PGM_H_ENTRIES_IN_FILTER = 1, PGM_H_ENTRIES_POST_FILTER = 2, PGM_H_TOTAL_DOCS = 3,
  PGM_H_FILTER_ENTRIES_EXACT = 4, PGM_H_NUM_AGGREGATIONS = 5, PGM_H_NUM_GROUPS = 6, PGM_H_GROUP_ID_UPPER_BOUND = 7,
  PGM_H_NUM_GROUPS_LIMIT_REACHED = 8, PGM_H_DOMINANT_KERNEL = 9, PGM_H_IS_GROUP_BY = 10, PGM_H_GROUP_KEY_KIND = 11, PGM_H_NUM_GROUP_BY = 12, PGM_HEADER_LEN = 13
};

/* Rows of the result: 1 for an aggregation-only result, num_groups for a group-by. */
int64_t pgm_result_rows(const pg_result* r, int32_t is_group_by);
void pgm_result_header(const pg_result* r, int32_t is_group_by, int64_t* header);      /* (PGM_H_NUM_GROUP_BY is the caller's to fill in) */
/* Copies the rows out; any output pointer may be NULL (not wanted).  Returns the number of rows written. */
int64_t pgm_result_fill(const pg_result* r, int32_t is_group_by, int32_t* group_ids, int64_t* counts, double* sums, int64_t* sums_i64,
                        int32_t* sum_exact, double* mins, double* maxs);
/* The dictId tuples of a group-by result: rows * num_group_by ints (num_group_by from the query the result answers). */
int64_t pgm_result_fill_keys(const pg_result* r, int32_t num_group_by, int32_t* group_keys);

const char* pgm_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
