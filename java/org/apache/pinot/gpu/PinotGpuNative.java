/**
 * Native methods of the MI355X segment executor: one per C-ABI entry point of include/pinot_gpu.h that the server path needs
 * (jni/pinot_gpu_jni.c holds the JNI functions, jni/pg_marshal.c the array marshalling they share with the tests).
 *
 * <p>Arrays instead of objects: a query crosses as the flat arrays documented in jni/pg_marshal.h (GpuQueryLowering writes them), a result
 * comes back as an {@code Object[]} of primitive arrays (GpuAggregationOperator reads them).  The only native state Java ever holds is the segment
 * handle, a {@code long}.
 */
package org.apache.pinot.gpu;

import java.nio.ByteBuffer;


public final class PinotGpuNative {
  private PinotGpuNative() {
  }

  static {
    System.loadLibrary("pinot_gpu_jni");
  }

  // ---- constants of the C boundary.  Every PG_* / PGM_* name below exists under the SAME name in include/pinot_gpu.h or
  // jni/pg_marshal.h; tests/test_java_constants.py parses both languages and fails on any difference, and no other class of this package
  // may spell one of these numbers as a literal.

  /** PG_ABI_VERSION (include/pinot_gpu.h): checked against pg_version() in GpuPlanMaker.init. */
  public static final int PG_ABI_VERSION = 3;

  /** pg_status */
  public static final int PG_OK = 0;
  public static final int PG_ERR_INVALID_ARGUMENT = 1;
  public static final int PG_ERR_UNSUPPORTED = 2;
  public static final int PG_ERR_DEVICE = 3;
  public static final int PG_ERR_OUT_OF_MEMORY = 4;
  public static final int PG_ERR_NOT_INITIALIZED = 5;
  public static final int PG_ERR_INTERNAL = 6;

  /** pg_data_type */
  public static final int PG_TYPE_INT = 0;
  public static final int PG_TYPE_LONG = 1;
  public static final int PG_TYPE_FLOAT = 2;
  public static final int PG_TYPE_DOUBLE = 3;

  /** pg_fwd_encoding */
  public static final int PG_FWD_FIXED_BIT_DICT = 0;
  public static final int PG_FWD_RAW_FIXED_BYTE = 1;

  /** pg_predicate_kind */
  public static final int PG_PRED_MATCH_ALL = 0;
  public static final int PG_PRED_MATCH_NONE = 1;
  public static final int PG_PRED_DICT_RANGE = 2;
  public static final int PG_PRED_DICT_SET = 3;
  public static final int PG_PRED_RAW_RANGE = 4;
  public static final int PG_PRED_DOC_RANGE = 5;
  public static final int PG_PRED_IS_NULL = 6;

  /** pg_leaf_eval */
  public static final int PG_EVAL_SCAN = 0;
  public static final int PG_EVAL_INVERTED = 1;

  /** pg_filter_op */
  public static final int PG_FILTER_LEAF = 0;
  public static final int PG_FILTER_AND = 1;
  public static final int PG_FILTER_OR = 2;
  public static final int PG_FILTER_NOT = 3;

  /** pg_agg_function */
  public static final int PG_AGG_COUNT = 0;
  public static final int PG_AGG_SUM = 1;
  public static final int PG_AGG_MIN = 2;
  public static final int PG_AGG_MAX = 3;
  public static final int PG_AGG_AVG = 4;

  /** pg_query.flags */
  public static final int PG_QUERY_NULL_HANDLING = 1;
  /** numEntriesScannedInFilter of a leap-frogging filter may be the upper bound (no pass behind the query): include/pinot_gpu.h. */
  public static final int PG_QUERY_STATS_UPPER_BOUND_OK = 2;

  /** Record sizes of the flat arrays and slots of the result array (jni/pg_marshal.h). */
  public static final int PGM_FILTER_NODE_INTS = 3;
  public static final int PGM_PRED_INTS = 4;
  public static final int PGM_PRED_LONGS = 2;
  public static final int PGM_AGG_INTS = 2;
  public static final int PGM_COLUMN_INTS = 6;
  public static final int PGM_COLUMN_BUFFERS = 8;
  public static final int PGM_RESULT_ARRAYS = 9;
  /** One query of executeBatch: Object[PGM_QUERY_ARRAYS], slots PGM_Q_*; the last slot is int[PGM_Q_LIMIT_FLAGS_LEN] {numGroupsLimit, flags}. */
  public static final int PGM_Q_FILTER_NODES = 0;
  public static final int PGM_Q_PRED_INTS = 1;
  public static final int PGM_Q_PRED_LONGS = 2;
  public static final int PGM_Q_SET_OFFSETS = 3;
  public static final int PGM_Q_SET_WORDS = 4;
  public static final int PGM_Q_AGGREGATIONS = 5;
  public static final int PGM_Q_GROUP_BY = 6;
  public static final int PGM_Q_LIMIT_FLAGS = 7;
  public static final int PGM_QUERY_ARRAYS = 8;
  public static final int PGM_Q_LIMIT_FLAGS_LEN = 2;
  public static final int PGM_R_HEADER = 0;
  public static final int PGM_R_GROUP_IDS = 1;
  public static final int PGM_R_COUNTS = 2;
  public static final int PGM_R_SUMS = 3;
  public static final int PGM_R_SUMS_I64 = 4;
  public static final int PGM_R_SUM_EXACT = 5;
  public static final int PGM_R_MINS = 6;
  public static final int PGM_R_MAXS = 7;
  public static final int PGM_R_GROUP_KEYS = 8;

  /** Indexes of the result header (PGM_H_* in jni/pg_marshal.h). */
  public static final int PGM_H_NUM_DOCS_SCANNED = 0;
  public static final int PGM_H_ENTRIES_IN_FILTER = 1;
  public static final int PGM_H_ENTRIES_POST_FILTER = 2;
  public static final int PGM_H_TOTAL_DOCS = 3;
  public static final int PGM_H_FILTER_ENTRIES_EXACT = 4;
  public static final int PGM_H_NUM_AGGREGATIONS = 5;
  public static final int PGM_H_NUM_GROUPS = 6;
  public static final int PGM_H_GROUP_ID_UPPER_BOUND = 7;
  public static final int PGM_H_NUM_GROUPS_LIMIT_REACHED = 8;
  public static final int PGM_H_DOMINANT_KERNEL = 9;
  public static final int PGM_H_IS_GROUP_BY = 10;
  public static final int PGM_H_GROUP_KEY_KIND = 11;
  public static final int PGM_H_NUM_GROUP_BY = 12;
  public static final int PGM_HEADER_LEN = 13;

  /** pg_init: once per JVM, from GpuPlanMaker.init. */
  static native void init(int device, int flags);

  /** pg_shutdown */
  static native void shutdown();

  /** pg_version, e.g. "pinot_gpu 0.2 gfx950" */
  static native String version();

  /** pg_last_error of the calling thread */
  static native String lastError();

  /** GetDirectBufferAddress: the address of a mapped index buffer (PinotDataBuffer.toDirectByteBuffer). */
  static native long directBufferAddress(ByteBuffer buffer);

  /**
   * pg_segment_open.  {@code columnInts}: PGM_COLUMN_INTS per column {storedType, fwdEncoding, bitsPerValue, cardinality, hasDictionary, 0};
   * {@code columnBuffers}: PGM_COLUMN_BUFFERS per column {fwd address, fwd size, dict address, dict size, inverted address, inverted size, null-vector
   * address, null-vector size}, 0 / 0 where an index does not exist.  The buffers are read during the call only.
   */
  static native long segmentOpen(String name, long crc, int device, int numDocs, String[] columnNames, int[] columnInts, long[] columnBuffers);

  /** pg_segment_close */
  static native void segmentClose(long handle);

  /** pg_segment_device_bytes */
  static native long segmentDeviceBytes(long handle);

  /**
   * pg_group_key_info: {base, isOffset, nullEntry} of a group-by column -- a dictionary column's key entries are dictIds (isOffset 0); a raw
   * INT / LONG column's are offsets from its smallest value (isOffset 1, key value = base + entry: the reference's no-dictionary key
   * generators key by value); nullEntry is the entry that means NULL under enableNullHandling.  isOffset 2: see
   * groupKeyValues.
   */
  static native long[] groupKeyInfo(long handle, int column);

  /**
   * pg_group_key_values: for a raw key column whose groupKeyInfo says isOffset 2 -- a FLOAT / DOUBLE column, or an INT / LONG column whose
   * values span more than an int: the device keys it by value through a dictionary it builds from the column
   * (NoDictionarySingleColumnGroupKeyGenerator.java:100-135) -- the column's distinct values in ascending order: the long values, or
   * Double.doubleToRawLongBits of the (widened) doubles.  A key entry of such a column is an index into this array.
   */
  static native long[] groupKeyValues(long handle, int column);

  /**
   * pg_query_check: PG_OK or PG_ERR_UNSUPPORTED.  Nothing is launched -- with one exception: the first check of a query that groups by a raw FLOAT /
   * DOUBLE column (or a raw INT / LONG column spanning more than an int) builds that column's dictionary and rank image on the device, once per
   * column and segment (its cardinality is what the plan is priced with); a build that fails answers PG_ERR_UNSUPPORTED: the CPU plan.
   */
  static native int queryCheck(long handle, int[] filterNodes, int[] predInts, long[] predLongs, int[] setOffsets, int[] setWords,
      int[] aggregations, int[] groupBy, int numGroupsLimit, int flags);

  /**
   * pg_execute.  Returns Object[PGM_RESULT_ARRAYS], slots PGM_R_*: {long[] header, int[] groupIds, long[] counts, double[] sums,
   * long[] sumsI64, int[] sumExact, double[] mins, double[] maxs, int[] groupKeys (rows x group-by columns dictIds)}; throws UnsupportedOperationException for PG_ERR_UNSUPPORTED, RuntimeException (pg_last_error) otherwise.
   */
  static native Object[] execute(long handle, int[] filterNodes, int[] predInts, long[] predLongs, int[] setOffsets, int[] setWords,
      int[] aggregations, int[] groupBy, int numGroupsLimit, int flags);

  /**
   * pg_execute_batch: {@code queries[i]} (Object[PGM_QUERY_ARRAYS], slots PGM_Q_*) over {@code handles[i]} -- the segments of ONE query, the
   * way BaseCombineOperator hands them to its worker threads, in one native call (aggregations over scan / sorted leaves share one
   * kernel launch over all the segments; everything else runs side by side on the library's own threads).  Returns Object[n]: element i
   * is the Object[PGM_RESULT_ARRAYS] execute() would have returned for item i, or a String {@code "<pg_status>\n<message>"} when that
   * item failed (the others do not stop for it).  Throws only when the call as a whole could not be made.
   */
  static native Object[] executeBatch(long[] handles, Object[][] queries);
}
