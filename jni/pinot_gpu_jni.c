/* JNI functions of org.apache.pinot.gpu.PinotGpuNative (java/org/apache/pinot/gpu/PinotGpuNative.java) over the C ABI of
 * include/pinot_gpu.h.  Every function is: pin the Java arrays, hand them to the JNI-free marshalling of pg_marshal.c, call ONE C-ABI
 * entry point, copy plain arrays back.  No native pointer other than the opaque segment handle (a jlong) escapes into Java, and no
 * Java object is retained past a call.
 *
 * Replaces, on the reference side: nothing that exists -- the reference has no native code.  It is what a maintainer adds next to
 * org.apache.pinot.gpu.GpuPlanMaker so that InstancePlanMakerImplV2.makeSegmentPlanNode (core/plan/maker/InstancePlanMakerImplV2.java:270-289)
 * can hand a segment's aggregation to the device.  Build: jni/Makefile (`make jni JAVA_HOME=...`); this environment has no JDK, so the
 * file is compiled here only against the type-level stand-in jni/stub/jni.h (tests/test_marshal.py), the code under it through ctypes.
 *
 * Errors: PG_ERR_UNSUPPORTED -> java.lang.UnsupportedOperationException (the plan maker asked queryCheck first, so this is a bug in the
 * caller), every other non-OK status -> java.lang.RuntimeException carrying pg_last_error(). */
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/pinot_gpu.h"
#include "pg_marshal.h"

static void throw_new(JNIEnv* env, const char* class_name, const char* message) {
  jclass cls = (*env)->FindClass(env, class_name);
  if (cls != NULL) (*env)->ThrowNew(env, cls, message ? message : "");
}

static void throw_status(JNIEnv* env, pg_status status) {
  throw_new(env, status == PG_ERR_UNSUPPORTED ? "java/lang/UnsupportedOperationException" : "java/lang/RuntimeException", pg_last_error());
}

JNIEXPORT void JNICALL Java_org_apache_pinot_gpu_PinotGpuNative_init(JNIEnv* env, jclass cls, jint device, jint flags) {
  (void)cls;
  pg_config config;
  memset(&config, 0, sizeof(config));
  config.abi_version = PG_ABI_VERSION;
  config.device_id = device;
  config.flags = (int32_t)flags;
  const pg_status status = pg_init(&config);
  if (status != PG_OK) throw_status(env, status);
}

JNIEXPORT void JNICALL Java_org_apache_pinot_gpu_PinotGpuNative_shutdown(JNIEnv* env, jclass cls) {
  (void)env; (void)cls;
  (void)pg_shutdown();
}

JNIEXPORT jstring JNICALL Java_org_apache_pinot_gpu_PinotGpuNative_version(JNIEnv* env, jclass cls) {
  (void)cls;
  return (*env)->NewStringUTF(env, pg_version());
}

/* The address behind a direct ByteBuffer: how the Java side learns where PinotDataBuffer mapped an index file. */
JNIEXPORT jlong JNICALL Java_org_apache_pinot_gpu_PinotGpuNative_directBufferAddress(JNIEnv* env, jclass cls, jobject buffer) {
  (void)cls;
  void* address = buffer != NULL ? (*env)->GetDirectBufferAddress(env, buffer) : NULL;
  if (address == NULL) { throw_new(env, "java/lang/IllegalArgumentException", "not a direct ByteBuffer"); return 0; }
  return (jlong)(intptr_t)address;
}

/* columnInts / columnBuffers: the layouts of pgm_segment_build.  The buffer addresses are those of the segment's memory-mapped index
 * buffers (PinotDataBuffer -> direct ByteBuffer -> GetDirectBufferAddress on the Java side, see PinotGpuNative.addressOf); they are
 * read once, here, while pg_segment_open copies them to HBM. */
JNIEXPORT jlong JNICALL Java_org_apache_pinot_gpu_PinotGpuNative_segmentOpen(JNIEnv* env, jclass cls, jstring name, jlong crc, jint device,
    jint numDocs, jobjectArray columnNames, jintArray columnInts, jlongArray columnBuffers) {
  (void)cls;
  if (name == NULL || columnNames == NULL || columnInts == NULL || columnBuffers == NULL) {
    throw_new(env, "java/lang/NullPointerException", "segmentOpen: null argument");
    return 0;
  }
  const jsize num_columns = (*env)->GetArrayLength(env, columnNames);
  if ((int64_t)(*env)->GetArrayLength(env, columnInts) != (int64_t)PGM_COLUMN_INTS * num_columns ||
      (int64_t)(*env)->GetArrayLength(env, columnBuffers) != (int64_t)PGM_COLUMN_BUFFERS * num_columns) {
    throw_new(env, "java/lang/IllegalArgumentException", "column arrays do not match the number of columns");
    return 0;
  }
  const char* segment_name = (*env)->GetStringUTFChars(env, name, NULL);
  const char** names = (const char**)calloc(num_columns ? (size_t)num_columns : 1, sizeof(char*));
  jstring* name_refs = (jstring*)calloc(num_columns ? (size_t)num_columns : 1, sizeof(jstring));
  jint* ints = (*env)->GetIntArrayElements(env, columnInts, NULL);
  jlong* buffers = (*env)->GetLongArrayElements(env, columnBuffers, NULL);
  jlong handle = 0;
  int ok = segment_name != NULL && names != NULL && name_refs != NULL && ints != NULL && buffers != NULL;
  for (jsize c = 0; ok && c < num_columns; c++) {
    name_refs[c] = (jstring)(*env)->GetObjectArrayElement(env, columnNames, c);
    names[c] = name_refs[c] ? (*env)->GetStringUTFChars(env, name_refs[c], NULL) : NULL;
    ok = names[c] != NULL;
  }
  if (ok) {
    pgm_segment* built = pgm_segment_build(segment_name, (int64_t)crc, (int32_t)device, (int32_t)numDocs, (int32_t)num_columns, names,
                                           (const int32_t*)ints, (const int64_t*)buffers);
    if (built == NULL) {
      throw_new(env, "java/lang/IllegalArgumentException", pgm_last_error());
    } else {
      pg_segment* segment = NULL;
      const pg_status status = pg_segment_open(pgm_segment_get(built), &segment);
      pgm_segment_free(built);                      /* pg_segment_open copied what it keeps */
      if (status != PG_OK) throw_status(env, status); else handle = (jlong)(intptr_t)segment;
    }
  } else if (!(*env)->ExceptionCheck(env)) {
    throw_new(env, "java/lang/OutOfMemoryError", "pinning the segment description failed");
  }
  for (jsize c = 0; c < num_columns; c++) {
    if (names != NULL && name_refs != NULL && names[c] != NULL) (*env)->ReleaseStringUTFChars(env, name_refs[c], names[c]);
    if (name_refs != NULL && name_refs[c] != NULL) (*env)->DeleteLocalRef(env, name_refs[c]);
  }
  if (buffers != NULL) (*env)->ReleaseLongArrayElements(env, columnBuffers, buffers, JNI_ABORT);
  if (ints != NULL) (*env)->ReleaseIntArrayElements(env, columnInts, ints, JNI_ABORT);
  if (segment_name != NULL) (*env)->ReleaseStringUTFChars(env, name, segment_name);
  free(name_refs);
  free((void*)names);
  return handle;
}

JNIEXPORT void JNICALL Java_org_apache_pinot_gpu_PinotGpuNative_segmentClose(JNIEnv* env, jclass cls, jlong handle) {
  (void)env; (void)cls;
  if (handle != 0) (void)pg_segment_close((pg_segment*)(intptr_t)handle);
}

JNIEXPORT jlong JNICALL Java_org_apache_pinot_gpu_PinotGpuNative_segmentDeviceBytes(JNIEnv* env, jclass cls, jlong handle) {
  (void)cls;
  uint64_t bytes = 0;
  const pg_status status = pg_segment_device_bytes((const pg_segment*)(intptr_t)handle, &bytes);
  if (status != PG_OK) { throw_status(env, status); return 0; }
  return (jlong)bytes;
}

JNIEXPORT jlongArray JNICALL Java_org_apache_pinot_gpu_PinotGpuNative_groupKeyInfo(JNIEnv* env, jclass cls, jlong handle, jint column) {
  (void)cls;
  int64_t base = 0;
  int32_t is_offset = 0, null_entry = 0;
  const pg_status status = pg_group_key_info((const pg_segment*)(intptr_t)handle, (int32_t)column, &base, &is_offset, &null_entry);
  if (status != PG_OK) { throw_status(env, status); return NULL; }
  const jlong values[3] = {(jlong)base, (jlong)is_offset, (jlong)null_entry};
  jlongArray out = (*env)->NewLongArray(env, 3);
  if (out != NULL) (*env)->SetLongArrayRegion(env, out, 0, 3, values);
  return out;
}

/* pg_group_key_values: the distinct values of a raw key column keyed through a rank image (groupKeyInfo's isOffset == 2), ascending, as
 * long values or the bits of doubles. */
JNIEXPORT jlongArray JNICALL Java_org_apache_pinot_gpu_PinotGpuNative_groupKeyValues(JNIEnv* env, jclass cls, jlong handle, jint column) {
  (void)cls;
  int32_t count = 0;
  pg_status status = pg_group_key_values((pg_segment*)(intptr_t)handle, (int32_t)column, NULL, 0, &count);
  if (status != PG_OK) { throw_status(env, status); return NULL; }
  jlongArray out = (*env)->NewLongArray(env, count);
  if (out == NULL || count == 0) return out;
  jlong* values = (*env)->GetLongArrayElements(env, out, NULL);
  if (values == NULL) return NULL;
  status = pg_group_key_values((pg_segment*)(intptr_t)handle, (int32_t)column, (int64_t*)values, count, &count);
  (*env)->ReleaseLongArrayElements(env, out, values, 0);
  if (status != PG_OK) { throw_status(env, status); return NULL; }
  return out;
}

/* The query arrays of pg_marshal.h, pinned for the duration of one call. */
typedef struct pinned_query {
  jint *nodes, *pred_ints, *set_offsets, *set_words, *aggregations, *group_by;
  jlong* pred_longs;
  pgm_query* built;
} pinned_query;

static void release_query(JNIEnv* env, pinned_query* p, jintArray filterNodes, jintArray predInts, jlongArray predLongs, jintArray setOffsets,
                          jintArray setWords, jintArray aggregations, jintArray groupBy) {
  pgm_query_free(p->built);
  if (p->group_by) (*env)->ReleaseIntArrayElements(env, groupBy, p->group_by, JNI_ABORT);
  if (p->aggregations) (*env)->ReleaseIntArrayElements(env, aggregations, p->aggregations, JNI_ABORT);
  if (p->set_words) (*env)->ReleaseIntArrayElements(env, setWords, p->set_words, JNI_ABORT);
  if (p->set_offsets) (*env)->ReleaseIntArrayElements(env, setOffsets, p->set_offsets, JNI_ABORT);
  if (p->pred_longs) (*env)->ReleaseLongArrayElements(env, predLongs, p->pred_longs, JNI_ABORT);
  if (p->pred_ints) (*env)->ReleaseIntArrayElements(env, predInts, p->pred_ints, JNI_ABORT);
  if (p->nodes) (*env)->ReleaseIntArrayElements(env, filterNodes, p->nodes, JNI_ABORT);
}

/* Returns 0 and leaves a Java exception pending on failure. */
static int pin_query(JNIEnv* env, pinned_query* p, jintArray filterNodes, jintArray predInts, jlongArray predLongs, jintArray setOffsets,
                     jintArray setWords, jintArray aggregations, jintArray groupBy, jint numGroupsLimit, jint flags) {
  memset(p, 0, sizeof(*p));
  if (filterNodes == NULL || predInts == NULL || predLongs == NULL || setOffsets == NULL || setWords == NULL || aggregations == NULL ||
      groupBy == NULL) {
    throw_new(env, "java/lang/NullPointerException", "query arrays must not be null (empty arrays stand for nothing)");
    return 0;
  }
  const jsize len_nodes = (*env)->GetArrayLength(env, filterNodes);
  const jsize len_pred_ints = (*env)->GetArrayLength(env, predInts);
  const jsize len_aggs = (*env)->GetArrayLength(env, aggregations);
  if (len_nodes % PGM_FILTER_NODE_INTS != 0 || len_pred_ints % PGM_PRED_INTS != 0 || len_aggs % PGM_AGG_INTS != 0) {
    throw_new(env, "java/lang/IllegalArgumentException", "filterNodes / predInts / aggregations: length is not a whole number of records");
    return 0;
  }
  const jsize num_nodes = len_nodes / PGM_FILTER_NODE_INTS;
  const jsize num_preds = len_pred_ints / PGM_PRED_INTS;
  const jsize num_set_words = (*env)->GetArrayLength(env, setWords);
  const jsize num_aggs = len_aggs / PGM_AGG_INTS;
  const jsize num_group_by = (*env)->GetArrayLength(env, groupBy);
  if ((int64_t)(*env)->GetArrayLength(env, predLongs) != (int64_t)PGM_PRED_LONGS * num_preds ||
      (*env)->GetArrayLength(env, setOffsets) != num_preds + 1) {
    throw_new(env, "java/lang/IllegalArgumentException", "predicate arrays of different lengths");
    return 0;
  }
  p->nodes = (*env)->GetIntArrayElements(env, filterNodes, NULL);
  p->pred_ints = (*env)->GetIntArrayElements(env, predInts, NULL);
  p->pred_longs = (*env)->GetLongArrayElements(env, predLongs, NULL);
  p->set_offsets = (*env)->GetIntArrayElements(env, setOffsets, NULL);
  p->set_words = (*env)->GetIntArrayElements(env, setWords, NULL);
  p->aggregations = (*env)->GetIntArrayElements(env, aggregations, NULL);
  p->group_by = (*env)->GetIntArrayElements(env, groupBy, NULL);
  if (!p->nodes || !p->pred_ints || !p->pred_longs || !p->set_offsets || !p->set_words || !p->aggregations || !p->group_by) {
    release_query(env, p, filterNodes, predInts, predLongs, setOffsets, setWords, aggregations, groupBy);
    if (!(*env)->ExceptionCheck(env)) throw_new(env, "java/lang/OutOfMemoryError", "pinning the query arrays failed");
    return 0;
  }
  p->built = pgm_query_build((const int32_t*)p->nodes, (int32_t)num_nodes, (const int32_t*)p->pred_ints, (const int64_t*)p->pred_longs,
                             (int32_t)num_preds, (const int32_t*)p->set_offsets, (const uint32_t*)p->set_words, (int32_t)num_set_words,
                             (const int32_t*)p->aggregations, (int32_t)num_aggs, (const int32_t*)p->group_by, (int32_t)num_group_by,
                             (int32_t)numGroupsLimit, (int32_t)flags);
  if (p->built == NULL) {
    release_query(env, p, filterNodes, predInts, predLongs, setOffsets, setWords, aggregations, groupBy);
    throw_new(env, "java/lang/IllegalArgumentException", pgm_last_error());
    return 0;
  }
  return 1;
}

/* pg_query_check: PG_OK (0) or PG_ERR_UNSUPPORTED (2) come back as the status; anything else is an exception. */
JNIEXPORT jint JNICALL Java_org_apache_pinot_gpu_PinotGpuNative_queryCheck(JNIEnv* env, jclass cls, jlong handle, jintArray filterNodes,
    jintArray predInts, jlongArray predLongs, jintArray setOffsets, jintArray setWords, jintArray aggregations, jintArray groupBy,
    jint numGroupsLimit, jint flags) {
  (void)cls;
  pinned_query p;
  if (!pin_query(env, &p, filterNodes, predInts, predLongs, setOffsets, setWords, aggregations, groupBy, numGroupsLimit, flags)) return PG_ERR_INTERNAL;
  const pg_status status = pg_query_check((const pg_segment*)(intptr_t)handle, pgm_query_get(p.built));
  release_query(env, &p, filterNodes, predInts, predLongs, setOffsets, setWords, aggregations, groupBy);
  if (status != PG_OK && status != PG_ERR_UNSUPPORTED) throw_status(env, status);
  return (jint)status;
}

JNIEXPORT jstring JNICALL Java_org_apache_pinot_gpu_PinotGpuNative_lastError(JNIEnv* env, jclass cls) {
  (void)cls;
  return (*env)->NewStringUTF(env, pg_last_error());
}

/* A pg_result as Object[PGM_RESULT_ARRAYS] (slots PGM_R_*): {long[] header (PGM_H_* layout), int[] groupIds, long[] counts, double[] sums, long[] sumsI64,
 * int[] sumExact, double[] mins, double[] maxs, int[] groupKeys (dictId tuples)}, the value arrays row-major [row * numAggregations + a].
 * Frees the result.  NULL with a Java exception pending on failure. */
static jobjectArray result_to_java(JNIEnv* env, pg_result* result, int32_t num_group_by) {
  const int32_t is_group_by = num_group_by > 0;
  const int64_t rows64 = pgm_result_rows(result, is_group_by);
  const int64_t cells64 = rows64 * (int64_t)result->num_aggregations;
  if (rows64 < 0 || cells64 < 0 || cells64 > (int64_t)INT32_MAX - 8 || rows64 * (int64_t)num_group_by > (int64_t)INT32_MAX - 8) {          /* a Java array holds fewer than 2^31 elements */
    pg_result_free(result);
    throw_new(env, "java/lang/IllegalStateException", "the result has more cells than a Java array holds");
    return NULL;
  }
  const jsize rows = (jsize)rows64;
  const jsize cells = (jsize)cells64;
  jobjectArray out = NULL;
  jclass object_class = (*env)->FindClass(env, "java/lang/Object");
  jlongArray header = (*env)->NewLongArray(env, PGM_HEADER_LEN);
  jintArray group_ids = (*env)->NewIntArray(env, is_group_by ? rows : 0);
  jintArray group_keys = (*env)->NewIntArray(env, is_group_by ? rows * (jsize)num_group_by : 0);
  jlongArray counts = (*env)->NewLongArray(env, cells);
  jdoubleArray sums = (*env)->NewDoubleArray(env, cells);
  jlongArray sums_i64 = (*env)->NewLongArray(env, cells);
  jintArray sum_exact = (*env)->NewIntArray(env, cells);
  jdoubleArray mins = (*env)->NewDoubleArray(env, cells);
  jdoubleArray maxs = (*env)->NewDoubleArray(env, cells);
  if (object_class && header && group_ids && group_keys && counts && sums && sums_i64 && sum_exact && mins && maxs) {
    int64_t h[PGM_HEADER_LEN];
    pgm_result_header(result, is_group_by, h);
    h[PGM_H_NUM_GROUP_BY] = num_group_by;
    (*env)->SetLongArrayRegion(env, header, 0, PGM_HEADER_LEN, (const jlong*)h);
    /* the value arrays are filled in place: GetPrimitiveArrayCritical would forbid the JNI calls in between, plain element access does not */
    jint* g = (*env)->GetIntArrayElements(env, group_ids, NULL);
    jint* gk = (*env)->GetIntArrayElements(env, group_keys, NULL);
    jlong* c = (*env)->GetLongArrayElements(env, counts, NULL);
    jdouble* s = (*env)->GetDoubleArrayElements(env, sums, NULL);
    jlong* si = (*env)->GetLongArrayElements(env, sums_i64, NULL);
    jint* se = (*env)->GetIntArrayElements(env, sum_exact, NULL);
    jdouble* mn = (*env)->GetDoubleArrayElements(env, mins, NULL);
    jdouble* mx = (*env)->GetDoubleArrayElements(env, maxs, NULL);
    if (g && gk && c && s && si && se && mn && mx) {
      (void)pgm_result_fill(result, is_group_by, (int32_t*)g, (int64_t*)c, (double*)s, (int64_t*)si, (int32_t*)se, (double*)mn, (double*)mx);
      if (is_group_by) (void)pgm_result_fill_keys(result, num_group_by, (int32_t*)gk);
      out = (*env)->NewObjectArray(env, PGM_RESULT_ARRAYS, object_class, NULL);
    }
    if (mx) (*env)->ReleaseDoubleArrayElements(env, maxs, mx, 0);
    if (mn) (*env)->ReleaseDoubleArrayElements(env, mins, mn, 0);
    if (se) (*env)->ReleaseIntArrayElements(env, sum_exact, se, 0);
    if (si) (*env)->ReleaseLongArrayElements(env, sums_i64, si, 0);
    if (s) (*env)->ReleaseDoubleArrayElements(env, sums, s, 0);
    if (c) (*env)->ReleaseLongArrayElements(env, counts, c, 0);
    if (gk) (*env)->ReleaseIntArrayElements(env, group_keys, gk, 0);
    if (g) (*env)->ReleaseIntArrayElements(env, group_ids, g, 0);
    if (out != NULL) {
      (*env)->SetObjectArrayElement(env, out, PGM_R_HEADER, header);
      (*env)->SetObjectArrayElement(env, out, PGM_R_GROUP_IDS, group_ids);
      (*env)->SetObjectArrayElement(env, out, PGM_R_COUNTS, counts);
      (*env)->SetObjectArrayElement(env, out, PGM_R_SUMS, sums);
      (*env)->SetObjectArrayElement(env, out, PGM_R_SUMS_I64, sums_i64);
      (*env)->SetObjectArrayElement(env, out, PGM_R_SUM_EXACT, sum_exact);
      (*env)->SetObjectArrayElement(env, out, PGM_R_MINS, mins);
      (*env)->SetObjectArrayElement(env, out, PGM_R_MAXS, maxs);
      (*env)->SetObjectArrayElement(env, out, PGM_R_GROUP_KEYS, group_keys);
    }
  }
  /* (a batch call builds one of these per segment: the local references are given back here, not at the end of the native call) */
  if (maxs) (*env)->DeleteLocalRef(env, maxs);
  if (mins) (*env)->DeleteLocalRef(env, mins);
  if (sum_exact) (*env)->DeleteLocalRef(env, sum_exact);
  if (sums_i64) (*env)->DeleteLocalRef(env, sums_i64);
  if (sums) (*env)->DeleteLocalRef(env, sums);
  if (counts) (*env)->DeleteLocalRef(env, counts);
  if (group_keys) (*env)->DeleteLocalRef(env, group_keys);
  if (group_ids) (*env)->DeleteLocalRef(env, group_ids);
  if (header) (*env)->DeleteLocalRef(env, header);
  if (object_class) (*env)->DeleteLocalRef(env, object_class);
  pg_result_free(result);
  if (out == NULL && !(*env)->ExceptionCheck(env)) throw_new(env, "java/lang/OutOfMemoryError", "allocating the result arrays failed");
  return out;
}

/* pg_execute.  Returns the Object[PGM_RESULT_ARRAYS] of result_to_java. */
JNIEXPORT jobjectArray JNICALL Java_org_apache_pinot_gpu_PinotGpuNative_execute(JNIEnv* env, jclass cls, jlong handle, jintArray filterNodes,
    jintArray predInts, jlongArray predLongs, jintArray setOffsets, jintArray setWords, jintArray aggregations, jintArray groupBy,
    jint numGroupsLimit, jint flags) {
  (void)cls;
  pinned_query p;
  if (!pin_query(env, &p, filterNodes, predInts, predLongs, setOffsets, setWords, aggregations, groupBy, numGroupsLimit, flags)) return NULL;
  const int32_t num_group_by = pgm_query_get(p.built)->num_group_by;
  pg_result result;
  const pg_status status = pg_execute((pg_segment*)(intptr_t)handle, pgm_query_get(p.built), &result);
  release_query(env, &p, filterNodes, predInts, predLongs, setOffsets, setWords, aggregations, groupBy);
  if (status != PG_OK) { throw_status(env, status); return NULL; }     /* pg_execute freed the result */
  return result_to_java(env, &result, num_group_by);
}

/* pg_execute_batch: queries[i] (Object[PGM_QUERY_ARRAYS], slots PGM_Q_*) over handles[i] -- the segments of ONE query as the combine operator
 * would hand them to its worker threads (BaseCombineOperator.java:85-142).  Returns Object[n]: element i is the Object[PGM_RESULT_ARRAYS]
 * execute() would have returned for item i, or -- the item failed, the others did not stop for it -- a String "<pg_status>\n<message>".
 * Throws only when the call as a whole could not be made (malformed arrays, library not initialised). */
JNIEXPORT jobjectArray JNICALL Java_org_apache_pinot_gpu_PinotGpuNative_executeBatch(JNIEnv* env, jclass cls, jlongArray handles, jobjectArray queries) {
  (void)cls;
  if (handles == NULL || queries == NULL) { throw_new(env, "java/lang/NullPointerException", "handles / queries must not be null"); return NULL; }
  const jsize n = (*env)->GetArrayLength(env, handles);
  if ((*env)->GetArrayLength(env, queries) != n) { throw_new(env, "java/lang/IllegalArgumentException", "one query per segment handle"); return NULL; }
  jclass object_class = (*env)->FindClass(env, "java/lang/Object");
  if (object_class == NULL) return NULL;
  jobjectArray out = (*env)->NewObjectArray(env, n, object_class, NULL);
  if (out == NULL || n == 0) return out;
  const size_t count = (size_t)n;
  pgm_query** built = (pgm_query**)calloc(count, sizeof(pgm_query*));
  pg_segment** segments = (pg_segment**)calloc(count, sizeof(pg_segment*));
  const pg_query** lowered = (const pg_query**)calloc(count, sizeof(pg_query*));
  pg_result* results = (pg_result*)calloc(count, sizeof(pg_result));
  pg_status* statuses = (pg_status*)calloc(count, sizeof(pg_status));
  jlong* h = (*env)->GetLongArrayElements(env, handles, NULL);
  int ok = built && segments && lowered && results && statuses && h;
  if (!ok && !(*env)->ExceptionCheck(env)) throw_new(env, "java/lang/OutOfMemoryError", "allocating the batch failed");
  for (jsize i = 0; ok && i < n; i++) {
    jobjectArray q = (jobjectArray)(*env)->GetObjectArrayElement(env, queries, i);
    if (q == NULL || (*env)->GetArrayLength(env, q) != PGM_QUERY_ARRAYS) {
      throw_new(env, "java/lang/IllegalArgumentException", "a batch query is Object[PGM_QUERY_ARRAYS]");
      ok = 0;
      break;
    }
    jintArray nodes = (jintArray)(*env)->GetObjectArrayElement(env, q, PGM_Q_FILTER_NODES);
    jintArray pred_ints = (jintArray)(*env)->GetObjectArrayElement(env, q, PGM_Q_PRED_INTS);
    jlongArray pred_longs = (jlongArray)(*env)->GetObjectArrayElement(env, q, PGM_Q_PRED_LONGS);
    jintArray set_offsets = (jintArray)(*env)->GetObjectArrayElement(env, q, PGM_Q_SET_OFFSETS);
    jintArray set_words = (jintArray)(*env)->GetObjectArrayElement(env, q, PGM_Q_SET_WORDS);
    jintArray aggregations = (jintArray)(*env)->GetObjectArrayElement(env, q, PGM_Q_AGGREGATIONS);
    jintArray group_by = (jintArray)(*env)->GetObjectArrayElement(env, q, PGM_Q_GROUP_BY);
    jintArray limit_flags = (jintArray)(*env)->GetObjectArrayElement(env, q, PGM_Q_LIMIT_FLAGS);
    jint limit = 0, flags = 0;
    jint* lf = (limit_flags != NULL && (*env)->GetArrayLength(env, limit_flags) == PGM_Q_LIMIT_FLAGS_LEN) ? (*env)->GetIntArrayElements(env, limit_flags, NULL) : NULL;
    if (lf == NULL) {
      if (!(*env)->ExceptionCheck(env)) throw_new(env, "java/lang/IllegalArgumentException", "a batch query ends with int[PGM_Q_LIMIT_FLAGS_LEN] {numGroupsLimit, flags}");
      ok = 0;
    } else {
      limit = lf[0];
      flags = lf[1];
      (*env)->ReleaseIntArrayElements(env, limit_flags, lf, JNI_ABORT);
      pinned_query p;
      if (!pin_query(env, &p, nodes, pred_ints, pred_longs, set_offsets, set_words, aggregations, group_by, limit, flags)) {
        ok = 0;
      } else {
        built[i] = p.built;              /* owns copies of everything (pg_marshal.c): the Java arrays are let go right away */
        p.built = NULL;
        release_query(env, &p, nodes, pred_ints, pred_longs, set_offsets, set_words, aggregations, group_by);
        segments[i] = (pg_segment*)(intptr_t)h[i];
        lowered[i] = pgm_query_get(built[i]);
      }
    }
    if (limit_flags) (*env)->DeleteLocalRef(env, limit_flags);
    if (group_by) (*env)->DeleteLocalRef(env, group_by);
    if (aggregations) (*env)->DeleteLocalRef(env, aggregations);
    if (set_words) (*env)->DeleteLocalRef(env, set_words);
    if (set_offsets) (*env)->DeleteLocalRef(env, set_offsets);
    if (pred_longs) (*env)->DeleteLocalRef(env, pred_longs);
    if (pred_ints) (*env)->DeleteLocalRef(env, pred_ints);
    if (nodes) (*env)->DeleteLocalRef(env, nodes);
    (*env)->DeleteLocalRef(env, q);
  }
  if (ok) {
    const pg_status status = pg_execute_batch(segments, lowered, (int32_t)n, results, statuses);
    if (status != PG_OK) {
      throw_status(env, status);           /* nothing ran: every result is still empty */
      ok = 0;
    }
  }
  if (ok) {
    int first_failure = 1;
    for (jsize i = 0; i < n; i++) {
      jobject element = NULL;
      if (statuses[i] == PG_OK) {
        if (ok) element = result_to_java(env, &results[i], lowered[i]->num_group_by);      /* frees results[i] */
        else pg_result_free(&results[i]);                                                  /* an exception is pending: only release */
        if (element == NULL) { ok = 0; continue; }
      } else if (ok) {
        /* pg_last_error() of this thread names the first failed item (pg_execute_batch); later failures carry their status only */
        char message[512];
        const char* text = first_failure ? pg_last_error() : "see the first failed item of the batch";
        first_failure = 0;
        size_t at = 0;
        int32_t st = (int32_t)statuses[i];
        char digits[12];
        int nd = 0;
        do { digits[nd++] = (char)('0' + st % 10); st /= 10; } while (st > 0 && nd < 11);
        while (nd > 0) message[at++] = digits[--nd];
        message[at++] = '\n';
        for (size_t k = 0; text && text[k] && at + 1 < sizeof(message); k++) message[at++] = text[k];
        message[at] = 0;
        element = (*env)->NewStringUTF(env, message);
        if (element == NULL) { ok = 0; continue; }
      }
      if (element != NULL) {
        (*env)->SetObjectArrayElement(env, out, i, element);
        (*env)->DeleteLocalRef(env, element);
      }
    }
  }
  if (h) (*env)->ReleaseLongArrayElements(env, handles, h, JNI_ABORT);
  if (built) for (size_t i = 0; i < count; i++) pgm_query_free(built[i]);
  free(built); free(segments); free((void*)lowered); free(results); free(statuses);
  (*env)->DeleteLocalRef(env, object_class);
  return ok ? out : NULL;
}
