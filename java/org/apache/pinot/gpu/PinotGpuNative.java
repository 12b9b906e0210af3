/**
 * Native methods of the MI355X segment executor: one per C-ABI entry point of include/pinot_gpu.h that the server path needs
 * (jni/pinot_gpu_jni.c holds the JNI functions, jni/pg_marshal.c the array marshalling they share with the tests).
 *
 * <p>Arrays instead of objects: a query crosses as the flat arrays documented in jni/pg_marshal.h (GpuQueryLowering writes them), a result
 * comes back as {@code Object[8]} of primitive arrays (GpuResult reads them).  The only native state Java ever holds is the segment
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

  /** pg_status values the Java side distinguishes (include/pinot_gpu.h:53-61). */
  public static final int PG_OK = 0;
  public static final int PG_ERR_UNSUPPORTED = 2;

  /** pg_query.flags */
  public static final int PG_QUERY_NULL_HANDLING = 1;

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
   * pg_segment_open.  {@code columnInts}: 6 per column {storedType, fwdEncoding, bitsPerValue, cardinality, hasDictionary, 0};
   * {@code columnBuffers}: 8 per column {fwd address, fwd size, dict address, dict size, inverted address, inverted size, null-vector
   * address, null-vector size}, 0 / 0 where an index does not exist.  The buffers are read during the call only.
   */
  static native long segmentOpen(String name, long crc, int device, int numDocs, String[] columnNames, int[] columnInts, long[] columnBuffers);

  /** pg_segment_close */
  static native void segmentClose(long handle);

  /** pg_segment_device_bytes */
  static native long segmentDeviceBytes(long handle);

  /** pg_query_check: PG_OK or PG_ERR_UNSUPPORTED; nothing is launched. */
  static native int queryCheck(long handle, int[] filterNodes, int[] predInts, long[] predLongs, int[] setOffsets, int[] setWords,
      int[] aggregations, int[] groupBy, int numGroupsLimit, int flags);

  /**
   * pg_execute.  Returns {long[] header, int[] groupIds, long[] counts, double[] sums, long[] sumsI64, int[] sumExact, double[] mins,
   * double[] maxs}; throws UnsupportedOperationException for PG_ERR_UNSUPPORTED, RuntimeException (pg_last_error) otherwise.
   */
  static native Object[] execute(long handle, int[] filterNodes, int[] predInts, long[] predLongs, int[] setOffsets, int[] setWords,
      int[] aggregations, int[] groupBy, int numGroupsLimit, int flags);
}
