/**
 * One immutable segment resident in HBM: the native handle of pg_segment_open plus the column table the lowering needs.
 *
 * <p>Opening reads every supported column's index buffers ONCE: the segment directory is re-opened read-only (the same mmap-ed files the
 * server already holds; SegmentDirectoryLoaderRegistry, pinot-segment-spi/.../loader/SegmentDirectoryLoaderRegistry.java:88), each
 * buffer's address is handed to pg_segment_open, which copies it to the device, and the directory is closed again.  Afterwards the JVM
 * keeps no reference to those buffers on behalf of the device.
 *
 * <p>What reaches the device per column (pg_column_desc, include/pinot_gpu.h:88-105): single-value columns of stored type INT / LONG /
 * FLOAT / DOUBLE with a dictionary (fixed-bit forward index + big-endian dictionary file, + the bitmap inverted index and the null value
 * vector when they exist) or raw (PASS_THROUGH fixed-byte chunks); STRING columns with a dictionary travel as their dictIds under a
 * placeholder dictionary {0..cardinality-1} (their values never reach the device: predicates are lowered to dictIds here, group keys
 * are mapped back through the Java dictionary).  Sorted columns store [start, end] docId pairs instead of a dictId per doc
 * (SortedIndexReaderImpl): the fixed-bit stream the device scans is packed here from those pairs, in the layout of
 * FixedBitSVForwardIndexWriter / PinotDataBitSet.  Anything else is left out, and a query that touches it keeps the CPU plan.
 */
package org.apache.pinot.gpu;

import java.io.Closeable;
import java.io.File;
import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.util.ArrayList;
import java.util.HashMap;
import java.util.List;
import java.util.Map;
import org.apache.pinot.segment.spi.IndexSegment;
import org.apache.pinot.segment.spi.compression.ChunkCompressionType;
import org.apache.pinot.segment.spi.datasource.DataSource;
import org.apache.pinot.segment.spi.datasource.DataSourceMetadata;
import org.apache.pinot.segment.spi.index.StandardIndexes;
import org.apache.pinot.segment.spi.index.reader.Dictionary;
import org.apache.pinot.segment.spi.index.reader.ForwardIndexReader;
import org.apache.pinot.segment.spi.index.reader.SortedIndexReader;
import org.apache.pinot.segment.spi.loader.SegmentDirectoryLoaderContext;
import org.apache.pinot.segment.spi.loader.SegmentDirectoryLoaderRegistry;
import org.apache.pinot.segment.spi.memory.PinotDataBuffer;
import org.apache.pinot.segment.spi.store.SegmentDirectory;
import org.apache.pinot.spi.data.FieldSpec.DataType;
import org.apache.pinot.spi.env.PinotConfiguration;
import org.apache.pinot.spi.utils.Pairs;


final class GpuSegment implements Closeable {
  // pg_data_type / pg_fwd_encoding (the numbers live in PinotGpuNative)
  private static final int TYPE_INT = PinotGpuNative.PG_TYPE_INT;
  private static final int TYPE_LONG = PinotGpuNative.PG_TYPE_LONG;
  private static final int TYPE_FLOAT = PinotGpuNative.PG_TYPE_FLOAT;
  private static final int TYPE_DOUBLE = PinotGpuNative.PG_TYPE_DOUBLE;
  private static final int FWD_FIXED_BIT_DICT = PinotGpuNative.PG_FWD_FIXED_BIT_DICT;
  private static final int FWD_RAW_FIXED_BYTE = PinotGpuNative.PG_FWD_RAW_FIXED_BYTE;

  // NO reference to the IndexSegment: GpuSegmentCache keys a WeakHashMap by it, and a value that reached its key would pin both (and the
  // HBM copy) for the JVM's lifetime.  Whoever needs the IndexSegment has it from the SegmentContext of the query at hand.
  private final String _segmentName;
  private final Map<String, Integer> _columnIndex = new HashMap<>();
  private final List<String> _columnNames = new ArrayList<>();
  private final List<Boolean> _hasDictionary = new ArrayList<>();
  private final List<Boolean> _numeric = new ArrayList<>();
  private final int _numDocs;
  private final int _device;
  // The native handle lives in a box of its own: the Cleaner action GpuSegmentCache registers against the IndexSegment shares the box (not
  // this object, which must stay collectable with its key) -- whoever closes first wins, nobody closes twice.
  private final HandleBox _box = new HandleBox();
  private int _pins;                                     // native calls in flight on the handle (guarded by `this`)
  private volatile long _lastUsedNanos = System.nanoTime();
  private volatile long _deviceBytes;
  private volatile GpuSegmentCache.Account _account;

  /** The native handle and its one-time close. */
  static final class HandleBox {
    private long _handle;

    synchronized long get() {
      return _handle;
    }

    synchronized void set(long handle) {
      _handle = handle;
    }

    /** pg_segment_close, once; returns whether this call closed it. */
    synchronized boolean close() {
      if (_handle == 0) {
        return false;
      }
      PinotGpuNative.segmentClose(_handle);
      _handle = 0;
      return true;
    }
  }

  private GpuSegment(IndexSegment indexSegment, int device) {
    _segmentName = indexSegment.getSegmentName();
    _numDocs = indexSegment.getSegmentMetadata().getTotalDocs();
    _device = device;
  }

  HandleBox box() {
    return _box;
  }

  int device() {
    return _device;
  }

  long deviceBytes() {
    return _deviceBytes;
  }

  /** This copy's share of the cache's per-device byte count (GpuSegmentCache.Account: given back once). */
  GpuSegmentCache.Account account() {
    return _account;
  }

  void setAccount(GpuSegmentCache.Account account) {
    _account = account;
  }

  /**
   * After a query ran on this copy: what it holds on the device may have grown (a plane, a key image or a rank image built by the query, the
   * scratch of a statistics pass) -- the cache's per-device byte count follows.  Called with the segment pinned (the handle is alive).
   */
  void refreshDeviceBytes() {
    GpuSegmentCache.Account account = _account;
    if (account == null) {
      return;
    }
    try {
      long now = PinotGpuNative.segmentDeviceBytes(handle());
      _deviceBytes = now;
      account.update(now);
    } catch (RuntimeException e) {
      // the byte count is bookkeeping: a failed read leaves it as it was
    }
  }

  long lastUsedNanos() {
    return _lastUsedNanos;
  }

  void touch() {
    _lastUsedNanos = System.nanoTime();
  }

  /**
   * Before a native call on the handle: false when the device copy is gone (evicted under the HBM budget between plan time and run time,
   * or released with its IndexSegment) -- the caller then runs its CPU plan.  pg_segment_close assumes no pg_execute in flight on the
   * handle (include/pinot_gpu.h): a pinned segment is never closed, closeIfIdle() refuses.
   */
  synchronized boolean tryPin() {
    if (_box.get() == 0) {
      return false;
    }
    _pins++;
    return true;
  }

  synchronized void unpin() {
    _pins--;
  }

  /** Eviction: closes the device copy unless a native call is using it. */
  synchronized boolean closeIfIdle() {
    if (_pins > 0) {
      return false;
    }
    _box.close();
    return true;
  }

  String getSegmentName() {
    return _segmentName;
  }

  long handle() {
    return _box.get();
  }

  int numDocs() {
    return _numDocs;
  }

  /** Index of a column in the device segment; NotOffloadable when the column did not qualify. */
  int columnIndex(String column) {
    Integer index = _columnIndex.get(column);
    if (index == null) {
      throw new GpuQueryLowering.NotOffloadable("column " + column + " is not on the device");
    }
    return index;
  }

  String columnName(int index) {
    return _columnNames.get(index);
  }

  boolean hasDictionary(int column) {
    return _hasDictionary.get(column);
  }

  boolean isNumeric(int column) {
    return _numeric.get(column);
  }

  @Override
  public synchronized void close() {
    _box.close();
  }

  static GpuSegment open(IndexSegment indexSegment, int device)
      throws Exception {
    GpuSegment segment = new GpuSegment(indexSegment, device);
    File indexDir = indexSegment.getSegmentMetadata().getIndexDir();
    SegmentDirectoryLoaderContext context = new SegmentDirectoryLoaderContext.Builder()
        .setSegmentName(indexSegment.getSegmentName())
        .setSegmentDirectoryConfigs(new PinotConfiguration(Map.of("readMode", "mmap")))
        .build();
    List<Integer> ints = new ArrayList<>();
    List<Long> buffers = new ArrayList<>();
    List<ByteBuffer> keepAlive = new ArrayList<>();            // direct buffers made here (placeholder dictionaries, packed sorted columns)
    try (SegmentDirectory directory = SegmentDirectoryLoaderRegistry.getDefaultSegmentDirectoryLoader().load(indexDir.toURI(), context);
        SegmentDirectory.Reader reader = directory.createReader()) {
      for (String column : indexSegment.getPhysicalColumnNames()) {
        DataSource dataSource = indexSegment.getDataSource(column);
        DataSourceMetadata metadata = dataSource.getDataSourceMetadata();
        ForwardIndexReader<?> forwardIndex = dataSource.getForwardIndex();
        if (forwardIndex == null || !metadata.isSingleValue()) {
          continue;
        }
        DataType storedType = metadata.getDataType().getStoredType();
        Dictionary dictionary = dataSource.getDictionary();
        boolean numeric = storedType == DataType.INT || storedType == DataType.LONG || storedType == DataType.FLOAT || storedType == DataType.DOUBLE;
        if (!numeric && !(storedType == DataType.STRING && dictionary != null)) {
          continue;
        }
        long[] fwd;
        long[] dict = {0, 0};
        long[] inverted = {0, 0};
        long[] nulls = {0, 0};
        int encoding;
        int bits = 0;
        int cardinality = 0;
        if (dictionary != null) {
          encoding = FWD_FIXED_BIT_DICT;
          cardinality = dictionary.length();
          bits = numBitsPerValue(cardinality - 1);
          if (metadata.isSorted() && dataSource.getInvertedIndex() instanceof SortedIndexReader) {
            ByteBuffer packed = packSorted((SortedIndexReader<?>) dataSource.getInvertedIndex(), cardinality, bits, segment._numDocs);
            keepAlive.add(packed);
            fwd = new long[]{PinotGpuNative.directBufferAddress(packed), packed.capacity()};
          } else {
            fwd = addressOf(reader.getIndexFor(column, StandardIndexes.forward()));
            if (reader.hasIndexFor(column, StandardIndexes.inverted())) {
              inverted = addressOf(reader.getIndexFor(column, StandardIndexes.inverted()));
            }
          }
          if (numeric) {
            // the .dict file is the header-less big-endian value array of Int / Long / Float / DoubleDictionary
            dict = addressOf(reader.getIndexFor(column, StandardIndexes.dictionary()));
          } else {
            ByteBuffer placeholder = ByteBuffer.allocateDirect(4 * cardinality).order(ByteOrder.BIG_ENDIAN);
            for (int d = 0; d < cardinality; d++) {
              placeholder.putInt(4 * d, d);
            }
            keepAlive.add(placeholder);
            dict = new long[]{PinotGpuNative.directBufferAddress(placeholder), placeholder.capacity()};
          }
        } else {
          if (forwardIndex.getCompressionType() != ChunkCompressionType.PASS_THROUGH) {
            continue;                                            // compressed raw chunks are decoded on the CPU plan
          }
          encoding = FWD_RAW_FIXED_BYTE;
          fwd = addressOf(reader.getIndexFor(column, StandardIndexes.forward()));
        }
        if (reader.hasIndexFor(column, StandardIndexes.nullValueVector())) {
          nulls = addressOf(reader.getIndexFor(column, StandardIndexes.nullValueVector()));
        }
        int storedCode = storedType == DataType.LONG ? TYPE_LONG : (storedType == DataType.FLOAT ? TYPE_FLOAT : (storedType == DataType.DOUBLE ? TYPE_DOUBLE : TYPE_INT));
        segment._columnIndex.put(column, segment._columnNames.size());
        segment._columnNames.add(column);
        segment._hasDictionary.add(dictionary != null);
        segment._numeric.add(numeric);
        int[] columnInts = {storedCode, encoding, bits, cardinality, dictionary != null ? 1 : 0, 0};
        long[][] columnBuffers = {fwd, dict, inverted, nulls};
        if (columnInts.length != PinotGpuNative.PGM_COLUMN_INTS || 2 * columnBuffers.length != PinotGpuNative.PGM_COLUMN_BUFFERS) {
          throw new IllegalStateException("column record does not match jni/pg_marshal.h");
        }
        for (int v : columnInts) {
          ints.add(v);
        }
        for (long[] pair : columnBuffers) {
          buffers.add(pair[0]);
          buffers.add(pair[1]);
        }
      }
      long crc = Long.parseLong(indexSegment.getSegmentMetadata().getCrc());
      segment._box.set(PinotGpuNative.segmentOpen(indexSegment.getSegmentName(), crc, device, segment._numDocs,
          segment._columnNames.toArray(new String[0]), ints.stream().mapToInt(Integer::intValue).toArray(),
          buffers.stream().mapToLong(Long::longValue).toArray()));
      segment._deviceBytes = PinotGpuNative.segmentDeviceBytes(segment._box.get());
    }
    keepAlive.clear();       // pg_segment_open has copied everything to the device
    return segment;
  }

  /** {address, size} of a mapped index buffer.  toDirectByteBuffer takes an int size: the base address is all that is needed. */
  private static long[] addressOf(PinotDataBuffer buffer) {
    long size = buffer.size();
    ByteBuffer head = buffer.toDirectByteBuffer(0, (int) Math.min(size, 1 << 20));
    return new long[]{PinotGpuNative.directBufferAddress(head), size};
  }

  /** PinotDataBitSet.getNumBitsPerValue (pinot-segment-local/.../io/util/PinotDataBitSet.java:61-72) */
  static int numBitsPerValue(int maxValue) {
    return maxValue <= 0 ? 1 : 32 - Integer.numberOfLeadingZeros(maxValue);
  }

  /** The dictId of every doc of a sorted column as the MSB-first fixed-bit stream of FixedBitSVForwardIndexWriter. */
  private static ByteBuffer packSorted(SortedIndexReader<?> sortedIndex, int cardinality, int bits, int numDocs) {
    long totalBits = (long) numDocs * bits;
    ByteBuffer out = ByteBuffer.allocateDirect((int) ((totalBits + 7) / 8));
    long bit = 0;
    for (int dictId = 0; dictId < cardinality; dictId++) {
      Pairs.IntPair range = sortedIndex.getDocIds(dictId);
      for (int doc = range.getLeft(); doc <= range.getRight(); doc++) {
        for (int b = bits - 1; b >= 0; b--, bit++) {
          if (((dictId >>> b) & 1) != 0) {
            int at = (int) (bit >>> 3);
            out.put(at, (byte) (out.get(at) | (0x80 >>> (int) (bit & 7))));
          }
        }
      }
    }
    return out;
  }
}
