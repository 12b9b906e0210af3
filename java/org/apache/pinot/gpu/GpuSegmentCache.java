/**
 * IndexSegment -> device-resident GpuSegment.  A segment is opened the first time a query reaches it and closed when the server drops
 * the IndexSegment: the cache holds the key weakly and registers the native handle with a Cleaner, so no hook into the segment data
 * manager is needed (IndexSegment.destroy() releases the mmap-ed buffers; the device copy does not depend on them).  For that to work the
 * VALUE must not reach the key: GpuSegment holds the segment's name and column table, never the IndexSegment, and the Cleaner's action
 * captures only the native handle.  (A WeakHashMap whose value references its key strongly never clears the entry.)
 *
 * <p>Segments that cannot be opened (mutable segments, unsupported layouts, out of device memory) are remembered as such: their queries
 * keep the CPU plan without trying again on every query.
 */
package org.apache.pinot.gpu;

import java.lang.ref.Cleaner;
import java.util.Collections;
import java.util.Map;
import java.util.Optional;
import java.util.WeakHashMap;
import org.apache.pinot.segment.spi.ImmutableSegment;
import org.apache.pinot.segment.spi.IndexSegment;
import org.slf4j.Logger;
import org.slf4j.LoggerFactory;


final class GpuSegmentCache {
  private static final Logger LOGGER = LoggerFactory.getLogger(GpuSegmentCache.class);
  private static final Cleaner CLEANER = Cleaner.create();

  private final Map<IndexSegment, Optional<GpuSegment>> _segments = Collections.synchronizedMap(new WeakHashMap<>());
  private final int _device;

  GpuSegmentCache(int device) {
    _device = device;
  }

  /** The device copy of the segment, or null when its queries keep the CPU plan. */
  GpuSegment get(IndexSegment indexSegment) {
    if (!(indexSegment instanceof ImmutableSegment) || indexSegment.getSegmentMetadata().getTotalDocs() == 0) {
      return null;
    }
    Optional<GpuSegment> cached = _segments.get(indexSegment);
    if (cached == null) {
      synchronized (this) {
        cached = _segments.get(indexSegment);
        if (cached == null) {
          cached = open(indexSegment);
          _segments.put(indexSegment, cached);
        }
      }
    }
    return cached.orElse(null);
  }

  private Optional<GpuSegment> open(IndexSegment indexSegment) {
    try {
      GpuSegment segment = GpuSegment.open(indexSegment, _device);
      long handle = segment.handle();
      // the action must not reference `segment` or `indexSegment` (it would keep them reachable)
      CLEANER.register(indexSegment, () -> PinotGpuNative.segmentClose(handle));
      LOGGER.info("Segment {} resident on device {}: {} bytes of HBM", indexSegment.getSegmentName(), _device,
          PinotGpuNative.segmentDeviceBytes(handle));
      return Optional.of(segment);
    } catch (Exception | UnsatisfiedLinkError e) {
      LOGGER.warn("Segment {} stays on the CPU plan: {}", indexSegment.getSegmentName(), e.toString());
      return Optional.empty();
    }
  }
}
