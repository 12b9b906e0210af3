/**
 * IndexSegment -> device-resident GpuSegment.  A segment is opened the first time a query reaches it and closed when the server drops
 * the IndexSegment: the cache holds the key weakly and registers the native handle with a Cleaner, so no hook into the segment data
 * manager is needed (IndexSegment.destroy() releases the mmap-ed buffers; the device copy does not depend on them).  For that to work the
 * VALUE must not reach the key: GpuSegment holds the segment's name and column table, never the IndexSegment, and the Cleaner's action
 * captures only the native handle.  (A WeakHashMap whose value references its key strongly never clears the entry.)
 *
 * <p><b>Devices.</b> One server process drives every device of the node (a Pinot server is one JVM; SURVEY.md section 8e: segment s on
 * device s mod N).  A segment goes to the device with the fewest resident bytes at the time it is opened -- for equal-sized segments
 * that IS s mod N in open order, and it stays balanced when sizes differ or segments come and go.  pg_segment_open places it; every
 * later native call switches to the segment's device by itself, and a query's lanes on different devices run side by side inside one
 * pg_execute_batch call.
 *
 * <p><b>HBM is released with the segment, not with the garbage collector.</b> Three ways, first one wins (the handle closes once:
 * GpuSegment.HandleBox): {@link #release(IndexSegment)} -- the call to add next to IndexSegment.destroy() where the deployment can
 * (INTEGRATION.md) --; the per-device budget ({@code gpu.hbm.budget.bytes}): opening a segment that would exceed it first closes the least
 * recently used segments of that device that no native call is using (they are re-opened when a query reaches them again); and the
 * Cleaner, as the safety net for an IndexSegment that simply becomes unreachable.
 *
 * <p>Segments that cannot be opened (mutable segments, unsupported layouts, out of device memory) are remembered as such: their queries
 * keep the CPU plan without trying again on every query.
 */
package org.apache.pinot.gpu;

import java.lang.ref.Cleaner;
import java.util.ArrayList;
import java.util.Collections;
import java.util.Comparator;
import java.util.List;
import java.util.Map;
import java.util.Optional;
import java.util.WeakHashMap;
import java.util.concurrent.atomic.AtomicLongArray;
import org.apache.pinot.segment.spi.ImmutableSegment;
import org.apache.pinot.segment.spi.IndexSegment;
import org.slf4j.Logger;
import org.slf4j.LoggerFactory;


final class GpuSegmentCache {
  private static final Logger LOGGER = LoggerFactory.getLogger(GpuSegmentCache.class);
  private static final Cleaner CLEANER = Cleaner.create();

  private final Map<IndexSegment, Optional<GpuSegment>> _segments = Collections.synchronizedMap(new WeakHashMap<>());
  private final int[] _devices;
  private final AtomicLongArray _residentBytes;          // per entry of _devices: bytes of the segments this cache holds there
  private final long _budgetBytesPerDevice;              // 0 = no budget

  GpuSegmentCache(int[] devices, long budgetBytesPerDevice) {
    if (devices == null || devices.length == 0) {
      throw new IllegalArgumentException("no device");
    }
    _devices = devices.clone();
    _residentBytes = new AtomicLongArray(_devices.length);
    _budgetBytesPerDevice = Math.max(0, budgetBytesPerDevice);
  }

  GpuSegmentCache(int device) {
    this(new int[]{device}, 0);
  }

  int[] devices() {
    return _devices.clone();
  }

  long residentBytes(int deviceSlot) {
    return _residentBytes.get(deviceSlot);
  }

  /** The device copy of the segment, or null when its queries keep the CPU plan. */
  GpuSegment get(IndexSegment indexSegment) {
    if (!(indexSegment instanceof ImmutableSegment) || indexSegment.getSegmentMetadata().getTotalDocs() == 0) {
      return null;
    }
    Optional<GpuSegment> cached = _segments.get(indexSegment);
    if (cached == null || (cached.isPresent() && cached.get().handle() == 0)) {       // never opened, or evicted under the budget since
      synchronized (this) {
        cached = _segments.get(indexSegment);
        if (cached == null || (cached.isPresent() && cached.get().handle() == 0)) {
          if (cached != null) {
            forget(cached.get());
          }
          cached = open(indexSegment);
          _segments.put(indexSegment, cached);
        }
      }
    }
    cached.ifPresent(GpuSegment::touch);
    return cached.orElse(null);
  }

  /**
   * The segment is going away (IndexSegment.destroy(): the server dropped, replaced or reloaded it; no query holds it any more --
   * SegmentDataManager's reference count guarantees that before destroy() runs): its HBM is given back now.
   */
  void release(IndexSegment indexSegment) {
    Optional<GpuSegment> cached = _segments.remove(indexSegment);
    if (cached != null && cached.isPresent()) {
      GpuSegment segment = cached.get();
      if (segment.closeIfIdle()) {
        forget(segment);
      }
      // (a pinned segment -- a caller broke the contract above -- is left to the Cleaner)
    }
  }

  /**
   * One open device copy's share of {@link #_residentBytes}: given back exactly once, whoever gets there first -- release(), eviction
   * under the budget, the re-open of an evicted segment, or the Cleaner.  (Round 4 subtracted an evicted segment's bytes a second time
   * when a query re-opened it, and never gave back the bytes of a garbage-collected segment: the counter drifted, makeRoom() stopped
   * evicting and leastLoadedSlot() was skewed.)  Holds no reference to the segment, so the Cleaner's action may capture it.
   */
  static final class Account {
    private final AtomicLongArray _resident;
    private final int _slot;
    private long _bytes;                                    // guarded by this
    private boolean _returned;                              // guarded by this

    Account(AtomicLongArray resident, int slot, long bytes) {
      _resident = resident;
      _slot = slot;
      _bytes = Math.max(0, bytes);
      _resident.addAndGet(slot, _bytes);
    }

    synchronized void giveBack() {
      if (!_returned) {
        _returned = true;
        _resident.addAndGet(_slot, -_bytes);
      }
    }

    /**
     * The copy's bytes as pg_segment_device_bytes reports them NOW: a copy grows after it was opened -- value planes, key images, the
     * dictionary and rank image of a raw FLOAT / DOUBLE group-by key, the scratch of the numEntriesScannedInFilter passes -- and the
     * per-device budget (makeRoom, leastLoadedSlot) has to see that memory.  No effect once the share was given back.
     */
    synchronized void update(long bytesNow) {
      if (!_returned) {
        long now = Math.max(0, bytesNow);
        _resident.addAndGet(_slot, now - _bytes);
        _bytes = now;
      }
    }
  }

  private static void forget(GpuSegment segment) {
    Account account = segment.account();
    if (account != null) {
      account.giveBack();
    }
  }

  /** The device a new segment goes to: the one with the fewest resident bytes (the lowest-numbered one among equals). */
  private int leastLoadedSlot() {
    int best = 0;
    for (int i = 1; i < _devices.length; i++) {
      if (_residentBytes.get(i) < _residentBytes.get(best)) {
        best = i;
      }
    }
    return best;
  }

  /** Under the budget: close least-recently-used idle segments of the device until `incomingBytes` more fit (or nothing idle is left). */
  private void makeRoom(int slot, long incomingBytes) {
    if (_budgetBytesPerDevice == 0 || _residentBytes.get(slot) + incomingBytes <= _budgetBytesPerDevice) {
      return;
    }
    List<GpuSegment> candidates = new ArrayList<>();
    synchronized (_segments) {
      for (Optional<GpuSegment> value : _segments.values()) {
        if (value.isPresent() && value.get().device() == _devices[slot] && value.get().handle() != 0) {
          candidates.add(value.get());
        }
      }
    }
    candidates.sort(Comparator.comparingLong(GpuSegment::lastUsedNanos));
    for (GpuSegment victim : candidates) {
      if (_residentBytes.get(slot) + incomingBytes <= _budgetBytesPerDevice) {
        return;
      }
      if (victim.closeIfIdle()) {
        forget(victim);
        LOGGER.info("Segment {} left device {} (HBM budget); it is re-opened by the next query that reaches it", victim.getSegmentName(), _devices[slot]);
      }
    }
  }

  /** On-disk bytes of the segment: what its device copy will roughly take (index buffers are copied as they are). */
  private static long estimatedBytes(IndexSegment indexSegment) {
    try {
      return indexSegment instanceof ImmutableSegment ? Math.max(0, ((ImmutableSegment) indexSegment).getSegmentSizeBytes()) : 0;
    } catch (RuntimeException e) {
      return 0;
    }
  }

  private Optional<GpuSegment> open(IndexSegment indexSegment) {
    int slot = leastLoadedSlot();
    int device = _devices[slot];
    try {
      makeRoom(slot, estimatedBytes(indexSegment));
      GpuSegment segment = GpuSegment.open(indexSegment, device);
      Account account = new Account(_residentBytes, slot, segment.deviceBytes());
      segment.setAccount(account);
      GpuSegment.HandleBox box = segment.box();
      // the action must not reference `segment` or `indexSegment` (it would keep them reachable): the box holds the handle and the
      // account a slot number and a byte count, nothing else
      CLEANER.register(indexSegment, () -> {
        box.close();
        account.giveBack();
      });
      LOGGER.info("Segment {} resident on device {}: {} bytes of HBM ({} bytes on that device now)", indexSegment.getSegmentName(), device,
          segment.deviceBytes(), _residentBytes.get(slot));
      return Optional.of(segment);
    } catch (Exception | UnsatisfiedLinkError e) {
      LOGGER.warn("Segment {} stays on the CPU plan: {}", indexSegment.getSegmentName(), e.toString());
      return Optional.empty();
    }
  }
}
