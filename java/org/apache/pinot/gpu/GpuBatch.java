/**
 * The native calls of ONE query over the segments of ONE server, made as one: PinotGpuNative.executeBatch (pg_execute_batch).
 *
 * <p>InstancePlanMakerImplV2.makeInstancePlan plans every segment of a query on one thread before anything runs
 * (core/plan/maker/InstancePlanMakerImplV2.java:166-193); BaseCombineOperator then calls the segment operators from the tasks of a
 * thread pool (core/operator/combine/BaseCombineOperator.java:85-142).  GpuPlanMaker registers every swim lane of every offloaded segment
 * here at plan time; whichever task asks for a lane's result first makes the native call for ALL of them, the other tasks wait for it
 * and take theirs.  A server holds hundreds of few-million-row segments per table: launched one by one each is a ~20 us kernel whose
 * launch latency, ramp and drain are most of its device time; in one call they share one launch (include/pinot_gpu.h, pg_execute_batch).
 *
 * <p>An item that fails makes only its own take() throw (the operator then re-plans that segment on the CPU, as it does for a failed
 * execute()); a call that fails as a whole makes every take() throw.
 */
package org.apache.pinot.gpu;

import java.util.ArrayList;
import java.util.List;


final class GpuBatch {
  private final List<GpuSegment> _segments = new ArrayList<>();
  private final List<GpuQueryLowering.Lowered> _queries = new ArrayList<>();
  private boolean _started;
  private boolean _done;
  private Object[] _results;
  private RuntimeException _failure;

  /** Plan time (one thread): the lane's slot in the batch. */
  synchronized int add(GpuSegment segment, GpuQueryLowering.Lowered query) {
    if (_started) {
      throw new IllegalStateException("the batch is already running");
    }
    _segments.add(segment);
    _queries.add(query);
    return _segments.size() - 1;
  }

  synchronized int size() {
    return _segments.size();
  }

  /** Run time (any combine task): the Object[PGM_RESULT_ARRAYS] of slot {@code index}; throws what PinotGpuNative.execute would have thrown. */
  Object[] take(int index) {
    boolean run = false;
    synchronized (this) {
      if (!_started) {
        _started = true;
        run = true;
      }
    }
    if (run) {
      Object[] results = null;
      RuntimeException failure = null;
      try {
        results = call();
      } catch (RuntimeException e) {
        failure = e;
      } catch (Error e) {
        failure = new RuntimeException(e);
      }
      synchronized (this) {
        _results = results;
        _failure = failure;
        _done = true;
        notifyAll();
      }
    }
    Object element;
    synchronized (this) {
      boolean interrupted = false;
      while (!_done) {
        try {
          wait();
        } catch (InterruptedException e) {
          interrupted = true;      // the query was cancelled: BaseOperator.nextBlock checks the flag again on the way out
        }
      }
      if (interrupted) {
        Thread.currentThread().interrupt();
      }
      if (_failure != null) {
        throw new RuntimeException("batch call failed: " + _failure.getMessage(), _failure);
      }
      element = _results[index];
      _results[index] = null;      // taken once
    }
    if (element instanceof Object[]) {
      return (Object[]) element;
    }
    if (element instanceof String) {
      String text = (String) element;
      int newline = text.indexOf('\n');
      int status = newline > 0 ? Integer.parseInt(text.substring(0, newline)) : PinotGpuNative.PG_ERR_INTERNAL;
      String message = newline >= 0 ? text.substring(newline + 1) : text;
      if (status == PinotGpuNative.PG_ERR_UNSUPPORTED) {
        throw new UnsupportedOperationException(message);
      }
      throw new RuntimeException(message);
    }
    throw new IllegalStateException("batch result " + index + " was already taken or is missing");
  }

  private Object[] call() {
    int n = _segments.size();
    long[] handles = new long[n];
    boolean[] pinned = new boolean[n];
    Object[][] queries = new Object[n][];
    try {
      return callPinned(n, handles, pinned, queries);
    } finally {
      for (int i = 0; i < n; i++) {
        if (pinned[i]) {
          _segments.get(i).refreshDeviceBytes();      // (what the copy holds may have grown: GpuSegmentCache.Account.update)
          _segments.get(i).unpin();
        }
      }
    }
  }

  private Object[] callPinned(int n, long[] handles, boolean[] pinned, Object[][] queries) {
    for (int i = 0; i < n; i++) {
      // a segment whose device copy left between plan time and now (HBM budget, release) travels as handle 0: pg_execute_batch fails that
      // item alone (PG_ERR_INVALID_ARGUMENT), its operator re-plans on the CPU like for any failed item
      pinned[i] = _segments.get(i).tryPin();
      handles[i] = pinned[i] ? _segments.get(i).handle() : 0;
      GpuQueryLowering.Lowered q = _queries.get(i);
      Object[] arrays = new Object[PinotGpuNative.PGM_QUERY_ARRAYS];
      arrays[PinotGpuNative.PGM_Q_FILTER_NODES] = q._filterNodes;
      arrays[PinotGpuNative.PGM_Q_PRED_INTS] = q._predInts;
      arrays[PinotGpuNative.PGM_Q_PRED_LONGS] = q._predLongs;
      arrays[PinotGpuNative.PGM_Q_SET_OFFSETS] = q._setOffsets;
      arrays[PinotGpuNative.PGM_Q_SET_WORDS] = q._setWords;
      arrays[PinotGpuNative.PGM_Q_AGGREGATIONS] = q._aggregations;
      arrays[PinotGpuNative.PGM_Q_GROUP_BY] = q._groupBy;
      int[] limitFlags = new int[PinotGpuNative.PGM_Q_LIMIT_FLAGS_LEN];
      limitFlags[0] = q._numGroupsLimit;
      limitFlags[1] = q._flags;
      arrays[PinotGpuNative.PGM_Q_LIMIT_FLAGS] = limitFlags;
      queries[i] = arrays;
    }
    Object[] results = PinotGpuNative.executeBatch(handles, queries);
    if (results == null || results.length != n) {
      throw new IllegalStateException("native batch result does not match jni/pinot_gpu_jni.c");
    }
    return results;
  }
}
