/**
 * The plan maker of the MI355X segment executor.  Named in the server configuration
 * ({@code pinot.server.query.executor.plan.maker.class=org.apache.pinot.gpu.GpuPlanMaker}; ServerQueryExecutorV1Impl instantiates it
 * reflectively, core/query/executor/ServerQueryExecutorV1Impl.java:116-123); everything except makeSegmentPlanNode is inherited, so
 * instance plans, combine operators, streaming, prefetch and query options stay the reference's.
 *
 * <p>makeSegmentPlanNode (InstancePlanMakerImplV2.java:270-289) swaps the per-segment plan node of an aggregation / group-by query for
 * the device operator when, at PLAN time, all of this holds: the segment is resident (GpuSegmentCache), the query lowers
 * (GpuQueryLowering) and pg_query_check admits it (PinotGpuNative.queryCheck -- the same decision pg_execute would take, nothing
 * launched).  Otherwise the reference's own node is returned, so an unsupported shape never fails at run time.
 */
package org.apache.pinot.gpu;

import org.apache.pinot.core.plan.PlanNode;
import org.apache.pinot.core.plan.maker.InstancePlanMakerImplV2;
import org.apache.pinot.core.query.aggregation.function.AggregationFunction;
import org.apache.pinot.core.query.request.context.QueryContext;
import org.apache.pinot.core.query.request.context.utils.QueryContextUtils;
import org.apache.pinot.segment.spi.SegmentContext;
import org.apache.pinot.spi.env.PinotConfiguration;
import org.slf4j.Logger;
import org.slf4j.LoggerFactory;


public class GpuPlanMaker extends InstancePlanMakerImplV2 {
  private static final Logger LOGGER = LoggerFactory.getLogger(GpuPlanMaker.class);
  public static final String DEVICE_KEY = "gpu.device";
  public static final String ENABLED_KEY = "gpu.enabled";

  private volatile GpuSegmentCache _segments;

  @Override
  public void init(PinotConfiguration queryExecutorConfig) {
    super.init(queryExecutorConfig);
    if (!queryExecutorConfig.getProperty(ENABLED_KEY, true)) {
      return;
    }
    int device = queryExecutorConfig.getProperty(DEVICE_KEY, 0);
    try {
      PinotGpuNative.init(device, 0);
      _segments = new GpuSegmentCache(device);
      LOGGER.info("Segment executor on device {}: {}", device, PinotGpuNative.version());
    } catch (RuntimeException | UnsatisfiedLinkError e) {
      LOGGER.warn("No device executor, every query keeps the CPU plan: {}", e.toString());
    }
  }

  @Override
  public PlanNode makeSegmentPlanNode(SegmentContext segmentContext, QueryContext queryContext) {
    GpuSegmentCache segments = _segments;
    if (segments == null || !QueryContextUtils.isAggregationQuery(queryContext) || queryContext.hasFilteredAggregations()) {
      // (FILTER (WHERE ...) aggregations: one native call per distinct filter, the swim lanes of FilteredAggregationOperator -- the C++
      //  host mirror implements it, pinot_amd/csrc/host/plan_maker.cpp GpuFilteredAggregationOperator; not wired here yet)
      return super.makeSegmentPlanNode(segmentContext, queryContext);
    }
    GpuSegment segment = segments.get(segmentContext.getIndexSegment());
    AggregationFunction[] functions = queryContext.getAggregationFunctions();
    if (segment == null || functions == null) {
      return super.makeSegmentPlanNode(segmentContext, queryContext);
    }
    GpuQueryLowering.Lowered lowered = GpuQueryLowering.lower(segment, queryContext, functions, queryContext.getFilter());
    if (lowered == null) {
      return super.makeSegmentPlanNode(segmentContext, queryContext);
    }
    int admitted = PinotGpuNative.queryCheck(segment.handle(), lowered._filterNodes, lowered._predInts, lowered._predLongs, lowered._setOffsets,
        lowered._setWords, lowered._aggregations, lowered._groupBy, lowered._numGroupsLimit, lowered._flags);
    if (admitted != PinotGpuNative.PG_OK) {
      LOGGER.debug("Segment {} keeps the CPU plan: {}", segment.getIndexSegment().getSegmentName(), PinotGpuNative.lastError());
      return super.makeSegmentPlanNode(segmentContext, queryContext);
    }
    return () -> new GpuAggregationOperator(segment, queryContext, functions, lowered);
  }
}
