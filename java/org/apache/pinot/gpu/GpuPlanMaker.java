/**
 * The plan maker of the MI355X segment executor.  Named in the server configuration
 * ({@code pinot.server.query.executor.plan.maker.class=org.apache.pinot.gpu.GpuPlanMaker}; ServerQueryExecutorV1Impl instantiates it
 * reflectively, core/query/executor/ServerQueryExecutorV1Impl.java:116-123); everything except makeSegmentPlanNode is inherited, so
 * instance plans, combine operators, streaming, prefetch and query options stay the reference's -- makeInstancePlan is overridden only to
 * open a GpuBatch around the reference's own planning loop, so that the query's segments share one native call ({@code gpu.batch=false}
 * turns that off).
 *
 * <p>makeSegmentPlanNode (InstancePlanMakerImplV2.java:270-289) swaps the per-segment plan node of an aggregation / group-by query for
 * the device operator when, at PLAN time, all of this holds: the segment is resident (GpuSegmentCache), every swim lane of the query
 * lowers (GpuQueryLowering) and pg_query_check admits it (PinotGpuNative.queryCheck -- the same decision pg_execute would take, nothing
 * launched).  Otherwise the reference's own node is returned.  The reference's node is ALSO kept inside the device operator: a run-time
 * failure of the native call (device out of memory for a group table, a tier pg_query_check admits by upper bound) re-plans the segment on
 * the CPU instead of failing the query (GpuAggregationOperator.getNextBlock).
 *
 * <p>Plan choices of the reference this class keeps or declines, one by one (AggregationPlanNode / GroupByPlanNode / FilterPlanNode):
 * <ul>
 *   <li>expression override hints: rewriteQueryContextWithHints runs first, as in the reference (:271);</li>
 *   <li>upsert / dedup tables: FilterPlanNode.run ANDs SegmentContext.getQueryableDocIdsSnapshot() into every filter (:89-102).  The
 *       device holds no valid-doc bitmap, so a segment context that carries one keeps the CPU plan;</li>
 *   <li>FILTER (WHERE ...) aggregations: one lane per distinct clause over mainFilter AND clause, match-all clauses folded into the
 *       unfiltered lane, plus -- for GROUP BY without skipEmptyGroups -- the lane that only creates the main filter's groups
 *       (AggregationFunctionUtils.buildFilteredAggregationInfos :312-402).  Every lane is one native call;</li>
 *   <li>star-tree: AggregationFunctionUtils.buildAggregationInfo prefers a fitting star-tree.  The device plan scans the forward indexes
 *       instead (same results; a segment whose queries should stay on their star-tree is left to the CPU plan with gpu.skip.startree
 *       segments, see INTEGRATION.md);</li>
 *   <li>metadata / dictionary fast paths and FastFilteredCount: pg_execute applies the same rules (include/pinot_gpu.h).</li>
 * </ul>
 */
package org.apache.pinot.gpu;

import java.util.ArrayList;
import java.util.Arrays;
import java.util.LinkedHashMap;
import java.util.List;
import java.util.Map;
import java.util.concurrent.ExecutorService;
import org.apache.commons.lang3.tuple.Pair;
import org.apache.pinot.common.metrics.ServerMetrics;
import org.apache.pinot.common.request.context.FilterContext;
import org.apache.pinot.common.utils.config.QueryOptionsUtils;
import org.apache.pinot.core.plan.Plan;
import org.apache.pinot.core.plan.PlanNode;
import org.apache.pinot.core.plan.maker.InstancePlanMakerImplV2;
import org.apache.pinot.core.query.aggregation.function.AggregationFunction;
import org.apache.pinot.core.query.request.context.QueryContext;
import org.apache.pinot.core.query.request.context.utils.QueryContextUtils;
import org.apache.pinot.segment.spi.IndexSegment;
import org.apache.pinot.segment.spi.SegmentContext;
import org.apache.pinot.spi.env.PinotConfiguration;
import org.slf4j.Logger;
import org.slf4j.LoggerFactory;


public class GpuPlanMaker extends InstancePlanMakerImplV2 {
  private static final Logger LOGGER = LoggerFactory.getLogger(GpuPlanMaker.class);
  public static final String DEVICE_KEY = "gpu.device";
  /** The devices this server drives: "0-7", "0,2,4" or a mix ("0-3,6"); default: the single device of {@code gpu.device}. */
  public static final String DEVICES_KEY = "gpu.devices";
  /** Bytes of HBM per device the resident segments may take; least recently used idle segments leave first (0 = no budget). */
  public static final String HBM_BUDGET_KEY = "gpu.hbm.budget.bytes";
  public static final String ENABLED_KEY = "gpu.enabled";
  public static final String SKIP_STAR_TREE_SEGMENTS_KEY = "gpu.skip.startree";

  public static final String BATCH_KEY = "gpu.batch";
  /**
   * ExecutionStatistics.numEntriesScannedInFilter of a filter whose iterators leap-frog (a root AND over scan-based leaves, OR / NOT children)
   * is a walk of its own behind the query on the device -- up to several times the query for a NOT child.  {@code false}: such filters run
   * nothing but the query and report the upper bound numDocs x scan leaves (PG_QUERY_STATS_UPPER_BOUND_OK); everything else of the result is
   * unchanged.  Default true (the reference's exact count).  Per query: the query option {@code gpuExactFilterStats=false}.
   */
  public static final String EXACT_FILTER_STATS_KEY = "gpu.exact.filter.stats";
  public static final String EXACT_FILTER_STATS_QUERY_OPTION = "gpuExactFilterStats";

  private volatile GpuSegmentCache _segments;
  private boolean _skipStarTreeSegments;
  private boolean _batchEnabled = true;
  private final ThreadLocal<GpuBatch> _planningBatch = new ThreadLocal<>();      // set while makeInstancePlan plans a query's segments

  @Override
  public void init(PinotConfiguration queryExecutorConfig) {
    super.init(queryExecutorConfig);
    if (!queryExecutorConfig.getProperty(ENABLED_KEY, true)) {
      return;
    }
    int device = queryExecutorConfig.getProperty(DEVICE_KEY, 0);
    _skipStarTreeSegments = queryExecutorConfig.getProperty(SKIP_STAR_TREE_SEGMENTS_KEY, false);
    _batchEnabled = queryExecutorConfig.getProperty(BATCH_KEY, true);
    GpuQueryLowering.setExactFilterStats(queryExecutorConfig.getProperty(EXACT_FILTER_STATS_KEY, true));
    try {
      // One process, every device of the node: pg_init once (its device is only the default for segments that name none), segments
      // placed over the devices by GpuSegmentCache, the library switching to a segment's device in every call.
      int[] devices = parseDevices(queryExecutorConfig.getProperty(DEVICES_KEY, Integer.toString(device)));
      long budget = queryExecutorConfig.getProperty(HBM_BUDGET_KEY, 0L);
      PinotGpuNative.init(devices[0], 0);
      _segments = new GpuSegmentCache(devices, budget);
      LOGGER.info("Segment executor on devices {}: {}", Arrays.toString(devices), PinotGpuNative.version());
    } catch (RuntimeException | UnsatisfiedLinkError e) {
      LOGGER.warn("No device executor, every query keeps the CPU plan: {}", e.toString());
    }
  }

  /** "0-7", "0,2,4", "0-3,6" -> the device numbers in the order written, duplicates dropped. */
  static int[] parseDevices(String text) {
    List<Integer> devices = new ArrayList<>();
    for (String part : text.split(",")) {
      String item = part.trim();
      if (item.isEmpty()) {
        continue;
      }
      int dash = item.indexOf('-', 1);
      int first = Integer.parseInt(dash > 0 ? item.substring(0, dash).trim() : item);
      int last = dash > 0 ? Integer.parseInt(item.substring(dash + 1).trim()) : first;
      if (first < 0 || last < first) {
        throw new IllegalArgumentException("bad device range: " + item);
      }
      for (int d = first; d <= last; d++) {
        if (!devices.contains(d)) {
          devices.add(d);
        }
      }
    }
    if (devices.isEmpty()) {
      throw new IllegalArgumentException("no device in: " + text);
    }
    int[] out = new int[devices.size()];
    for (int i = 0; i < out.length; i++) {
      out[i] = devices.get(i);
    }
    return out;
  }

  /**
   * For the deployment's segment-drop path (next to IndexSegment.destroy(), INTEGRATION.md): gives the segment's HBM back at once instead
   * of when the garbage collector finds the IndexSegment unreachable.
   */
  public void releaseSegment(IndexSegment indexSegment) {
    GpuSegmentCache segments = _segments;
    if (segments != null) {
      segments.release(indexSegment);
    }
  }

  @Override
  public PlanNode makeSegmentPlanNode(SegmentContext segmentContext, QueryContext queryContext) {
    PlanNode cpuPlan = super.makeSegmentPlanNode(segmentContext, queryContext);     // also applies rewriteQueryContextWithHints (:271)
    GpuSegmentCache segments = _segments;
    IndexSegment indexSegment = segmentContext.getIndexSegment();
    if (segments == null || !QueryContextUtils.isAggregationQuery(queryContext)
        || segmentContext.getQueryableDocIdsSnapshot() != null                       // upsert / dedup: FilterPlanNode.java:89-102
        || (_skipStarTreeSegments && indexSegment.getStarTrees() != null && !indexSegment.getStarTrees().isEmpty())) {
      return cpuPlan;
    }
    AggregationFunction[] functions = queryContext.getAggregationFunctions();
    GpuSegment segment = segments.get(indexSegment);
    if (segment == null || functions == null) {
      return cpuPlan;
    }
    List<GpuAggregationOperator.Lane> lanes = queryContext.hasFilteredAggregations()
        ? filteredLanes(segment, indexSegment, queryContext, functions)
        : unfilteredLane(segment, indexSegment, queryContext, functions);
    if (lanes == null) {
      return cpuPlan;
    }
    for (GpuAggregationOperator.Lane lane : lanes) {
      GpuQueryLowering.Lowered q = lane._query;
      int admitted = PinotGpuNative.queryCheck(segment.handle(), q._filterNodes, q._predInts, q._predLongs, q._setOffsets, q._setWords,
          q._aggregations, q._groupBy, q._numGroupsLimit, q._flags);
      if (admitted != PinotGpuNative.PG_OK) {
        LOGGER.debug("Segment {} keeps the CPU plan: {}", segment.getSegmentName(), PinotGpuNative.lastError());
        return cpuPlan;
      }
    }
    // makeInstancePlan below is planning a whole query: its lanes join the query's batch, in plan order
    GpuBatch batch = _planningBatch.get();
    int[] batchSlots = null;
    if (batch != null) {
      batchSlots = new int[lanes.size()];
      for (int i = 0; i < batchSlots.length; i++) {
        batchSlots[i] = batch.add(segment, lanes.get(i)._query);
      }
    }
    final int[] slots = batchSlots;
    return () -> new GpuAggregationOperator(segment, indexSegment, queryContext, functions, lanes, cpuPlan, batch, slots);
  }

  /**
   * InstancePlanMakerImplV2.makeInstancePlan (:166-193) plans every segment of the query on this thread -- through makeSegmentPlanNode above
   * -- and hands the nodes to one CombinePlanNode.  While it does, the lanes of every offloaded segment register with one GpuBatch, so
   * that the combine operator's tasks make ONE native call for the whole query (pg_execute_batch) instead of one per segment and lane.
   * Everything else -- prefetch, the combine operator chosen, the instance response -- is the reference's.
   */
  @Override
  public Plan makeInstancePlan(List<SegmentContext> segmentContexts, QueryContext queryContext, ExecutorService executorService,
      ServerMetrics serverMetrics) {
    if (_segments == null || !_batchEnabled || segmentContexts.size() < 2 || !QueryContextUtils.isAggregationQuery(queryContext)) {
      return super.makeInstancePlan(segmentContexts, queryContext, executorService, serverMetrics);
    }
    _planningBatch.set(new GpuBatch());
    try {
      return super.makeInstancePlan(segmentContexts, queryContext, executorService, serverMetrics);
    } finally {
      _planningBatch.remove();
    }
  }

  private static List<GpuAggregationOperator.Lane> unfilteredLane(GpuSegment segment, IndexSegment indexSegment, QueryContext queryContext,
      AggregationFunction[] functions) {
    GpuQueryLowering.Lowered lowered = GpuQueryLowering.lower(segment, indexSegment, queryContext, functions, queryContext.getFilter());
    if (lowered == null) {
      return null;
    }
    int[] positions = new int[functions.length];
    for (int i = 0; i < positions.length; i++) {
      positions[i] = i;
    }
    List<GpuAggregationOperator.Lane> lanes = new ArrayList<>(1);
    lanes.add(new GpuAggregationOperator.Lane(lowered, positions));
    return lanes;
  }

  /**
   * The swim lanes of FilteredAggregationOperator / FilteredGroupByOperator (AggregationFunctionUtils.buildFilteredAggregationInfos
   * :312-402): functions grouped by their FILTER clause in first-appearance order; a clause that matches every doc of this segment joins
   * the unfiltered lane (:351-358, :376-378); the unfiltered lane runs last and exists when it has functions or -- for GROUP BY without
   * the skipEmptyGroups option -- to create every group of the main filter (:388-400).
   */
  private static List<GpuAggregationOperator.Lane> filteredLanes(GpuSegment segment, IndexSegment indexSegment, QueryContext queryContext,
      AggregationFunction[] functions) {
    List<Pair<AggregationFunction, FilterContext>> pairs = queryContext.getFilteredAggregationFunctions();
    if (pairs == null || pairs.size() != functions.length) {
      return null;
    }
    FilterContext mainFilter = queryContext.getFilter();
    Map<FilterContext, List<Integer>> byClause = new LinkedHashMap<>();
    List<Integer> unfiltered = new ArrayList<>();
    for (int i = 0; i < functions.length; i++) {
      FilterContext clause = pairs.get(i).getRight();
      if (pairs.get(i).getLeft() != functions[i]) {
        return null;                          // the two views of the query disagree on the order: not a shape this class knows
      }
      if (clause == null) {
        unfiltered.add(i);
        continue;
      }
      Integer folds = GpuQueryLowering.foldsTo(segment, indexSegment, queryContext, clause);
      if (folds == null) {
        return null;
      }
      if (folds > 0) {
        unfiltered.add(i);
      } else {
        byClause.computeIfAbsent(clause, k -> new ArrayList<>()).add(i);
      }
    }
    List<GpuAggregationOperator.Lane> lanes = new ArrayList<>();
    for (Map.Entry<FilterContext, List<Integer>> entry : byClause.entrySet()) {
      FilterContext combined = mainFilter == null ? entry.getKey() : FilterContext.forAnd(List.of(mainFilter, entry.getKey()));
      if (!addLane(lanes, segment, indexSegment, queryContext, functions, entry.getValue(), combined)) {
        return null;
      }
    }
    boolean groupBy = queryContext.getGroupByExpressions() != null;
    if (!unfiltered.isEmpty() || (groupBy && !QueryOptionsUtils.isFilteredAggregationsSkipEmptyGroups(queryContext.getQueryOptions()))) {
      if (!addLane(lanes, segment, indexSegment, queryContext, functions, unfiltered, mainFilter)) {
        return null;
      }
    }
    return lanes.isEmpty() ? null : lanes;
  }

  private static boolean addLane(List<GpuAggregationOperator.Lane> lanes, GpuSegment segment, IndexSegment indexSegment,
      QueryContext queryContext, AggregationFunction[] functions, List<Integer> members, FilterContext filter) {
    AggregationFunction[] laneFunctions = new AggregationFunction[members.size()];
    int[] positions = new int[members.size()];
    for (int i = 0; i < positions.length; i++) {
      positions[i] = members.get(i);
      laneFunctions[i] = functions[positions[i]];
    }
    GpuQueryLowering.Lowered lowered = GpuQueryLowering.lower(segment, indexSegment, queryContext, laneFunctions, filter);
    if (lowered == null) {
      return false;
    }
    lanes.add(new GpuAggregationOperator.Lane(lowered, positions));
    return true;
  }
}
