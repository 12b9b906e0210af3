/**
 * The segment-level operator of an offloaded aggregation or group-by query: what AggregationOperator / GroupByOperator -- and, when the
 * query has FILTER (WHERE ...) aggregations, FilteredAggregationOperator / FilteredGroupByOperator -- are on the CPU plan
 * (core/operator/query/AggregationOperator.java:44-106, GroupByOperator.java:52-170, FilteredAggregationOperator.java:47-110,
 * FilteredGroupByOperator.java:108-190), with the whole filter -> projection -> aggregation subtree of every swim lane behind one native
 * call.  It extends BaseOperator so that nextBlock() keeps the interruption check and the trace scope
 * (core/operator/BaseOperator.java:43-57), returns the reference's own results blocks (AggregationResultsBlock / GroupByResultsBlock with
 * the intermediate-result objects extractAggregationResult would produce: Long, Double, AvgPair), and reports ExecutionStatistics the way
 * the combine operator reads them (Operator.java:120-122; lanes add up like FilteredAggregationOperator.java:96-99).
 *
 * <p>A run-time failure of a native call -- device out of memory, a tier pg_query_check admits by upper bound -- never fails the query:
 * the segment is re-planned with the reference's own node (the one InstancePlanMakerImplV2 would have returned) and that operator's block
 * and statistics are reported instead.
 */
package org.apache.pinot.gpu;

import java.util.ArrayList;
import java.util.Arrays;
import java.util.Collection;
import java.util.Collections;
import java.util.List;
import org.apache.commons.lang3.tuple.Pair;
import org.apache.pinot.common.request.context.ExpressionContext;
import org.apache.pinot.common.request.context.FilterContext;
import org.apache.pinot.common.utils.DataSchema;
import org.apache.pinot.core.common.Operator;
import org.apache.pinot.core.data.table.IntermediateRecord;
import org.apache.pinot.core.data.table.TableResizer;
import org.apache.pinot.core.operator.BaseOperator;
import org.apache.pinot.core.operator.ExecutionStatistics;
import org.apache.pinot.core.operator.blocks.results.AggregationResultsBlock;
import org.apache.pinot.core.operator.blocks.results.BaseResultsBlock;
import org.apache.pinot.core.operator.blocks.results.GroupByResultsBlock;
import org.apache.pinot.core.plan.PlanNode;
import org.apache.pinot.core.query.aggregation.function.AggregationFunction;
import org.apache.pinot.core.query.aggregation.function.AggregationFunctionUtils;
import org.apache.pinot.core.query.aggregation.groupby.AggregationGroupByResult;
import org.apache.pinot.core.query.aggregation.groupby.DoubleGroupByResultHolder;
import org.apache.pinot.core.query.aggregation.groupby.GroupByResultHolder;
import org.apache.pinot.core.query.aggregation.groupby.ObjectGroupByResultHolder;
import org.apache.pinot.core.query.request.context.QueryContext;
import org.apache.pinot.core.util.GroupByUtils;
import org.apache.pinot.segment.local.customobject.AvgPair;
import org.apache.pinot.segment.spi.AggregationFunctionType;
import org.apache.pinot.segment.spi.IndexSegment;
import org.apache.pinot.segment.spi.index.reader.Dictionary;
import org.apache.pinot.spi.data.FieldSpec;
import org.slf4j.Logger;
import org.slf4j.LoggerFactory;


final class GpuAggregationOperator extends BaseOperator<BaseResultsBlock> {
  private static final Logger LOGGER = LoggerFactory.getLogger(GpuAggregationOperator.class);

  /** One native call: a lowered query and, for each of its aggregations, the position of the function in the query's function array. */
  static final class Lane {
    final GpuQueryLowering.Lowered _query;
    final int[] _positions;       // empty: the lane only creates groups (FilteredGroupByOperator's main-filter lane without functions)

    Lane(GpuQueryLowering.Lowered query, int[] positions) {
      _query = query;
      _positions = positions;
    }
  }

  /** What one lane brought back (the slots of PinotGpuNative.execute's Object[]). */
  private static final class LaneResult {
    long[] _header;
    int[] _groupIds;
    int[] _groupKeys;              // dictId tuples, row-major: what identifies a group whichever holder the key space calls for
    long[] _counts;
    double[] _sums;
    double[] _mins;
    double[] _maxs;
    int _width;                   // aggregations per row in this lane's arrays
  }

  private final GpuSegment _segment;
  private final IndexSegment _indexSegment;
  private final QueryContext _queryContext;
  private final AggregationFunction[] _functions;
  private final List<Lane> _lanes;
  private final PlanNode _cpuPlan;
  private final GpuBatch _batch;                        // the native calls of the whole query made as one (GpuBatch), or null
  private final int[] _batchSlots;                      // this operator's lanes in the batch, lane order
  private final long[] _statistics = new long[4];       // numDocsScanned, entriesInFilter, entriesPostFilter, totalDocs
  private Operator<?> _cpuOperator;                     // set when a native call failed and the segment ran on the CPU plan

  GpuAggregationOperator(GpuSegment segment, IndexSegment indexSegment, QueryContext queryContext, AggregationFunction[] functions,
      List<Lane> lanes, PlanNode cpuPlan, GpuBatch batch, int[] batchSlots) {
    _segment = segment;
    _indexSegment = indexSegment;
    _queryContext = queryContext;
    _functions = functions;
    _lanes = lanes;
    _cpuPlan = cpuPlan;
    _batch = batch;
    _batchSlots = batchSlots;
  }

  @Override
  protected BaseResultsBlock getNextBlock() {
    List<LaneResult> results = new ArrayList<>(_lanes.size());
    try {
      for (int i = 0; i < _lanes.size(); i++) {
        results.add(execute(i));
      }
    } catch (RuntimeException e) {
      // UnsupportedOperationException (PG_ERR_UNSUPPORTED at run time) and RuntimeException (pg_last_error: device / out of memory) alike
      LOGGER.warn("Segment {} falls back to the CPU plan: {}", _segment.getSegmentName(), e.toString());
      Arrays.fill(_statistics, 0);
      _cpuOperator = _cpuPlan.run();
      return (BaseResultsBlock) _cpuOperator.nextBlock();
    }
    for (LaneResult result : results) {
      // FilteredAggregationOperator.java:96-99 / FilteredGroupByOperator.java:151-153: lanes add up; one lane is the plain operator
      _statistics[0] += result._header[PinotGpuNative.PGM_H_NUM_DOCS_SCANNED];
      _statistics[1] += result._header[PinotGpuNative.PGM_H_ENTRIES_IN_FILTER];
      _statistics[2] += result._header[PinotGpuNative.PGM_H_ENTRIES_POST_FILTER];
      _statistics[3] = result._header[PinotGpuNative.PGM_H_TOTAL_DOCS];
    }
    if (_queryContext.getGroupByExpressions() == null) {
      return aggregationBlock(results);
    }
    // FILTER (WHERE ...) lanes under GROUP BY: the reference shares ONE GroupKeyGenerator across the lanes (FilteredGroupByOperator.java
    // :121-143), so numGroupsLimit bounds the groups of all lanes TOGETHER and ids go to keys by first appearance across lanes.  Here every
    // lane applied the limit on its own: whenever the limit can have shaped the result -- a lane reached it, or the union of the lanes'
    // groups exceeds it -- the segment runs on the CPU plan instead of returning other (or more) groups than the reference.
    if (results.size() > 1) {
      int numKeyColumns = _queryContext.getGroupByExpressions().size();
      int limit = 0;
      boolean reached = false;
      for (int l = 0; l < results.size(); l++) {
        reached |= results.get(l)._header[PinotGpuNative.PGM_H_NUM_GROUPS_LIMIT_REACHED] != 0;
        limit = Math.max(limit, _lanes.get(l)._query._numGroupsLimit);
      }
      if (!reached && limit > 0 && numKeyColumns > 0) {
        reached = unionOf(results, numKeyColumns).length / numKeyColumns >= limit;
      }
      if (reached) {
        LOGGER.debug("Segment {}: numGroupsLimit binds across FILTER lanes; CPU plan", _segment.getSegmentName());
        Arrays.fill(_statistics, 0);
        _cpuOperator = _cpuPlan.run();
        return (BaseResultsBlock) _cpuOperator.nextBlock();
      }
    }
    // the group-by block asks the device how each key column's digits map to values (groupKeyInfo): the segment must still be resident
    if (!_segment.tryPin()) {
      LOGGER.warn("Segment {} left the device while its group-by result was assembled; it runs on the CPU plan", _segment.getSegmentName());
      Arrays.fill(_statistics, 0);
      _cpuOperator = _cpuPlan.run();
      return (BaseResultsBlock) _cpuOperator.nextBlock();
    }
    try {
      return groupByBlock(results);
    } finally {
      _segment.unpin();
    }
  }

  /** One pg_execute; the segment stays pinned for its duration (GpuSegmentCache may otherwise evict it under the HBM budget). */
  private Object[] executeAlone(GpuQueryLowering.Lowered q) {
    if (!_segment.tryPin()) {
      throw new UnsupportedOperationException("segment " + _segment.getSegmentName() + " is no longer resident on the device");
    }
    try {
      Object[] raw = PinotGpuNative.execute(_segment.handle(), q._filterNodes, q._predInts, q._predLongs, q._setOffsets, q._setWords,
          q._aggregations, q._groupBy, q._numGroupsLimit, q._flags);
      _segment.refreshDeviceBytes();
      return raw;
    } finally {
      _segment.unpin();
    }
  }

  private LaneResult execute(int laneIndex) {
    GpuQueryLowering.Lowered q = _lanes.get(laneIndex)._query;
    // in a batch: the first lane of the first segment a combine task reaches makes the native call for every lane of every segment
    Object[] raw = _batch != null ? _batch.take(_batchSlots[laneIndex]) : executeAlone(q);
    if (raw == null || raw.length != PinotGpuNative.PGM_RESULT_ARRAYS) {
      throw new IllegalStateException("native result does not match jni/pg_marshal.h");
    }
    LaneResult result = new LaneResult();
    result._header = (long[]) raw[PinotGpuNative.PGM_R_HEADER];
    result._groupIds = (int[]) raw[PinotGpuNative.PGM_R_GROUP_IDS];
    result._groupKeys = (int[]) raw[PinotGpuNative.PGM_R_GROUP_KEYS];
    result._counts = (long[]) raw[PinotGpuNative.PGM_R_COUNTS];
    result._sums = (double[]) raw[PinotGpuNative.PGM_R_SUMS];
    result._mins = (double[]) raw[PinotGpuNative.PGM_R_MINS];
    result._maxs = (double[]) raw[PinotGpuNative.PGM_R_MAXS];
    result._width = (int) result._header[PinotGpuNative.PGM_H_NUM_AGGREGATIONS];
    if (result._header.length != PinotGpuNative.PGM_HEADER_LEN) {
      throw new IllegalStateException("native result header does not match jni/pg_marshal.h");
    }
    return result;
  }

  private AggregationResultsBlock aggregationBlock(List<LaneResult> results) {
    boolean nullHandling = _queryContext.isNullHandlingEnabled();
    Object[] out = new Object[_functions.length];
    for (int l = 0; l < _lanes.size(); l++) {
      int[] positions = _lanes.get(l)._positions;
      LaneResult result = results.get(l);
      for (int i = 0; i < positions.length; i++) {
        out[positions[i]] = intermediate(_functions[positions[i]].getType(), result, i, nullHandling);
      }
    }
    return new AggregationResultsBlock(_functions, Arrays.asList(out), _queryContext);
  }

  /** The object extractAggregationResult of the reference's function returns (null for an empty SUM / MIN / MAX / AVG under null handling). */
  private static Object intermediate(AggregationFunctionType type, LaneResult r, int at, boolean nullHandling) {
    switch (type) {
      case COUNT:
        return r._counts[at];
      case SUM:
        return nullHandling && r._counts[at] == 0 ? null : (Object) r._sums[at];
      case MIN:
        return nullHandling && r._counts[at] == 0 ? null : (Object) r._mins[at];
      case MAX:
        return nullHandling && r._counts[at] == 0 ? null : (Object) r._maxs[at];
      case AVG:
        return nullHandling && r._counts[at] == 0 ? null : new AvgPair(r._sums[at], r._counts[at]);
      default:
        throw new IllegalStateException("not offloadable: " + type);
    }
  }

  /**
   * Group-by: the lanes share one key space (a key's dictIds are the same in every lane), so the block's groups are the union of the
   * lanes' groups in ascending raw-key order -- what sharing one GroupKeyGenerator across lanes gives the reference
   * (FilteredGroupByOperator.java:121-143) -- and a function whose lane never saw a group keeps its holder's default there.
   */
  private GroupByResultsBlock groupByBlock(List<LaneResult> results) {
    boolean nullHandling = _queryContext.isNullHandlingEnabled();
    int numFunctions = _functions.length;
    int numKeyColumns = _queryContext.getGroupByExpressions().size();
    // the block's groups: one lane's rows as they are, or the union of the lanes' keys (dictId tuples, compared like the raw key:
    // the last column is the most significant digit)
    int[] groupKeys = results.size() == 1 ? results.get(0)._groupKeys : unionOf(results, numKeyColumns);
    int numGroups = numKeyColumns == 0 ? 0 : groupKeys.length / numKeyColumns;
    int capacity = Math.max(numGroups, 1);
    GroupByResultHolder[] holders = new GroupByResultHolder[numFunctions];
    for (int l = 0; l < _lanes.size(); l++) {
      int[] positions = _lanes.get(l)._positions;
      LaneResult r = results.get(l);
      int laneGroups = r._groupIds.length;
      int[] rowOf = new int[laneGroups];                          // lane row -> row of the block
      for (int g = 0; g < laneGroups; g++) {
        rowOf[g] = results.size() == 1 ? g : find(groupKeys, numGroups, numKeyColumns, r._groupKeys, g);
      }
      for (int i = 0; i < positions.length; i++) {
        AggregationFunctionType type = _functions[positions[i]].getType();
        boolean object = type == AggregationFunctionType.AVG || (nullHandling && type != AggregationFunctionType.COUNT);
        if (object) {
          // AVG always, and SUM / MIN / MAX under null handling (NullableSingleInputAggregationFunction), keep an ObjectGroupByResultHolder
          // that stays null until a value arrives
          ObjectGroupByResultHolder holder = new ObjectGroupByResultHolder(capacity, capacity);
          for (int g = 0; g < rowOf.length; g++) {
            int at = g * r._width + i;
            if (nullHandling && r._counts[at] == 0) {
              continue;
            }
            Object value = type == AggregationFunctionType.AVG ? new AvgPair(r._sums[at], r._counts[at])
                : (Object) Double.valueOf(type == AggregationFunctionType.SUM ? r._sums[at] : (type == AggregationFunctionType.MIN ? r._mins[at] : r._maxs[at]));
            holder.setValueForKey(rowOf[g], value);
          }
          holders[positions[i]] = holder;
        } else {
          // COUNT / SUM / MIN / MAX read getDoubleResult (CountAggregationFunction.extractGroupByResult casts it back to long); the default
          // is what the function's own holder starts from: 0 for COUNT / SUM, +inf for MIN, -inf for MAX
          double initial = type == AggregationFunctionType.MIN ? Double.POSITIVE_INFINITY : (type == AggregationFunctionType.MAX ? Double.NEGATIVE_INFINITY : 0.0);
          DoubleGroupByResultHolder holder = new DoubleGroupByResultHolder(capacity, capacity, initial);
          for (int g = 0; g < rowOf.length; g++) {
            int at = g * r._width + i;
            double value = type == AggregationFunctionType.COUNT ? r._counts[at]
                : (type == AggregationFunctionType.SUM ? r._sums[at] : (type == AggregationFunctionType.MIN ? r._mins[at] : r._maxs[at]));
            holder.setValueForKey(rowOf[g], value);
          }
          holders[positions[i]] = holder;
        }
      }
    }
    List<ExpressionContext> groupBy = _queryContext.getGroupByExpressions();
    Dictionary[] dictionaries = new Dictionary[groupBy.size()];      // null: a raw key column (the no-dictionary key generators' case)
    long[][] keyInfo = new long[groupBy.size()][];
    boolean[] longKeys = new boolean[groupBy.size()];
    long[][] rankValues = new long[groupBy.size()][];
    FieldSpec.DataType[] storedTypes = new FieldSpec.DataType[groupBy.size()];
    String[] columnNames = new String[groupBy.size() + numFunctions];
    DataSchema.ColumnDataType[] columnTypes = new DataSchema.ColumnDataType[groupBy.size() + numFunctions];
    for (int i = 0; i < groupBy.size(); i++) {
      String column = groupBy.get(i).getIdentifier();
      dictionaries[i] = _indexSegment.getDataSource(column).getDictionary();
      keyInfo[i] = PinotGpuNative.groupKeyInfo(_segment.handle(), _segment.columnIndex(column));
      storedTypes[i] = _indexSegment.getDataSource(column).getDataSourceMetadata().getDataType().getStoredType();
      longKeys[i] = storedTypes[i] == FieldSpec.DataType.LONG;
      if (keyInfo[i][1] == 2) {
        rankValues[i] = PinotGpuNative.groupKeyValues(_segment.handle(), _segment.columnIndex(column));      // FLOAT / DOUBLE / wide INT / LONG raw key
      }
      columnNames[i] = groupBy.get(i).toString();
      columnTypes[i] = DataSchema.ColumnDataType.fromDataTypeSV(_indexSegment.getDataSource(column).getDataSourceMetadata().getDataType());
    }
    List<Pair<AggregationFunction, FilterContext>> filtered = _queryContext.getFilteredAggregationFunctions();
    for (int i = 0; i < numFunctions; i++) {
      // FilteredGroupByOperator.java:98-104 names a filtered function after its clause; unfiltered ones keep getResultColumnName
      FilterContext clause = filtered != null && _queryContext.hasFilteredAggregations() ? filtered.get(i).getRight() : null;
      columnNames[groupBy.size() + i] = AggregationFunctionUtils.getResultColumnName(_functions[i], clause);
      columnTypes[groupBy.size() + i] = _functions[i].getIntermediateResultColumnType();
    }
    long upperBound = 0;
    boolean limitReached = false;
    for (LaneResult r : results) {
      upperBound = Math.max(upperBound, r._header[PinotGpuNative.PGM_H_GROUP_ID_UPPER_BOUND]);
      limitReached |= r._header[PinotGpuNative.PGM_H_NUM_GROUPS_LIMIT_REACHED] != 0;
    }
    DataSchema dataSchema = new DataSchema(columnNames, columnTypes);
    GpuGroupKeyGenerator keys = new GpuGroupKeyGenerator(numGroups, groupKeys, dictionaries, keyInfo, longKeys, rankValues, storedTypes, (int) upperBound);
    // In-segment trim, exactly GroupByOperator.java:119-135: ORDER BY + minSegmentGroupTrimSize > 0 + more groups than the trim size
    int minGroupTrimSize = _queryContext.getMinSegmentGroupTrimSize();
    if (_queryContext.getOrderByExpressions() != null && minGroupTrimSize > 0) {
      int trimSize = GroupByUtils.getTableCapacity(_queryContext.getLimit(), minGroupTrimSize);
      if (numGroups > trimSize) {
        Collection<IntermediateRecord> records = new TableResizer(dataSchema, _queryContext).trimInSegmentResults(keys, holders, trimSize);
        GroupByResultsBlock trimmed = new GroupByResultsBlock(dataSchema, records, _queryContext);
        trimmed.setNumGroupsLimitReached(limitReached);
        return trimmed;
      }
    }
    GroupByResultsBlock block = new GroupByResultsBlock(dataSchema, new AggregationGroupByResult(keys, _functions, holders), _queryContext);
    block.setNumGroupsLimitReached(limitReached);      // GroupByOperator.java:114-115
    return block;
  }

  /** The order of the native rows: ascending raw key = the dictId tuples compared from the last column (most significant digit) down. */
  private static int compareKeys(int[] a, int rowA, int[] b, int rowB, int columns) {
    for (int c = columns - 1; c >= 0; c--) {
      int x = a[rowA * columns + c];
      int y = b[rowB * columns + c];
      if (x != y) {
        return x < y ? -1 : 1;
      }
    }
    return 0;
  }

  /** The sorted union of the lanes' (sorted) key tuples. */
  private static int[] unionOf(List<LaneResult> results, int columns) {
    int[] merged = new int[0];
    for (LaneResult r : results) {
      int rowsA = columns == 0 ? 0 : merged.length / columns;
      int rowsB = r._groupIds.length;
      int[] out = new int[(rowsA + rowsB) * columns];
      int i = 0;
      int j = 0;
      int k = 0;
      while (i < rowsA || j < rowsB) {
        int cmp = i >= rowsA ? 1 : (j >= rowsB ? -1 : compareKeys(merged, i, r._groupKeys, j, columns));
        if (cmp <= 0) {
          System.arraycopy(merged, i * columns, out, k * columns, columns);
          i++;
          if (cmp == 0) {
            j++;
          }
        } else {
          System.arraycopy(r._groupKeys, j * columns, out, k * columns, columns);
          j++;
        }
        k++;
      }
      merged = Arrays.copyOf(out, k * columns);
    }
    return merged;
  }

  /** Row of the key tuple keys[row] in the sorted union. */
  private static int find(int[] union, int unionRows, int columns, int[] keys, int row) {
    int lo = 0;
    int hi = unionRows - 1;
    while (lo <= hi) {
      int mid = (lo + hi) >>> 1;
      int cmp = compareKeys(union, mid, keys, row, columns);
      if (cmp == 0) {
        return mid;
      }
      if (cmp < 0) {
        lo = mid + 1;
      } else {
        hi = mid - 1;
      }
    }
    throw new IllegalStateException("a lane's group is missing from the union of the lanes");
  }

  @Override
  public ExecutionStatistics getExecutionStatistics() {
    if (_cpuOperator != null) {
      return _cpuOperator.getExecutionStatistics();
    }
    return new ExecutionStatistics(_statistics[0], _statistics[1], _statistics[2], _statistics[3]);
  }

  @Override
  public IndexSegment getIndexSegment() {
    return _indexSegment;
  }

  @Override
  @SuppressWarnings("rawtypes")
  public List<Operator> getChildOperators() {
    return _cpuOperator != null ? Collections.singletonList(_cpuOperator) : Collections.emptyList();
  }

  @Override
  public String toExplainString() {
    String name = _queryContext.getGroupByExpressions() == null ? "GPU_AGGREGATE" : "GPU_GROUP_BY";
    return _lanes.size() > 1 ? name + "_FILTERED" : name;
  }
}
