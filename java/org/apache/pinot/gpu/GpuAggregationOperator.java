/**
 * The segment-level operator of an offloaded aggregation or group-by query: what AggregationOperator / GroupByOperator are on the CPU
 * plan (core/operator/query/AggregationOperator.java:44-106, GroupByOperator.java:52-170), with the whole
 * filter -> projection -> aggregation subtree behind one native call.  It extends BaseOperator so that nextBlock() keeps the
 * interruption check and the trace scope (core/operator/BaseOperator.java:43-57), returns the reference's own results blocks
 * (AggregationResultsBlock / GroupByResultsBlock with the intermediate-result objects extractAggregationResult would produce: Long,
 * Double, AvgPair), and reports ExecutionStatistics the way the combine operator reads them (Operator.java:120-122).
 */
package org.apache.pinot.gpu;

import java.util.ArrayList;
import java.util.Collections;
import java.util.List;
import org.apache.pinot.common.request.context.ExpressionContext;
import org.apache.pinot.common.utils.DataSchema;
import org.apache.pinot.core.common.Operator;
import org.apache.pinot.core.operator.BaseOperator;
import org.apache.pinot.core.operator.ExecutionStatistics;
import org.apache.pinot.core.operator.blocks.results.AggregationResultsBlock;
import org.apache.pinot.core.operator.blocks.results.BaseResultsBlock;
import org.apache.pinot.core.operator.blocks.results.GroupByResultsBlock;
import org.apache.pinot.core.query.aggregation.function.AggregationFunction;
import org.apache.pinot.core.query.aggregation.groupby.AggregationGroupByResult;
import org.apache.pinot.core.query.aggregation.groupby.DoubleGroupByResultHolder;
import org.apache.pinot.core.query.aggregation.groupby.GroupByResultHolder;
import org.apache.pinot.core.query.aggregation.groupby.ObjectGroupByResultHolder;
import org.apache.pinot.core.query.request.context.QueryContext;
import org.apache.pinot.segment.local.customobject.AvgPair;
import org.apache.pinot.segment.spi.AggregationFunctionType;
import org.apache.pinot.segment.spi.IndexSegment;
import org.apache.pinot.segment.spi.index.reader.Dictionary;
import org.apache.pinot.segment.spi.index.reader.NullValueVectorReader;


final class GpuAggregationOperator extends BaseOperator<BaseResultsBlock> {
  // header indexes of the native result (PGM_H_* in jni/pg_marshal.h)
  private static final int H_NUM_DOCS_SCANNED = 0;
  private static final int H_ENTRIES_IN_FILTER = 1;
  private static final int H_ENTRIES_POST_FILTER = 2;
  private static final int H_TOTAL_DOCS = 3;
  private static final int H_GROUP_ID_UPPER_BOUND = 7;
  private static final int H_NUM_GROUPS_LIMIT_REACHED = 8;

  private final GpuSegment _segment;
  private final QueryContext _queryContext;
  private final AggregationFunction[] _functions;
  private final GpuQueryLowering.Lowered _query;
  private long[] _header;

  GpuAggregationOperator(GpuSegment segment, QueryContext queryContext, AggregationFunction[] functions, GpuQueryLowering.Lowered query) {
    _segment = segment;
    _queryContext = queryContext;
    _functions = functions;
    _query = query;
  }

  @Override
  protected BaseResultsBlock getNextBlock() {
    Object[] result = PinotGpuNative.execute(_segment.handle(), _query._filterNodes, _query._predInts, _query._predLongs, _query._setOffsets,
        _query._setWords, _query._aggregations, _query._groupBy, _query._numGroupsLimit, _query._flags);
    _header = (long[]) result[0];
    int[] groupIds = (int[]) result[1];
    long[] counts = (long[]) result[2];
    double[] sums = (double[]) result[3];
    double[] mins = (double[]) result[6];
    double[] maxs = (double[]) result[7];
    int numFunctions = _functions.length;
    boolean nullHandling = _queryContext.isNullHandlingEnabled();
    if (_query._groupBy.length == 0) {
      List<Object> results = new ArrayList<>(numFunctions);
      for (int i = 0; i < numFunctions; i++) {
        results.add(intermediate(i, 0, numFunctions, counts, sums, mins, maxs, nullHandling));
      }
      return new AggregationResultsBlock(_functions, results, _queryContext);
    }
    // group-by: holders indexed by the row of the native result, keys mapped back to dictionary values
    int numGroups = groupIds.length;
    GroupByResultHolder[] holders = new GroupByResultHolder[numFunctions];
    for (int i = 0; i < numFunctions; i++) {
      switch (_functions[i].getType()) {
        case AVG: {
          ObjectGroupByResultHolder holder = new ObjectGroupByResultHolder(Math.max(numGroups, 1), Math.max(numGroups, 1));
          for (int g = 0; g < numGroups; g++) {
            if (!nullHandling || counts[g * numFunctions + i] != 0) {
              holder.setValueForKey(g, new AvgPair(sums[g * numFunctions + i], counts[g * numFunctions + i]));
            }
          }
          holders[i] = holder;
          break;
        }
        case SUM:
        case MIN:
        case MAX:
          if (nullHandling) {
            // NullableSingleInputAggregationFunction keeps these in an ObjectGroupByResultHolder that stays null until a value arrives
            ObjectGroupByResultHolder nullable = new ObjectGroupByResultHolder(Math.max(numGroups, 1), Math.max(numGroups, 1));
            for (int g = 0; g < numGroups; g++) {
              int at = g * numFunctions + i;
              if (counts[at] != 0) {
                nullable.setValueForKey(g, (Object) Double.valueOf(_functions[i].getType() == AggregationFunctionType.SUM ? sums[at]
                    : (_functions[i].getType() == AggregationFunctionType.MIN ? mins[at] : maxs[at])));
              }
            }
            holders[i] = nullable;
            break;
          }
          // fall through
        default: {
          // COUNT / SUM / MIN / MAX read getDoubleResult (CountAggregationFunction.extractGroupByResult casts it back to long)
          DoubleGroupByResultHolder holder = new DoubleGroupByResultHolder(Math.max(numGroups, 1), Math.max(numGroups, 1), 0.0);
          for (int g = 0; g < numGroups; g++) {
            int at = g * numFunctions + i;
            double value;
            switch (_functions[i].getType()) {
              case COUNT:
                value = counts[at];
                break;
              case SUM:
                value = sums[at];
                break;
              case MIN:
                value = mins[at];
                break;
              default:
                value = maxs[at];
                break;
            }
            holder.setValueForKey(g, value);
          }
          holders[i] = holder;
          break;
        }
      }
    }
    List<ExpressionContext> groupBy = _queryContext.getGroupByExpressions();
    Dictionary[] dictionaries = new Dictionary[groupBy.size()];
    boolean[] nullableKeys = new boolean[groupBy.size()];
    String[] columnNames = new String[groupBy.size() + numFunctions];
    DataSchema.ColumnDataType[] columnTypes = new DataSchema.ColumnDataType[groupBy.size() + numFunctions];
    IndexSegment indexSegment = _segment.getIndexSegment();
    for (int i = 0; i < groupBy.size(); i++) {
      String column = groupBy.get(i).getIdentifier();
      dictionaries[i] = indexSegment.getDataSource(column).getDictionary();
      NullValueVectorReader nullVector = indexSegment.getDataSource(column).getNullValueVector();
      nullableKeys[i] = nullHandling && nullVector != null && nullVector.getNullBitmap() != null && !nullVector.getNullBitmap().isEmpty();
      columnNames[i] = groupBy.get(i).toString();
      columnTypes[i] = DataSchema.ColumnDataType.fromDataTypeSV(indexSegment.getDataSource(column).getDataSourceMetadata().getDataType());
    }
    for (int i = 0; i < numFunctions; i++) {
      columnNames[groupBy.size() + i] = _functions[i].getResultColumnName();
      columnTypes[groupBy.size() + i] = _functions[i].getIntermediateResultColumnType();
    }
    GpuGroupKeyGenerator keys = new GpuGroupKeyGenerator(groupIds, dictionaries, nullableKeys, (int) _header[H_GROUP_ID_UPPER_BOUND]);
    GroupByResultsBlock block = new GroupByResultsBlock(new DataSchema(columnNames, columnTypes), new AggregationGroupByResult(keys, _functions, holders), _queryContext);
    block.setNumGroupsLimitReached(_header[H_NUM_GROUPS_LIMIT_REACHED] != 0);      // GroupByOperator.java:114-115
    return block;
  }

  /** The object extractAggregationResult of the reference's function returns (null for an empty SUM / MIN / MAX / AVG under null handling). */
  private Object intermediate(int function, int row, int numFunctions, long[] counts, double[] sums, double[] mins, double[] maxs, boolean nullHandling) {
    int at = row * numFunctions + function;
    switch (_functions[function].getType()) {
      case COUNT:
        return counts[at];
      case SUM:
        return nullHandling && counts[at] == 0 ? null : (Object) sums[at];
      case MIN:
        return nullHandling && counts[at] == 0 ? null : (Object) mins[at];
      case MAX:
        return nullHandling && counts[at] == 0 ? null : (Object) maxs[at];
      case AVG:
        return nullHandling && counts[at] == 0 ? null : new AvgPair(sums[at], counts[at]);
      default:
        throw new IllegalStateException("not offloadable: " + _functions[function].getType());
    }
  }

  @Override
  public ExecutionStatistics getExecutionStatistics() {
    long[] h = _header != null ? _header : new long[11];
    return new ExecutionStatistics(h[H_NUM_DOCS_SCANNED], h[H_ENTRIES_IN_FILTER], h[H_ENTRIES_POST_FILTER], h[H_TOTAL_DOCS]);
  }

  @Override
  public IndexSegment getIndexSegment() {
    return _segment.getIndexSegment();
  }

  @Override
  @SuppressWarnings("rawtypes")
  public List<Operator> getChildOperators() {
    return Collections.emptyList();
  }

  @Override
  public String toExplainString() {
    return _query._groupBy.length == 0 ? "GPU_AGGREGATE" : "GPU_GROUP_BY";
  }
}
