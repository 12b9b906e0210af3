/**
 * Lowers a QueryContext over one segment into the flat arrays of jni/pg_marshal.h -- or says the query keeps the CPU plan.
 *
 * <p>The filter is lowered into the PHYSICAL operator tree the reference itself would build: FilterPlanNode.constructPhysicalOperator
 * calls FilterOperatorUtils.get{Leaf,And,Or,Not}FilterOperator (core/operator/filter/FilterOperatorUtils.java:68-193), which pick the
 * leaf operator from the column's indexes, drop MatchAll / Empty children and re-order the children of an AND by priority (:196-245).
 * The order travels with the query: AndDocIdSet.iterator() applies scan-based children in list order, and the engine's
 * numEntriesScannedInFilter follows it.  The predicates are evaluated by the reference's own PredicateEvaluators: this class only reads
 * what they decided (always true / false, dictId range, matching dictIds), it never compares values itself -- except for raw
 * (no-dictionary) columns, whose evaluators are private classes: there the inclusive bounds are derived from the RangePredicate /
 * EqPredicate strings with the same java.lang parsers the evaluators use.
 *
 * <p>C++ twin (tested here, where no JDK exists): pinot_amd/csrc/host/plan_maker.cpp, lowerFilter / physAnd / physOr / physNot.
 */
package org.apache.pinot.gpu;

import java.util.ArrayList;
import java.util.Arrays;
import java.util.List;
import java.util.Map;
import javax.annotation.Nullable;
import org.apache.pinot.common.request.context.ExpressionContext;
import org.apache.pinot.common.request.context.FilterContext;
import org.apache.pinot.common.request.context.predicate.EqPredicate;
import org.apache.pinot.common.request.context.predicate.NotEqPredicate;
import org.apache.pinot.common.request.context.predicate.Predicate;
import org.apache.pinot.common.request.context.predicate.RangePredicate;
import org.apache.pinot.core.operator.filter.predicate.PredicateEvaluator;
import org.apache.pinot.core.operator.filter.predicate.PredicateEvaluatorProvider;
import org.apache.pinot.core.operator.filter.predicate.RangePredicateEvaluatorFactory.SortedDictionaryBasedRangePredicateEvaluator;
import org.apache.pinot.core.query.aggregation.function.AggregationFunction;
import org.apache.pinot.core.query.request.context.QueryContext;
import org.apache.pinot.segment.spi.IndexSegment;
import org.apache.pinot.segment.spi.datasource.DataSource;
import org.apache.pinot.segment.spi.index.reader.NullValueVectorReader;
import org.apache.pinot.segment.spi.index.reader.SortedIndexReader;
import org.apache.pinot.spi.data.FieldSpec.DataType;
import org.apache.pinot.spi.utils.Pairs;
import org.roaringbitmap.buffer.ImmutableRoaringBitmap;


final class GpuQueryLowering {
  // pg_predicate_kind / pg_leaf_eval / pg_filter_op / pg_agg_function: the numbers live in PinotGpuNative, next to the header they mirror
  static final int PRED_MATCH_ALL = PinotGpuNative.PG_PRED_MATCH_ALL;
  static final int PRED_MATCH_NONE = PinotGpuNative.PG_PRED_MATCH_NONE;
  static final int PRED_DICT_RANGE = PinotGpuNative.PG_PRED_DICT_RANGE;
  static final int PRED_DICT_SET = PinotGpuNative.PG_PRED_DICT_SET;
  static final int PRED_RAW_RANGE = PinotGpuNative.PG_PRED_RAW_RANGE;
  static final int PRED_DOC_RANGE = PinotGpuNative.PG_PRED_DOC_RANGE;
  static final int PRED_IS_NULL = PinotGpuNative.PG_PRED_IS_NULL;
  static final int EVAL_SCAN = PinotGpuNative.PG_EVAL_SCAN;
  static final int EVAL_INVERTED = PinotGpuNative.PG_EVAL_INVERTED;
  static final int OP_LEAF = PinotGpuNative.PG_FILTER_LEAF;
  static final int OP_AND = PinotGpuNative.PG_FILTER_AND;
  static final int OP_OR = PinotGpuNative.PG_FILTER_OR;
  static final int OP_NOT = PinotGpuNative.PG_FILTER_NOT;
  static final int AGG_COUNT = PinotGpuNative.PG_AGG_COUNT;
  static final int AGG_SUM = PinotGpuNative.PG_AGG_SUM;
  static final int AGG_MIN = PinotGpuNative.PG_AGG_MIN;
  static final int AGG_MAX = PinotGpuNative.PG_AGG_MAX;
  static final int AGG_AVG = PinotGpuNative.PG_AGG_AVG;
  private static final int NODE_INTS = PinotGpuNative.PGM_FILTER_NODE_INTS;
  private static final int PRED_INTS = PinotGpuNative.PGM_PRED_INTS;
  private static final int PRED_LONGS = PinotGpuNative.PGM_PRED_LONGS;
  private static final int AGG_INTS = PinotGpuNative.PGM_AGG_INTS;

  // PrioritizedFilterOperator.java:32-39
  private static final int SORTED_PRIORITY = 0;
  private static final int BITMAP_PRIORITY = 100;
  private static final int AND_PRIORITY = 300;
  private static final int OR_PRIORITY = 400;
  private static final int SCAN_PRIORITY = 500;
  private static final int UNKNOWN_PRIORITY = 10000;

  /** The arrays PinotGpuNative.queryCheck / execute take. */
  static final class Lowered {
    int[] _filterNodes;
    int[] _predInts;
    long[] _predLongs;
    int[] _setOffsets;
    int[] _setWords;
    int[] _aggregations;
    int[] _groupBy;
    int _numGroupsLimit;
    int _flags;
  }

  /** Thrown inside the lowering when a construct has no device form; the plan maker keeps the CPU plan. */
  static final class NotOffloadable extends RuntimeException {
    NotOffloadable(String why) {
      super(why, null, false, false);
    }
  }

  private enum Kind { MATCH_ALL, EMPTY, LEAF, AND, OR, NOT }

  private static final class Node {
    Kind _kind = Kind.LEAF;
    int _predicate = -1;
    int _priority = UNKNOWN_PRIORITY;
    List<Node> _children = new ArrayList<>();
  }

  private final GpuSegment _segment;
  private final IndexSegment _indexSegment;
  private final QueryContext _queryContext;
  private final List<int[]> _predInts = new ArrayList<>();
  private final List<long[]> _predLongs = new ArrayList<>();
  private final List<int[]> _predSets = new ArrayList<>();

  // gpu.exact.filter.stats of the plan maker's configuration (GpuPlanMaker.init); a query turns it off for itself with the option gpuExactFilterStats=false
  private static volatile boolean _exactFilterStats = true;

  static void setExactFilterStats(boolean exact) {
    _exactFilterStats = exact;
  }

  /** PG_QUERY_STATS_UPPER_BOUND_OK for this query: the server's setting, or the query's own option. */
  static boolean statsUpperBoundOk(QueryContext queryContext) {
    if (!_exactFilterStats) {
      return true;
    }
    Map<String, String> options = queryContext.getQueryOptions();
    String option = options == null ? null : options.get(GpuPlanMaker.EXACT_FILTER_STATS_QUERY_OPTION);
    return option != null && "false".equalsIgnoreCase(option.trim());
  }

  private GpuQueryLowering(GpuSegment segment, IndexSegment indexSegment, QueryContext queryContext) {
    _segment = segment;
    _indexSegment = indexSegment;          // handed in per query: the device copy keeps no reference to the IndexSegment (GpuSegmentCache)
    _queryContext = queryContext;
  }

  /** The lowered query, or null when it keeps the CPU plan (the reason is logged by the caller at debug level). */
  @Nullable
  static Lowered lower(GpuSegment segment, IndexSegment indexSegment, QueryContext queryContext, AggregationFunction[] functions,
      @Nullable FilterContext filter) {
    try {
      return new GpuQueryLowering(segment, indexSegment, queryContext).run(functions, filter);
    } catch (NotOffloadable e) {
      return null;
    }
  }

  /**
   * What a FILTER (WHERE ...) clause folds to on this segment: +1 matches every doc (MatchAllFilterOperator), -1 matches none
   * (EmptyFilterOperator), 0 neither.  AggregationFunctionUtils.buildFilteredAggregationInfos (:351-358, :376-378) treats the functions of
   * a match-all clause as non-filtered; the plan maker does the same with this answer.  Null when the clause has no device form.
   */
  @Nullable
  static Integer foldsTo(GpuSegment segment, IndexSegment indexSegment, QueryContext queryContext, FilterContext filter) {
    try {
      Node root = new GpuQueryLowering(segment, indexSegment, queryContext).lowerFilter(filter);
      return root._kind == Kind.MATCH_ALL ? 1 : (root._kind == Kind.EMPTY ? -1 : 0);
    } catch (NotOffloadable e) {
      return null;
    }
  }

  private Lowered run(AggregationFunction[] functions, @Nullable FilterContext filter) {
    Lowered out = new Lowered();
    // ---- aggregations: COUNT / SUM / MIN / MAX / AVG over one identifier (AggregationFunctionType) ----
    // (no functions at all: the group-by lane FilteredGroupByOperator runs over the main filter only to create its groups,
    //  AggregationFunctionUtils.java:388-400 -- the device needs something to aggregate, so it counts; the operator ignores the value)
    out._aggregations = functions.length == 0 ? new int[]{AGG_COUNT, -1} : new int[AGG_INTS * functions.length];
    for (int i = 0; i < functions.length; i++) {
      AggregationFunction function = functions[i];
      int code;
      switch (function.getType()) {
        case COUNT:
          code = AGG_COUNT;
          break;
        case SUM:
          code = AGG_SUM;
          break;
        case MIN:
          code = AGG_MIN;
          break;
        case MAX:
          code = AGG_MAX;
          break;
        case AVG:
          code = AGG_AVG;
          break;
        default:
          throw new NotOffloadable("aggregation function " + function.getType());
      }
      int column = -1;
      List<ExpressionContext> inputs = function.getInputExpressions();
      if (code == AGG_COUNT) {
        // COUNT(col) is COUNT(*) unless null handling is on (CountAggregationFunction.java:44-50)
        if (_queryContext.isNullHandlingEnabled() && !inputs.isEmpty() && inputs.get(0).getType() == ExpressionContext.Type.IDENTIFIER
            && !"*".equals(inputs.get(0).getIdentifier())) {
          column = _segment.columnIndex(inputs.get(0).getIdentifier());
        }
      } else {
        if (inputs.size() != 1 || inputs.get(0).getType() != ExpressionContext.Type.IDENTIFIER) {
          throw new NotOffloadable("aggregation over an expression");
        }
        column = _segment.columnIndex(inputs.get(0).getIdentifier());
        if (!_segment.isNumeric(column)) {
          throw new NotOffloadable("aggregation of a non-numeric column");   // the reference throws BadQueryRequestException itself
        }
      }
      out._aggregations[AGG_INTS * i] = code;
      out._aggregations[AGG_INTS * i + 1] = column;
    }
    // ---- group-by keys: dictionary-encoded identifiers ----
    List<ExpressionContext> groupBy = _queryContext.getGroupByExpressions();
    out._groupBy = new int[groupBy == null ? 0 : groupBy.size()];
    for (int i = 0; i < out._groupBy.length; i++) {
      ExpressionContext expression = groupBy.get(i);
      if (expression.getType() != ExpressionContext.Type.IDENTIFIER) {
        throw new NotOffloadable("group-by over an expression");
      }
      int column = _segment.columnIndex(expression.getIdentifier());
      if (!_segment.hasDictionary(column)) {
        // DefaultGroupByExecutor.java:106-121: the no-dictionary key generators (keys by value).  The device groups a raw INT / LONG
        // column through its key image (value - min as the dictId, include/pinot_gpu.h pg_group_key_info); whether the column's value
        // range allows one is pg_query_check's decision (GpuPlanMaker keeps the CPU plan on PG_ERR_UNSUPPORTED).
        // (round 5: FLOAT / DOUBLE columns and INT / LONG columns over more than an int too -- through a dictionary the device builds from
        //  the column's own values, PinotGpuNative.groupKeyValues; STRING / BYTES raw keys keep the CPU plan)
        DataType storedType = _indexSegment.getDataSource(expression.getIdentifier()).getDataSourceMetadata().getDataType().getStoredType();
        if (storedType != DataType.INT && storedType != DataType.LONG && storedType != DataType.FLOAT && storedType != DataType.DOUBLE) {
          throw new NotOffloadable("group-by on a raw " + storedType + " column (NoDictionary key generators)");
        }
      }
      out._groupBy[i] = column;
    }
    // ---- filter ----
    List<int[]> nodes = new ArrayList<>();
    if (filter != null) {
      Node root = lowerFilter(filter);
      if (root._kind != Kind.MATCH_ALL) {          // a filter that matches everything is no filter (MatchAllFilterOperator)
        flatten(root, nodes);
      }
    }
    out._filterNodes = new int[NODE_INTS * nodes.size()];
    for (int i = 0; i < nodes.size(); i++) {
      System.arraycopy(nodes.get(i), 0, out._filterNodes, NODE_INTS * i, NODE_INTS);
    }
    int numPredicates = _predInts.size();
    out._predInts = new int[PRED_INTS * numPredicates];
    out._predLongs = new long[PRED_LONGS * numPredicates];
    out._setOffsets = new int[numPredicates + 1];
    int totalWords = 0;
    for (int[] set : _predSets) {
      totalWords += set.length;
    }
    out._setWords = new int[totalWords];
    int at = 0;
    for (int i = 0; i < numPredicates; i++) {
      System.arraycopy(_predInts.get(i), 0, out._predInts, PRED_INTS * i, PRED_INTS);
      System.arraycopy(_predLongs.get(i), 0, out._predLongs, PRED_LONGS * i, PRED_LONGS);
      int[] set = _predSets.get(i);
      System.arraycopy(set, 0, out._setWords, at, set.length);
      at += set.length;
      out._setOffsets[i + 1] = at;
    }
    out._numGroupsLimit = _queryContext.getNumGroupsLimit();
    out._flags = (_queryContext.isNullHandlingEnabled() ? PinotGpuNative.PG_QUERY_NULL_HANDLING : 0)
        | (statsUpperBoundOk(_queryContext) ? PinotGpuNative.PG_QUERY_STATS_UPPER_BOUND_OK : 0);
    return out;
  }

  // ---- the physical tree: FilterOperatorUtils.getAndFilterOperator / getOrFilterOperator / getNotFilterOperator ----

  private static Node constant(boolean all) {
    Node n = new Node();
    n._kind = all ? Kind.MATCH_ALL : Kind.EMPTY;
    return n;
  }

  private static Node and(List<Node> children) {
    List<Node> kept = new ArrayList<>();
    for (Node child : children) {
      if (child._kind == Kind.EMPTY) {
        return constant(false);
      }
      if (child._kind != Kind.MATCH_ALL) {
        kept.add(child);
      }
    }
    if (kept.isEmpty()) {
      return constant(true);
    }
    if (kept.size() == 1) {
      return kept.get(0);
    }
    kept.sort((a, b) -> Integer.compare(a._priority, b._priority));        // List.sort is stable, like the reference's
    Node n = new Node();
    n._kind = Kind.AND;
    n._priority = AND_PRIORITY;
    n._children = kept;
    return n;
  }

  private static Node or(List<Node> children) {
    List<Node> kept = new ArrayList<>();
    for (Node child : children) {
      if (child._kind == Kind.MATCH_ALL) {
        return constant(true);
      }
      if (child._kind != Kind.EMPTY) {
        kept.add(child);
      }
    }
    if (kept.isEmpty()) {
      return constant(false);
    }
    if (kept.size() == 1) {
      return kept.get(0);
    }
    Node n = new Node();
    n._kind = Kind.OR;
    n._priority = OR_PRIORITY;
    n._children = kept;
    return n;
  }

  private static Node not(Node child) {
    if (child._kind == Kind.MATCH_ALL) {
      return constant(false);
    }
    if (child._kind == Kind.EMPTY) {
      return constant(true);
    }
    Node n = new Node();
    n._kind = Kind.NOT;
    n._priority = child._priority;             // getPriority(NotFilterOperator) = priority of its child (:228-230)
    n._children.add(child);
    return n;
  }

  private Node leaf(int kind, int column, int eval, boolean exclusive, long lo, long hi, int[] setWords, int priority) {
    _predInts.add(new int[]{kind, column, eval, exclusive ? 1 : 0});
    _predLongs.add(new long[]{lo, hi});
    _predSets.add(setWords == null ? new int[0] : setWords);
    Node n = new Node();
    n._kind = Kind.LEAF;
    n._predicate = _predInts.size() - 1;
    n._priority = priority;
    return n;
  }

  private Node lowerFilter(FilterContext filter) {
    switch (filter.getType()) {
      case AND: {
        List<Node> children = new ArrayList<>();
        for (FilterContext child : filter.getChildren()) {
          children.add(lowerFilter(child));
        }
        return and(children);
      }
      case OR: {
        List<Node> children = new ArrayList<>();
        for (FilterContext child : filter.getChildren()) {
          children.add(lowerFilter(child));
        }
        return or(children);
      }
      case NOT:
        return not(lowerFilter(filter.getChildren().get(0)));
      case CONSTANT:
        return constant(filter.isConstantTrue());
      case PREDICATE:
        return lowerPredicate(filter.getPredicate());
      default:
        throw new NotOffloadable("filter type " + filter.getType());
    }
  }

  private Node lowerPredicate(Predicate predicate) {
    ExpressionContext lhs = predicate.getLhs();
    if (lhs.getType() != ExpressionContext.Type.IDENTIFIER) {
      throw new NotOffloadable("predicate over an expression");
    }
    String columnName = lhs.getIdentifier();
    int column = _segment.columnIndex(columnName);
    DataSource dataSource = _indexSegment.getDataSource(columnName);
    boolean hasNulls = hasNulls(dataSource);
    Predicate.Type type = predicate.getType();
    if (type == Predicate.Type.IS_NULL || type == Predicate.Type.IS_NOT_NULL) {
      // FilterPlanNode.java:294-310: the null bitmap as a BitmapBasedFilterOperator; Empty / MatchAll without a null vector
      if (!hasNulls) {
        return constant(type == Predicate.Type.IS_NOT_NULL);
      }
      return leaf(PRED_IS_NULL, column, EVAL_SCAN, type == Predicate.Type.IS_NOT_NULL, 0, 0, null, BITMAP_PRIORITY);
    }
    if (type != Predicate.Type.EQ && type != Predicate.Type.NOT_EQ && type != Predicate.Type.IN && type != Predicate.Type.NOT_IN
        && type != Predicate.Type.RANGE) {
      throw new NotOffloadable("predicate type " + type);
    }
    if (!_segment.hasDictionary(column)) {
      return lowerRawPredicate(predicate, column, dataSource);
    }
    PredicateEvaluator evaluator = PredicateEvaluatorProvider.getPredicateEvaluator(predicate, dataSource, _queryContext);
    if (evaluator.isAlwaysFalse()) {
      return constant(false);                                                              // EmptyFilterOperator (FilterOperatorUtils.java:72-73)
    }
    if (evaluator.isAlwaysTrue()) {
      if (_queryContext.isNullHandlingEnabled() && hasNulls) {
        return leaf(PRED_IS_NULL, column, EVAL_SCAN, true, 0, 0, null, BITMAP_PRIORITY);   // :75-86
      }
      return constant(true);
    }
    int cardinality = dataSource.getDataSourceMetadata().getCardinality();
    boolean exclusive = evaluator.isExclusive();
    boolean isRange = evaluator instanceof SortedDictionaryBasedRangePredicateEvaluator;
    int start = 0;
    int end = 0;
    int[] dictIds = null;
    if (isRange) {
      start = ((SortedDictionaryBasedRangePredicateEvaluator) evaluator).getStartDictId();
      end = ((SortedDictionaryBasedRangePredicateEvaluator) evaluator).getEndDictId();
    } else if (type == Predicate.Type.RANGE) {
      throw new NotOffloadable("range predicate on an unsorted dictionary");
    } else {
      dictIds = exclusive ? evaluator.getNonMatchingDictIds() : evaluator.getMatchingDictIds();
      dictIds = dictIds.clone();
      Arrays.sort(dictIds);
      if (dictIds.length == 1) {
        isRange = true;                                                                    // EQ / NOT_EQ: the dictId range [d, d + 1)
        start = dictIds[0];
        end = start + 1;
      }
    }
    boolean sorted = dataSource.getDataSourceMetadata().isSorted() && dataSource.getInvertedIndex() instanceof SortedIndexReader;
    if (sorted) {
      // SortedIndexBasedFilterOperator (priority 0, :96-104, :123-126): docId ranges of the matching dictIds, adjacent ones merged,
      // complemented over [0, numDocs) for exclusive predicates (SortedIndexBasedFilterOperator.java:60-125)
      SortedIndexReader<?> sortedIndex = (SortedIndexReader<?>) dataSource.getInvertedIndex();
      List<int[]> ranges = new ArrayList<>();
      if (isRange) {
        ranges.add(new int[]{sortedIndex.getDocIds(start).getLeft(), sortedIndex.getDocIds(end - 1).getRight()});
      } else {
        for (int dictId : dictIds) {
          Pairs.IntPair pair = sortedIndex.getDocIds(dictId);
          if (!ranges.isEmpty() && pair.getLeft() == ranges.get(ranges.size() - 1)[1] + 1) {
            ranges.get(ranges.size() - 1)[1] = pair.getRight();
          } else {
            ranges.add(new int[]{pair.getLeft(), pair.getRight()});
          }
        }
      }
      if (exclusive) {
        List<int[]> rest = new ArrayList<>();
        int next = 0;
        int numDocs = _segment.numDocs();
        for (int[] range : ranges) {
          if (range[0] > next) {
            rest.add(new int[]{next, range[0] - 1});
          }
          next = range[1] + 1;
        }
        if (next < numDocs) {
          rest.add(new int[]{next, numDocs - 1});
        }
        ranges = rest;
        if (ranges.isEmpty()) {
          return constant(false);
        }
      }
      List<Node> leaves = new ArrayList<>();
      for (int[] range : ranges) {
        leaves.add(leaf(PRED_DOC_RANGE, column, EVAL_SCAN, false, range[0], range[1], null, SORTED_PRIORITY));
      }
      if (leaves.size() == 1) {
        return leaves.get(0);
      }
      Node n = new Node();                       // ONE sorted operator in the reference: an OR the iterators merge back (OrDocIdSet.java:94-112)
      n._kind = Kind.OR;
      n._priority = SORTED_PRIORITY;
      n._children = leaves;
      return n;
    }
    // :106-133: RANGE predicates scan (no range index on this path); every other type prefers the inverted index.
    // InvertedIndexFilterOperator is none of the classes reorderAndFilterChildOperators knows: it sorts last.
    boolean inverted = type != Predicate.Type.RANGE && dataSource.getInvertedIndex() != null;
    int eval = inverted ? EVAL_INVERTED : EVAL_SCAN;
    int priority = inverted ? UNKNOWN_PRIORITY : SCAN_PRIORITY;
    if (isRange) {
      return leaf(PRED_DICT_RANGE, column, eval, exclusive, start, end, null, priority);
    }
    int[] words = new int[(cardinality + 31) >>> 5];
    for (int dictId : dictIds) {
      words[dictId >>> 5] |= 1 << (dictId & 31);
    }
    return leaf(PRED_DICT_SET, column, eval, exclusive, 0, 0, words, priority);
  }

  /** Raw (no-dictionary) numeric column: RANGE / EQ / NOT_EQ as inclusive bounds, the way the *RawValueBasedRangePredicateEvaluators hold them. */
  private Node lowerRawPredicate(Predicate predicate, int column, DataSource dataSource) {
    DataType storedType = dataSource.getDataSourceMetadata().getDataType().getStoredType();
    String lower;
    String upper;
    boolean lowerInclusive = true;
    boolean upperInclusive = true;
    boolean exclusive = false;
    switch (predicate.getType()) {
      case RANGE: {
        RangePredicate range = (RangePredicate) predicate;
        lower = range.getLowerBound();
        upper = range.getUpperBound();
        lowerInclusive = range.isLowerInclusive();
        upperInclusive = range.isUpperInclusive();
        break;
      }
      case EQ:
        lower = ((EqPredicate) predicate).getValue();
        upper = lower;
        break;
      case NOT_EQ:
        lower = ((NotEqPredicate) predicate).getValue();
        upper = lower;
        exclusive = true;
        break;
      default:
        throw new NotOffloadable("IN / NOT_IN on a raw column");
    }
    boolean lowerUnbounded = RangePredicate.UNBOUNDED.equals(lower);
    boolean upperUnbounded = RangePredicate.UNBOUNDED.equals(upper);
    long lo;
    long hi;
    switch (storedType) {
      case INT: {
        // IntRawValueBasedRangePredicateEvaluator (RangePredicateEvaluatorFactory.java:326-380): exclusive bounds move by one
        int l = lowerUnbounded ? Integer.MIN_VALUE : Integer.parseInt(lower);
        int u = upperUnbounded ? Integer.MAX_VALUE : Integer.parseInt(upper);
        if (!lowerUnbounded && !lowerInclusive) {
          if (l == Integer.MAX_VALUE) {
            return constant(exclusive);
          }
          l++;
        }
        if (!upperUnbounded && !upperInclusive) {
          if (u == Integer.MIN_VALUE) {
            return constant(exclusive);
          }
          u--;
        }
        lo = l;
        hi = u;
        break;
      }
      case LONG: {
        long l = lowerUnbounded ? Long.MIN_VALUE : Long.parseLong(lower);
        long u = upperUnbounded ? Long.MAX_VALUE : Long.parseLong(upper);
        if (!lowerUnbounded && !lowerInclusive) {
          if (l == Long.MAX_VALUE) {
            return constant(exclusive);
          }
          l++;
        }
        if (!upperUnbounded && !upperInclusive) {
          if (u == Long.MIN_VALUE) {
            return constant(exclusive);
          }
          u--;
        }
        lo = l;
        hi = u;
        break;
      }
      case FLOAT: {
        // bounds widen to double exactly; exclusive bounds step with Math.nextUp / nextDown in float precision (:438-492)
        float l = lowerUnbounded ? Float.NEGATIVE_INFINITY : Float.parseFloat(lower);
        float u = upperUnbounded ? Float.POSITIVE_INFINITY : Float.parseFloat(upper);
        if (!lowerUnbounded && !lowerInclusive) {
          l = Math.nextUp(l);
        }
        if (!upperUnbounded && !upperInclusive) {
          u = Math.nextDown(u);
        }
        lo = Double.doubleToRawLongBits((double) l);
        hi = Double.doubleToRawLongBits((double) u);
        break;
      }
      case DOUBLE: {
        double l = lowerUnbounded ? Double.NEGATIVE_INFINITY : Double.parseDouble(lower);
        double u = upperUnbounded ? Double.POSITIVE_INFINITY : Double.parseDouble(upper);
        if (!lowerUnbounded && !lowerInclusive) {
          l = Math.nextUp(l);
        }
        if (!upperUnbounded && !upperInclusive) {
          u = Math.nextDown(u);
        }
        lo = Double.doubleToRawLongBits(l);
        hi = Double.doubleToRawLongBits(u);
        break;
      }
      default:
        throw new NotOffloadable("predicate on a raw " + storedType + " column");
    }
    return leaf(PRED_RAW_RANGE, column, EVAL_SCAN, exclusive, lo, hi, null, SCAN_PRIORITY);
  }

  private static boolean hasNulls(DataSource dataSource) {
    NullValueVectorReader reader = dataSource.getNullValueVector();
    if (reader == null) {
      return false;
    }
    ImmutableRoaringBitmap nulls = reader.getNullBitmap();
    return nulls != null && !nulls.isEmpty();
  }

  private static void flatten(Node node, List<int[]> out) {
    switch (node._kind) {
      case LEAF:
        out.add(new int[]{OP_LEAF, node._predicate, 0});
        return;
      case NOT:
        flatten(node._children.get(0), out);
        out.add(new int[]{OP_NOT, -1, 1});
        return;
      case AND:
      case OR:
        for (Node child : node._children) {
          flatten(child, out);
        }
        out.add(new int[]{node._kind == Kind.AND ? OP_AND : OP_OR, -1, node._children.size()});
        return;
      default:
        throw new IllegalStateException("constant nodes are folded before flattening");
    }
  }
}
