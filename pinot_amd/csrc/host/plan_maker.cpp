// plan_maker.cpp -- GpuPlanMaker, plan nodes, operators, results blocks and the combine step, above the C ABI.
// Mirrors (paths under /root/reference/pinot-core/src/main/java/org/apache/pinot/core/):
//   plan/maker/InstancePlanMakerImplV2.java:166-193,270-289   makeInstancePlan / makeSegmentPlanNode
//   plan/AggregationPlanNode.java:71-121, plan/GroupByPlanNode.java:49-74, plan/FilterPlanNode.java:88-106,195-320
//   operator/filter/FilterOperatorUtils.java:74-159 (leaf operator choice: inverted index before scan for non-range)
//   operator/query/AggregationOperator.java:64-93, operator/query/GroupByOperator.java:101-140
//   operator/combine/BaseCombineOperator.java:85-142, merger/AggregationResultsBlockMerger.java:34-44,
//   operator/combine/GroupByCombineOperator.java:132-147
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <limits>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "pinot_host.h"

namespace pinot {

// ---------------------------------------------------------------------------------------------------------------
// C ABI loader: libpinot_gpu.so sits next to this library.  No fallback: a missing library is a hard error.
// ---------------------------------------------------------------------------------------------------------------
const GpuAbi& gpuAbi() {
  static GpuAbi abi;
  static std::once_flag once;
  static std::string error;
  std::call_once(once, [] {
    Dl_info info;
    std::string dir = ".";
    if (dladdr((void*)&gpuAbi, &info) && info.dli_fname) {
      std::string p = info.dli_fname;
      size_t slash = p.rfind('/');
      if (slash != std::string::npos) dir = p.substr(0, slash);
    }
    const std::string path = dir + "/libpinot_gpu.so";
    void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_GLOBAL);
    if (!h) { error = std::string("cannot load ") + path + ": " + dlerror(); return; }
    auto sym = [&](const char* name) { void* s = dlsym(h, name); if (!s && error.empty()) error = std::string("missing symbol ") + name; return s; };
    abi.init = (decltype(abi.init))sym("pg_init");
    abi.last_error = (decltype(abi.last_error))sym("pg_last_error");
    abi.segment_open = (decltype(abi.segment_open))sym("pg_segment_open");
    abi.segment_close = (decltype(abi.segment_close))sym("pg_segment_close");
    abi.query_check = (decltype(abi.query_check))sym("pg_query_check");
    abi.execute = (decltype(abi.execute))sym("pg_execute");
    abi.execute_batch = (decltype(abi.execute_batch))sym("pg_execute_batch");
    abi.result_free = (decltype(abi.result_free))sym("pg_result_free");
    abi.filter_bitmap = (decltype(abi.filter_bitmap))sym("pg_filter_bitmap");
    abi.group_key_info = (decltype(abi.group_key_info))sym("pg_group_key_info");
    abi.group_key_values = (decltype(abi.group_key_values))sym("pg_group_key_values");
  });
  if (!error.empty()) throw std::runtime_error("pinot GPU engine unavailable (no CPU fallback in this library): " + error);
  return abi;
}

static void checkStatus(pg_status st, const char* what) {
  if (st == PG_OK) return;
  const std::string msg = std::string(what) + ": " + gpuAbi().last_error();
  if (st == PG_ERR_UNSUPPORTED) throw UnsupportedOperationException(msg);
  if (st == PG_ERR_INVALID_ARGUMENT) throw QueryException(msg);
  throw std::runtime_error(msg);   // BaseCombineOperator.wrapOperatorException attaches the segment name (:185-199)
}

// ---------------------------------------------------------------------------------------------------------------
// ImmutableSegment
// ---------------------------------------------------------------------------------------------------------------
ImmutableSegment::~ImmutableSegment() {
  if (_handle) { try { gpuAbi().segment_close(_handle); } catch (...) {} }
}

const DataSource& ImmutableSegment::getDataSource(const std::string& column) const {
  for (const auto& c : _columns) if (c.name == column) return c;
  throw QueryException("Cannot find data source for column: " + column);   // same text as the reference's segment impl
}

int ImmutableSegment::getColumnIndex(const std::string& column) const {
  for (size_t i = 0; i < _columns.size(); ++i) if (_columns[i].name == column) return (int)i;
  throw QueryException("Cannot find data source for column: " + column);
}

void ImmutableSegment::load(int deviceId) {
  if (_handle) return;
  std::vector<pg_column_desc> descs(_columns.size());
  for (size_t i = 0; i < _columns.size(); ++i) {
    DataSource& ds = _columns[i];
    pg_column_desc& d = descs[i];
    memset(&d, 0, sizeof(d));
    d.name = ds.name.c_str();
    d.stored_type = ds.dataType == DataType::LONG ? PG_TYPE_LONG : (ds.dataType == DataType::FLOAT ? PG_TYPE_FLOAT : (ds.dataType == DataType::DOUBLE ? PG_TYPE_DOUBLE : PG_TYPE_INT));
    d.fwd_encoding = ds.hasDictionary ? PG_FWD_FIXED_BIT_DICT : PG_FWD_RAW_FIXED_BYTE;
    d.bits_per_value = ds.bitsPerElement;
    d.cardinality = ds.hasDictionary ? ds.cardinality : 0;
    d.fwd_data = ds.forwardIndex;
    d.fwd_size = ds.forwardIndexSize;
    if (ds.hasDictionary) {
      if (ds.dataType == DataType::STRING) {
        // the device only ever sees dictIds of STRING columns; give it a 0..C-1 big-endian placeholder dictionary
        ds.placeholderDictionary.resize((size_t)ds.cardinality * 4);
        for (int k = 0; k < ds.cardinality; ++k) {
          uint8_t* p = ds.placeholderDictionary.data() + (size_t)k * 4;
          p[0] = (uint8_t)(k >> 24); p[1] = (uint8_t)(k >> 16); p[2] = (uint8_t)(k >> 8); p[3] = (uint8_t)k;
        }
        d.dict_data = ds.placeholderDictionary.data();
        d.dict_size = ds.placeholderDictionary.size();
      } else {
        d.dict_data = ds.dictionaryBuffer;
        d.dict_size = ds.dictionaryBufferSize;
      }
    }
    if (ds.hasInvertedIndex) { d.inv_data = ds.invertedIndex; d.inv_size = ds.invertedIndexSize; }
    if (ds.nullValueVector && ds.nullValueVectorSize) { d.null_data = ds.nullValueVector; d.null_size = ds.nullValueVectorSize; }
  }
  pg_segment_desc sd;
  memset(&sd, 0, sizeof(sd));
  sd.name = _name.c_str();
  sd.num_docs = _totalDocs;
  sd.num_columns = (int32_t)descs.size();
  sd.columns = descs.data();
  sd.device_id = deviceId;
  checkStatus(gpuAbi().segment_open(&sd, &_handle), ("loading segment " + _name).c_str());
  _deviceId = deviceId;
}

void ImmutableSegment::destroy() {
  if (_handle) { gpuAbi().segment_close(_handle); _handle = nullptr; }
}

// ---------------------------------------------------------------------------------------------------------------
// AggregationFunction
// ---------------------------------------------------------------------------------------------------------------
std::string AggregationFunction::getResultColumnName() const {
  static const char* names[] = {"count", "sum", "min", "max", "avg"};
  return std::string(names[(int)_type]) + "(" + _column + ")";
}

IntermediateResult AggregationFunction::fromDevice(const pg_agg_value& v) const {
  // null handling: the holder is an ObjectAggregationResultHolder that stays null until a non-null value arrives
  // (SumAggregationFunction.java:52-57,147-157; same in Min / Max / Avg); COUNT is never null
  if (_nullHandlingEnabled && _type != AggregationFunctionType::COUNT && v.count == 0) return std::monostate{};
  switch (_type) {
    case AggregationFunctionType::COUNT: return (int64_t)v.count;          // CountAggregationFunction.extractAggregationResult -> Long
    case AggregationFunctionType::SUM: return v.sum;                       // Double
    case AggregationFunctionType::MIN: return v.min;                       // Double, +inf when nothing matched
    case AggregationFunctionType::MAX: return v.max;                       // Double, -inf when nothing matched
    case AggregationFunctionType::AVG: return AvgPair{v.sum, v.count};     // AvgPair(sum, count)
  }
  return 0.0;
}

// java.lang.Math.min / max on doubles (MinAggregationFunction.merge / MaxAggregationFunction.merge): a NaN operand gives NaN, and
// the zeros are ordered -0.0 < +0.0 -- std::fmin / fmax drop NaNs and leave the zeros' order unspecified.
static double javaMin(double a, double b) {
  if (a != a || b != b) return std::numeric_limits<double>::quiet_NaN();
  if (a == 0.0 && b == 0.0) return std::signbit(a) ? a : b;
  return a < b ? a : b;
}
static double javaMax(double a, double b) {
  if (a != a || b != b) return std::numeric_limits<double>::quiet_NaN();
  if (a == 0.0 && b == 0.0) return std::signbit(a) ? b : a;
  return a > b ? a : b;
}

IntermediateResult AggregationFunction::merge(const IntermediateResult& a, const IntermediateResult& b) const {
  // SumAggregationFunction.merge :223-233 and friends under null handling: a null side yields the other side
  if (isNullResult(a)) return b;
  if (isNullResult(b)) return a;
  switch (_type) {
    case AggregationFunctionType::COUNT: return std::get<int64_t>(a) + std::get<int64_t>(b);   // CountAggregationFunction.merge
    case AggregationFunctionType::SUM: return std::get<double>(a) + std::get<double>(b);        // SumAggregationFunction.merge :223-233
    case AggregationFunctionType::MIN: return javaMin(std::get<double>(a), std::get<double>(b));      // Math.min: NaN wins, -0.0 < +0.0
    case AggregationFunctionType::MAX: return javaMax(std::get<double>(a), std::get<double>(b));
    case AggregationFunctionType::AVG: {                                                       // AvgAggregationFunction.merge -> AvgPair.apply
      AvgPair r = std::get<AvgPair>(a);
      const AvgPair& o = std::get<AvgPair>(b);
      r.sum += o.sum; r.count += o.count;
      return r;
    }
  }
  return a;
}

double AggregationFunction::extractFinalResult(const IntermediateResult& r) const {
  if (isNullResult(r)) return std::nan("");           // the final result is null (printed as such by the callers)
  switch (_type) {
    case AggregationFunctionType::COUNT: return (double)std::get<int64_t>(r);
    case AggregationFunctionType::AVG: {                                    // AvgAggregationFunction.extractFinalResult :209-218
      const AvgPair& p = std::get<AvgPair>(r);
      return p.count == 0 ? -INFINITY : p.sum / (double)p.count;            // DEFAULT_FINAL_RESULT = Double.NEGATIVE_INFINITY
    }
    default: return std::get<double>(r);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Filter lowering: FilterPlanNode.constructPhysicalOperator + FilterOperatorUtils.getLeafFilterOperator
// ---------------------------------------------------------------------------------------------------------------
namespace {

struct LoweredQuery {
  std::vector<pg_filter_node> nodes;
  std::vector<pg_predicate> predicates;
  std::vector<std::vector<uint32_t>> setWords;   // owns DICT_SET bitsets
  std::vector<pg_aggregation> aggregations;
  std::vector<int32_t> groupBy;
  pg_query query;
};

// The physical filter tree before it is flattened: FilterPlanNode.constructPhysicalOperator builds it bottom-up through
// FilterOperatorUtils.get{Leaf,And,Or,Not}FilterOperator (core/operator/filter/FilterOperatorUtils.java:68-193), which drop
// MatchAll / Empty children and re-order the children of an AND by priority (:196-245).  The order is part of the contract:
// AndDocIdSet.iterator() applies the scan-based children in list order, so numEntriesScannedInFilter depends on it.
struct PhysNode {
  enum Kind { MATCH_ALL, EMPTY, LEAF, AND, OR, NOT } kind = LEAF;
  int predicate = -1;
  int priority = 10000;                      // PrioritizedFilterOperator.java:32-39
  std::vector<PhysNode> children;
};

constexpr int kSortedPriority = 0, kBitmapPriority = 100, kAndPriority = 300, kOrPriority = 400, kScanPriority = 500, kUnknownPriority = 10000;

PhysNode physLeaf(LoweredQuery* out, const pg_predicate& p, int priority) {
  out->predicates.push_back(p);
  PhysNode n;
  n.kind = PhysNode::LEAF; n.predicate = (int)out->predicates.size() - 1; n.priority = priority;
  return n;
}

PhysNode physAnd(std::vector<PhysNode> children) {       // FilterOperatorUtils.getAndFilterOperator :136-158
  std::vector<PhysNode> kept;
  for (auto& c : children) {
    if (c.kind == PhysNode::EMPTY) { PhysNode e; e.kind = PhysNode::EMPTY; return e; }
    if (c.kind != PhysNode::MATCH_ALL) kept.push_back(std::move(c));
  }
  if (kept.empty()) { PhysNode m; m.kind = PhysNode::MATCH_ALL; return m; }
  if (kept.size() == 1) return std::move(kept[0]);
  std::stable_sort(kept.begin(), kept.end(), [](const PhysNode& a, const PhysNode& b) { return a.priority < b.priority; });   // List.sort is stable
  PhysNode n;
  n.kind = PhysNode::AND; n.priority = kAndPriority; n.children = std::move(kept);
  return n;
}

PhysNode physOr(std::vector<PhysNode> children) {        // FilterOperatorUtils.getOrFilterOperator :161-183
  std::vector<PhysNode> kept;
  for (auto& c : children) {
    if (c.kind == PhysNode::MATCH_ALL) { PhysNode m; m.kind = PhysNode::MATCH_ALL; return m; }
    if (c.kind != PhysNode::EMPTY) kept.push_back(std::move(c));
  }
  if (kept.empty()) { PhysNode e; e.kind = PhysNode::EMPTY; return e; }
  if (kept.size() == 1) return std::move(kept[0]);
  PhysNode n;
  n.kind = PhysNode::OR; n.priority = kOrPriority; n.children = std::move(kept);
  return n;
}

PhysNode physNot(PhysNode child) {                       // FilterOperatorUtils.getNotFilterOperator :186-194
  PhysNode n;
  if (child.kind == PhysNode::MATCH_ALL) { n.kind = PhysNode::EMPTY; return n; }
  if (child.kind == PhysNode::EMPTY) { n.kind = PhysNode::MATCH_ALL; return n; }
  n.kind = PhysNode::NOT; n.priority = child.priority;  // getPriority(NotFilterOperator) = priority of its child (:228-230)
  n.children.push_back(std::move(child));
  return n;
}

PhysNode lowerFilter(const FilterContext& f, const ImmutableSegment& seg, bool nullHandling, LoweredQuery* out) {
  switch (f.type) {
    case FilterContext::Type::AND:
    case FilterContext::Type::OR: {
      std::vector<PhysNode> children;
      for (const auto& c : f.children) children.push_back(lowerFilter(c, seg, nullHandling, out));
      return f.type == FilterContext::Type::AND ? physAnd(std::move(children)) : physOr(std::move(children));
    }
    case FilterContext::Type::NOT: return physNot(lowerFilter(f.children.at(0), seg, nullHandling, out));
    case FilterContext::Type::PREDICATE: {
      const DataSource& ds = seg.getDataSource(f.predicate.column);
      pg_predicate p;
      memset(&p, 0, sizeof(p));
      p.column = seg.getColumnIndex(f.predicate.column);
      const bool hasNulls = ds.nullValueVector != nullptr && ds.nullValueVectorSize > 0;
      PhysNode constant;
      if (f.predicate.type == Predicate::Type::IS_NULL || f.predicate.type == Predicate::Type::IS_NOT_NULL) {
        // FilterPlanNode.java:294-310: the null bitmap as a BitmapBasedFilterOperator, Empty / MatchAll without a null vector
        if (!hasNulls) { constant.kind = f.predicate.type == Predicate::Type::IS_NULL ? PhysNode::EMPTY : PhysNode::MATCH_ALL; return constant; }
        p.kind = PG_PRED_IS_NULL;
        p.exclusive = f.predicate.type == Predicate::Type::IS_NOT_NULL;
        return physLeaf(out, p, kBitmapPriority);
      }
      const PredicateEvaluator ev = getPredicateEvaluator(f.predicate, ds);
      if (ev.alwaysTrue && nullHandling && hasNulls) { p.kind = PG_PRED_IS_NULL; p.exclusive = 1; return physLeaf(out, p, kBitmapPriority); }   // FilterOperatorUtils.java:78-86
      if (ev.alwaysTrue) { constant.kind = PhysNode::MATCH_ALL; return constant; }     // MatchAllFilterOperator (FilterOperatorUtils.java:79-92)
      if (ev.alwaysFalse) { constant.kind = PhysNode::EMPTY; return constant; }        // EmptyFilterOperator
      if (ev.rawRange) {
        p.kind = PG_PRED_RAW_RANGE; p.lo = ev.rawLower; p.hi = ev.rawUpper; p.exclusive = ev.exclusive;
        return physLeaf(out, p, kScanPriority);
      }
      if (ds.isSorted && ev.isRange && (int)ds.sortedDocIdRanges.size() == 2 * ds.cardinality && ev.endDictId > ev.startDictId) {
        // SortedIndexBasedFilterOperator (priority 0, FilterOperatorUtils.java:96-104): RANGE / EQ / NOT_EQ on a sorted column are the docId
        // range [start of startDictId, end of endDictId - 1] (SortedIndexBasedFilterOperator.java:60-85); nothing is scanned
        p.kind = PG_PRED_DOC_RANGE;
        p.lo = ds.sortedDocIdRanges[2 * (size_t)ev.startDictId];
        p.hi = ds.sortedDocIdRanges[2 * (size_t)(ev.endDictId - 1) + 1];
        p.exclusive = ev.exclusive;
        return physLeaf(out, p, kSortedPriority);
      }
      if (ds.isSorted && !ev.isRange && (int)ds.sortedDocIdRanges.size() == 2 * ds.cardinality && !ev.matchingDictIds.empty() &&
          ev.matchingDictIds.size() <= 6) {
        // IN / NOT IN on a sorted column: ONE SortedIndexBasedFilterOperator whose SortedDocIdSet holds the docId ranges of the
        // matching dictIds, adjacent ones merged (SortedIndexBasedFilterOperator.java:86-125); NOT IN holds the complementary
        // ranges.  Here: an OR of docId-range leaves with the sorted operator's priority -- the iterators merge such an OR back
        // into one index-based docId set (OrDocIdSet.java:94-112).
        std::vector<std::pair<int32_t, int32_t>> ranges;
        for (int d : ev.matchingDictIds) {      // ascending dictIds
          const int32_t s0 = ds.sortedDocIdRanges[2 * (size_t)d], e0 = ds.sortedDocIdRanges[2 * (size_t)d + 1];
          if (!ranges.empty() && s0 == ranges.back().second + 1) ranges.back().second = e0;
          else ranges.push_back({s0, e0});
        }
        if (ev.exclusive) {
          std::vector<std::pair<int32_t, int32_t>> rest;
          int32_t next = 0;
          for (const auto& r : ranges) { if (r.first > next) rest.push_back({next, r.first - 1}); next = r.second + 1; }
          if (next < seg.getTotalDocs()) rest.push_back({next, seg.getTotalDocs() - 1});
          ranges.swap(rest);
          if (ranges.empty()) { constant.kind = PhysNode::EMPTY; return constant; }
        }
        std::vector<PhysNode> leaves;
        for (const auto& r : ranges) {
          pg_predicate dr;
          memset(&dr, 0, sizeof(dr));
          dr.kind = PG_PRED_DOC_RANGE; dr.lo = r.first; dr.hi = r.second; dr.column = p.column;
          leaves.push_back(physLeaf(out, dr, kSortedPriority));
        }
        if (leaves.size() == 1) return std::move(leaves[0]);
        PhysNode n;
        n.kind = PhysNode::OR; n.priority = kSortedPriority; n.children = std::move(leaves);
        return n;
      }
      p.exclusive = ev.exclusive;
      // FilterOperatorUtils.java:96-133: RANGE predicates scan (no sorted / range index on this path); every other predicate type
      // prefers the inverted index when the column has one.  InvertedIndexFilterOperator is none of the classes
      // reorderAndFilterChildOperators knows (:204-243), so it sorts last (UNKNOWN_FILTER_PRIORITY).
      p.eval = (f.predicate.type != Predicate::Type::RANGE && ds.hasInvertedIndex) ? PG_EVAL_INVERTED : PG_EVAL_SCAN;
      if (ev.isRange) { p.kind = PG_PRED_DICT_RANGE; p.lo = ev.startDictId; p.hi = ev.endDictId; }
      else {
        p.kind = PG_PRED_DICT_SET;
        std::vector<uint32_t> words(((size_t)ds.cardinality + 31) / 32, 0u);
        for (int d : ev.matchingDictIds) words[(size_t)d >> 5] |= 1u << (d & 31);
        out->setWords.push_back(std::move(words));
        p.set_words = out->setWords.back().data();
        p.num_set_words = (int32_t)out->setWords.back().size();
      }
      return physLeaf(out, p, p.eval == PG_EVAL_INVERTED ? kUnknownPriority : kScanPriority);
    }
  }
  throw std::logic_error("unreachable filter type");
}

// The physical filter tree as text: which operator FilterOperatorUtils would build for every node, in the order the AND children run.
//   SORTED(col docs a..b)   SortedIndexBasedFilterOperator     BITMAP(col IS [NOT] NULL)  BitmapBasedFilterOperator over the null vector
//   INVERTED(col ...)       InvertedIndexFilterOperator        SCAN(col ...)              ScanBasedFilterOperator
std::string explainPhysical(const PhysNode& n, const LoweredQuery& lq, const ImmutableSegment& seg) {
  switch (n.kind) {
    case PhysNode::MATCH_ALL: return "MATCH_ALL";
    case PhysNode::EMPTY: return "EMPTY";
    case PhysNode::NOT: return "NOT(" + explainPhysical(n.children[0], lq, seg) + ")";
    case PhysNode::AND: case PhysNode::OR: {
      std::string s = n.kind == PhysNode::AND ? "AND(" : "OR(";
      for (size_t i = 0; i < n.children.size(); ++i) s += (i ? ", " : "") + explainPhysical(n.children[i], lq, seg);
      return s + ")";
    }
    default: break;
  }
  const pg_predicate& p = lq.predicates.at((size_t)n.predicate);
  const std::string col = p.column >= 0 && p.column < (int)seg.getDataSources().size() ? seg.getDataSources()[(size_t)p.column].name : std::string("?");
  const std::string neg = p.exclusive ? " NOT" : "";
  switch (p.kind) {
    case PG_PRED_DOC_RANGE: return "SORTED(" + col + neg + " docs " + std::to_string(p.lo) + ".." + std::to_string(p.hi) + ")";
    case PG_PRED_IS_NULL: return "BITMAP(" + col + " IS" + neg + " NULL)";
    case PG_PRED_RAW_RANGE: return "SCAN(" + col + neg + " raw " + std::to_string(p.lo) + ".." + std::to_string(p.hi) + ")";
    case PG_PRED_DICT_RANGE:
      return std::string(p.eval == PG_EVAL_INVERTED ? "INVERTED(" : "SCAN(") + col + neg + " dictIds " + std::to_string(p.lo) + ".." + std::to_string(p.hi - 1) + ")";
    default: {
      int count = 0;
      for (int w = 0; w < p.num_set_words; ++w) count += __builtin_popcount(p.set_words[w]);
      return std::string(p.eval == PG_EVAL_INVERTED ? "INVERTED(" : "SCAN(") + col + neg + " in " + std::to_string(count) + " dictIds)";
    }
  }
}

void flattenFilter(const PhysNode& n, LoweredQuery* out) {
  switch (n.kind) {
    case PhysNode::MATCH_ALL: case PhysNode::EMPTY: {
      pg_predicate p;
      memset(&p, 0, sizeof(p));
      p.kind = n.kind == PhysNode::MATCH_ALL ? PG_PRED_MATCH_ALL : PG_PRED_MATCH_NONE;
      p.column = -1;
      out->predicates.push_back(p);
      out->nodes.push_back(pg_filter_node{PG_FILTER_LEAF, (int32_t)out->predicates.size() - 1, 0, 0});
      return;
    }
    case PhysNode::LEAF: out->nodes.push_back(pg_filter_node{PG_FILTER_LEAF, n.predicate, 0, 0}); return;
    case PhysNode::NOT: flattenFilter(n.children[0], out); out->nodes.push_back(pg_filter_node{PG_FILTER_NOT, -1, 1, 0}); return;
    default:
      for (const auto& c : n.children) flattenFilter(c, out);
      out->nodes.push_back(pg_filter_node{n.kind == PhysNode::AND ? PG_FILTER_AND : PG_FILTER_OR, -1, (int32_t)n.children.size(), 0});
      return;
  }
}

std::unique_ptr<LoweredQuery> lowerQuery(const ImmutableSegment& seg, const QueryContext& qc) {
  auto lq = std::make_unique<LoweredQuery>();
  // reserve so that set_words pointers taken during lowering stay valid
  lq->setWords.reserve(64);
  lq->predicates.reserve(256);
  if (qc.hasFilter) {
    const PhysNode root = lowerFilter(qc.filter, seg, qc.nullHandlingEnabled, lq.get());
    if (root.kind != PhysNode::MATCH_ALL) flattenFilter(root, lq.get());      // a filter that matches everything is no filter (MatchAllFilterOperator)
  }
  for (const auto& a : qc.aggregations) {
    pg_aggregation pa;
    pa.function = (int32_t)a.function;
    pa.column = a.column == "*" ? -1 : seg.getColumnIndex(a.column);
    if (pa.column >= 0) {
      const DataSource& ds = seg.getDataSource(a.column);
      if (!isNumeric(ds.dataType) && a.function != AggregationFunctionType::COUNT)
        throw QueryException("Cannot compute " + AggregationFunction(a.function, a.column).getResultColumnName() + " for non-numeric type: STRING");
      // COUNT(col) == COUNT(*) unless null handling is on (CountAggregationFunction.java:44-50)
      if (a.function == AggregationFunctionType::COUNT && !qc.nullHandlingEnabled) pa.column = -1;
    }
    lq->aggregations.push_back(pa);
  }
  long long product = 1;
  for (const auto& g : qc.groupByExpressions) {
    const DataSource& ds = seg.getDataSource(g);
    if (!ds.hasDictionary) {
      // DefaultGroupByExecutor.java:106-121: the no-dictionary key generators (keys by value).  The device groups a raw INT / LONG column
      // through its key image (value - min as the dictId); whether the column's value range allows one is pg_query_check's decision.
      // (round 5: FLOAT / DOUBLE columns and INT / LONG columns over more than an int too -- through a dictionary the device builds from the
      //  column's own values, pg_group_key_values; NoDictionarySingleColumnGroupKeyGenerator.java:100-135 keys all four types by value)
      if (ds.dataType == DataType::STRING)
        throw UnsupportedOperationException("group-by on a raw STRING column uses the no-dictionary key generator (CPU plan)");
      lq->groupBy.push_back(seg.getColumnIndex(g));
      continue;
    }
    // (a key space beyond an int -- the Long / ArrayMap holders -- is the device's hashed table; pg_query_check prices either table
    //  against the device's budget: DictionaryBasedGroupKeyGenerator.java:164-184)
    if (product <= 0x7FFFFFFFll) product *= ds.cardinality + ((qc.nullHandlingEnabled && ds.nullValueVector != nullptr && ds.nullValueVectorSize > 0) ? 1 : 0);   // NULL is a key value of its own
    lq->groupBy.push_back(seg.getColumnIndex(g));
  }
  pg_query& q = lq->query;
  memset(&q, 0, sizeof(q));
  q.filter = lq->nodes.data();
  q.num_filter_nodes = (int32_t)lq->nodes.size();
  q.predicates = lq->predicates.data();
  q.num_predicates = (int32_t)lq->predicates.size();
  q.aggregations = lq->aggregations.data();
  q.num_aggregations = (int32_t)lq->aggregations.size();
  q.group_by_columns = lq->groupBy.data();
  q.num_group_by = (int32_t)lq->groupBy.size();
  q.num_groups_limit = qc.numGroupsLimit;
  q.flags = (qc.nullHandlingEnabled ? PG_QUERY_NULL_HANDLING : PG_QUERY_DEFAULT) | (qc.gpuExactFilterStats ? 0 : PG_QUERY_STATS_UPPER_BOUND_OK);
  return lq;
}

// The segment operators of ONE query share one pg_execute_batch.  CombinePlanNode plans every segment of a query before anything runs
// (core/plan/CombinePlanNode.java:92-110) and BaseCombineOperator's tasks then call the operators from a thread pool
// (core/operator/combine/BaseCombineOperator.java:85-142): the operators register here at plan time; whichever task asks for its block
// first runs the whole batch -- one launch over all the query's resident segments where the library can share it (include/pinot_gpu.h,
// pg_execute_batch) -- and the others take their results as they come to ask.  A batch of one is a plain pg_execute.
class GpuBatch {
 public:
  ~GpuBatch() {
    for (size_t i = 0; i < _results.size(); ++i) if (_ran && !_taken[i] && _statuses[i] == PG_OK) gpuAbi().result_free(&_results[i]);
  }
  int add(pg_segment* handle, const pg_query* query) {
    _handles.push_back(handle);
    _queries.push_back(query);
    return (int)_handles.size() - 1;
  }
  // The result of item `index` (the caller owns it: pg_result_free); throws what a pg_execute of the item would have thrown.
  void take(int index, const std::string& what, pg_result* out) {
    std::unique_lock<std::mutex> lk(_mu);
    if (!_started) {
      _started = true;
      lk.unlock();
      const size_t n = _handles.size();
      std::vector<pg_result> results(n);
      std::vector<pg_status> statuses(n, PG_ERR_INTERNAL);
      pg_status st;
      std::string error;
      if (n == 1) {
        memset(&results[0], 0, sizeof(pg_result));
        st = statuses[0] = gpuAbi().execute(_handles[0], _queries[0], &results[0]);
        if (st != PG_OK) error = gpuAbi().last_error();
        st = PG_OK;
      } else {
        st = gpuAbi().execute_batch(_handles.data(), _queries.data(), (int32_t)n, results.data(), statuses.data());
        bool anyFailed = st != PG_OK;
        for (pg_status s : statuses) anyFailed = anyFailed || s != PG_OK;
        if (anyFailed) error = gpuAbi().last_error();      // the call's, or the first failed item's (pg_execute_batch)
      }
      lk.lock();
      _results = std::move(results);
      _statuses = std::move(statuses);
      _taken.assign(n, false);
      _callStatus = st;
      _error = std::move(error);
      _ran = true;
      _cv.notify_all();
    } else {
      _cv.wait(lk, [&] { return _ran; });
    }
    const pg_status st = _callStatus != PG_OK ? _callStatus : _statuses[(size_t)index];
    if (st != PG_OK) {
      const std::string msg = what + ": " + _error;
      if (st == PG_ERR_UNSUPPORTED) throw UnsupportedOperationException(msg);
      if (st == PG_ERR_INVALID_ARGUMENT) throw QueryException(msg);
      throw std::runtime_error(msg);
    }
    if (_taken[(size_t)index]) throw std::runtime_error(what + ": the block of this operator was already taken");
    _taken[(size_t)index] = true;
    *out = _results[(size_t)index];
  }
 private:
  std::vector<pg_segment*> _handles;
  std::vector<const pg_query*> _queries;
  std::mutex _mu;
  std::condition_variable _cv;
  bool _started = false, _ran = false;
  pg_status _callStatus = PG_OK;
  std::string _error;
  std::vector<pg_result> _results;
  std::vector<pg_status> _statuses;
  std::vector<bool> _taken;
};
// Set by executeCombined around the planning of a query's segments (one thread plans: InstancePlanMakerImplV2.makeInstancePlan :166-193).
static thread_local std::shared_ptr<GpuBatch> t_planningBatch;

// One operator class serves both AggregationOperator and GroupByOperator roles: the device does the whole
// filter -> project -> aggregate pull loop in one fused launch, nextBlock() is called exactly once (Operator.java:35-44).
class GpuAggregationOperator : public Operator {
 public:
  GpuAggregationOperator(const ImmutableSegment* seg, QueryContext qc, std::unique_ptr<LoweredQuery> lq, std::shared_ptr<GpuBatch> batch)
      : _segment(seg), _queryContext(std::move(qc)), _lowered(std::move(lq)), _batch(std::move(batch)) {
    if (_batch) _batchIndex = _batch->add((pg_segment*)_segment->handle(), &_lowered->query);
  }

  ResultsBlock nextBlock() override {
    pg_result res{};      // zero-initialised: pg_execute also clears it first thing, whatever path fails
    const std::string what = "executing on segment " + _segment->getSegmentName();
    if (_batch) _batch->take(_batchIndex, what, &res);
    else checkStatus(gpuAbi().execute(_segment->handle(), &_lowered->query, &res), what.c_str());
    ResultsBlock block;
    std::vector<AggregationFunction> functions;
    for (const auto& a : _queryContext.aggregations) functions.emplace_back(a.function, a.column, _queryContext.nullHandlingEnabled);
    block.stats.numDocsScanned = res.stats.num_docs_scanned;
    block.stats.numEntriesScannedInFilter = res.stats.num_entries_scanned_in_filter;
    block.stats.numEntriesScannedPostFilter = res.stats.num_entries_scanned_post_filter;
    block.stats.numTotalDocs = res.stats.num_total_docs;
    block.numGroupsLimitReached = res.num_groups_limit_reached != 0;
    block.deviceMs = res.device_ms;
    block.kernelMs = res.dominant_kernel_ms;
    const int na = (int)functions.size();
    if (_queryContext.groupByExpressions.empty()) {
      block.isGroupBy = false;
      block.aggregation.functions = functions;
      for (int a = 0; a < na; ++a) block.aggregation.results.push_back(functions[(size_t)a].fromDevice(res.aggregations[a]));
    } else {
      block.isGroupBy = true;
      GroupByResultsBlock& g = block.groupBy;
      g.groupByColumns = _queryContext.groupByExpressions;
      g.functions = functions;
      std::vector<const DataSource*> keyCols;
      for (const auto& c : g.groupByColumns) keyCols.push_back(&_segment->getDataSource(c));
      for (const DataSource* ds : keyCols) g.groupByTypes.push_back(ds->dataType);
      // raw key columns: the entry is an offset from the column's smallest value (pg_group_key_info); dictionary columns: a dictId
      std::vector<int64_t> keyBase(keyCols.size(), 0);
      std::vector<int32_t> keyIsOffset(keyCols.size(), 0), keyNullEntry(keyCols.size(), 0);
      for (size_t j = 0; j < keyCols.size(); ++j)
        checkStatus(gpuAbi().group_key_info((const pg_segment*)_segment->handle(), _lowered->groupBy[j], &keyBase[j], &keyIsOffset[j], &keyNullEntry[j]), "reading the group keys");
      // is_offset 2: the entry is the value's rank among the column's distinct values (a raw FLOAT / DOUBLE / wide INT / LONG key column)
      std::vector<std::vector<int64_t>> keyValueBits(keyCols.size());
      for (size_t j = 0; j < keyCols.size(); ++j) {
        if (keyIsOffset[j] != 2) continue;
        int32_t count = 0;
        checkStatus(gpuAbi().group_key_values((pg_segment*)_segment->handle(), _lowered->groupBy[j], nullptr, 0, &count), "reading the group key values");
        keyValueBits[j].assign((size_t)std::max(count, 1), 0);
        checkStatus(gpuAbi().group_key_values((pg_segment*)_segment->handle(), _lowered->groupBy[j], keyValueBits[j].data(), (int32_t)keyValueBits[j].size(), &count), "reading the group key values");
      }
      for (int i = 0; i < res.num_groups; ++i) {
        GroupKey key;
        key.groupId = res.group_ids[i];
        // DictionaryBasedGroupKeyGenerator.getKeys: groupId -> dictIds (mixed radix) -> dictionary VALUES
        // Under null handling a nullable key column has one more digit value, `cardinality` = NULL (include/pinot_gpu.h, pg_query.flags):
        // the no-dictionary key generators of DefaultGroupByExecutor.java:106-121 treat NULL as a key value of its own.
        // (the device hands the digits over as they are -- pg_result.group_key_dict_ids -- for every holder: int, long and array keyed)
        const size_t nk = keyCols.size();
        for (size_t j = 0; j < nk; ++j) {
          const DataSource* ds = keyCols[j];
          const int d = res.group_key_dict_ids[(size_t)i * nk + j];
          key.dictIds.push_back(d);
          if (d == keyNullEntry[j]) key.keys.emplace_back(std::monostate{});
          else if (keyIsOffset[j] == 2) {
            const int64_t bits = keyValueBits[j][(size_t)d];
            if (ds->dataType == DataType::FLOAT || ds->dataType == DataType::DOUBLE) { double v; memcpy(&v, &bits, 8); key.keys.emplace_back(v); }
            else key.keys.emplace_back(bits);
          }
          else if (keyIsOffset[j]) key.keys.emplace_back((int64_t)(keyBase[j] + (int64_t)d));      // NoDictionary*GroupKeyGenerator: the key IS the value
          else if (ds->dataType == DataType::STRING) key.keys.emplace_back(ds->dictionary->getStringValue(d));
          else if (ds->dataType == DataType::FLOAT || ds->dataType == DataType::DOUBLE) key.keys.emplace_back(ds->dictionary->getDoubleValue(d));
          else key.keys.emplace_back(ds->dictionary->getLongValue(d));
        }
        g.groupKeys.push_back(std::move(key));
        std::vector<IntermediateResult> row;
        for (int a = 0; a < na; ++a) row.push_back(functions[(size_t)a].fromDevice(res.group_aggregations[(size_t)i * (size_t)na + (size_t)a]));
        g.results.push_back(std::move(row));
      }
    }
    gpuAbi().result_free(&res);
    _stats = block.stats;
    trimSegmentGroupByBlock(&block, _queryContext);      // GroupByOperator.java:119-135 (ORDER BY + minSegmentGroupTrimSize)
    return block;
  }

  std::string toExplainString() const override { return _queryContext.groupByExpressions.empty() ? "GPU_AGGREGATE" : "GPU_GROUP_BY"; }
  ExecutionStatistics getExecutionStatistics() const override { return _stats; }
  const ImmutableSegment* getIndexSegment() const override { return _segment; }

 private:
  const ImmutableSegment* _segment;
  QueryContext _queryContext;
  std::unique_ptr<LoweredQuery> _lowered;
  std::shared_ptr<GpuBatch> _batch;       // null: the operator runs its own pg_execute
  int _batchIndex = -1;
  ExecutionStatistics _stats;
};

class GpuAggregationPlanNode : public PlanNode {
 public:
  // Plan-time eligibility: pg_query_check takes the decision pg_execute would take (leaf / node / column-stream tables, key spaces,
  // nullable group-by ...) without launching anything; PG_ERR_UNSUPPORTED becomes the UnsupportedOperationException on which the
  // caller keeps the CPU plan (InstancePlanMakerImplV2.makeSegmentPlanNode :270-289).  Nothing is rejected at run time.
  GpuAggregationPlanNode(const ImmutableSegment* seg, QueryContext qc, std::unique_ptr<LoweredQuery> lq)
      : _segment(seg), _queryContext(std::move(qc)), _lowered(std::move(lq)), _batch(t_planningBatch) {
    checkStatus(gpuAbi().query_check((const pg_segment*)seg->handle(), &_lowered->query), "planning the segment query");
  }
  std::unique_ptr<Operator> run() override { return std::make_unique<GpuAggregationOperator>(_segment, _queryContext, std::move(_lowered), _batch); }
 private:
  const ImmutableSegment* _segment;
  QueryContext _queryContext;
  std::unique_ptr<LoweredQuery> _lowered;
  std::shared_ptr<GpuBatch> _batch;       // the batch of the query being planned (executeCombined), if any
};

}  // namespace

// FilterPlanNode.run() of the query's WHERE clause over this segment, as text (no device involved)
std::string explainFilter(const ImmutableSegment& seg, const QueryContext& qc) {
  if (!qc.hasFilter) return "MATCH_ALL";
  LoweredQuery lq;
  lq.setWords.reserve(64);
  lq.predicates.reserve(256);
  const PhysNode root = lowerFilter(qc.filter, seg, qc.nullHandlingEnabled, &lq);
  return explainPhysical(root, lq, seg);
}

// ---------------------------------------------------------------------------------------------------------------
// GpuPlanMaker
// ---------------------------------------------------------------------------------------------------------------
void GpuPlanMaker::init(const std::map<std::string, std::string>& cfg) {
  pg_config c;
  memset(&c, 0, sizeof(c));
  c.abi_version = PG_ABI_VERSION;
  auto it = cfg.find(kConfigDevice);
  _device = it == cfg.end() ? 0 : atoi(it->second.c_str());
  c.device_id = _device;
  it = cfg.find(kConfigDevices);
  setDevices(it == cfg.end() ? std::vector<int>{_device} : parseDevices(it->second));
  c.device_id = _device = _devices[0];      // pg_init's device is only the default of segments that name none: the library switches to a segment's device in every call
  it = cfg.find(kConfigTimeKernels);
  if (it != cfg.end() && it->second == "true") c.flags |= PG_CFG_TIME_KERNELS;
  it = cfg.find(kConfigBatch);
  _batch = it == cfg.end() || it->second != "false";
  it = cfg.find(kConfigExactFilterStats);
  _exactFilterStats = it == cfg.end() || it->second != "false";
  checkStatus(gpuAbi().init(&c), "initialising the GPU plan maker");
}

std::vector<int> GpuPlanMaker::parseDevices(const std::string& text) {
  std::vector<int> out;
  size_t pos = 0;
  auto number = [](const std::string& t) {
    size_t used = 0;
    const int v = std::stoi(t, &used);
    while (used < t.size() && isspace((unsigned char)t[used])) ++used;
    if (used != t.size() || v < 0) throw std::invalid_argument("bad device number: " + t);
    return v;
  };
  while (pos <= text.size()) {
    const size_t comma = std::min(text.find(',', pos), text.size());
    std::string item = text.substr(pos, comma - pos);
    pos = comma + 1;
    const size_t b = item.find_first_not_of(" \t"), e = item.find_last_not_of(" \t");
    if (b == std::string::npos) continue;
    item = item.substr(b, e - b + 1);
    const size_t dash = item.find('-', 1);
    const int first = number(dash == std::string::npos ? item : item.substr(0, dash));
    const int last = dash == std::string::npos ? first : number(item.substr(dash + 1));
    if (last < first) throw std::invalid_argument("bad device range: " + item);
    for (int d = first; d <= last; ++d) if (std::find(out.begin(), out.end(), d) == out.end()) out.push_back(d);
  }
  if (out.empty()) throw std::invalid_argument("no device in: " + text);
  return out;
}

int GpuPlanMaker::placeSegment(long long bytes) {
  std::lock_guard<std::mutex> lk(_placementMu);
  size_t best = 0;
  for (size_t i = 1; i < _devices.size(); ++i) if (_residentBytes[i] < _residentBytes[best]) best = i;
  _residentBytes[best] += std::max<long long>(bytes, 0);
  return _devices[best];
}

void GpuPlanMaker::releaseSegment(int device, long long bytes) {
  std::lock_guard<std::mutex> lk(_placementMu);
  for (size_t i = 0; i < _devices.size(); ++i) if (_devices[i] == device) { _residentBytes[i] = std::max<long long>(0, _residentBytes[i] - std::max<long long>(bytes, 0)); return; }
}

// FilteredAggregationOperator (core/operator/query/FilteredAggregationOperator.java:68-110; lanes built by
// AggregationFunctionUtils.buildFilteredAggregationInfos): the aggregations are split into one "swim lane" per distinct FILTER clause
// (plus one for the unfiltered ones), each lane runs over main-filter AND lane-filter, the results go back to their positions and
// the execution statistics of the lanes are added up.  Every lane is one pg_execute.
class GpuFilteredAggregationOperator : public Operator {
 public:
  struct Lane { std::unique_ptr<Operator> op; std::vector<int> positions; };
  GpuFilteredAggregationOperator(const ImmutableSegment* seg, QueryContext qc, std::vector<Lane> lanes)
      : _segment(seg), _queryContext(std::move(qc)), _lanes(std::move(lanes)) {}

  ResultsBlock nextBlock() override {
    if (!_queryContext.groupByExpressions.empty()) return nextGroupByBlock();
    ResultsBlock block;
    block.isGroupBy = false;
    for (const auto& a : _queryContext.aggregations) block.aggregation.functions.emplace_back(a.function, a.column, _queryContext.nullHandlingEnabled);
    block.aggregation.results.resize(_queryContext.aggregations.size());
    for (auto& lane : _lanes) {
      ResultsBlock b = lane.op->nextBlock();
      for (size_t i = 0; i < lane.positions.size(); ++i) block.aggregation.results[(size_t)lane.positions[i]] = b.aggregation.results[i];
      block.stats.numDocsScanned += b.stats.numDocsScanned;
      block.stats.numEntriesScannedInFilter += b.stats.numEntriesScannedInFilter;
      block.stats.numEntriesScannedPostFilter += b.stats.numEntriesScannedPostFilter;
      block.stats.numTotalDocs = b.stats.numTotalDocs;
      block.deviceMs += b.deviceMs;
      block.kernelMs += b.kernelMs;
    }
    _stats = block.stats;
    return block;
  }
  // FilteredGroupByOperator (core/operator/query/FilteredGroupByOperator.java:108-150): the lanes share one group key generator, so
  // the result holds every group some lane saw; a function whose lane never saw a group keeps its holder's default there
  // (COUNT 0, SUM 0.0, MIN +inf, MAX -inf, AVG (0, 0)).  Raw group ids are the same in every lane: rows are merged on them.
  ResultsBlock nextGroupByBlock() {
    ResultsBlock block;
    block.isGroupBy = true;
    GroupByResultsBlock& g = block.groupBy;
    g.groupByColumns = _queryContext.groupByExpressions;
    for (const auto& c : g.groupByColumns) g.groupByTypes.push_back(_segment->getDataSource(c).dataType);
    for (const auto& a : _queryContext.aggregations) g.functions.emplace_back(a.function, a.column, _queryContext.nullHandlingEnabled);
    pg_agg_value empty;
    memset(&empty, 0, sizeof(empty));
    empty.min = INFINITY; empty.max = -INFINITY;
    std::vector<IntermediateResult> defaults;
    for (const auto& f : g.functions) defaults.push_back(AggregationFunction(f.getType(), f.getColumn(), _queryContext.nullHandlingEnabled).fromDevice(empty));
    std::map<std::vector<int32_t>, size_t> rowOf;         // the key's dictIds -> row (the same in every lane, whichever holder the key space calls for)
    for (auto& lane : _lanes) {
      ResultsBlock b = lane.op->nextBlock();
      for (size_t r = 0; r < b.groupBy.groupKeys.size(); ++r) {
        const GroupKey& key = b.groupBy.groupKeys[r];
        auto it = rowOf.find(key.dictIds);
        if (it == rowOf.end()) {
          it = rowOf.emplace(key.dictIds, g.groupKeys.size()).first;
          g.groupKeys.push_back(key);
          g.results.push_back(defaults);
        }
        for (size_t i = 0; i < lane.positions.size(); ++i) g.results[it->second][(size_t)lane.positions[i]] = b.groupBy.results[r][i];
      }
      block.stats.numDocsScanned += b.stats.numDocsScanned;
      block.stats.numEntriesScannedInFilter += b.stats.numEntriesScannedInFilter;
      block.stats.numEntriesScannedPostFilter += b.stats.numEntriesScannedPostFilter;
      block.stats.numTotalDocs = b.stats.numTotalDocs;
      block.numGroupsLimitReached = block.numGroupsLimitReached || b.numGroupsLimitReached;
      block.deviceMs += b.deviceMs;
      block.kernelMs += b.kernelMs;
    }
    // ascending raw group id, like every other group-by block of this path
    std::vector<size_t> order(g.groupKeys.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    // ascending raw key: the last key column is the most significant digit
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b2) {
      const auto& x = g.groupKeys[a].dictIds; const auto& y = g.groupKeys[b2].dictIds;
      for (size_t j = x.size(); j-- > 0;) if (x[j] != y[j]) return x[j] < y[j];
      return false;
    });
    std::vector<GroupKey> keys;
    std::vector<std::vector<IntermediateResult>> rows;
    for (size_t i : order) { keys.push_back(std::move(g.groupKeys[i])); rows.push_back(std::move(g.results[i])); }
    g.groupKeys = std::move(keys);
    g.results = std::move(rows);
    _stats = block.stats;
    trimSegmentGroupByBlock(&block, _queryContext);
    return block;
  }
  std::string toExplainString() const override { return _queryContext.groupByExpressions.empty() ? "GPU_AGGREGATE_FILTERED" : "GPU_GROUP_BY_FILTERED"; }
  ExecutionStatistics getExecutionStatistics() const override { return _stats; }
  const ImmutableSegment* getIndexSegment() const override { return _segment; }

 private:
  const ImmutableSegment* _segment;
  QueryContext _queryContext;
  std::vector<Lane> _lanes;
  ExecutionStatistics _stats;
};

class GpuFilteredAggregationPlanNode : public PlanNode {
 public:
  GpuFilteredAggregationPlanNode(const ImmutableSegment* seg, QueryContext qc, std::vector<std::unique_ptr<PlanNode>> lanePlans, std::vector<std::vector<int>> positions)
      : _segment(seg), _queryContext(std::move(qc)), _lanePlans(std::move(lanePlans)), _positions(std::move(positions)) {}
  std::unique_ptr<Operator> run() override {
    std::vector<GpuFilteredAggregationOperator::Lane> lanes;
    for (size_t i = 0; i < _lanePlans.size(); ++i) lanes.push_back({_lanePlans[i]->run(), _positions[i]});
    return std::make_unique<GpuFilteredAggregationOperator>(_segment, _queryContext, std::move(lanes));
  }
 private:
  const ImmutableSegment* _segment;
  QueryContext _queryContext;
  std::vector<std::unique_ptr<PlanNode>> _lanePlans;
  std::vector<std::vector<int>> _positions;
};

std::unique_ptr<PlanNode> GpuPlanMaker::makeSegmentPlanNode(const SegmentContext& sc, const QueryContext& qc) {
  const ImmutableSegment* seg = sc.indexSegment;
  if (!seg || !seg->handle()) throw std::runtime_error("segment is not loaded on a device");
  if (qc.aggregations.empty()) throw UnsupportedOperationException("only aggregation / group-by queries are offloaded (selection stays on the CPU plan)");
  if (!_exactFilterStats && qc.gpuExactFilterStats) {      // the server's setting: every lane of the query is lowered with PG_QUERY_STATS_UPPER_BOUND_OK
    QueryContext relaxed = qc;
    relaxed.gpuExactFilterStats = false;
    return makeSegmentPlanNode(sc, relaxed);
  }
  // Anything the device cannot run is rejected HERE, at plan time, never at run time.
  bool anyFiltered = false;
  for (const auto& a : qc.aggregations) anyFiltered |= a.hasFilter;
  if (!anyFiltered) return std::make_unique<GpuAggregationPlanNode>(seg, qc, lowerQuery(*seg, qc));
  std::vector<std::string> keys;                       // lane order = first appearance, the unfiltered lane keyed ""
  std::vector<QueryContext> laneQueries;
  std::vector<std::vector<int>> positions;
  for (size_t i = 0; i < qc.aggregations.size(); ++i) {
    const AggregationExpression& a = qc.aggregations[i];
    const std::string key = a.hasFilter ? a.filterText : std::string();
    size_t lane = std::find(keys.begin(), keys.end(), key) - keys.begin();
    if (lane == keys.size()) {
      keys.push_back(key);
      QueryContext lq = qc;
      lq.aggregations.clear();
      lq.orderByExpressions.clear();                   // the lanes return every group; the operator above them orders and trims
      if (a.hasFilter) {
        if (qc.hasFilter) {
          FilterContext both;
          both.type = FilterContext::Type::AND;
          both.children = {qc.filter, a.filter};
          lq.filter = both;
        } else {
          lq.filter = a.filter;
        }
        lq.hasFilter = true;
      }
      laneQueries.push_back(std::move(lq));
      positions.emplace_back();
    }
    AggregationExpression plain = a;
    plain.hasFilter = false;
    laneQueries[lane].aggregations.push_back(plain);
    positions[lane].push_back((int)i);
  }
  std::vector<std::unique_ptr<PlanNode>> lanePlans;
  for (const auto& lq : laneQueries) lanePlans.push_back(std::make_unique<GpuAggregationPlanNode>(seg, lq, lowerQuery(*seg, lq)));
  return std::make_unique<GpuFilteredAggregationPlanNode>(seg, qc, std::move(lanePlans), std::move(positions));
}

// AggregationResultsBlockMerger.mergeResultsBlocks (:34-44) and GroupByCombineOperator (:132-147, keyed by VALUES).
void mergeResultsBlocks(ResultsBlock* merged, const ResultsBlock& other) {
  merged->stats.merge(other.stats);
  merged->numGroupsLimitReached = merged->numGroupsLimitReached || other.numGroupsLimitReached;   // GroupByCombineOperator: any segment
  merged->deviceMs = std::max(merged->deviceMs, other.deviceMs);
  merged->kernelMs = std::max(merged->kernelMs, other.kernelMs);
  if (!merged->isGroupBy) {
    auto& m = merged->aggregation;
    for (size_t a = 0; a < m.functions.size(); ++a) m.results[a] = m.functions[a].merge(m.results[a], other.aggregation.results[a]);
    return;
  }
  auto& m = merged->groupBy;
  std::map<std::vector<GroupKeyValue>, size_t> index;
  for (size_t i = 0; i < m.groupKeys.size(); ++i) index[m.groupKeys[i].keys] = i;
  for (size_t i = 0; i < other.groupBy.groupKeys.size(); ++i) {
    const auto& keys = other.groupBy.groupKeys[i].keys;
    auto it = index.find(keys);
    if (it == index.end()) {
      index[keys] = m.groupKeys.size();
      GroupKey k = other.groupBy.groupKeys[i];
      k.groupId = (int)m.groupKeys.size();           // group ids are segment-local; after the merge they are just ordinals
      m.groupKeys.push_back(std::move(k));
      m.results.push_back(other.groupBy.results[i]);
    } else {
      for (size_t a = 0; a < m.functions.size(); ++a) m.results[it->second][a] = m.functions[a].merge(m.results[it->second][a], other.groupBy.results[i][a]);
    }
  }
}

ResultsBlock GpuPlanMaker::executeCombined(const std::vector<SegmentContext>& segments, const QueryContext& qc, int maxExecutionThreads) {
  if (segments.empty()) throw QueryException("no segments");
  // CombinePlanNode: plan every segment first (plan-time rejection), then BaseCombineOperator: one task per segment,
  // numTasks = min(numSegments, maxExecutionThreads) (QueryMultiThreadingUtils.java:46-65).
  std::vector<std::unique_ptr<Operator>> operators;
  {
    // every device operator planned below (FILTER (WHERE) lanes included) joins ONE pg_execute_batch (GpuBatch)
    struct Planning { Planning(bool on) { if (on) t_planningBatch = std::make_shared<GpuBatch>(); } ~Planning() { t_planningBatch.reset(); } } planning(_batch && segments.size() > 1);
    for (const auto& sc : segments) operators.push_back(makeSegmentPlanNode(sc, qc)->run());
  }
  std::vector<ResultsBlock> blocks(operators.size());
  std::vector<std::string> errors(operators.size());
  std::vector<int> errorKinds(operators.size(), 0);       // 1 UnsupportedOperationException, 2 QueryException, 3 anything else
  const int numTasks = std::max(1, std::min((int)operators.size(), maxExecutionThreads > 0 ? maxExecutionThreads : (int)operators.size()));
  std::vector<std::thread> workers;
  for (int t = 0; t < numTasks; ++t) {
    workers.emplace_back([&, t] {
      for (size_t i = (size_t)t; i < operators.size(); i += (size_t)numTasks) {
        try { blocks[i] = operators[i]->nextBlock(); }
        catch (const std::exception& e) {   // wrapOperatorException: attach the segment name
          errorKinds[i] = dynamic_cast<const UnsupportedOperationException*>(&e) ? 1 : (dynamic_cast<const QueryException*>(&e) ? 2 : 3);
          errors[i] = std::string("Caught exception while doing operator: ") + operators[i]->toExplainString() + " on segment " +
                      operators[i]->getIndexSegment()->getSegmentName() + ": " + e.what();
        }
      }
    });
  }
  for (auto& w : workers) w.join();
  for (size_t i = 0; i < errors.size(); ++i) {
    if (errors[i].empty()) continue;
    if (errorKinds[i] == 1) throw UnsupportedOperationException(errors[i]);      // the caller's fallback signal keeps its class
    if (errorKinds[i] == 2) throw QueryException(errors[i]);
    throw std::runtime_error(errors[i]);
  }
  // deterministic merge order (segment order); for integer sums below 2^53 any order gives the same doubles
  if (!blocks[0].isGroupBy) {
    ResultsBlock merged = blocks[0];
    for (size_t i = 1; i < blocks.size(); ++i) mergeResultsBlocks(&merged, blocks[i]);
    return merged;
  }
  return combineGroupByBlocks(blocks, qc);
}

// GroupByCombineOperator.processSegments + mergeResults (:102-162,186-222): every segment's groups are upserted into one IndexedTable
// keyed by the key VALUES; the table decides which groups survive (LIMIT without ORDER BY: the first LIMIT keys; ORDER BY: trims to
// max(5 * LIMIT, minServerGroupTrimSize) whenever groupTrimThreshold records pile up, and once more at the end).
ResultsBlock combineGroupByBlocks(const std::vector<ResultsBlock>& blocks, const QueryContext& qc) {
  if (blocks.empty() || !blocks[0].isGroupBy) throw QueryException("combineGroupByBlocks needs group-by blocks");
  ResultsBlock merged;
  merged.isGroupBy = true;
  GroupByResultsBlock& m = merged.groupBy;
  m.groupByColumns = blocks[0].groupBy.groupByColumns;
  m.groupByTypes = blocks[0].groupBy.groupByTypes;
  m.functions = blocks[0].groupBy.functions;
  IndexedTable table = IndexedTable::forCombineOperator(m.functions, qc);
  for (const ResultsBlock& b : blocks) {
    merged.stats.merge(b.stats);
    merged.numGroupsLimitReached = merged.numGroupsLimitReached || b.numGroupsLimitReached;      // any segment
    merged.deviceMs = std::max(merged.deviceMs, b.deviceMs);
    merged.kernelMs = std::max(merged.kernelMs, b.kernelMs);
    for (size_t i = 0; i < b.groupBy.groupKeys.size(); ++i) table.upsert(Record{b.groupBy.groupKeys[i].keys, b.groupBy.results[i]});
  }
  table.finish(false);
  for (const Record& r : table.records()) {
    m.groupKeys.push_back(GroupKey{(int)m.groupKeys.size(), r.keys});      // group ids are segment-local; after the merge they are ordinals
    m.results.push_back(r.values);
  }
  return merged;
}

}  // namespace pinot
