// The group-by table behind the combine operator and the broker's reducer: which groups survive ORDER BY / LIMIT and the trims.
//
// Mirrors pinot-core/src/main/java/org/apache/pinot/core/data/table/{IndexedTable,SimpleIndexedTable,TableResizer}.java and
// core/util/GroupByUtils.java for the slice this path produces (group-by columns and COUNT / SUM / MIN / MAX / AVG in ORDER BY).
// Where the reference leaves the outcome to HashMap iteration order (which of several records that compare equal is evicted),
// this table is deterministic: records keep their insertion order and ties go to the earlier record.
#include <climits>
#include <cmath>
#include <cstring>

#include "pinot_host.h"

namespace pinot {

// ---- GroupByUtils.java:48-73 ---------------------------------------------------------------------------------------
int GroupByUtils::getTableCapacity(int limit, int minNumGroups) {
  const long long capacityByLimit = (long long)limit * 5ll;
  return capacityByLimit > INT_MAX ? INT_MAX : std::max((int)capacityByLimit, minNumGroups);
}

int GroupByUtils::getIndexedTableTrimThreshold(int trimSize, int trimThreshold) {
  // at least 2 * trimSize to avoid excessive trimming; non-positive or above 10^9: trim disabled
  if (trimThreshold <= 0 || trimThreshold > MAX_TRIM_THRESHOLD || trimSize > MAX_TRIM_THRESHOLD / 2) return INT_MAX;
  return std::max(trimThreshold, 2 * trimSize);
}

// ---- TableResizer --------------------------------------------------------------------------------------------------
TableResizer::TableResizer(const std::vector<AggregationFunction>& functions, const QueryContext& queryContext)
    : _functions(functions), _orderBy(queryContext.orderByExpressions), _nullHandlingEnabled(queryContext.nullHandlingEnabled) {}

std::vector<OrderByValue> TableResizer::orderByValues(const Record& r) const {
  std::vector<OrderByValue> out;
  out.reserve(_orderBy.size());
  for (const auto& ob : _orderBy) {
    if (!ob.isAggregation) {                          // GroupByExpressionExtractor
      const GroupKeyValue& k = r.keys.at((size_t)ob.index);
      if (std::holds_alternative<int64_t>(k)) out.emplace_back(std::get<int64_t>(k));
      else if (std::holds_alternative<double>(k)) out.emplace_back(std::get<double>(k));
      else if (std::holds_alternative<std::string>(k)) out.emplace_back(std::get<std::string>(k));
      else out.emplace_back(std::monostate{});
    } else {                                          // AggregationFunctionExtractor: the FINAL result is what gets compared
      const IntermediateResult& v = r.values.at((size_t)ob.index);
      const AggregationFunction& f = _functions.at((size_t)ob.index);
      if (isNullResult(v)) out.emplace_back(std::monostate{});
      else if (f.getType() == AggregationFunctionType::COUNT) out.emplace_back(std::get<int64_t>(v));
      else out.emplace_back(f.extractFinalResult(v));
    }
  }
  return out;
}

namespace {
// Comparable.compareTo of two values of one column: Long.compare, Double.compare (NaN above everything, -0.0 below 0.0),
// String.compareTo (UTF-16 code units; the same order as bytes for ASCII keys)
int compareValues(const OrderByValue& a, const OrderByValue& b) {
  if (std::holds_alternative<int64_t>(a) && std::holds_alternative<int64_t>(b)) {
    const int64_t x = std::get<int64_t>(a), y = std::get<int64_t>(b);
    return x < y ? -1 : (x > y ? 1 : 0);
  }
  if (std::holds_alternative<std::string>(a) && std::holds_alternative<std::string>(b)) {
    const int c = std::get<std::string>(a).compare(std::get<std::string>(b));
    return c < 0 ? -1 : (c > 0 ? 1 : 0);
  }
  auto asDouble = [](const OrderByValue& v) { return std::holds_alternative<double>(v) ? std::get<double>(v) : (double)std::get<int64_t>(v); };
  const double x = asDouble(a), y = asDouble(b);
  if (x < y) return -1;
  if (x > y) return 1;
  // Double.compare: equal or unordered -> by doubleToLongBits as signed longs (one canonical NaN above everything, -0.0 below 0.0)
  auto bits = [](double v) { int64_t b; if (v != v) return (int64_t)0x7ff8000000000000ll; memcpy(&b, &v, 8); return b; };
  const int64_t xb = bits(x), yb = bits(y);
  return xb < yb ? -1 : (xb > yb ? 1 : 0);
}
}  // namespace

int TableResizer::compare(const std::vector<OrderByValue>& a, const std::vector<OrderByValue>& b) const {
  for (size_t i = 0; i < _orderBy.size(); ++i) {
    const bool n1 = std::holds_alternative<std::monostate>(a[i]), n2 = std::holds_alternative<std::monostate>(b[i]);
    if (n1 || n2) {
      // TableResizer.java:98-118 (the null-aware comparator is only installed under null handling; without it no value is null)
      if (n1 && n2) continue;
      const int nullComparisonResult = _orderBy[i].isNullsLast() ? -1 : 1;
      return n1 ? -nullComparisonResult : nullComparisonResult;
    }
    int c = compareValues(a[i], b[i]);
    if (!_orderBy[i].isAsc) c = -c;
    if (c != 0) return c;
  }
  return 0;
}

std::vector<Record> TableResizer::topRecords(std::vector<Record> records, size_t size, bool sort) const {
  if (records.size() <= size && !sort) return records;
  struct Entry { std::vector<OrderByValue> values; size_t at; };
  std::vector<Entry> entries(records.size());
  for (size_t i = 0; i < records.size(); ++i) entries[i] = Entry{orderByValues(records[i]), i};
  auto before = [&](const Entry& x, const Entry& y) {
    const int c = compare(x.values, y.values);
    return c != 0 ? c < 0 : x.at < y.at;
  };
  const size_t keep = std::min(size, entries.size());
  if (sort) std::partial_sort(entries.begin(), entries.begin() + (long)keep, entries.end(), before);
  else if (keep < entries.size()) {
    std::nth_element(entries.begin(), entries.begin() + (long)keep, entries.end(), before);
    std::sort(entries.begin(), entries.begin() + (long)keep, [](const Entry& x, const Entry& y) { return x.at < y.at; });   // keep insertion order
  }
  std::vector<Record> out;
  out.reserve(keep);
  for (size_t i = 0; i < keep; ++i) out.push_back(std::move(records[entries[i].at]));
  return out;
}

// ---- IndexedTable ---------------------------------------------------------------------------------------------------
IndexedTable::IndexedTable(std::vector<AggregationFunction> functions, const QueryContext& queryContext, int resultSize, int trimSize, int trimThreshold)
    : _functions(std::move(functions)), _hasOrderBy(queryContext.hasOrderBy()), _resizer(_functions, queryContext), _resultSize(resultSize),
      _trimSize(trimSize), _trimThreshold(trimThreshold) {
  if (resultSize < 0 || trimSize < 0 || trimThreshold < 0) throw QueryException("Result size, trim size and trim threshold can't be negative");
  // trim is disabled when there is no ORDER BY (IndexedTable.java:84-85)
  if (!_hasOrderBy) { _trimSize = INT_MAX; _trimThreshold = INT_MAX; }
}

IndexedTable IndexedTable::forCombineOperator(std::vector<AggregationFunction> functions, const QueryContext& qc) {
  const int limit = qc.getLimit();
  const int trimSize = qc.minServerGroupTrimSize > 0 ? GroupByUtils::getTableCapacity(limit, qc.minServerGroupTrimSize) : INT_MAX;
  // no ORDER BY: the table stops accepting new groups once LIMIT of them exist, nothing is trimmed (GroupByUtils.java:108-122; no HAVING here)
  if (!qc.hasOrderBy()) return IndexedTable(std::move(functions), qc, limit, INT_MAX, INT_MAX);
  // ORDER BY: the server keeps trimSize groups for the broker (it does not return final results on this path)
  const int trimThreshold = GroupByUtils::getIndexedTableTrimThreshold(trimSize, qc.groupTrimThreshold);
  return IndexedTable(std::move(functions), qc, trimSize, trimThreshold == INT_MAX ? INT_MAX : trimSize, trimThreshold);
}

IndexedTable IndexedTable::forDataTableReducer(std::vector<AggregationFunction> functions, const QueryContext& qc) {
  const int limit = qc.getLimit();
  // the broker's minGroupTrimSize / groupByTrimThreshold defaults are the server's (BrokerReduceService -> DataTableReducerContext)
  const int trimSize = qc.minServerGroupTrimSize > 0 ? GroupByUtils::getTableCapacity(limit, qc.minServerGroupTrimSize) : INT_MAX;
  if (!qc.hasOrderBy()) return IndexedTable(std::move(functions), qc, limit, INT_MAX, INT_MAX);
  const int trimThreshold = GroupByUtils::getIndexedTableTrimThreshold(trimSize, qc.groupTrimThreshold);
  return IndexedTable(std::move(functions), qc, limit, trimThreshold == INT_MAX ? INT_MAX : trimSize, trimThreshold);
}

bool IndexedTable::upsert(const Record& record) {
  if (_finished) throw std::runtime_error("IndexedTable: upsert after finish");
  auto it = _lookup.find(record.keys);
  if (it != _lookup.end()) {                          // updateRecord: merge every aggregation into the existing record
    Record& existing = _records[it->second];
    for (size_t a = 0; a < _functions.size(); ++a) existing.values[a] = _functions[a].merge(existing.values[a], record.values[a]);
    return true;
  }
  if (_hasOrderBy) {
    _lookup.emplace(record.keys, _records.size());
    _records.push_back(record);
    if (_records.size() >= (size_t)_trimThreshold) resize();
  } else if (_records.size() < (size_t)_resultSize) {
    _lookup.emplace(record.keys, _records.size());
    _records.push_back(record);
  }                                                   // else: a new key past the result size is ignored (updateExistingRecord)
  return true;
}

void IndexedTable::resize() {
  _records = _resizer.topRecords(std::move(_records), (size_t)_trimSize, false);
  _lookup.clear();
  for (size_t i = 0; i < _records.size(); ++i) _lookup.emplace(_records[i].keys, i);
  _numResizes++;
}

void IndexedTable::finish(bool sort) {
  if (_hasOrderBy) { _topRecords = _resizer.topRecords(_records, (size_t)_resultSize, sort); _numResizes++; }
  else _topRecords = _records;
  _finished = true;
}

// ---- GroupByOperator's in-segment trim -------------------------------------------------------------------------------
void trimSegmentGroupByBlock(ResultsBlock* block, const QueryContext& qc) {
  if (!block->isGroupBy || !qc.hasOrderBy() || qc.minSegmentGroupTrimSize <= 0) return;
  const int trimSize = GroupByUtils::getTableCapacity(qc.getLimit(), qc.minSegmentGroupTrimSize);
  GroupByResultsBlock& g = block->groupBy;
  if ((long long)g.groupKeys.size() <= (long long)trimSize) return;
  std::vector<Record> records(g.groupKeys.size());
  for (size_t i = 0; i < records.size(); ++i) records[i] = Record{g.groupKeys[i].keys, g.results[i]};
  const TableResizer resizer(g.functions, qc);
  records = resizer.topRecords(std::move(records), (size_t)trimSize, false);
  g.groupKeys.clear(); g.results.clear();
  for (size_t i = 0; i < records.size(); ++i) {
    g.groupKeys.push_back(GroupKey{(int)i, std::move(records[i].keys)});
    g.results.push_back(std::move(records[i].values));
  }
}

// ---- the broker's reduce ----------------------------------------------------------------------------------------------
std::vector<ReducedRow> reduceGroupBy(const ResultsBlock& combined, const QueryContext& qc) {
  if (!combined.isGroupBy) throw QueryException("reduceGroupBy needs a group-by block");
  const GroupByResultsBlock& g = combined.groupBy;
  IndexedTable table = IndexedTable::forDataTableReducer(g.functions, qc);
  for (size_t i = 0; i < g.groupKeys.size(); ++i) table.upsert(Record{g.groupKeys[i].keys, g.results[i]});
  table.finish(true);
  std::vector<ReducedRow> rows;
  const size_t limit = (size_t)qc.getLimit();
  for (const Record& r : table.records()) {
    if (rows.size() >= limit) break;
    ReducedRow row;
    row.keys = r.keys;
    for (size_t a = 0; a < g.functions.size(); ++a) {
      if (isNullResult(r.values[a])) row.finals.emplace_back(std::monostate{});
      else if (g.functions[a].getType() == AggregationFunctionType::COUNT) row.finals.emplace_back(std::get<int64_t>(r.values[a]));
      else row.finals.emplace_back(g.functions[a].extractFinalResult(r.values[a]));
    }
    rows.push_back(std::move(row));
  }
  return rows;
}

ResultTable toResultTable(const std::vector<ReducedRow>& rows, const ResultsBlock& combined, const QueryContext& qc) {
  const GroupByResultsBlock& g = combined.groupBy;
  ResultTable t;
  // a query that selects only aggregations shows only them (the group keys are not part of the result: PostAggregationHandler)
  std::vector<SelectExpression> select = qc.selectExpressions;
  for (const auto& se : select)
    t.columnNames.push_back(se.isAggregation ? g.functions.at((size_t)se.index).getResultColumnName() : g.groupByColumns.at((size_t)se.index));
  for (const ReducedRow& r : rows) {
    std::vector<OrderByValue> row;
    for (const auto& se : select) {
      if (se.isAggregation) { row.push_back(r.finals.at((size_t)se.index)); continue; }
      const GroupKeyValue& k = r.keys.at((size_t)se.index);
      if (std::holds_alternative<int64_t>(k)) row.emplace_back(std::get<int64_t>(k));
      else if (std::holds_alternative<double>(k)) row.emplace_back(std::get<double>(k));
      else if (std::holds_alternative<std::string>(k)) row.emplace_back(std::get<std::string>(k));
      else row.emplace_back(std::monostate{});
    }
    t.rows.push_back(std::move(row));
  }
  return t;
}

}  // namespace pinot
