// pinot_host.h -- C++ host mirror of the reference's per-segment operator interface, sitting ABOVE the C ABI.
//
// The reference is Java and no JDK exists in the build environment, so the host side that a Pinot server would
// run in the JVM is mirrored here in C++ with the reference's names, argument meaning and error behaviour.  Every
// class cites the Java type it mirrors (paths under /root/reference/pinot-core/src/main/java/org/apache/pinot/core/
// unless noted).  The only way this layer computes anything is by calling the pg_* C ABI (include/pinot_gpu.h),
// resolved with dlopen from libpinot_gpu.so: there is no CPU execution path here.
#pragma once
#include <algorithm>
#include <cstdint>
#include <map>
#include <mutex>
#include <memory>
#include <stdexcept>
#include <string>
#include <variant>
#include <vector>

#include "../../../include/pinot_gpu.h"

namespace pinot {

// ---- errors -------------------------------------------------------------------------------------------------
struct QueryException : std::runtime_error { using std::runtime_error::runtime_error; };          // BadQueryRequestException
struct UnsupportedOperationException : std::runtime_error { using std::runtime_error::runtime_error; };

// ---- pinot-spi FieldSpec.DataType (stored types on this path) -------------------------------------------------
enum class DataType { INT, LONG, FLOAT, DOUBLE, STRING };
inline bool isNumeric(DataType t) { return t != DataType::STRING; }
const char* dataTypeName(DataType t);

// ---- sspi/index/reader/Dictionary.java:37-301; segl/segment/index/readers/BaseImmutableDictionary.java ----------
class Dictionary {
 public:
  virtual ~Dictionary() = default;
  virtual DataType getValueType() const = 0;
  virtual int length() const = 0;
  virtual bool isSorted() const { return true; }
  // Returns the dictId of the value, or -(insertionIndex + 1) when absent (BaseImmutableDictionary.java:124-140).
  virtual int insertionIndexOf(const std::string& stringValue) const = 0;
  int indexOf(const std::string& stringValue) const { int i = insertionIndexOf(stringValue); return i >= 0 ? i : -1; }
  virtual int32_t getIntValue(int dictId) const = 0;
  virtual int64_t getLongValue(int dictId) const { return (int64_t)getIntValue(dictId); }
  virtual double getDoubleValue(int dictId) const = 0;
  virtual std::string getStringValue(int dictId) const = 0;
};

// segl/segment/index/readers/IntDictionary.java:38-70 over a big-endian buffer.
class IntDictionary : public Dictionary {
 public:
  IntDictionary(const uint8_t* buffer, int length) : _buffer(buffer), _length(length) {}
  DataType getValueType() const override { return DataType::INT; }
  int length() const override { return _length; }
  int insertionIndexOf(const std::string& stringValue) const override;
  int binarySearch(int32_t value) const;
  int32_t getIntValue(int dictId) const override;
  double getDoubleValue(int dictId) const override { return (double)getIntValue(dictId); }
  std::string getStringValue(int dictId) const override { return std::to_string(getIntValue(dictId)); }
 private:
  const uint8_t* _buffer;
  int _length;
};

// segl/segment/index/readers/{Long,Float,Double}Dictionary.java over a big-endian fixed-width buffer;
// insertionIndexOf(String) = binarySearch(Long.parseLong / Float.parseFloat / Double.parseDouble), BaseImmutableDictionary.java:142-195.
class LongDictionary : public Dictionary {
 public:
  LongDictionary(const uint8_t* buffer, int length) : _buffer(buffer), _length(length) {}
  DataType getValueType() const override { return DataType::LONG; }
  int length() const override { return _length; }
  int insertionIndexOf(const std::string& stringValue) const override;
  int32_t getIntValue(int dictId) const override { return (int32_t)getLongValue(dictId); }
  int64_t getLongValue(int dictId) const override;
  double getDoubleValue(int dictId) const override { return (double)getLongValue(dictId); }
  std::string getStringValue(int dictId) const override { return std::to_string(getLongValue(dictId)); }
 private:
  const uint8_t* _buffer;
  int _length;
};
class FloatDictionary : public Dictionary {
 public:
  FloatDictionary(const uint8_t* buffer, int length) : _buffer(buffer), _length(length) {}
  DataType getValueType() const override { return DataType::FLOAT; }
  int length() const override { return _length; }
  int insertionIndexOf(const std::string& stringValue) const override;
  float getFloatValue(int dictId) const;
  int32_t getIntValue(int dictId) const override { return (int32_t)getFloatValue(dictId); }
  int64_t getLongValue(int dictId) const override { return (int64_t)getFloatValue(dictId); }
  double getDoubleValue(int dictId) const override { return (double)getFloatValue(dictId); }
  std::string getStringValue(int dictId) const override;
 private:
  const uint8_t* _buffer;
  int _length;
};
class DoubleDictionary : public Dictionary {
 public:
  DoubleDictionary(const uint8_t* buffer, int length) : _buffer(buffer), _length(length) {}
  DataType getValueType() const override { return DataType::DOUBLE; }
  int length() const override { return _length; }
  int insertionIndexOf(const std::string& stringValue) const override;
  int32_t getIntValue(int dictId) const override { return (int32_t)getDoubleValue(dictId); }
  int64_t getLongValue(int dictId) const override { return (int64_t)getDoubleValue(dictId); }
  double getDoubleValue(int dictId) const override;
  std::string getStringValue(int dictId) const override;
 private:
  const uint8_t* _buffer;
  int _length;
};

// StringDictionary values never reach the device: predicates are lowered to dictIds on the host and group keys are
// mapped back here (segl/segment/index/readers/StringDictionary.java).
class StringDictionary : public Dictionary {
 public:
  // Values in dictId order.  Legacy segments pad strings with '%', and the padded order (what the file is sorted by) can differ
  // from the order of the un-padded values ("lynda 2.0" < "lynda%%%%"): then lookups are linear and range predicates are not offloaded.
  explicit StringDictionary(std::vector<std::string> values) : _values(std::move(values)), _sorted(std::is_sorted(_values.begin(), _values.end())) {}
  bool isSorted() const override { return _sorted; }
  DataType getValueType() const override { return DataType::STRING; }
  int length() const override { return (int)_values.size(); }
  int insertionIndexOf(const std::string& stringValue) const override;
  int32_t getIntValue(int) const override { throw UnsupportedOperationException("getIntValue on STRING dictionary"); }
  double getDoubleValue(int) const override { throw UnsupportedOperationException("getDoubleValue on STRING dictionary"); }
  std::string getStringValue(int dictId) const override { return _values.at((size_t)dictId); }
 private:
  std::vector<std::string> _values;
  bool _sorted;
};

// ---- sspi/datasource/DataSource.java:46-132 + DataSourceMetadata ------------------------------------------------
struct DataSource {
  std::string name;
  DataType dataType = DataType::INT;
  bool hasDictionary = true;
  bool hasInvertedIndex = false;
  int cardinality = 0;
  int bitsPerElement = 0;
  std::shared_ptr<Dictionary> dictionary;     // null for raw columns
  const uint8_t* forwardIndex = nullptr; uint64_t forwardIndexSize = 0;
  const uint8_t* dictionaryBuffer = nullptr; uint64_t dictionaryBufferSize = 0;   // INT dictionaries only
  const uint8_t* invertedIndex = nullptr; uint64_t invertedIndexSize = 0;
  // DataSource.getNullValueVector(): the <column>.bitmap.nullvalue file (one RoaringBitmap of null docIds), absent when there are none
  const uint8_t* nullValueVector = nullptr; uint64_t nullValueVectorSize = 0;
  std::vector<uint8_t> placeholderDictionary;  // STRING columns hand the device a 0..C-1 int dictionary
  // sorted column: SortedIndexReaderImpl's [startDocId, endDocId] per dictId (2 * cardinality ints); predicates become docId ranges
  bool isSorted = false;
  std::vector<int32_t> sortedDocIdRanges;
};

// ---- sspi/IndexSegment.java / ImmutableSegment: the buffers stay caller-owned, the HBM copy is made by load() ----
class ImmutableSegment {
 public:
  ImmutableSegment(std::string name, int totalDocs) : _name(std::move(name)), _totalDocs(totalDocs) {}
  ~ImmutableSegment();
  const std::string& getSegmentName() const { return _name; }
  int getTotalDocs() const { return _totalDocs; }
  void addDataSource(DataSource ds) { _columns.push_back(std::move(ds)); }
  const DataSource& getDataSource(const std::string& column) const;      // throws QueryException for unknown columns
  DataSource& mutableDataSource(const std::string& column) { return _columns.at((size_t)getColumnIndex(column)); }
  int getColumnIndex(const std::string& column) const;
  const std::vector<DataSource>& getDataSources() const { return _columns; }
  void load(int deviceId);       // pg_segment_open
  void destroy();                // IndexSegment.destroy() -> pg_segment_close
  pg_segment* handle() const { return _handle; }
  int deviceId() const { return _deviceId; }
  void keepAlive(std::shared_ptr<std::vector<uint8_t>> buffer) { _owned.push_back(std::move(buffer)); }   // loader-owned index buffers
  std::vector<std::string> notOffloaded;      // "<column>: <reason>" for columns of a loaded directory that stay on the CPU plan
 private:
  std::string _name;
  int _totalDocs;
  std::vector<DataSource> _columns;
  std::vector<std::shared_ptr<std::vector<uint8_t>>> _owned;
  pg_segment* _handle = nullptr;
  int _deviceId = -1;
};

// ImmutableSegmentLoader.load(indexDir, ReadMode) for the single-value numeric / string columns of a v1 or v3 segment directory
// (segment_loader.cpp).  Columns that are not offloaded are reported by name and reason.
std::unique_ptr<ImmutableSegment> loadSegmentDirectory(const std::string& indexDir, std::vector<std::string>* notOffloaded);
extern "C" int32_t ph_num_bits_per_value(int32_t max_value);

// ---- common/request/context: ExpressionContext (identifiers only on this path), predicates, FilterContext -------
struct Predicate {                                   // common/request/context/predicate/Predicate.java
  enum class Type { EQ, NOT_EQ, IN, NOT_IN, RANGE, IS_NULL, IS_NOT_NULL };
  Type type = Type::EQ;
  std::string column;
  std::vector<std::string> values;                  // EQ / NOT_EQ: 1 value; IN / NOT_IN: n values
  // RangePredicate: "*" = UNBOUNDED (RangePredicate.java)
  std::string lowerBound = "*", upperBound = "*";
  bool lowerInclusive = false, upperInclusive = false;
};

struct FilterContext {                               // common/request/context/FilterContext.java
  enum class Type { AND, OR, NOT, PREDICATE };
  Type type = Type::PREDICATE;
  std::vector<FilterContext> children;
  Predicate predicate;
};

enum class AggregationFunctionType { COUNT, SUM, MIN, MAX, AVG };   // sspi/AggregationFunctionType.java

struct AggregationExpression {
  AggregationFunctionType function;
  std::string column;                                // "*" for COUNT(*)
  // SUM(x) FILTER (WHERE ...): FilteredAggregationFunction (core/query/aggregation/function/FilterableAggregationFunction / QueryContext
  // filtered aggregations); evaluated as one "swim lane" per distinct filter like FilteredAggregationOperator
  bool hasFilter = false;
  FilterContext filter;
  std::string filterText;                            // canonical text of `filter`: the lane key
  std::string alias;                                 // SELECT SUM(x) AS v1: ORDER BY may name it
};

// query/request/context/QueryContext.java (the slice this path needs)
// common/request/context/OrderByExpressionContext.java: an ORDER BY item of a group-by query is a group-by column or one of the
// query's aggregations (post-aggregation expressions are not on this path)
struct OrderByExpressionContext {
  bool isAggregation = false;
  int index = 0;                                     // into groupByExpressions, or into aggregations
  bool isAsc = true;
  int nullsLast = -1;                                // -1: not given -> NULLS LAST for ASC, NULLS FIRST for DESC (:54-62)
  bool isNullsLast() const { return nullsLast < 0 ? isAsc : nullsLast != 0; }
};

struct SelectExpression { bool isAggregation; int index; };   // one item of the SELECT list: aggregations[index] or groupByExpressions[index]

struct QueryContext {
  std::string tableName;
  std::vector<SelectExpression> selectExpressions;   // what the result table shows, in SELECT order
  std::vector<AggregationExpression> aggregations;   // the selected aggregations, then those that are only ordered by (QueryContext._aggregationFunctions)
  std::vector<std::string> groupByExpressions;
  bool hasFilter = false;
  FilterContext filter;
  int maxInitialResultHolderCapacity = 10000;        // InstancePlanMakerImplV2.java:69-91 defaults
  int numGroupsLimit = 100000;
  bool nullHandlingEnabled = false;                  // query option enableNullHandling (QueryContext.isNullHandlingEnabled)
  bool gpuExactFilterStats = true;                   // query option gpuExactFilterStats (default true): false = numEntriesScannedInFilter of a leap-frogging
                                                     //   filter may be the upper bound (PG_QUERY_STATS_UPPER_BOUND_OK) -- GpuQueryLowering.java reads the same option
  std::vector<OrderByExpressionContext> orderByExpressions;   // empty = no ORDER BY (getOrderByExpressions() == null)
  int limit = 10;                                    // LIMIT n; the parser's default of 10 rows without a LIMIT clause (CalciteSqlParser / PinotQuery.limit)
  int minSegmentGroupTrimSize = -1;                  // InstancePlanMakerImplV2.java:82-91 defaults; query options of the same names override
  int minServerGroupTrimSize = 5000;
  int groupTrimThreshold = 1000000;
  bool hasOrderBy() const { return !orderByExpressions.empty(); }
  int getLimit() const { return limit; }
};

// QueryContextConverterUtils.getQueryContext(sql) for the SQL subset of this path:
//   [SET enableNullHandling = true;] SELECT agg(col|*) [FILTER (WHERE ...)] [, ...] FROM t
//   [WHERE <AND/OR/NOT tree of =, !=, <>, <, <=, >, >=, BETWEEN, IN, NOT IN, IS NULL, IS NOT NULL>] [GROUP BY c [, ...]]
QueryContext getQueryContext(const std::string& sql);

// ---- operator/filter/predicate: PredicateEvaluator lowering -----------------------------------------------------
struct PredicateEvaluator {                          // operator/filter/predicate/PredicateEvaluator.java
  Predicate::Type predicateType = Predicate::Type::EQ;
  bool alwaysTrue = false, alwaysFalse = false;      // BaseDictionaryBasedPredicateEvaluator.java:69-89
  bool exclusive = false;                            // NOT_EQ / NOT_IN
  bool isRange = false;                              // dictId range [startDictId, endDictId)
  int startDictId = 0, endDictId = 0;
  std::vector<int> matchingDictIds;                  // sorted (IN / NOT_IN inner set)
  bool rawRange = false; int64_t rawLower = 0, rawUpper = 0;   // raw INT columns: inclusive bounds
  int getNumMatchingItems() const;
};
// PredicateEvaluatorProvider.getPredicateEvaluator (operator/filter/predicate/PredicateEvaluatorProvider.java)
PredicateEvaluator getPredicateEvaluator(const Predicate& predicate, const DataSource& dataSource);

// ---- query/aggregation/function ----------------------------------------------------------------------------------
struct AvgPair { double sum = 0.0; int64_t count = 0; };                  // segl/customobject/AvgPair.java:26-45
using IntermediateResult = std::variant<int64_t, double, AvgPair, std::monostate>;   // Long / Double / AvgPair / null (null handling only)
inline bool isNullResult(const IntermediateResult& r) { return std::holds_alternative<std::monostate>(r); }

class AggregationFunction {                         // query/aggregation/function/AggregationFunction.java:42-145
 public:
  AggregationFunction(AggregationFunctionType type, std::string column, bool nullHandlingEnabled = false)
      : _type(type), _column(std::move(column)), _nullHandlingEnabled(nullHandlingEnabled) {}
  AggregationFunctionType getType() const { return _type; }
  const std::string& getColumn() const { return _column; }
  std::string getResultColumnName() const;
  IntermediateResult fromDevice(const pg_agg_value& v) const;            // extractAggregationResult / extractGroupByResult
  IntermediateResult merge(const IntermediateResult& a, const IntermediateResult& b) const;
  double extractFinalResult(const IntermediateResult& r) const;          // COUNT returns the long as a double-exact value
 private:
  AggregationFunctionType _type;
  std::string _column;
  bool _nullHandlingEnabled;                          // NullableSingleInputAggregationFunction._nullHandlingEnabled
};

// ---- operator/ExecutionStatistics.java:25-64 ---------------------------------------------------------------------
struct ExecutionStatistics {
  int64_t numDocsScanned = 0, numEntriesScannedInFilter = 0, numEntriesScannedPostFilter = 0, numTotalDocs = 0;
  void merge(const ExecutionStatistics& o) {
    numDocsScanned += o.numDocsScanned; numEntriesScannedInFilter += o.numEntriesScannedInFilter;
    numEntriesScannedPostFilter += o.numEntriesScannedPostFilter; numTotalDocs += o.numTotalDocs;
  }
};

// ---- operator/blocks/results --------------------------------------------------------------------------------------
using GroupKeyValue = std::variant<int64_t, std::string, double, std::monostate>;   // INT / LONG keys, STRING keys, FLOAT / DOUBLE keys, NULL (null handling)
struct GroupKey { int groupId; std::vector<GroupKeyValue> keys; std::vector<int32_t> dictIds; };        // groupby/GroupKeyGenerator.GroupKey (+ the key's dictIds: what identifies a group of ONE segment whatever the holder, int / long / array keyed)

struct AggregationResultsBlock {                    // operator/blocks/results/AggregationResultsBlock.java:54-59
  std::vector<AggregationFunction> functions;
  std::vector<IntermediateResult> results;
};

struct GroupByResultsBlock {                        // GroupByResultsBlock.java:68-76 + AggregationGroupByResult.java:31-56
  std::vector<std::string> groupByColumns;
  std::vector<DataType> groupByTypes;                                     // data type of every key column (the DataSchema's key column types)
  std::vector<AggregationFunction> functions;
  std::vector<GroupKey> groupKeys;                                       // ascending raw group id (ArrayBasedHolder iterator)
  std::vector<std::vector<IntermediateResult>> results;                   // [group][function]
};

struct ResultsBlock {
  bool isGroupBy = false;
  AggregationResultsBlock aggregation;
  GroupByResultsBlock groupBy;
  ExecutionStatistics stats;
  bool numGroupsLimitReached = false;                 // GroupByResultsBlock.setNumGroupsLimitReached (GroupByOperator.java:114-146)
  double deviceMs = 0.0, kernelMs = 0.0;
};

// ---- common/Operator.java:35-122, operator/BaseOperator.java:39-50 -------------------------------------------------
class Operator {
 public:
  virtual ~Operator() = default;
  virtual ResultsBlock nextBlock() = 0;              // called exactly once
  virtual std::string toExplainString() const = 0;
  virtual ExecutionStatistics getExecutionStatistics() const = 0;
  virtual const ImmutableSegment* getIndexSegment() const = 0;
};

class PlanNode {                                     // plan/PlanNode.java
 public:
  virtual ~PlanNode() = default;
  virtual std::unique_ptr<Operator> run() = 0;
};

struct SegmentContext { ImmutableSegment* indexSegment; };   // sspi/SegmentContext.java

// ---- plan/maker/PlanMaker.java:37-67 ------------------------------------------------------------------------------
class PlanMaker {
 public:
  virtual ~PlanMaker() = default;
  virtual void init(const std::map<std::string, std::string>& queryExecutorConfig) = 0;
  virtual std::unique_ptr<PlanNode> makeSegmentPlanNode(const SegmentContext& segmentContext, const QueryContext& queryContext) = 0;
};

// The drop-in: what `pinot.server.query.executor.plan.maker.class` would name.  Mirrors a subclass of
// InstancePlanMakerImplV2 that overrides makeSegmentPlanNode (plan/maker/InstancePlanMakerImplV2.java:270-289):
// eligible aggregation / group-by queries get a device plan node; anything else throws
// UnsupportedOperationException at PLAN time so the caller keeps the stock CPU plan.
class GpuPlanMaker : public PlanMaker {
 public:
  void init(const std::map<std::string, std::string>& queryExecutorConfig) override;
  std::unique_ptr<PlanNode> makeSegmentPlanNode(const SegmentContext& segmentContext, const QueryContext& queryContext) override;
  // makeInstancePlan + CombinePlanNode: one worker per segment, results merged like
  // AggregationResultsBlockMerger.java:34-44 / GroupByCombineOperator.java:132-147 (keys are VALUES, not dictIds).
  ResultsBlock executeCombined(const std::vector<SegmentContext>& segments, const QueryContext& queryContext, int maxExecutionThreads);
  // pinot.server.query.executor.gpu.batch (default true): the segment operators of one executeCombined share ONE pg_execute_batch
  static constexpr const char* kConfigBatch = "pinot.server.query.executor.gpu.batch";
  // pinot.server.query.executor.gpu.exact.filter.stats (default true): false = every query runs with PG_QUERY_STATS_UPPER_BOUND_OK (the exact
  // numEntriesScannedInFilter of a leap-frogging filter is a pass of its own behind the query: DESIGN.md section 5); per query: SET gpuExactFilterStats = false
  static constexpr const char* kConfigExactFilterStats = "pinot.server.query.executor.gpu.exact.filter.stats";
  bool exactFilterStats() const { return _exactFilterStats; }
  static constexpr const char* kConfigDevice = "pinot.server.query.executor.gpu.device";
  static constexpr const char* kConfigTimeKernels = "pinot.server.query.executor.gpu.time.kernels";
  // pinot.server.query.executor.gpu.devices: "0-7", "0,2,4", "0-3,6" -- the devices ONE server process drives (a Pinot server is one JVM;
  // segment s on device s mod N).  Default: the single device of gpu.device.  Twin of GpuPlanMaker.java / GpuSegmentCache.java.
  static constexpr const char* kConfigDevices = "pinot.server.query.executor.gpu.devices";
  static std::vector<int> parseDevices(const std::string& text);      // throws std::invalid_argument
  const std::vector<int>& devices() const { return _devices; }
  void setDevices(std::vector<int> devices) { _devices = std::move(devices); _residentBytes.assign(_devices.size(), 0); }
  // The device a segment of `bytes` goes to: the one with the fewest resident bytes, the lowest-numbered among equals (for equal-sized
  // segments that is s mod N in open order).  The bytes are booked there; releaseSegment gives them back.
  int placeSegment(long long bytes);
  void releaseSegment(int device, long long bytes);
 private:
  int _device = 0;
  bool _batch = true;
  bool _exactFilterStats = true;
  std::vector<int> _devices{0};
  std::vector<long long> _residentBytes{0};
  std::mutex _placementMu;
};

// DataTable V4 bytes of a results block (host/datatable_v4.cpp): what InstanceResponseBlock.toDataTable().toBytes() hands the broker.
std::vector<uint8_t> toDataTableV4(const ResultsBlock& block, bool nullHandlingEnabled, int numSegmentsProcessed, int numSegmentsMatched);

// Merge helpers (operator/combine/merger/AggregationResultsBlockMerger.java, combine/GroupByCombineOperator.java)
void mergeResultsBlocks(ResultsBlock* merged, const ResultsBlock& toMerge);

// ---- core/util/GroupByUtils.java:38-73 ----------------------------------------------------------------------------
struct GroupByUtils {
  static constexpr int DEFAULT_MIN_NUM_GROUPS = 5000;
  static constexpr int MAX_TRIM_THRESHOLD = 1000000000;
  static int getTableCapacity(int limit, int minNumGroups = DEFAULT_MIN_NUM_GROUPS);      // max(limit * 5, minNumGroups), saturating
  static int getIndexedTableTrimThreshold(int trimSize, int trimThreshold);              // Integer.MAX_VALUE = trim disabled
};

// ---- core/data/table/{Record,Key,IndexedTable,SimpleIndexedTable,TableResizer}.java --------------------------------
struct Record {                                       // key columns in front, then one intermediate result per aggregation
  std::vector<GroupKeyValue> keys;
  std::vector<IntermediateResult> values;
};

// An ORDER BY value: what TableResizer's extractors hand the comparators -- a key value, or an aggregation's FINAL result
// (AggregationFunctionExtractor.extract = extractFinalResult: Long for COUNT, Double for the rest), null under null handling.
using OrderByValue = std::variant<std::monostate, int64_t, double, std::string>;

class TableResizer {                                  // core/data/table/TableResizer.java:60-330
 public:
  TableResizer(const std::vector<AggregationFunction>& functions, const QueryContext& queryContext);
  std::vector<OrderByValue> orderByValues(const Record& r) const;                       // getIntermediateRecord
  int compare(const std::vector<OrderByValue>& a, const std::vector<OrderByValue>& b) const;     // _intermediateRecordComparator
  // the `size` records that sort first (resizeRecordsMap / getTopRecords); sorted when `sort`
  std::vector<Record> topRecords(std::vector<Record> records, size_t size, bool sort) const;
 private:
  std::vector<AggregationFunction> _functions;
  std::vector<OrderByExpressionContext> _orderBy;
  bool _nullHandlingEnabled;
};

// SimpleIndexedTable: the single-threaded table of GroupByCombineOperator / GroupByDataTableReducer.  Records are merged in the order
// they are upserted (segment order here: deterministic where the reference's concurrent table depends on thread timing).
class IndexedTable {
 public:
  IndexedTable(std::vector<AggregationFunction> functions, const QueryContext& queryContext, int resultSize, int trimSize, int trimThreshold);
  // GroupByUtils.createIndexedTableForCombineOperator (:95-140): sizes from LIMIT, minServerGroupTrimSize, groupTrimThreshold
  static IndexedTable forCombineOperator(std::vector<AggregationFunction> functions, const QueryContext& queryContext);
  // GroupByUtils.createIndexedTableForDataTableReducer (:145-175): the broker's table (resultSize = LIMIT)
  static IndexedTable forDataTableReducer(std::vector<AggregationFunction> functions, const QueryContext& queryContext);
  bool upsert(const Record& record);                  // SimpleIndexedTable.upsert :47-62
  void finish(bool sort);                             // IndexedTable.finish :143-170 (intermediate results are kept; finals are extracted by the caller)
  size_t size() const { return _finished ? _topRecords.size() : _records.size(); }
  const std::vector<Record>& records() const { return _finished ? _topRecords : _records; }
  int getNumResizes() const { return _numResizes; }
  int resultSize() const { return _resultSize; }
  int trimSize() const { return _trimSize; }
  int trimThreshold() const { return _trimThreshold; }
 private:
  void resize();
  std::vector<AggregationFunction> _functions;
  bool _hasOrderBy;
  TableResizer _resizer;
  int _resultSize, _trimSize, _trimThreshold;
  std::map<std::vector<GroupKeyValue>, size_t> _lookup;      // key -> index into _records
  std::vector<Record> _records, _topRecords;
  bool _finished = false;
  int _numResizes = 0;
};

// The physical filter operator tree FilterPlanNode / FilterOperatorUtils would build for the query's WHERE clause on this segment, as text
// (SORTED / BITMAP / INVERTED / SCAN leaves, AND children in execution order); needs no device
std::string explainFilter(const ImmutableSegment& segment, const QueryContext& queryContext);

// GroupByCombineOperator: the segments' group-by blocks through one IndexedTable (plan_maker.cpp)
ResultsBlock combineGroupByBlocks(const std::vector<ResultsBlock>& blocks, const QueryContext& queryContext);

// GroupByOperator.java:119-135: ORDER BY + minSegmentGroupTrimSize > 0 + more groups than max(5 * LIMIT, minSegmentGroupTrimSize):
// keep that many groups, the ones that sort first (TableResizer.trimInSegmentResults)
void trimSegmentGroupByBlock(ResultsBlock* block, const QueryContext& queryContext);

// GroupByDataTableReducer: the broker's rows for a combined block -- merged into the reducer's table, sorted by ORDER BY, first LIMIT rows,
// aggregations as final results
struct ReducedRow { std::vector<GroupKeyValue> keys; std::vector<OrderByValue> finals; };
// BrokerResponseNative.resultTable: the reduced rows projected onto the SELECT list (column names as the reference prints them)
struct ResultTable { std::vector<std::string> columnNames; std::vector<std::vector<OrderByValue>> rows; };
ResultTable toResultTable(const std::vector<ReducedRow>& rows, const ResultsBlock& combined, const QueryContext& queryContext);
std::vector<ReducedRow> reduceGroupBy(const ResultsBlock& combined, const QueryContext& queryContext);

// ---- the C ABI, resolved at run time from libpinot_gpu.so ---------------------------------------------------------
struct GpuAbi {
  decltype(&pg_init) init;
  decltype(&pg_last_error) last_error;
  decltype(&pg_segment_open) segment_open;
  decltype(&pg_segment_close) segment_close;
  decltype(&pg_query_check) query_check;
  decltype(&pg_execute) execute;
  decltype(&pg_execute_batch) execute_batch;
  decltype(&pg_result_free) result_free;
  decltype(&pg_filter_bitmap) filter_bitmap;
  decltype(&pg_group_key_info) group_key_info;
  decltype(&pg_group_key_values) group_key_values;
};
const GpuAbi& gpuAbi();   // throws std::runtime_error when libpinot_gpu.so cannot be loaded (no fallback)

}  // namespace pinot

extern "C" int64_t ph_roaring_serialize(const int32_t* sorted_doc_ids, int64_t n, int32_t run_optimize, uint8_t* out);
