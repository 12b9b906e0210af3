// c_api.cpp -- flat C facade over the C++ host mirror so that Python tests can drive
// GpuPlanMaker.makeSegmentPlanNode(...).run().nextBlock() and the combine step the way the reference's
// BaseQueriesTest.getOperator(sql) / getBrokerResponse(sql, planMaker) do (BaseQueriesTest.java:100-105,154-156).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>

#include "pinot_host.h"

using namespace pinot;

namespace {
thread_local std::string g_hostError;
GpuPlanMaker g_planMaker;

std::string jsonEscape(const std::string& s) {
  std::string o;
  for (char c : s) {
    if (c == '"' || c == '\\') { o += '\\'; o += c; }
    else if ((unsigned char)c < 0x20) { char b[8]; snprintf(b, sizeof(b), "\\u%04x", c); o += b; }
    else o += c;
  }
  return o;
}

std::string num(double v) {
  if (std::isinf(v)) return v > 0 ? "\"Infinity\"" : "\"-Infinity\"";
  if (std::isnan(v)) return "\"NaN\"";        // bare nan is not JSON
  char b[64];
  snprintf(b, sizeof(b), "%.17g", v);
  return b;
}

std::string intermediateJson(const IntermediateResult& r) {
  if (isNullResult(r)) return "null";
  if (std::holds_alternative<int64_t>(r)) return std::to_string(std::get<int64_t>(r));
  if (std::holds_alternative<double>(r)) return num(std::get<double>(r));
  const AvgPair& p = std::get<AvgPair>(r);
  return "[" + num(p.sum) + ", " + std::to_string(p.count) + "]";
}

std::string blockJson(const ResultsBlock& b) {
  std::ostringstream o;
  o << "{\"isGroupBy\": " << (b.isGroupBy ? "true" : "false");
  const auto& functions = b.isGroupBy ? b.groupBy.functions : b.aggregation.functions;
  o << ", \"columns\": [";
  for (size_t i = 0; i < functions.size(); ++i) o << (i ? ", " : "") << "\"" << jsonEscape(functions[i].getResultColumnName()) << "\"";
  o << "]";
  if (!b.isGroupBy) {
    o << ", \"intermediate\": [";
    for (size_t i = 0; i < functions.size(); ++i) o << (i ? ", " : "") << intermediateJson(b.aggregation.results[i]);
    o << "], \"final\": [";
    for (size_t i = 0; i < functions.size(); ++i)
      o << (i ? ", " : "") << (isNullResult(b.aggregation.results[i]) ? std::string("null") : num(functions[i].extractFinalResult(b.aggregation.results[i])));
    o << "]";
  } else {
    o << ", \"groupByColumns\": [";
    for (size_t i = 0; i < b.groupBy.groupByColumns.size(); ++i) o << (i ? ", " : "") << "\"" << jsonEscape(b.groupBy.groupByColumns[i]) << "\"";
    o << "], \"groups\": [";
    for (size_t g = 0; g < b.groupBy.groupKeys.size(); ++g) {
      o << (g ? ", " : "") << "{\"key\": [";
      const auto& keys = b.groupBy.groupKeys[g].keys;
      for (size_t k = 0; k < keys.size(); ++k) {
        o << (k ? ", " : "");
        if (std::holds_alternative<std::monostate>(keys[k])) o << "null";
        else if (std::holds_alternative<int64_t>(keys[k])) o << std::get<int64_t>(keys[k]);
        else if (std::holds_alternative<double>(keys[k])) o << num(std::get<double>(keys[k]));
        else o << "\"" << jsonEscape(std::get<std::string>(keys[k])) << "\"";
      }
      o << "], \"intermediate\": [";
      for (size_t i = 0; i < functions.size(); ++i) o << (i ? ", " : "") << intermediateJson(b.groupBy.results[g][i]);
      o << "], \"final\": [";
      for (size_t i = 0; i < functions.size(); ++i)
        o << (i ? ", " : "") << (isNullResult(b.groupBy.results[g][i]) ? std::string("null") : num(functions[i].extractFinalResult(b.groupBy.results[g][i])));
      o << "]}";
    }
    o << "]";
  }
  o << ", \"stats\": {\"numDocsScanned\": " << b.stats.numDocsScanned << ", \"numEntriesScannedInFilter\": " << b.stats.numEntriesScannedInFilter
    << ", \"numEntriesScannedPostFilter\": " << b.stats.numEntriesScannedPostFilter << ", \"numTotalDocs\": " << b.stats.numTotalDocs << "}";
  if (b.isGroupBy) o << ", \"numGroupsLimitReached\": " << (b.numGroupsLimitReached ? "true" : "false");
  o << ", \"deviceMs\": " << num(b.deviceMs) << ", \"kernelMs\": " << num(b.kernelMs) << "}";
  return o.str();
}

// the broker's rows (GroupByDataTableReducer): [[key..., final...], ...] in ORDER BY order, at most LIMIT of them
std::string reducedJson(const std::vector<ReducedRow>& rows) {
  std::ostringstream o;
  o << "[";
  for (size_t r = 0; r < rows.size(); ++r) {
    o << (r ? ", " : "") << "[";
    bool first = true;
    for (const auto& k : rows[r].keys) {
      o << (first ? "" : ", ");
      first = false;
      if (std::holds_alternative<std::monostate>(k)) o << "null";
      else if (std::holds_alternative<int64_t>(k)) o << std::get<int64_t>(k);
      else if (std::holds_alternative<double>(k)) o << num(std::get<double>(k));
      else o << "\"" << jsonEscape(std::get<std::string>(k)) << "\"";
    }
    for (const auto& v : rows[r].finals) {
      o << (first ? "" : ", ");
      first = false;
      if (std::holds_alternative<std::monostate>(v)) o << "null";
      else if (std::holds_alternative<int64_t>(v)) o << std::get<int64_t>(v);
      else if (std::holds_alternative<double>(v)) o << num(std::get<double>(v));
      else o << "\"" << jsonEscape(std::get<std::string>(v)) << "\"";
    }
    o << "]";
  }
  o << "]";
  return o.str();
}

// A results block from flat arrays (ph_datatable_v4_build / ph_group_by_combine: tests drive the writers and the combine without a
// device): rows [row_begin, row_end) of the arrays.
ResultsBlock blockFromArrays(bool is_group_by, const std::vector<AggregationFunction>& functions, int32_t num_keys, const char* const* key_names,
                             const int32_t* key_types, int64_t row_begin, int64_t row_end, const int64_t* key_longs, const double* key_doubles,
                             const char* const* key_strings, const uint8_t* key_is_null, const int64_t* counts, const double* sums, const double* mins,
                             const double* maxs, const uint8_t* is_null) {
  ResultsBlock block;
  block.isGroupBy = is_group_by;
  const int num_functions = (int)functions.size();
  auto value = [&](int64_t row, int f) -> IntermediateResult {
    const size_t at = (size_t)row * (size_t)num_functions + (size_t)f;
    if (is_null && is_null[at]) return std::monostate{};
    switch (functions[(size_t)f].getType()) {
      case AggregationFunctionType::COUNT: return counts[at];
      case AggregationFunctionType::SUM: return sums[at];
      case AggregationFunctionType::MIN: return mins[at];
      case AggregationFunctionType::MAX: return maxs[at];
      default: return AvgPair{sums[at], counts[at]};
    }
  };
  if (!block.isGroupBy) {
    block.aggregation.functions = functions;
    for (int f = 0; f < num_functions; ++f) block.aggregation.results.push_back(value(row_begin, f));
    return block;
  }
  GroupByResultsBlock& g = block.groupBy;
  g.functions = functions;
  for (int k = 0; k < num_keys; ++k) { g.groupByColumns.push_back(key_names[k]); g.groupByTypes.push_back((DataType)key_types[k]); }
  for (int64_t r = row_begin; r < row_end; ++r) {
    GroupKey key;
    key.groupId = (int)(r - row_begin);
    for (int k = 0; k < num_keys; ++k) {
      const size_t at = (size_t)r * (size_t)num_keys + (size_t)k;
      const DataType t = (DataType)key_types[k];
      if (key_is_null && key_is_null[at]) key.keys.emplace_back(std::monostate{});
      else if (t == DataType::INT || t == DataType::LONG) key.keys.emplace_back(key_longs[at]);
      else if (t == DataType::STRING) key.keys.emplace_back(std::string(key_strings[at]));
      else key.keys.emplace_back(key_doubles[at]);
    }
    g.groupKeys.push_back(std::move(key));
    std::vector<IntermediateResult> row;
    for (int f = 0; f < num_functions; ++f) row.push_back(value(r, f));
    g.results.push_back(std::move(row));
  }
  return block;
}

std::string resultTableJson(const ResultTable& t) {
  std::ostringstream o;
  o << "{\"columns\": [";
  for (size_t i = 0; i < t.columnNames.size(); ++i) o << (i ? ", " : "") << "\"" << jsonEscape(t.columnNames[i]) << "\"";
  o << "], \"rows\": [";
  for (size_t r = 0; r < t.rows.size(); ++r) {
    o << (r ? ", " : "") << "[";
    for (size_t i = 0; i < t.rows[r].size(); ++i) {
      const OrderByValue& v = t.rows[r][i];
      o << (i ? ", " : "");
      if (std::holds_alternative<std::monostate>(v)) o << "null";
      else if (std::holds_alternative<int64_t>(v)) o << std::get<int64_t>(v);
      else if (std::holds_alternative<double>(v)) o << num(std::get<double>(v));
      else o << "\"" << jsonEscape(std::get<std::string>(v)) << "\"";
    }
    o << "]";
  }
  o << "]}";
  return o.str();
}

template <typename F>
int guarded(F f) {
  try { f(); return 0; }
  catch (const QueryException& e) { g_hostError = e.what(); return 1; }
  catch (const UnsupportedOperationException& e) { g_hostError = e.what(); return 2; }
  catch (const std::exception& e) { g_hostError = e.what(); return 3; }
}
}  // namespace

extern "C" {

const char* ph_last_error(void) { return g_hostError.c_str(); }
void ph_free(char* p) { free(p); }

// Attaches the null value vector file of a column added earlier (the buffer stays caller-owned like the other index buffers).
int32_t ph_segment_set_null_vector(void* seg, const char* column, const void* data, uint64_t size) {
  return guarded([&] {
    ImmutableSegment* s = static_cast<ImmutableSegment*>(seg);
    DataSource& ds = s->mutableDataSource(column ? column : "");
    ds.nullValueVector = (const uint8_t*)data;
    ds.nullValueVectorSize = size;
  });
}

// Segments assembled column by column have no metadata.properties: isSorted is derived from the data the way the segment creator's
// column statistics do (a dictionary column whose dictIds never decrease in docId order, AbstractColumnStatisticsCollector), together
// with the [start, end] docId pair of every dictId that SortedIndexReaderImpl would hold.  Unsorted columns leave at the first descent.
static void detectSorted(DataSource* ds, int numDocs) {
  if (!ds->hasDictionary || ds->forwardIndex == nullptr || ds->bitsPerElement <= 0 || ds->bitsPerElement > 31 || numDocs <= 0 || ds->cardinality <= 0) return;
  const int b = ds->bitsPerElement;
  if (ds->forwardIndexSize < ((uint64_t)numDocs * (uint64_t)b + 7) / 8) return;
  std::vector<int32_t> ranges((size_t)ds->cardinality * 2, -1);
  int32_t previous = -1;
  for (int64_t doc = 0; doc < numDocs; ++doc) {
    const uint64_t bit = (uint64_t)doc * (uint64_t)b;
    uint64_t window = 0;                                      // the 5 bytes that hold the value, MSB first (PinotDataBitSet layout)
    for (int k = 0; k < 5; ++k) { const uint64_t at = (bit >> 3) + (uint64_t)k; window = (window << 8) | (at < ds->forwardIndexSize ? ds->forwardIndex[at] : 0u); }
    const int32_t id = (int32_t)((window >> (40 - (int)(bit & 7) - b)) & ((1ull << b) - 1ull));
    if (id < previous || id >= ds->cardinality) return;
    if (id != previous) ranges[2 * (size_t)id] = (int32_t)doc;
    ranges[2 * (size_t)id + 1] = (int32_t)doc;
    previous = id;
  }
  for (int32_t r : ranges) if (r < 0) return;                 // a dictionary entry no doc uses: not a creator-built segment
  ds->isSorted = true;
  ds->sortedDocIdRanges = std::move(ranges);
}

void* ph_segment_create(const char* name, int32_t num_docs) { return new ImmutableSegment(name ? name : "", num_docs); }

int32_t ph_segment_add_int_column(void* seg, const char* name, int32_t has_dictionary, int32_t bits, int32_t cardinality, const void* fwd,
                                  uint64_t fwd_size, const void* dict, uint64_t dict_size, const void* inv, uint64_t inv_size) {
  return guarded([&] {
    DataSource ds;
    ds.name = name;
    ds.dataType = DataType::INT;
    ds.hasDictionary = has_dictionary != 0;
    ds.bitsPerElement = bits;
    ds.cardinality = cardinality;
    ds.forwardIndex = (const uint8_t*)fwd; ds.forwardIndexSize = fwd_size;
    ds.dictionaryBuffer = (const uint8_t*)dict; ds.dictionaryBufferSize = dict_size;
    if (ds.hasDictionary) ds.dictionary = std::make_shared<IntDictionary>((const uint8_t*)dict, cardinality);
    ds.hasInvertedIndex = inv != nullptr && inv_size > 0;
    ds.invertedIndex = (const uint8_t*)inv; ds.invertedIndexSize = inv_size;
    detectSorted(&ds, ((ImmutableSegment*)seg)->getTotalDocs());
    ((ImmutableSegment*)seg)->addDataSource(std::move(ds));
  });
}

// data_type: 0 INT, 1 LONG, 2 FLOAT, 3 DOUBLE (pg_data_type).  Dictionary columns pass the big-endian fixed-width dictionary
// buffer; raw columns the PASS_THROUGH chunk file.
int32_t ph_segment_add_numeric_column(void* seg, const char* name, int32_t data_type, int32_t has_dictionary, int32_t bits, int32_t cardinality,
                                      const void* fwd, uint64_t fwd_size, const void* dict, uint64_t dict_size, const void* inv, uint64_t inv_size) {
  return guarded([&] {
    if (data_type < 0 || data_type > 3) throw QueryException("unknown data type");
    static const DataType kTypes[] = {DataType::INT, DataType::LONG, DataType::FLOAT, DataType::DOUBLE};
    DataSource ds;
    ds.name = name;
    ds.dataType = kTypes[data_type];
    ds.hasDictionary = has_dictionary != 0;
    ds.bitsPerElement = bits;
    ds.cardinality = cardinality;
    ds.forwardIndex = (const uint8_t*)fwd; ds.forwardIndexSize = fwd_size;
    ds.dictionaryBuffer = (const uint8_t*)dict; ds.dictionaryBufferSize = dict_size;
    if (ds.hasDictionary) {
      const uint8_t* d = (const uint8_t*)dict;
      switch (ds.dataType) {
        case DataType::INT: ds.dictionary = std::make_shared<IntDictionary>(d, cardinality); break;
        case DataType::LONG: ds.dictionary = std::make_shared<LongDictionary>(d, cardinality); break;
        case DataType::FLOAT: ds.dictionary = std::make_shared<FloatDictionary>(d, cardinality); break;
        default: ds.dictionary = std::make_shared<DoubleDictionary>(d, cardinality); break;
      }
    }
    ds.hasInvertedIndex = inv != nullptr && inv_size > 0;
    ds.invertedIndex = (const uint8_t*)inv; ds.invertedIndexSize = inv_size;
    detectSorted(&ds, ((ImmutableSegment*)seg)->getTotalDocs());
    ((ImmutableSegment*)seg)->addDataSource(std::move(ds));
  });
}

// values: `cardinality` NUL-terminated strings back to back, sorted ascending.
int32_t ph_segment_add_string_column(void* seg, const char* name, int32_t bits, int32_t cardinality, const void* fwd, uint64_t fwd_size,
                                     const char* values, const void* inv, uint64_t inv_size) {
  return guarded([&] {
    std::vector<std::string> vals;
    const char* p = values;
    for (int i = 0; i < cardinality; ++i) { vals.emplace_back(p); p += vals.back().size() + 1; }
    DataSource ds;
    ds.name = name;
    ds.dataType = DataType::STRING;
    ds.hasDictionary = true;
    ds.bitsPerElement = bits;
    ds.cardinality = cardinality;
    ds.forwardIndex = (const uint8_t*)fwd; ds.forwardIndexSize = fwd_size;
    ds.dictionary = std::make_shared<StringDictionary>(std::move(vals));
    ds.hasInvertedIndex = inv != nullptr && inv_size > 0;
    ds.invertedIndex = (const uint8_t*)inv; ds.invertedIndexSize = inv_size;
    detectSorted(&ds, ((ImmutableSegment*)seg)->getTotalDocs());
    ((ImmutableSegment*)seg)->addDataSource(std::move(ds));
  });
}

// ImmutableSegmentLoader.load(indexDir): opens a v1 / v3 segment directory and makes it HBM resident.  NULL + status on error.
void* ph_segment_load_directory(const char* index_dir, int32_t device, int32_t* status) {
  ImmutableSegment* out = nullptr;
  *status = guarded([&] {
    std::vector<std::string> skipped;
    std::unique_ptr<ImmutableSegment> seg = loadSegmentDirectory(index_dir ? index_dir : "", &skipped);
    seg->notOffloaded = std::move(skipped);
    if (device >= 0) seg->load(device);      // device < 0: host-side open only (metadata, dictionaries, buffers)
    out = seg.release();
  });
  return out;
}

// JSON description of a segment of the host mirror (columns as DataSourceMetadata sees them).
char* ph_segment_describe(void* segment, int32_t* status) {
  std::string out;
  *status = guarded([&] {
    const ImmutableSegment* seg = (const ImmutableSegment*)segment;
    std::ostringstream o;
    o << "{\"name\": \"" << jsonEscape(seg->getSegmentName()) << "\", \"totalDocs\": " << seg->getTotalDocs() << ", \"columns\": [";
    bool first = true;
    for (const DataSource& ds : seg->getDataSources()) {
      o << (first ? "" : ", ") << "{\"name\": \"" << jsonEscape(ds.name) << "\", \"dataType\": \"" << dataTypeName(ds.dataType) << "\", \"hasDictionary\": "
        << (ds.hasDictionary ? "true" : "false") << ", \"cardinality\": " << ds.cardinality << ", \"bitsPerElement\": " << ds.bitsPerElement
        << ", \"hasInvertedIndex\": " << (ds.hasInvertedIndex ? "true" : "false") << ", \"isSorted\": " << (ds.isSorted ? "true" : "false")
        << ", \"hasNullValueVector\": " << (ds.nullValueVector && ds.nullValueVectorSize ? "true" : "false");
      if (ds.dictionary && ds.cardinality > 0)
        o << ", \"minValue\": \"" << jsonEscape(ds.dictionary->getStringValue(0)) << "\", \"maxValue\": \"" << jsonEscape(ds.dictionary->getStringValue(ds.cardinality - 1)) << "\"";
      o << "}";
      first = false;
    }
    o << "], \"notOffloaded\": [";
    for (size_t i = 0; i < seg->notOffloaded.size(); ++i) o << (i ? ", " : "") << "\"" << jsonEscape(seg->notOffloaded[i]) << "\"";
    o << "]}";
    out = o.str();
  });
  return *status == 0 ? strdup(out.c_str()) : nullptr;
}

int32_t ph_segment_load(void* seg, int32_t device) { return guarded([&] { ((ImmutableSegment*)seg)->load(device); }); }
void ph_segment_destroy(void* seg) { delete (ImmutableSegment*)seg; }

int32_t ph_plan_maker_init(int32_t device, int32_t time_kernels) {
  return guarded([&] {
    std::map<std::string, std::string> cfg;
    cfg[GpuPlanMaker::kConfigDevice] = std::to_string(device);
    cfg[GpuPlanMaker::kConfigTimeKernels] = time_kernels ? "true" : "false";
    const char* batch = getenv("PINOT_GPU_HOST_BATCH");          // tests: "0" = every segment operator runs its own pg_execute
    cfg[GpuPlanMaker::kConfigBatch] = (batch && batch[0] == '0') ? "false" : "true";
    const char* exact = getenv("PINOT_GPU_HOST_EXACT_FILTER_STATS");      // tests: "0" = gpu.exact.filter.stats=false
    cfg[GpuPlanMaker::kConfigExactFilterStats] = (exact && exact[0] == '0') ? "false" : "true";
    g_planMaker.init(cfg);
  });
}

// gpu.devices parsing and least-loaded placement (no device touched): "[d0, d1, ...]" for `text`, then the device each of the `count`
// segment sizes goes to when opened in order -- the logic GpuPlanMaker.java / GpuSegmentCache.java carry in Java.
char* ph_plan_maker_placement(const char* devices_text, const int64_t* segment_bytes, int32_t count, int32_t* status) {
  std::string out;
  *status = guarded([&] {
    GpuPlanMaker pm;      // (not initialised: no device, no pg_init -- only its placement book)
    try { pm.setDevices(GpuPlanMaker::parseDevices(devices_text ? devices_text : "")); } catch (const std::invalid_argument& e) { throw QueryException(e.what()); }
    std::ostringstream o;
    o << "{\"devices\": [";
    for (size_t i = 0; i < pm.devices().size(); ++i) o << (i ? ", " : "") << pm.devices()[i];
    o << "], \"placement\": [";
    for (int32_t s = 0; s < count; ++s) {
      if (segment_bytes[s] < 0) pm.releaseSegment((int)(-segment_bytes[s] >> 48), (-segment_bytes[s]) & ((1ll << 48) - 1));      // tests: -(device << 48 | bytes) releases
      o << (s ? ", " : "") << (segment_bytes[s] < 0 ? -1 : pm.placeSegment((long long)segment_bytes[s]));
    }
    o << "]}";
    out = o.str();
  });
  return *status == 0 ? strdup(out.c_str()) : nullptr;
}

// Parse only (no device): returns a JSON description of the QueryContext; used by the CPU-side tests.
char* ph_parse_sql(const char* sql, int32_t* status) {
  std::string out;
  *status = guarded([&] {
    const QueryContext q = getQueryContext(sql);
    std::ostringstream o;
    o << "{\"table\": \"" << jsonEscape(q.tableName) << "\", \"aggregations\": [";
    for (size_t i = 0; i < q.aggregations.size(); ++i)
      o << (i ? ", " : "") << "\"" << jsonEscape(AggregationFunction(q.aggregations[i].function, q.aggregations[i].column).getResultColumnName()
                                                + (q.aggregations[i].hasFilter ? " FILTER(WHERE " + q.aggregations[i].filterText + ")" : std::string())) << "\"";
    o << "], \"groupBy\": [";
    for (size_t i = 0; i < q.groupByExpressions.size(); ++i) o << (i ? ", " : "") << "\"" << jsonEscape(q.groupByExpressions[i]) << "\"";
    o << "], \"hasFilter\": " << (q.hasFilter ? "true" : "false");
    if (q.nullHandlingEnabled) o << ", \"nullHandling\": true";
    if (q.hasOrderBy()) {
      o << ", \"orderBy\": [";
      for (size_t i = 0; i < q.orderByExpressions.size(); ++i) {
        const OrderByExpressionContext& ob = q.orderByExpressions[i];
        const std::string text = ob.isAggregation ? AggregationFunction(q.aggregations[(size_t)ob.index].function, q.aggregations[(size_t)ob.index].column).getResultColumnName()
                                                  : q.groupByExpressions[(size_t)ob.index];
        o << (i ? ", " : "") << "{\"expression\": \"" << jsonEscape(text) << "\", \"asc\": " << (ob.isAsc ? "true" : "false") << ", \"nullsLast\": "
          << (ob.isNullsLast() ? "true" : "false") << "}";
      }
      o << "]";
    }
    if (!q.groupByExpressions.empty()) o << ", \"limit\": " << q.limit;      // (aggregation-only results are one row whatever the limit)
    o << "}";
    out = o.str();
  });
  return *status == 0 ? strdup(out.c_str()) : nullptr;
}

// Lower one predicate against a dictionary buffer (no device): "[alwaysTrue, alwaysFalse, exclusive, isRange, start, end, n]"
char* ph_lower_predicate(const char* sql_predicate, const void* dict, int32_t cardinality, int32_t* status) {
  std::string out;
  *status = guarded([&] {
    const QueryContext q = getQueryContext(std::string("SELECT COUNT(*) FROM t WHERE ") + sql_predicate);
    if (q.filter.type != FilterContext::Type::PREDICATE) throw QueryException("expected a single predicate");
    DataSource ds;
    ds.name = q.filter.predicate.column;
    ds.cardinality = cardinality;
    ds.dictionary = std::make_shared<IntDictionary>((const uint8_t*)dict, cardinality);
    const PredicateEvaluator ev = getPredicateEvaluator(q.filter.predicate, ds);
    std::ostringstream o;
    o << "{\"alwaysTrue\": " << (ev.alwaysTrue ? "true" : "false") << ", \"alwaysFalse\": " << (ev.alwaysFalse ? "true" : "false")
      << ", \"exclusive\": " << (ev.exclusive ? "true" : "false") << ", \"isRange\": " << (ev.isRange ? "true" : "false")
      << ", \"start\": " << ev.startDictId << ", \"end\": " << ev.endDictId << ", \"dictIds\": [";
    for (size_t i = 0; i < ev.matchingDictIds.size(); ++i) o << (i ? ", " : "") << ev.matchingDictIds[i];
    o << "]}";
    out = o.str();
  });
  return *status == 0 ? strdup(out.c_str()) : nullptr;
}

// The physical filter tree of `sql`'s WHERE clause over a segment (which need not be loaded on a device): see explainFilter
char* ph_explain_filter(void* segment, const char* sql, int32_t* status) {
  std::string out;
  *status = guarded([&] { out = explainFilter(*(const ImmutableSegment*)segment, getQueryContext(sql)); });
  return *status == 0 ? strdup(out.c_str()) : nullptr;
}

// A RangePredicate given by its bounds ("*" = unbounded) against an INT dictionary buffer: RangePredicateEvaluatorFactory.newDictionaryBasedEvaluator
// (the reference's RangeOfflineDictionaryPredicateEvaluatorTest builds its predicates this way; SQL can only say one side at a time)
char* ph_lower_range_predicate(const void* dict, int32_t cardinality, const char* lower, int32_t lower_inclusive, const char* upper, int32_t upper_inclusive, int32_t* status) {
  std::string out;
  *status = guarded([&] {
    Predicate p;
    p.column = "column";
    p.type = Predicate::Type::RANGE;
    p.lowerBound = lower; p.lowerInclusive = lower_inclusive != 0;
    p.upperBound = upper; p.upperInclusive = upper_inclusive != 0;
    DataSource ds;
    ds.name = p.column;
    ds.cardinality = cardinality;
    ds.dictionary = std::make_shared<IntDictionary>((const uint8_t*)dict, cardinality);
    const PredicateEvaluator ev = getPredicateEvaluator(p, ds);
    std::ostringstream o;
    o << "{\"alwaysTrue\": " << (ev.alwaysTrue ? "true" : "false") << ", \"alwaysFalse\": " << (ev.alwaysFalse ? "true" : "false")
      << ", \"isRange\": " << (ev.isRange ? "true" : "false") << ", \"start\": " << ev.startDictId << ", \"end\": " << ev.endDictId
      << ", \"numMatchingItems\": " << ev.getNumMatchingItems() << "}";
    out = o.str();
  });
  return *status == 0 ? strdup(out.c_str()) : nullptr;
}

// A RangePredicate on a RAW (no-dictionary) INT (data_type 0) or LONG (1) column: RangePredicateEvaluatorFactory.newRawValueBasedEvaluator;
// the evaluator matches lower <= value <= upper (both inclusive after the exclusive bounds were stepped inwards)
char* ph_lower_raw_range_predicate(int32_t data_type, const char* lower, int32_t lower_inclusive, const char* upper, int32_t upper_inclusive, int32_t* status) {
  std::string out;
  *status = guarded([&] {
    Predicate p;
    p.column = "column";
    p.type = Predicate::Type::RANGE;
    p.lowerBound = lower; p.lowerInclusive = lower_inclusive != 0;
    p.upperBound = upper; p.upperInclusive = upper_inclusive != 0;
    DataSource ds;
    ds.name = p.column;
    ds.hasDictionary = false;
    ds.dataType = data_type == 1 ? DataType::LONG : DataType::INT;
    const PredicateEvaluator ev = getPredicateEvaluator(p, ds);
    std::ostringstream o;
    o << "{\"alwaysTrue\": " << (ev.alwaysTrue ? "true" : "false") << ", \"alwaysFalse\": " << (ev.alwaysFalse ? "true" : "false")
      << ", \"rawLower\": " << ev.rawLower << ", \"rawUpper\": " << ev.rawUpper << "}";
    out = o.str();
  });
  return *status == 0 ? strdup(out.c_str()) : nullptr;
}

// getOperator(sql).nextBlock() per segment + the combined block: {"segments": [...], "combined": {...}}
char* ph_execute_sql(void** segments, int32_t num_segments, const char* sql, int32_t max_execution_threads, int32_t* status) {
  std::string out;
  *status = guarded([&] {
    const QueryContext q = getQueryContext(sql);
    std::vector<SegmentContext> ctxs;
    for (int i = 0; i < num_segments; ++i) ctxs.push_back(SegmentContext{(ImmutableSegment*)segments[i]});
    std::ostringstream o;
    o << "{\"segments\": [";
    for (int i = 0; i < num_segments; ++i) {
      auto op = g_planMaker.makeSegmentPlanNode(ctxs[(size_t)i], q)->run();
      const ResultsBlock b = op->nextBlock();
      o << (i ? ", " : "") << blockJson(b);
    }
    const ResultsBlock combined = g_planMaker.executeCombined(ctxs, q, max_execution_threads);
    o << "], \"combined\": " << blockJson(combined);
    if (combined.isGroupBy) {      // what the broker would answer: every key and final result ("reduced"), and the SELECT list's projection of it
      const std::vector<ReducedRow> rows = reduceGroupBy(combined, q);
      o << ", \"reduced\": " << reducedJson(rows) << ", \"resultTable\": " << resultTableJson(toResultTable(rows, combined, q));
    }
    o << "}";
    out = o.str();
  });
  return *status == 0 ? strdup(out.c_str()) : nullptr;
}

// DataTable V4 bytes of a results block given as flat arrays (tests drive the writer without a device through this):
// function_types: AggregationFunctionType ordinals (COUNT 0, SUM 1, MIN 2, MAX 3, AVG 4); key_types: DataType ordinals (INT 0, LONG 1,
// FLOAT 2, DOUBLE 3, STRING 4); per row and key the value comes from key_longs (INT / LONG), key_doubles (FLOAT / DOUBLE) or
// key_strings; per row and function: COUNT counts[], SUM sums[], MIN mins[], MAX maxs[], AVG (sums[], counts[]); is_null marks a null
// intermediate result (null handling).  stats: numDocsScanned, numEntriesScannedInFilter, numEntriesScannedPostFilter, numTotalDocs.
uint8_t* ph_datatable_v4_build(int32_t is_group_by, int32_t num_functions, const int32_t* function_types, const char* const* function_columns, int32_t num_keys,
                               const char* const* key_names, const int32_t* key_types, int64_t num_rows, const int64_t* key_longs, const double* key_doubles,
                               const char* const* key_strings, const int64_t* counts, const double* sums, const double* mins, const double* maxs,
                               const uint8_t* is_null, const int64_t* stats, int32_t null_handling, int32_t limit_reached, int32_t segments_processed,
                               int32_t segments_matched, int64_t* out_size, int32_t* status) {
  std::vector<uint8_t> bytes;
  *status = guarded([&] {
    std::vector<AggregationFunction> functions;
    for (int f = 0; f < num_functions; ++f) functions.emplace_back((AggregationFunctionType)function_types[f], function_columns[f], null_handling != 0);
    ResultsBlock block = blockFromArrays(is_group_by != 0, functions, num_keys, key_names, key_types, 0, num_rows, key_longs, key_doubles, key_strings, nullptr,
                                         counts, sums, mins, maxs, is_null);
    block.stats.numDocsScanned = stats[0]; block.stats.numEntriesScannedInFilter = stats[1];
    block.stats.numEntriesScannedPostFilter = stats[2]; block.stats.numTotalDocs = stats[3];
    block.numGroupsLimitReached = limit_reached != 0;
    bytes = toDataTableV4(block, null_handling != 0, segments_processed, segments_matched);
  });
  if (*status != 0) return nullptr;
  uint8_t* out = (uint8_t*)malloc(bytes.size() ? bytes.size() : 1);
  memcpy(out, bytes.data(), bytes.size());
  *out_size = (int64_t)bytes.size();
  return out;
}

// The DataTable V4 bytes the server would send for `sql` over these segments (combine, then InstanceResponseBlock.toDataTable().toBytes()).
// Returns a malloc-ed buffer (*out_size bytes, ph_free it) or NULL with *status set.
uint8_t* ph_execute_sql_datatable(void** segments, int32_t num_segments, const char* sql, int32_t max_execution_threads, int64_t* out_size, int32_t* status) {
  std::vector<uint8_t> bytes;
  *status = guarded([&] {
    const QueryContext q = getQueryContext(sql);
    std::vector<SegmentContext> ctxs;
    for (int i = 0; i < num_segments; ++i) ctxs.push_back(SegmentContext{(ImmutableSegment*)segments[i]});
    // numSegmentsMatched: segments with at least one doc scanned (BaseCombineOperator / InstanceResponseOperator bookkeeping)
    int matched = 0;
    for (int i = 0; i < num_segments; ++i) {
      const ResultsBlock b = g_planMaker.makeSegmentPlanNode(ctxs[(size_t)i], q)->run()->nextBlock();
      matched += b.stats.numDocsScanned > 0 ? 1 : 0;
    }
    bytes = toDataTableV4(g_planMaker.executeCombined(ctxs, q, max_execution_threads), q.nullHandlingEnabled, num_segments, matched);
  });
  if (*status != 0) return nullptr;
  uint8_t* out = (uint8_t*)malloc(bytes.size() ? bytes.size() : 1);
  memcpy(out, bytes.data(), bytes.size());
  *out_size = (int64_t)bytes.size();
  return out;
}

// GroupByCombineOperator + GroupByDataTableReducer over group-by blocks given as flat arrays (block b holds rows
// [sum(block_rows[0..b)), +block_rows[b]) of the arrays; same array conventions as ph_datatable_v4_build, key_is_null marks NULL keys).
// `sql` supplies the query context: aggregations (in the arrays' function order), GROUP BY columns, ORDER BY, LIMIT and the trim
// options.  Returns {"combined": <block>, "reduced": [[key..., final...], ...], "table": {resultSize, trimSize, trimThreshold, numResizes}}.
char* ph_group_by_combine(const char* sql, int32_t num_blocks, const int64_t* block_rows, const int32_t* key_types, const int64_t* key_longs,
                          const double* key_doubles, const char* const* key_strings, const uint8_t* key_is_null, const int64_t* counts, const double* sums,
                          const double* mins, const double* maxs, const uint8_t* is_null, int32_t* status) {
  std::string out;
  *status = guarded([&] {
    const QueryContext q = getQueryContext(sql);
    if (q.groupByExpressions.empty()) throw QueryException("ph_group_by_combine needs a GROUP BY query");
    std::vector<AggregationFunction> functions;
    for (const auto& a : q.aggregations) functions.emplace_back(a.function, a.column, q.nullHandlingEnabled);
    std::vector<const char*> key_names;
    for (const auto& g : q.groupByExpressions) key_names.push_back(g.c_str());
    std::vector<ResultsBlock> blocks;
    int64_t row = 0;
    for (int b = 0; b < num_blocks; ++b) {
      ResultsBlock block = blockFromArrays(true, functions, (int32_t)key_names.size(), key_names.data(), key_types, row, row + block_rows[b], key_longs, key_doubles,
                                           key_strings, key_is_null, counts, sums, mins, maxs, is_null);
      trimSegmentGroupByBlock(&block, q);          // the segment operator's own trim comes first (GroupByOperator.java:119-135)
      blocks.push_back(std::move(block));
      row += block_rows[b];
    }
    if (blocks.empty()) throw QueryException("no blocks");
    const IndexedTable sizes = IndexedTable::forCombineOperator(functions, q);
    // count the resizes the way the combine operator's table does
    IndexedTable table = IndexedTable::forCombineOperator(functions, q);
    for (const auto& b : blocks) for (size_t i = 0; i < b.groupBy.groupKeys.size(); ++i) table.upsert(Record{b.groupBy.groupKeys[i].keys, b.groupBy.results[i]});
    table.finish(false);
    const ResultsBlock combined = combineGroupByBlocks(blocks, q);
    std::ostringstream o;
    const std::vector<ReducedRow> reduced = reduceGroupBy(combined, q);
    o << "{\"combined\": " << blockJson(combined) << ", \"reduced\": " << reducedJson(reduced) << ", \"resultTable\": " << resultTableJson(toResultTable(reduced, combined, q))
      << ", \"table\": {\"resultSize\": " << sizes.resultSize()
      << ", \"trimSize\": " << sizes.trimSize() << ", \"trimThreshold\": " << sizes.trimThreshold() << ", \"numResizes\": " << table.getNumResizes() << "}}";
    out = o.str();
  });
  return *status == 0 ? strdup(out.c_str()) : nullptr;
}

// GroupByUtils.getTableCapacity / getIndexedTableTrimThreshold (core/util/GroupByUtils.java:48-73)
int32_t ph_group_by_table_capacity(int32_t limit, int32_t min_num_groups) { return GroupByUtils::getTableCapacity(limit, min_num_groups); }
int32_t ph_group_by_trim_threshold(int32_t trim_size, int32_t trim_threshold) { return GroupByUtils::getIndexedTableTrimThreshold(trim_size, trim_threshold); }

}  // extern "C"
