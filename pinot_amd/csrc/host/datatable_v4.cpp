// DataTable V4 emission of segment / instance level aggregation and group-by results: the bytes a server sends the broker
// (InstanceResponseBlock.toDataTable -> BaseResultsBlock.getDataTable -> DataTableImplV4.toBytes).  Follows
//   pinot-common/.../datatable/DataTableImplV4.java:51-84 (layout), :422-518 (toBytes / writeLeadingSections), :532-558 (metadata),
//     :375-391 (string dictionary), :589-606 (exceptions)
//   pinot-core/.../common/datatable/BaseDataTableBuilder.java:61-129,193-197 and DataTableBuilderV4.java:38-83 (rows, objects, null bitmaps)
//   pinot-common/.../datatable/DataTableUtils.java:41-65 (column offsets), pinot-common/.../utils/DataSchema.java:118-143 (schema bytes)
//   pinot-core/.../operator/blocks/results/AggregationResultsBlock.java:79-171, GroupByResultsBlock.java:186-316, BaseResultsBlock.java:190-202
//   pinot-core/.../common/ObjectSerDeUtils.java:121 (AvgPair = object type 4), segl/customobject/AvgPair.java:57-62 (double sum, long count)
// Everything is big-endian (java.io.DataOutputStream / ByteBuffer).  Two orders the reference leaves to hash maps are fixed here and
// accepted by its readers: metadata entries go out in ascending key id (the reference iterates a HashMap; deserializeMetadata reads
// by key id), group-by rows in this library's merged order (the reference iterates a ConcurrentIndexedTable).
#include <cmath>
#include <cstring>
#include <map>

#include "pinot_host.h"

namespace pinot {
namespace {

struct Out {
  std::vector<uint8_t> bytes;
  void i32(int32_t v) { for (int s = 24; s >= 0; s -= 8) bytes.push_back((uint8_t)((uint32_t)v >> s)); }
  void i64(int64_t v) { for (int s = 56; s >= 0; s -= 8) bytes.push_back((uint8_t)((uint64_t)v >> s)); }
  void f32(float v) { int32_t b; memcpy(&b, &v, 4); i32(b); }
  void f64(double v) { int64_t b; memcpy(&b, &v, 8); i64(b); }
  void str(const std::string& s) { i32((int32_t)s.size()); bytes.insert(bytes.end(), s.begin(), s.end()); }
  void raw(const std::vector<uint8_t>& b) { bytes.insert(bytes.end(), b.begin(), b.end()); }
  int32_t size() const { return (int32_t)bytes.size(); }
};

enum class ColumnType { INT, LONG, FLOAT, DOUBLE, STRING, OBJECT };
const char* columnTypeName(ColumnType t) {
  switch (t) {
    case ColumnType::INT: return "INT"; case ColumnType::LONG: return "LONG"; case ColumnType::FLOAT: return "FLOAT";
    case ColumnType::DOUBLE: return "DOUBLE"; case ColumnType::STRING: return "STRING"; default: return "OBJECT";
  }
}
int columnBytes(ColumnType t) { return (t == ColumnType::INT || t == ColumnType::FLOAT || t == ColumnType::STRING) ? 4 : 8; }   // DataTableUtils.java:48-61

// AggregationFunction.getIntermediateResultColumnType: COUNT LONG, SUM / MIN / MAX DOUBLE, AVG OBJECT (AvgPair)
ColumnType intermediateType(AggregationFunctionType t) {
  return t == AggregationFunctionType::COUNT ? ColumnType::LONG : (t == AggregationFunctionType::AVG ? ColumnType::OBJECT : ColumnType::DOUBLE);
}

constexpr int32_t kObjectTypeAvgPair = 4;      // ObjectSerDeUtils.ObjectType.AvgPair
constexpr int32_t kObjectTypeNull = 100;       // CustomObject.NULL_TYPE_VALUE

// One RoaringBitmap holding the given ascending row ids, in the portable format RoaringBitmapUtils.serialize writes
// (array / bitmap containers; no run containers: the builder's bitmaps are never run-optimised).
std::vector<uint8_t> serializeRowIds(const std::vector<int32_t>& rows) {
  std::vector<uint8_t> out((size_t)ph_roaring_serialize(rows.data(), (int64_t)rows.size(), 0, nullptr));
  ph_roaring_serialize(rows.data(), (int64_t)rows.size(), 0, out.data());
  return out;
}

class Builder {      // DataTableBuilderV4 + BaseDataTableBuilder
 public:
  Builder(std::vector<std::string> names, std::vector<ColumnType> types) : _names(std::move(names)), _types(std::move(types)) {
    int at = 0;
    for (ColumnType t : _types) { _offsets.push_back(at); at += columnBytes(t); }
    _rowSize = at;
  }
  void startRow() { _numRows++; _row.assign((size_t)_rowSize, 0); }
  void finishRow() { _fixed.raw(_row); }
  void setInt(int c, int32_t v) { put32(_offsets[(size_t)c], v); }
  void setLong(int c, int64_t v) { put64(_offsets[(size_t)c], v); }
  void setFloat(int c, float v) { int32_t b; memcpy(&b, &v, 4); put32(_offsets[(size_t)c], b); }
  void setDouble(int c, double v) { int64_t b; memcpy(&b, &v, 8); put64(_offsets[(size_t)c], b); }
  void setString(int c, const std::string& v) {                          // DataTableBuilderV4.setColumn(int, String): dictionary id
    auto it = _dictionary.find(v);
    if (it == _dictionary.end()) { it = _dictionary.emplace(v, (int32_t)_reverse.size()).first; _reverse.push_back(v); }
    put32(_offsets[(size_t)c], it->second);
  }
  void setAvgPair(int c, const AvgPair* v) {                             // BaseDataTableBuilder.setColumn(int, Object)
    put32(_offsets[(size_t)c], _variable.size());
    if (!v) { put32(_offsets[(size_t)c] + 4, 0); _variable.i32(kObjectTypeNull); return; }
    put32(_offsets[(size_t)c] + 4, 16);
    _variable.i32(kObjectTypeAvgPair);
    _variable.f64(v->sum);
    _variable.i64(v->count);
  }
  void setNullRowIds(const std::vector<int32_t>& rows) {                 // DataTableBuilderV4.setNullRowIds: after the rows, one (offset, length) per column
    _fixed.i32(_variable.size());
    if (rows.empty()) { _fixed.i32(0); return; }
    const std::vector<uint8_t> bitmap = serializeRowIds(rows);
    _fixed.i32((int32_t)bitmap.size());
    _variable.raw(bitmap);
  }
  std::vector<uint8_t> build(const std::map<int, std::pair<char, std::string>>& metadata) const {
    Out exceptions; exceptions.i32(0);
    Out dictionary; dictionary.i32((int32_t)_reverse.size());
    for (const auto& s : _reverse) dictionary.str(s);
    Out schema; schema.i32((int32_t)_names.size());
    for (const auto& n : _names) schema.str(n);
    for (ColumnType t : _types) schema.str(columnTypeName(t));
    Out out;
    out.i32(4);                                      // DataTableFactory.VERSION_4
    out.i32(_numRows);
    out.i32((int32_t)_names.size());
    int32_t at = 13 * 4;                             // HEADER_SIZE
    out.i32(at); out.i32(exceptions.size()); at += exceptions.size();
    out.i32(at); out.i32(dictionary.size()); at += dictionary.size();
    out.i32(at); out.i32(schema.size()); at += schema.size();
    out.i32(at); out.i32(_fixed.size()); at += _fixed.size();
    out.i32(at); out.i32(_variable.size());
    out.raw(exceptions.bytes); out.raw(dictionary.bytes); out.raw(schema.bytes); out.raw(_fixed.bytes); out.raw(_variable.bytes);
    Out meta; meta.i32((int32_t)metadata.size());
    for (const auto& e : metadata) {                 // [key id, value]: INT 4 bytes, LONG 8 bytes, STRING length + UTF-8
      meta.i32(e.first);
      if (e.second.first == 'i') meta.i32((int32_t)std::stol(e.second.second));
      else if (e.second.first == 'l') meta.i64((int64_t)std::stoll(e.second.second));
      else meta.str(e.second.second);
    }
    out.i32(meta.size());
    out.raw(meta.bytes);
    return out.bytes;
  }
 private:
  void put32(int at, int32_t v) { for (int k = 0; k < 4; ++k) _row[(size_t)at + k] = (uint8_t)((uint32_t)v >> (24 - 8 * k)); }
  void put64(int at, int64_t v) { for (int k = 0; k < 8; ++k) _row[(size_t)at + k] = (uint8_t)((uint64_t)v >> (56 - 8 * k)); }
  std::vector<std::string> _names;
  std::vector<ColumnType> _types;
  std::vector<int> _offsets;
  int _rowSize = 0, _numRows = 0;
  std::vector<uint8_t> _row;
  Out _fixed, _variable;
  std::map<std::string, int32_t> _dictionary;
  std::vector<std::string> _reverse;
};

// MetadataKey ids and value types (pinot-common/.../datatable/DataTable.java:104-142)
void put(std::map<int, std::pair<char, std::string>>* m, int id, char type, const std::string& value) { (*m)[id] = {type, value}; }

void setIntermediate(Builder* b, int column, const AggregationFunction& f, const IntermediateResult& r, std::vector<int32_t>* nullRows, int row, bool isGroupBy) {
  const ColumnType t = intermediateType(f.getType());
  if (isNullResult(r)) {                               // null handling: placeholder + the column's null bitmap (AggregationResultsBlock.java:119-122)
    // AggregationResultsBlock.getDataTable adds row 0 to the null bitmap for EVERY null result, OBJECT included (:119-122); only
    // GroupByResultsBlock leaves OBJECT columns out of the bitmaps (a null object already says so).
    if (t == ColumnType::OBJECT) { if (!isGroupBy) nullRows->push_back(row); b->setAvgPair(column, nullptr); }
    else { nullRows->push_back(row); if (t == ColumnType::LONG) b->setLong(column, 0); else b->setDouble(column, 0.0); }
    return;
  }
  if (t == ColumnType::LONG) b->setLong(column, std::get<int64_t>(r));
  else if (t == ColumnType::DOUBLE) b->setDouble(column, std::get<double>(r));
  else { const AvgPair p = std::get<AvgPair>(r); b->setAvgPair(column, &p); }
}

}  // namespace

std::vector<uint8_t> toDataTableV4(const ResultsBlock& block, bool nullHandlingEnabled, int numSegmentsProcessed, int numSegmentsMatched) {
  std::vector<std::string> names;
  std::vector<ColumnType> types;
  const std::vector<AggregationFunction>& functions = block.isGroupBy ? block.groupBy.functions : block.aggregation.functions;
  if (block.isGroupBy) {
    // GroupByOperator.java:76-96: group-by columns first (ColumnDataType.fromDataTypeSV), then the functions' intermediate types
    for (size_t k = 0; k < block.groupBy.groupByColumns.size(); ++k) {
      names.push_back(block.groupBy.groupByColumns[k]);
      switch (block.groupBy.groupByTypes.at(k)) {
        case DataType::INT: types.push_back(ColumnType::INT); break;
        case DataType::LONG: types.push_back(ColumnType::LONG); break;
        case DataType::FLOAT: types.push_back(ColumnType::FLOAT); break;
        case DataType::DOUBLE: types.push_back(ColumnType::DOUBLE); break;
        default: types.push_back(ColumnType::STRING); break;
      }
    }
  }
  for (const auto& f : functions) { names.push_back(f.getResultColumnName()); types.push_back(intermediateType(f.getType())); }
  Builder builder(names, types);
  std::vector<std::vector<int32_t>> nullRows(names.size());
  if (!block.isGroupBy) {
    builder.startRow();
    for (size_t a = 0; a < functions.size(); ++a) setIntermediate(&builder, (int)a, functions[a], block.aggregation.results[a], &nullRows[a], 0, false);
    builder.finishRow();
  } else {
    const size_t nk = block.groupBy.groupByColumns.size();
    for (size_t r = 0; r < block.groupBy.groupKeys.size(); ++r) {
      builder.startRow();
      for (size_t k = 0; k < nk; ++k) {
        const GroupKeyValue& v = block.groupBy.groupKeys[r].keys[k];
        if (std::holds_alternative<std::monostate>(v)) {        // NULL key: the stored type's placeholder + the key column's null bitmap (GroupByResultsBlock.java:208-214)
          nullRows[k].push_back((int32_t)r);
          switch (types[k]) {
            case ColumnType::INT: builder.setInt((int)k, 0); break;
            case ColumnType::LONG: builder.setLong((int)k, 0); break;
            case ColumnType::FLOAT: builder.setFloat((int)k, 0.0f); break;
            case ColumnType::DOUBLE: builder.setDouble((int)k, 0.0); break;
            default: builder.setString((int)k, ""); break;
          }
          continue;
        }
        switch (types[k]) {
          case ColumnType::INT: builder.setInt((int)k, (int32_t)std::get<int64_t>(v)); break;
          case ColumnType::LONG: builder.setLong((int)k, std::get<int64_t>(v)); break;
          case ColumnType::FLOAT: builder.setFloat((int)k, (float)std::get<double>(v)); break;
          case ColumnType::DOUBLE: builder.setDouble((int)k, std::get<double>(v)); break;
          default: builder.setString((int)k, std::get<std::string>(v)); break;
        }
      }
      for (size_t a = 0; a < functions.size(); ++a) setIntermediate(&builder, (int)(nk + a), functions[a], block.groupBy.results[r][a], &nullRows[nk + a], (int)r, true);
      builder.finishRow();
    }
  }
  if (nullHandlingEnabled) for (const auto& rows : nullRows) builder.setNullRowIds(rows);
  std::map<int, std::pair<char, std::string>> metadata;                  // BaseResultsBlock.getResultsMetadata :190-202
  put(&metadata, 10, 'l', std::to_string(block.stats.numTotalDocs));
  put(&metadata, 2, 'l', std::to_string(block.stats.numDocsScanned));
  put(&metadata, 3, 'l', std::to_string(block.stats.numEntriesScannedInFilter));
  put(&metadata, 4, 'l', std::to_string(block.stats.numEntriesScannedPostFilter));
  put(&metadata, 6, 'i', std::to_string(numSegmentsProcessed));
  put(&metadata, 7, 'i', std::to_string(numSegmentsMatched));
  put(&metadata, 26, 'i', "0");
  put(&metadata, 27, 'i', "0");
  if (block.isGroupBy) {                                                 // GroupByResultsBlock.getResultsMetadata :308-316
    if (block.numGroupsLimitReached) put(&metadata, 11, 's', "true");
    put(&metadata, 15, 'i', "0");                                        // numResizes / resizeTimeMs: the device table is never resized
    put(&metadata, 16, 'l', "0");
  }
  return builder.build(metadata);
}

}  // namespace pinot
