// segment_loader.cpp -- opens a Pinot segment DIRECTORY (what a server has on disk) as an ImmutableSegment of the host mirror:
// the native counterpart of ImmutableSegmentLoader.load(indexDir, ReadMode) for the parts this path needs.
//
//   layouts   v1: one file per index            <col>.dict  <col>.sv.unsorted.fwd  <col>.sv.sorted.fwd  <col>.sv.raw.fwd  <col>.bitmap.inv
//                                               (sspi/V1Constants.java:38-49; segl/segment/store/FilePerIndexDirectory.java)
//             v3: v3/columns.psf + v3/index_map (segl/segment/store/SingleFileIndexDirectory.java:72-73,174-185,216-310): every index
//                 is a slice of columns.psf that starts with the 8-byte magic marker 0xdeadbeefdeafbead; index_map lines are
//                 "<column>.<index>.startOffset = N" / "<column>.<index>.size = M" (size includes the marker; column names may
//                 contain dots, so keys are parsed from the right -- sspi/store/ColumnIndexUtils.java:33-50)
//   metadata  metadata.properties, keys of sspi/V1Constants.java:63-138 (segment.total.docs, column.<c>.cardinality / dataType /
//             bitsPerElement / lengthOfEachEntry / isSorted / hasDictionary / isSingleValues, segment.padding.character)
//
// Single-value INT / LONG / FLOAT / DOUBLE / STRING (and BOOLEAN / TIMESTAMP through their stored types) columns are opened; the
// rest (multi-value, BYTES, JSON, BIG_DECIMAL, var-length dictionaries) are listed as not offloaded and queries that name them
// keep the CPU plan.  A sorted forward index (<col>.sv.sorted.fwd: C pairs of big-endian [startDocId, endDocId],
// segl/segment/index/readers/sorted/SortedIndexReaderImpl.java:33-42) is expanded to the fixed-bit dictId stream at load.
#include <sys/stat.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

#include "pinot_host.h"

namespace pinot {

extern "C" void ph_fixedbit_pack(const int32_t* dict_ids, int64_t num_docs, int32_t bits, uint8_t* out, int32_t threads);
extern "C" int64_t ph_fixedbit_size(int64_t num_docs, int32_t bits);

namespace {

constexpr uint64_t kMagicMarker = 0xdeadbeefdeafbeadull;   // SingleFileIndexDirectory.java:72

bool fileExists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode); }

std::shared_ptr<std::vector<uint8_t>> readFile(const std::string& path) {
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if (!f) throw QueryException("cannot open " + path);
  const std::streamsize n = f.tellg();
  auto buf = std::make_shared<std::vector<uint8_t>>((size_t)n);
  f.seekg(0);
  if (n > 0 && !f.read(reinterpret_cast<char*>(buf->data()), n)) throw QueryException("cannot read " + path);
  return buf;
}

std::string trim(const std::string& s) {
  size_t a = 0, b = s.size();
  while (a < b && isspace((unsigned char)s[a])) a++;
  while (b > a && isspace((unsigned char)s[b - 1])) b--;
  return s.substr(a, b - a);
}

// The subset of commons-configuration's properties syntax the segment writers emit: "key = value" lines, '#' / '!' comments,
// backslash escapes of separators inside values.
std::map<std::string, std::string> readProperties(const std::string& path) {
  std::ifstream f(path);
  if (!f) throw QueryException("cannot open " + path);
  std::map<std::string, std::string> out;
  std::string line;
  while (std::getline(f, line)) {
    const std::string t = trim(line);
    if (t.empty() || t[0] == '#' || t[0] == '!') continue;
    const size_t eq = t.find('=');
    if (eq == std::string::npos) continue;
    std::string value = trim(t.substr(eq + 1)), unescaped;
    for (size_t i = 0; i < value.size(); ++i) {
      if (value[i] == '\\' && i + 1 < value.size()) { unescaped += value[++i]; continue; }
      unescaped += value[i];
    }
    out[trim(t.substr(0, eq))] = unescaped;
  }
  return out;
}

struct Slice { const uint8_t* data = nullptr; uint64_t size = 0; };

// Where the index buffers of one segment come from (either layout).
class IndexDirectory {
 public:
  IndexDirectory(const std::string& dir, bool v3) : _dir(dir), _v3(v3) {
    if (!v3) return;
    _psf = readFile(dir + "/columns.psf");
    for (const auto& kv : readProperties(dir + "/index_map")) {
      // "<column>.<index>.<startOffset|size>", parsed from the right (ColumnIndexUtils.parseIndexMapKeys)
      const size_t last = kv.first.rfind('.');
      if (last == std::string::npos || last == 0) throw QueryException("index_map: key separator not found: " + kv.first);
      const size_t mid = kv.first.rfind('.', last - 1);
      if (mid == std::string::npos) throw QueryException("index_map: index separator not found: " + kv.first);
      const std::string prop = kv.first.substr(last + 1), key = kv.first.substr(0, last);   // key = "<column>.<index>"
      Entry& e = _entries[key];
      if (prop == "startOffset") e.start = strtoll(kv.second.c_str(), nullptr, 10);
      else if (prop == "size") e.size = strtoll(kv.second.c_str(), nullptr, 10);
      else throw QueryException("index_map: invalid key " + kv.first);
    }
    for (const auto& kv : _entries) {
      const Entry& e = kv.second;
      if (e.start < 0 || e.size < 8 || (uint64_t)(e.start + e.size) > _psf->size()) throw QueryException("index_map: invalid entry for " + kv.first);
      uint64_t marker = 0;
      for (int i = 0; i < 8; ++i) marker = (marker << 8) | (*_psf)[(size_t)e.start + (size_t)i];
      if (marker != kMagicMarker) throw QueryException("columns.psf: missing magic marker at " + std::to_string(e.start) + " (" + kv.first + ")");
    }
  }

  // v1 extension / v3 index id of the same index
  Slice get(const std::string& column, const char* v1Extension, const char* v3Index, std::vector<std::shared_ptr<std::vector<uint8_t>>>* keep) {
    Slice s;
    if (_v3) {
      auto it = _entries.find(column + "." + v3Index);
      if (it == _entries.end()) return s;
      s.data = _psf->data() + it->second.start + 8;
      s.size = (uint64_t)it->second.size - 8;
      if (std::find(keep->begin(), keep->end(), _psf) == keep->end()) keep->push_back(_psf);
      return s;
    }
    const std::string path = _dir + "/" + column + v1Extension;
    if (!fileExists(path)) return s;
    auto buf = readFile(path);
    keep->push_back(buf);
    s.data = buf->data();
    s.size = buf->size();
    return s;
  }

 private:
  struct Entry { long long start = -1, size = -1; };
  std::string _dir;
  bool _v3;
  std::shared_ptr<std::vector<uint8_t>> _psf;
  std::map<std::string, Entry> _entries;
};

int32_t beInt(const uint8_t* p) { return (int32_t)(((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3]); }

bool parseBool(const std::map<std::string, std::string>& m, const std::string& key, bool dflt) {
  auto it = m.find(key);
  if (it == m.end()) return dflt;
  return it->second == "true" || it->second == "TRUE";
}
int parseIntKey(const std::map<std::string, std::string>& m, const std::string& key, int dflt) {
  auto it = m.find(key);
  return it == m.end() || it->second.empty() ? dflt : atoi(it->second.c_str());
}

}  // namespace

std::unique_ptr<ImmutableSegment> loadSegmentDirectory(const std::string& indexDir, std::vector<std::string>* notOffloaded) {
  // SegmentDirectoryPaths.findMetadataFile: the v3 sub-directory wins when it exists
  std::string dir = indexDir;
  bool v3 = false;
  if (fileExists(indexDir + "/v3/metadata.properties")) { dir = indexDir + "/v3"; v3 = true; }
  else if (fileExists(indexDir + "/columns.psf")) v3 = true;
  if (!fileExists(dir + "/metadata.properties")) throw QueryException("no metadata.properties under " + indexDir);
  const auto meta = readProperties(dir + "/metadata.properties");
  auto need = [&](const std::string& k) { auto it = meta.find(k); if (it == meta.end()) throw QueryException("metadata.properties: missing " + k); return it->second; };
  const int totalDocs = atoi(need("segment.total.docs").c_str());
  std::string name = meta.count("segment.name") ? meta.at("segment.name") : indexDir;
  // V1Constants.Str.DEFAULT_STRING_PAD_CHAR is '\0'; segments written before that carry '%' (segment.padding.character)
  char pad = '\0';
  if (meta.count("segment.padding.character") && !meta.at("segment.padding.character").empty()) {
    const std::string& pc = meta.at("segment.padding.character");
    pad = pc == "\\u0000" || pc == "u0000" ? '\0' : pc[0];
  } else if (!meta.count("segment.padding.character")) {
    pad = '%';   // SegmentMetadataImpl: legacy segments without the key used '%'
  }
  auto seg = std::make_unique<ImmutableSegment>(name, totalDocs);
  IndexDirectory index(dir, v3);

  // column names: every "column.<name>.dataType" key (names may contain dots)
  std::vector<std::string> columns;
  const std::string prefix = "column.", suffix = ".dataType";
  for (const auto& kv : meta) {
    if (kv.first.compare(0, prefix.size(), prefix) != 0 || kv.first.size() <= prefix.size() + suffix.size()) continue;
    if (kv.first.compare(kv.first.size() - suffix.size(), suffix.size(), suffix) != 0) continue;
    columns.push_back(kv.first.substr(prefix.size(), kv.first.size() - prefix.size() - suffix.size()));
  }
  std::sort(columns.begin(), columns.end());

  for (const std::string& col : columns) {
    const std::string k = "column." + col + ".";
    auto skip = [&](const std::string& why) { if (notOffloaded) notOffloaded->push_back(col + ": " + why); };
    const std::string type = meta.at(k + "dataType");
    if (!parseBool(meta, k + "isSingleValues", true)) { skip("multi-value column"); continue; }
    DataSource ds;
    ds.name = col;
    if (type == "INT" || type == "BOOLEAN") ds.dataType = DataType::INT;               // FieldSpec.DataType.getStoredType
    else if (type == "LONG" || type == "TIMESTAMP") ds.dataType = DataType::LONG;
    else if (type == "FLOAT") ds.dataType = DataType::FLOAT;
    else if (type == "DOUBLE") ds.dataType = DataType::DOUBLE;
    else if (type == "STRING") ds.dataType = DataType::STRING;
    else { skip("data type " + type); continue; }
    ds.hasDictionary = parseBool(meta, k + "hasDictionary", true);
    ds.cardinality = parseIntKey(meta, k + "cardinality", 0);
    ds.bitsPerElement = parseIntKey(meta, k + "bitsPerElement", 0);
    const bool sorted = parseBool(meta, k + "isSorted", false);
    const int valueBytes = ds.dataType == DataType::INT || ds.dataType == DataType::FLOAT ? 4 : 8;
    std::vector<std::shared_ptr<std::vector<uint8_t>>> keep;

    if (ds.hasDictionary) {
      const Slice dict = index.get(col, ".dict", "dictionary", &keep);
      if (!dict.data || ds.cardinality < 1) { skip("dictionary buffer not found"); continue; }
      if (ds.dataType == DataType::STRING) {
        const int width = parseIntKey(meta, k + "lengthOfEachEntry", 0);
        if (width <= 0 || dict.size != (uint64_t)width * (uint64_t)ds.cardinality) { skip("variable-length string dictionary"); continue; }
        std::vector<std::string> values;
        for (int d = 0; d < ds.cardinality; ++d) {
          const char* p = reinterpret_cast<const char*>(dict.data) + (size_t)d * (size_t)width;
          size_t len = (size_t)width;
          while (len > 0 && p[len - 1] == pad) len--;        // StringDictionary un-pads with the segment's padding character
          values.emplace_back(p, len);
        }
        ds.dictionary = std::make_shared<StringDictionary>(std::move(values));
      } else {
        if (dict.size != (uint64_t)valueBytes * (uint64_t)ds.cardinality) { skip("dictionary size does not match the data type"); continue; }
        ds.dictionaryBuffer = dict.data; ds.dictionaryBufferSize = dict.size;
        switch (ds.dataType) {
          case DataType::INT: ds.dictionary = std::make_shared<IntDictionary>(dict.data, ds.cardinality); break;
          case DataType::LONG: ds.dictionary = std::make_shared<LongDictionary>(dict.data, ds.cardinality); break;
          case DataType::FLOAT: ds.dictionary = std::make_shared<FloatDictionary>(dict.data, ds.cardinality); break;
          default: ds.dictionary = std::make_shared<DoubleDictionary>(dict.data, ds.cardinality); break;
        }
      }
      // Which reader the reference picks is decided by the metadata, not by the file size (ForwardIndexReaderFactory.java:75-91:
      // dictionary + single-value + isSorted -> SortedIndexReaderImpl over [start, end] pairs).  v1 names the file after the format
      // (.sv.sorted.fwd / .sv.unsorted.fwd); in v3 the index map has one forward_index entry and isSorted says what it holds.
      Slice fwd = index.get(col, sorted ? ".sv.sorted.fwd" : ".sv.unsorted.fwd", "forward_index", &keep);
      bool pairFile = sorted && fwd.data != nullptr;
      bool sniffed = false;
      if (!fwd.data && !v3) {
        // the other extension had to be used (metadata and file name disagree): only here the size decides
        fwd = index.get(col, sorted ? ".sv.unsorted.fwd" : ".sv.sorted.fwd", "forward_index", &keep);
        sniffed = true;
      }
      if (!fwd.data && totalDocs > 0) { skip("forward index not found"); continue; }
      if (ds.bitsPerElement < 1) ds.bitsPerElement = ph_num_bits_per_value(ds.cardinality - 1);
      const uint64_t packedSize = (uint64_t)ph_fixedbit_size(totalDocs, ds.bitsPerElement);
      const uint64_t pairSize = 2ull * 4ull * (uint64_t)ds.cardinality;
      if (sniffed) pairFile = fwd.size == pairSize && fwd.size != packedSize;
      if (pairFile && fwd.size != pairSize) { skip("sorted forward index size does not match 8 * cardinality"); continue; }
      if (pairFile) {
        // SortedIndexReaderImpl: [startDocId, endDocId] per dictId -> the dictId of every doc, packed like an unsorted column
        std::vector<int32_t> ids((size_t)totalDocs, 0);
        ds.isSorted = true;
        for (int d = 0; d < ds.cardinality; ++d) {
          const int32_t s = beInt(fwd.data + 8 * (size_t)d), e = beInt(fwd.data + 8 * (size_t)d + 4);
          if (s < 0 || e >= totalDocs || s > e + 1) throw QueryException("column " + col + ": corrupt sorted forward index");
          ds.sortedDocIdRanges.push_back(s);
          ds.sortedDocIdRanges.push_back(e);
          for (int32_t doc = s; doc <= e; ++doc) ids[(size_t)doc] = d;
        }
        auto packed = std::make_shared<std::vector<uint8_t>>((size_t)packedSize, 0);
        if (totalDocs > 0) ph_fixedbit_pack(ids.data(), totalDocs, ds.bitsPerElement, packed->data(), 4);
        keep.push_back(packed);
        fwd.data = packed->data(); fwd.size = packed->size();
      } else if (fwd.size != packedSize) {
        skip("forward index size does not match numDocs * bitsPerElement");
        continue;
      }
      ds.forwardIndex = fwd.data; ds.forwardIndexSize = fwd.size;
      const Slice nulls = index.get(col, ".bitmap.nullvalue", "nullvalue_vector", &keep);   // V1Constants.Indexes.NULLVALUE_VECTOR_FILE_EXTENSION
      ds.nullValueVector = nulls.data; ds.nullValueVectorSize = nulls.size;
      const Slice inv = index.get(col, ".bitmap.inv", "inverted_index", &keep);
      ds.hasInvertedIndex = inv.data != nullptr && inv.size > 0;
      ds.invertedIndex = inv.data; ds.invertedIndexSize = inv.size;
    } else {
      if (ds.dataType == DataType::STRING) { skip("raw STRING column"); continue; }
      const Slice fwd = index.get(col, ".sv.raw.fwd", "forward_index", &keep);
      if (!fwd.data) { skip("raw forward index not found"); continue; }
      if (fwd.size >= 28 && beInt(fwd.data + 20) != 0) { skip("compressed raw forward index"); continue; }   // PASS_THROUGH only
      ds.forwardIndex = fwd.data; ds.forwardIndexSize = fwd.size;
      ds.bitsPerElement = 8 * valueBytes;
      const Slice nulls = index.get(col, ".bitmap.nullvalue", "nullvalue_vector", &keep);
      ds.nullValueVector = nulls.data; ds.nullValueVectorSize = nulls.size;
    }
    for (auto& b : keep) seg->keepAlive(b);
    seg->addDataSource(std::move(ds));
  }
  return seg;
}

}  // namespace pinot
