// query_context.cpp -- SQL subset -> QueryContext, dictionaries, predicate evaluators.
// Mirrors (paths under /root/reference/):
//   pinot-core/.../query/request/context/utils/QueryContextConverterUtils.java (getQueryContext)
//   pinot-common/.../request/context/RequestContextUtils.java (filter tree, BETWEEN -> RANGE, <,<=,>,>= -> RANGE)
//   pinot-core/.../operator/filter/predicate/{Equals,NotEquals,In,NotIn,Range}PredicateEvaluatorFactory.java
#include <algorithm>
#include <cctype>
#include <climits>
#include <cstdlib>
#include <cstring>

#include "pinot_host.h"

namespace pinot {

// ---------------------------------------------------------------------------------------------------------------
// Dictionaries
// ---------------------------------------------------------------------------------------------------------------
static inline int32_t be_int(const uint8_t* p) {
  return (int32_t)(((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3]);
}

int32_t IntDictionary::getIntValue(int dictId) const { return be_int(_buffer + (size_t)dictId * 4); }

// BaseImmutableDictionary.binarySearch(int), BaseImmutableDictionary.java:124-140
int IntDictionary::binarySearch(int32_t value) const {
  int low = 0, high = _length - 1;
  while (low <= high) {
    const int mid = (int)(((unsigned)low + (unsigned)high) >> 1);
    const int32_t midValue = getIntValue(mid);
    if (midValue < value) low = mid + 1;
    else if (midValue > value) high = mid - 1;
    else return mid;
  }
  return -(low + 1);
}

static bool parseInt32(const std::string& s, int32_t* out) {
  if (s.empty()) return false;
  char* end = nullptr;
  errno = 0;
  const long long v = strtoll(s.c_str(), &end, 10);
  if (errno != 0 || *end != 0 || v < INT_MIN || v > INT_MAX) return false;
  *out = (int32_t)v;
  return true;
}

// IntDictionary.insertionIndexOf(String) = binarySearch(Integer.parseInt(stringValue)), IntDictionary.java:38-46.
// The reference converts the literal with PredicateUtils.getStoredValue; a non-integer literal on an INT column is a
// BadQueryRequestException there, a QueryException here.
int IntDictionary::insertionIndexOf(const std::string& stringValue) const {
  int32_t v;
  if (!parseInt32(stringValue, &v)) throw QueryException("Cannot convert value: '" + stringValue + "' to INT");
  return binarySearch(v);
}

const char* dataTypeName(DataType t) {
  switch (t) { case DataType::INT: return "INT"; case DataType::LONG: return "LONG"; case DataType::FLOAT: return "FLOAT";
               case DataType::DOUBLE: return "DOUBLE"; default: return "STRING"; }
}

static inline int64_t be_long(const uint8_t* p) { return (int64_t)(((uint64_t)(uint32_t)be_int(p) << 32) | (uint64_t)(uint32_t)be_int(p + 4)); }
int64_t LongDictionary::getLongValue(int dictId) const { return be_long(_buffer + (size_t)dictId * 8); }
float FloatDictionary::getFloatValue(int dictId) const { const int32_t b = be_int(_buffer + (size_t)dictId * 4); float f; memcpy(&f, &b, 4); return f; }
double DoubleDictionary::getDoubleValue(int dictId) const { const int64_t b = be_long(_buffer + (size_t)dictId * 8); double d; memcpy(&d, &b, 8); return d; }
std::string FloatDictionary::getStringValue(int dictId) const { char buf[64]; snprintf(buf, sizeof(buf), "%.9g", (double)getFloatValue(dictId)); return buf; }
std::string DoubleDictionary::getStringValue(int dictId) const { char buf[64]; snprintf(buf, sizeof(buf), "%.17g", getDoubleValue(dictId)); return buf; }

// BaseImmutableDictionary.binarySearch(long | float | double), BaseImmutableDictionary.java:142-195
template <typename T, typename Get>
static int binarySearchTyped(int length, T value, Get get) {
  int low = 0, high = length - 1;
  while (low <= high) {
    const int mid = (int)(((unsigned)low + (unsigned)high) >> 1);
    const T midValue = get(mid);
    if (midValue < value) low = mid + 1;
    else if (midValue > value) high = mid - 1;
    else return mid;
  }
  return -(low + 1);
}
bool parseInt64(const std::string& s, int64_t* out) {
  if (s.empty()) return false;
  char* end = nullptr;
  errno = 0;
  const long long v = strtoll(s.c_str(), &end, 10);
  if (errno != 0 || *end != 0) return false;
  *out = (int64_t)v;
  return true;
}
bool parseFloating(const std::string& s, double* out) {
  if (s.empty()) return false;
  char* end = nullptr;
  errno = 0;
  const double v = strtod(s.c_str(), &end);
  if (*end != 0) return false;
  *out = v;
  return true;
}
int LongDictionary::insertionIndexOf(const std::string& stringValue) const {
  int64_t v;
  if (!parseInt64(stringValue, &v)) throw QueryException("Cannot convert value: '" + stringValue + "' to LONG");
  return binarySearchTyped<int64_t>(_length, v, [this](int d) { return getLongValue(d); });
}
int FloatDictionary::insertionIndexOf(const std::string& stringValue) const {
  double v;
  if (!parseFloating(stringValue, &v)) throw QueryException("Cannot convert value: '" + stringValue + "' to FLOAT");
  return binarySearchTyped<float>(_length, (float)v, [this](int d) { return getFloatValue(d); });     // Float.parseFloat rounds to float
}
int DoubleDictionary::insertionIndexOf(const std::string& stringValue) const {
  double v;
  if (!parseFloating(stringValue, &v)) throw QueryException("Cannot convert value: '" + stringValue + "' to DOUBLE");
  return binarySearchTyped<double>(_length, v, [this](int d) { return getDoubleValue(d); });
}

int StringDictionary::insertionIndexOf(const std::string& stringValue) const {
  if (!_sorted) {
    for (size_t i = 0; i < _values.size(); ++i) if (_values[i] == stringValue) return (int)i;
    return -1;     // absent; the insertion point is meaningless here and RANGE predicates are rejected before they ask for it
  }
  auto it = std::lower_bound(_values.begin(), _values.end(), stringValue);
  if (it != _values.end() && *it == stringValue) return (int)(it - _values.begin());
  return -((int)(it - _values.begin()) + 1);
}

// ---------------------------------------------------------------------------------------------------------------
// SQL subset parser
// ---------------------------------------------------------------------------------------------------------------
namespace {

struct Token {
  enum Kind { END, IDENT, NUMBER, STRING, SYMBOL } kind = END;
  std::string text;
};

class Lexer {
 public:
  explicit Lexer(const std::string& s) : _s(s) { advance(); }
  const Token& peek() const { return _tok; }
  Token next() { Token t = _tok; advance(); return t; }
  bool isKeyword(const char* kw) const {
    if (_tok.kind != Token::IDENT) return false;
    std::string u = _tok.text;
    for (auto& c : u) c = (char)toupper((unsigned char)c);
    return u == kw;
  }
  bool acceptKeyword(const char* kw) { if (isKeyword(kw)) { advance(); return true; } return false; }
  bool acceptSymbol(const char* sym) { if (_tok.kind == Token::SYMBOL && _tok.text == sym) { advance(); return true; } return false; }
  void expectSymbol(const char* sym) { if (!acceptSymbol(sym)) throw QueryException(std::string("expected '") + sym + "' near '" + _tok.text + "'"); }
  void expectKeyword(const char* kw) { if (!acceptKeyword(kw)) throw QueryException(std::string("expected ") + kw + " near '" + _tok.text + "'"); }

 private:
  void advance() {
    while (_pos < _s.size() && isspace((unsigned char)_s[_pos])) _pos++;
    _tok = Token();
    if (_pos >= _s.size()) return;
    const char c = _s[_pos];
    if (isalpha((unsigned char)c) || c == '_' || c == '"') {
      if (c == '"') {
        size_t e = _s.find('"', _pos + 1);
        if (e == std::string::npos) throw QueryException("unterminated quoted identifier");
        _tok.kind = Token::IDENT; _tok.text = _s.substr(_pos + 1, e - _pos - 1); _pos = e + 1;
        return;
      }
      size_t e = _pos;
      while (e < _s.size() && (isalnum((unsigned char)_s[e]) || _s[e] == '_' || _s[e] == '.')) e++;
      _tok.kind = Token::IDENT; _tok.text = _s.substr(_pos, e - _pos); _pos = e;
    } else if (isdigit((unsigned char)c) || (c == '-' && _pos + 1 < _s.size() && isdigit((unsigned char)_s[_pos + 1]))) {
      // integer or decimal literal with an optional exponent (what Calcite hands the reference as a numeric literal)
      size_t e = _pos + 1;
      while (e < _s.size() && isdigit((unsigned char)_s[e])) e++;
      if (e + 1 < _s.size() && _s[e] == '.' && isdigit((unsigned char)_s[e + 1])) { e++; while (e < _s.size() && isdigit((unsigned char)_s[e])) e++; }
      if (e < _s.size() && (_s[e] == 'e' || _s[e] == 'E')) {
        size_t x = e + 1;
        if (x < _s.size() && (_s[x] == '+' || _s[x] == '-')) x++;
        if (x < _s.size() && isdigit((unsigned char)_s[x])) { while (x < _s.size() && isdigit((unsigned char)_s[x])) x++; e = x; }
      }
      _tok.kind = Token::NUMBER; _tok.text = _s.substr(_pos, e - _pos); _pos = e;
    } else if (c == '\'') {
      std::string out;
      size_t e = _pos + 1;
      for (;;) {
        if (e >= _s.size()) throw QueryException("unterminated string literal");
        if (_s[e] == '\'') { if (e + 1 < _s.size() && _s[e + 1] == '\'') { out += '\''; e += 2; continue; } break; }
        out += _s[e++];
      }
      _tok.kind = Token::STRING; _tok.text = out; _pos = e + 1;
    } else {
      static const char* two[] = {"<=", ">=", "<>", "!="};
      for (const char* t : two) if (_s.compare(_pos, 2, t) == 0) { _tok.kind = Token::SYMBOL; _tok.text = t; _pos += 2; return; }
      _tok.kind = Token::SYMBOL; _tok.text = std::string(1, c); _pos++;
    }
  }
  const std::string& _s;
  size_t _pos = 0;
  Token _tok;
};

std::string literal(Lexer& lx) {
  const Token t = lx.next();
  if (t.kind != Token::NUMBER && t.kind != Token::STRING) throw QueryException("expected a literal near '" + t.text + "'");
  return t.text;
}

FilterContext parseOr(Lexer& lx);

FilterContext parsePredicate(Lexer& lx) {
  if (lx.acceptSymbol("(")) {
    FilterContext f = parseOr(lx);
    lx.expectSymbol(")");
    return f;
  }
  const Token col = lx.next();
  if (col.kind != Token::IDENT) throw QueryException("expected a column name near '" + col.text + "'");
  FilterContext f;
  f.type = FilterContext::Type::PREDICATE;
  Predicate& p = f.predicate;
  p.column = col.text;
  if (lx.acceptKeyword("BETWEEN")) {
    // RequestContextUtils: BETWEEN a AND b -> RANGE [a, b]
    p.type = Predicate::Type::RANGE;
    p.lowerBound = literal(lx); p.lowerInclusive = true;
    lx.expectKeyword("AND");
    p.upperBound = literal(lx); p.upperInclusive = true;
    return f;
  }
  if (lx.acceptKeyword("IS")) {
    // IS NULL / IS NOT NULL (Predicate.Type.IS_NULL / IS_NOT_NULL)
    const bool isNot = lx.acceptKeyword("NOT");
    lx.expectKeyword("NULL");
    p.type = isNot ? Predicate::Type::IS_NOT_NULL : Predicate::Type::IS_NULL;
    return f;
  }
  bool negated = lx.acceptKeyword("NOT");
  if (lx.acceptKeyword("IN")) {
    p.type = negated ? Predicate::Type::NOT_IN : Predicate::Type::IN;
    lx.expectSymbol("(");
    do { p.values.push_back(literal(lx)); } while (lx.acceptSymbol(","));
    lx.expectSymbol(")");
    return f;
  }
  if (negated && lx.acceptKeyword("BETWEEN")) {
    // col NOT BETWEEN a AND b -> NOT(RANGE [a, b]) (CalciteSqlParser rewrites it into a NOT filter over the BETWEEN)
    FilterContext inner;
    inner.type = FilterContext::Type::PREDICATE;
    inner.predicate.column = col.text;
    inner.predicate.type = Predicate::Type::RANGE;
    inner.predicate.lowerBound = literal(lx); inner.predicate.lowerInclusive = true;
    lx.expectKeyword("AND");
    inner.predicate.upperBound = literal(lx); inner.predicate.upperInclusive = true;
    FilterContext n;
    n.type = FilterContext::Type::NOT;
    n.children.push_back(std::move(inner));
    return n;
  }
  if (negated) throw QueryException("expected IN or BETWEEN after NOT");
  const Token op = lx.next();
  if (op.kind != Token::SYMBOL) throw QueryException("expected a comparison operator near '" + op.text + "'");
  const std::string v = literal(lx);
  if (op.text == "=") { p.type = Predicate::Type::EQ; p.values = {v}; }
  else if (op.text == "!=" || op.text == "<>") { p.type = Predicate::Type::NOT_EQ; p.values = {v}; }
  else if (op.text == ">") { p.type = Predicate::Type::RANGE; p.lowerBound = v; p.lowerInclusive = false; }
  else if (op.text == ">=") { p.type = Predicate::Type::RANGE; p.lowerBound = v; p.lowerInclusive = true; }
  else if (op.text == "<") { p.type = Predicate::Type::RANGE; p.upperBound = v; p.upperInclusive = false; }
  else if (op.text == "<=") { p.type = Predicate::Type::RANGE; p.upperBound = v; p.upperInclusive = true; }
  else throw QueryException("unsupported operator '" + op.text + "'");
  return f;
}

FilterContext parseNot(Lexer& lx) {
  if (lx.acceptKeyword("NOT")) {
    FilterContext f;
    f.type = FilterContext::Type::NOT;
    f.children.push_back(parseNot(lx));
    return f;
  }
  return parsePredicate(lx);
}

FilterContext parseAnd(Lexer& lx) {
  FilterContext first = parseNot(lx);
  if (!lx.isKeyword("AND")) return first;
  FilterContext f;
  f.type = FilterContext::Type::AND;
  f.children.push_back(std::move(first));
  while (lx.acceptKeyword("AND")) f.children.push_back(parseNot(lx));
  return f;
}

FilterContext parseOr(Lexer& lx) {
  FilterContext first = parseAnd(lx);
  if (!lx.isKeyword("OR")) return first;
  FilterContext f;
  f.type = FilterContext::Type::OR;
  f.children.push_back(std::move(first));
  while (lx.acceptKeyword("OR")) f.children.push_back(parseAnd(lx));
  return f;
}

}  // namespace

// canonical text of a filter tree (used to find aggregations that share a FILTER clause)
static std::string filterToString(const FilterContext& f) {
  switch (f.type) {
    case FilterContext::Type::PREDICATE: {
      const Predicate& p = f.predicate;
      std::string s = p.column + "#" + std::to_string((int)p.type) + "#" + p.lowerBound + (p.lowerInclusive ? "[" : "(") + p.upperBound + (p.upperInclusive ? "]" : ")");
      for (const auto& v : p.values) s += "|" + v;
      return "{" + s + "}";
    }
    case FilterContext::Type::NOT: return "NOT(" + filterToString(f.children.at(0)) + ")";
    default: {
      std::string s = f.type == FilterContext::Type::AND ? "AND(" : "OR(";
      for (const auto& c : f.children) s += filterToString(c) + ",";
      return s + ")";
    }
  }
}

QueryContext getQueryContext(const std::string& sql) {
  Lexer lx(sql);
  QueryContext q;
  // SET key = value; statements ahead of the query are query options (CalciteSqlParser SET statements -> QueryOptions)
  while (lx.acceptKeyword("SET")) {
    const Token key = lx.next();
    if (key.kind != Token::IDENT) throw QueryException("expected an option name near '" + key.text + "'");
    lx.expectSymbol("=");
    const Token value = lx.next();
    if (value.kind == Token::END) throw QueryException("expected an option value");
    lx.expectSymbol(";");
    std::string k = key.text, v = value.text;
    for (auto& c : k) c = (char)tolower((unsigned char)c);
    for (auto& c : v) c = (char)tolower((unsigned char)c);
    if (k == "enablenullhandling") q.nullHandlingEnabled = v == "true";
    else if (k == "numgroupslimit") q.numGroupsLimit = atoi(v.c_str());
    else if (k == "gpuexactfilterstats") q.gpuExactFilterStats = v != "false";
    // QueryOptionsUtils: minSegmentGroupTrimSize / minServerGroupTrimSize / groupTrimThreshold override the plan maker's defaults
    else if (k == "minsegmentgrouptrimsize") q.minSegmentGroupTrimSize = atoi(v.c_str());
    else if (k == "minservergrouptrimsize") q.minServerGroupTrimSize = atoi(v.c_str());
    else if (k == "grouptrimthreshold") q.groupTrimThreshold = atoi(v.c_str());
    else throw UnsupportedOperationException("query option '" + key.text + "' is not handled on this path");
  }
  lx.expectKeyword("SELECT");
  std::vector<std::string> selectedColumns;      // plain identifiers of the SELECT list, resolved against GROUP BY below
  do {
    const Token fn = lx.next();
    if (fn.kind != Token::IDENT) throw QueryException("expected an aggregation function near '" + fn.text + "'");
    if (!(lx.peek().kind == Token::SYMBOL && lx.peek().text == "(")) {
      // SELECT column11, SUM(column1) ... GROUP BY column11: a group-by column in the result (anything else is a selection query)
      if (lx.acceptKeyword("AS")) lx.next();
      q.selectExpressions.push_back(SelectExpression{false, (int)selectedColumns.size()});
      selectedColumns.push_back(fn.text);
      continue;
    }
    std::string u = fn.text;
    for (auto& c : u) c = (char)toupper((unsigned char)c);
    AggregationExpression e;
    if (u == "COUNT") e.function = AggregationFunctionType::COUNT;
    else if (u == "SUM") e.function = AggregationFunctionType::SUM;
    else if (u == "MIN") e.function = AggregationFunctionType::MIN;
    else if (u == "MAX") e.function = AggregationFunctionType::MAX;
    else if (u == "AVG") e.function = AggregationFunctionType::AVG;
    else throw UnsupportedOperationException("only COUNT/SUM/MIN/MAX/AVG are offloaded, got " + fn.text);
    lx.expectSymbol("(");
    if (lx.acceptSymbol("*")) e.column = "*";
    else {
      const Token c = lx.next();
      if (c.kind != Token::IDENT) throw UnsupportedOperationException("only identifier arguments are offloaded (ProjectPlanNode.java:85-86)");
      e.column = c.text;
      if (!(lx.peek().kind == Token::SYMBOL && lx.peek().text == ")"))
        throw UnsupportedOperationException("only identifier arguments are offloaded (transform expressions keep the CPU plan, ProjectPlanNode.java:85-86)");
    }
    lx.expectSymbol(")");
    if (e.function != AggregationFunctionType::COUNT && e.column == "*") throw QueryException("'*' is only valid in COUNT(*)");
    if (lx.acceptKeyword("FILTER")) {
      lx.expectSymbol("(");
      lx.expectKeyword("WHERE");
      e.filter = parseOr(lx);
      e.hasFilter = true;
      e.filterText = filterToString(e.filter);
      lx.expectSymbol(")");
    }
    if (lx.acceptKeyword("AS")) {
      const Token alias = lx.next();
      if (alias.kind != Token::IDENT) throw QueryException("expected an alias near '" + alias.text + "'");
      e.alias = alias.text;
    }
    q.selectExpressions.push_back(SelectExpression{true, (int)q.aggregations.size()});
    q.aggregations.push_back(e);
  } while (lx.acceptSymbol(","));
  lx.expectKeyword("FROM");
  const Token t = lx.next();
  if (t.kind != Token::IDENT) throw QueryException("expected a table name");
  q.tableName = t.text;
  if (lx.acceptKeyword("WHERE")) {
    q.hasFilter = true;
    q.filter = parseOr(lx);
  }
  if (lx.acceptKeyword("GROUP")) {
    lx.expectKeyword("BY");
    do {
      const Token c = lx.next();
      if (c.kind != Token::IDENT) throw QueryException("expected a group-by column");
      q.groupByExpressions.push_back(c.text);
    } while (lx.acceptSymbol(","));
  }
  // identifiers of the SELECT list must be group-by columns; without aggregations nothing is offloaded
  if (!selectedColumns.empty() && q.groupByExpressions.empty())
    throw UnsupportedOperationException("only aggregation / group-by queries are offloaded (selection stays on the CPU plan)");
  for (auto& se : q.selectExpressions) {
    if (se.isAggregation) continue;
    int found = -1;
    for (size_t g = 0; g < q.groupByExpressions.size(); ++g) if (q.groupByExpressions[g] == selectedColumns[(size_t)se.index]) found = (int)g;
    if (found < 0) throw QueryException("'" + selectedColumns[(size_t)se.index] + "' should appear in GROUP BY clause.");
    se.index = found;
  }
  if (lx.acceptKeyword("ORDER")) {
    // ORDER BY <group-by column | one of the selected aggregations> [ASC | DESC] [NULLS FIRST | NULLS LAST] [, ...]
    lx.expectKeyword("BY");
    if (q.groupByExpressions.empty()) throw UnsupportedOperationException("ORDER BY without GROUP BY is a selection query (CPU plan)");
    do {
      const Token first = lx.next();
      if (first.kind != Token::IDENT) throw QueryException("expected an ORDER BY expression near '" + first.text + "'");
      OrderByExpressionContext ob;
      if (lx.acceptSymbol("(")) {
        std::string u = first.text;
        for (auto& c : u) c = (char)toupper((unsigned char)c);
        std::string column;
        if (lx.acceptSymbol("*")) column = "*";
        else {
          const Token c = lx.next();
          if (c.kind != Token::IDENT) throw UnsupportedOperationException("only identifier arguments are offloaded (post-aggregation ORDER BY keeps the CPU plan)");
          column = c.text;
        }
        lx.expectSymbol(")");
        int found = -1;
        for (size_t a = 0; a < q.aggregations.size() && found < 0; ++a) {
          const AggregationExpression& e = q.aggregations[a];
          static const char* const names[] = {"COUNT", "SUM", "MIN", "MAX", "AVG"};
          if (!e.hasFilter && u == names[(int)e.function] && (e.column == column || (e.function == AggregationFunctionType::COUNT && (column == "*" || e.column == "*")))) found = (int)a;
        }
        if (found < 0) {
          // an aggregation that is only ordered by joins the query's functions after the selected ones (QueryContext.Builder
          // generateAggregationFunctions: SELECT expressions first, then HAVING / ORDER BY); the result does not show it
          AggregationExpression e;
          int kind = -1;
          static const char* const names[] = {"COUNT", "SUM", "MIN", "MAX", "AVG"};
          for (int k = 0; k < 5; ++k) if (u == names[k]) kind = k;
          if (kind < 0) throw UnsupportedOperationException("only COUNT/SUM/MIN/MAX/AVG are offloaded, got " + first.text);
          e.function = (AggregationFunctionType)kind;
          e.column = column;
          if (e.function != AggregationFunctionType::COUNT && column == "*") throw QueryException("'*' is only valid in COUNT(*)");
          found = (int)q.aggregations.size();
          q.aggregations.push_back(e);
        }
        ob.isAggregation = true;
        ob.index = found;
      } else {
        int found = -1;
        for (size_t g = 0; g < q.groupByExpressions.size(); ++g) if (q.groupByExpressions[g] == first.text) found = (int)g;
        if (found < 0) {
          // an alias of a selected aggregation (CalciteSqlParser rewrites ORDER BY aliases into the aliased expression)
          for (size_t a = 0; a < q.aggregations.size() && found < 0; ++a) if (!q.aggregations[a].alias.empty() && q.aggregations[a].alias == first.text) found = (int)a;
          if (found >= 0 && q.aggregations[(size_t)found].hasFilter) throw UnsupportedOperationException("ORDER BY a filtered aggregation keeps the CPU plan");
          if (found >= 0) ob.isAggregation = true;
        }
        if (found < 0) throw QueryException("Failed to find ORDER-BY expression: " + first.text + " in the GROUP-BY clause");      // TableResizer.java:150-151
        ob.index = found;
      }
      if (lx.acceptKeyword("DESC")) ob.isAsc = false;
      else (void)lx.acceptKeyword("ASC");
      if (lx.acceptKeyword("NULLS")) {
        if (lx.acceptKeyword("LAST")) ob.nullsLast = 1;
        else { lx.expectKeyword("FIRST"); ob.nullsLast = 0; }
      }
      q.orderByExpressions.push_back(ob);
    } while (lx.acceptSymbol(","));
  }
  if (lx.acceptKeyword("LIMIT")) {
    const Token n = lx.next();
    int32_t v = 0;
    if (!parseInt32(n.text, &v) || v < 0) throw QueryException("expected a non-negative LIMIT near '" + n.text + "'");
    q.limit = v;
  }
  if (lx.peek().kind != Token::END) throw UnsupportedOperationException("unsupported clause near '" + lx.peek().text + "'");
  return q;
}

// ---------------------------------------------------------------------------------------------------------------
// Predicate evaluators
// ---------------------------------------------------------------------------------------------------------------
int PredicateEvaluator::getNumMatchingItems() const {
  if (isRange) return std::max(endDictId - startDictId, 0);
  return (int)matchingDictIds.size();
}

PredicateEvaluator getPredicateEvaluator(const Predicate& predicate, const DataSource& ds) {
  PredicateEvaluator ev;
  ev.predicateType = predicate.type;
  // IS_NULL / IS_NOT_NULL have no predicate evaluator in the reference either: FilterPlanNode.java:294-310 turns them into bitmap
  // operators over the null value vector (lowerFilter does the same before it gets here)
  if (predicate.type == Predicate::Type::IS_NULL || predicate.type == Predicate::Type::IS_NOT_NULL)
    throw QueryException("IS NULL / IS NOT NULL are evaluated on the null value vector, not through a predicate evaluator");
  if (!ds.hasDictionary) {
    // raw INT / LONG column: Int / LongRawValueBasedRangePredicateEvaluator (RangePredicateEvaluatorFactory.java:68-81,331-446);
    // EQ is the degenerate range, the other raw evaluators are not offloaded.
    if (ds.dataType != DataType::INT && ds.dataType != DataType::LONG)
      throw UnsupportedOperationException(std::string("predicates on a raw ") + dataTypeName(ds.dataType) + " column are not offloaded");
    const bool isLong = ds.dataType == DataType::LONG;
    auto toInt = [isLong](const std::string& s) {
      if (isLong) { int64_t v; if (!parseInt64(s, &v)) throw QueryException("Cannot convert value: '" + s + "' to LONG"); return v; }
      int32_t v; if (!parseInt32(s, &v)) throw QueryException("Cannot convert value: '" + s + "' to INT"); return (int64_t)v; };
    const int64_t typeMin = isLong ? INT64_MIN : (int64_t)INT_MIN, typeMax = isLong ? INT64_MAX : (int64_t)INT_MAX;
    ev.rawRange = true;
    if (predicate.type == Predicate::Type::RANGE) {
      const bool lowerUnbounded = predicate.lowerBound == "*", upperUnbounded = predicate.upperBound == "*";
      ev.rawLower = lowerUnbounded ? typeMin : toInt(predicate.lowerBound);
      ev.rawUpper = upperUnbounded ? typeMax : toInt(predicate.upperBound);
      // an exclusive bound at the type's extreme is "Invalid range" in the reference (Preconditions.checkArgument)
      if ((!lowerUnbounded && !predicate.lowerInclusive && ev.rawLower == typeMax) || (!upperUnbounded && !predicate.upperInclusive && ev.rawUpper == typeMin))
        throw QueryException("Invalid range");
      if (!lowerUnbounded && !predicate.lowerInclusive) ev.rawLower += 1;
      if (!upperUnbounded && !predicate.upperInclusive) ev.rawUpper -= 1;
    } else if (predicate.type == Predicate::Type::EQ || predicate.type == Predicate::Type::NOT_EQ) {
      ev.rawLower = ev.rawUpper = toInt(predicate.values.at(0));
      ev.exclusive = predicate.type == Predicate::Type::NOT_EQ;
    } else {
      throw UnsupportedOperationException("IN / NOT IN on a raw column is not offloaded");
    }
    if (ev.rawLower > ev.rawUpper) { if (ev.exclusive) ev.alwaysTrue = true; else ev.alwaysFalse = true; }
    return ev;
  }
  const Dictionary& dict = *ds.dictionary;
  switch (predicate.type) {
    case Predicate::Type::EQ:
    case Predicate::Type::NOT_EQ: {
      // EqualsPredicateEvaluatorFactory.java:92-124 / NotEqualsPredicateEvaluatorFactory
      const int d = dict.indexOf(predicate.values.at(0));
      ev.exclusive = predicate.type == Predicate::Type::NOT_EQ;
      ev.isRange = true;
      if (d >= 0) {
        ev.startDictId = d; ev.endDictId = d + 1;
        if (dict.length() == 1) { if (ev.exclusive) ev.alwaysFalse = true; else ev.alwaysTrue = true; }
      } else {
        if (ev.exclusive) ev.alwaysTrue = true; else ev.alwaysFalse = true;
      }
      break;
    }
    case Predicate::Type::IN:
    case Predicate::Type::NOT_IN: {
      // InPredicateEvaluatorFactory.java:161-188 (PredicateUtils.getDictIdSet drops values absent from the dictionary)
      ev.exclusive = predicate.type == Predicate::Type::NOT_IN;
      for (const auto& v : predicate.values) { const int d = dict.indexOf(v); if (d >= 0) ev.matchingDictIds.push_back(d); }
      std::sort(ev.matchingDictIds.begin(), ev.matchingDictIds.end());
      ev.matchingDictIds.erase(std::unique(ev.matchingDictIds.begin(), ev.matchingDictIds.end()), ev.matchingDictIds.end());
      const int n = (int)ev.matchingDictIds.size();
      if (n == 0) { if (ev.exclusive) ev.alwaysTrue = true; else ev.alwaysFalse = true; }
      else if (n == dict.length()) { if (ev.exclusive) ev.alwaysFalse = true; else ev.alwaysTrue = true; }
      break;
    }
    case Predicate::Type::IS_NULL:
    case Predicate::Type::IS_NOT_NULL:
      break;      // rejected above
    case Predicate::Type::RANGE: {
      // SortedDictionaryBasedRangePredicateEvaluator, RangePredicateEvaluatorFactory.java:126-169
      if (!dict.isSorted()) throw UnsupportedOperationException("range predicate on a dictionary whose un-padded values are not sorted");
      ev.isRange = true;
      if (predicate.lowerBound == "*") ev.startDictId = 0;
      else {
        const int ins = dict.insertionIndexOf(predicate.lowerBound);
        if (ins < 0) ev.startDictId = -(ins + 1);
        else ev.startDictId = predicate.lowerInclusive ? ins : ins + 1;
      }
      if (predicate.upperBound == "*") ev.endDictId = dict.length();
      else {
        const int ins = dict.insertionIndexOf(predicate.upperBound);
        if (ins < 0) ev.endDictId = -(ins + 1);
        else ev.endDictId = predicate.upperInclusive ? ins + 1 : ins;
      }
      const int n = std::max(ev.endDictId - ev.startDictId, 0);
      if (n == 0) ev.alwaysFalse = true;
      else if (n == dict.length()) ev.alwaysTrue = true;
      break;
    }
  }
  return ev;
}

}  // namespace pinot
