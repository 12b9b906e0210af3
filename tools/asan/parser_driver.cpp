#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include "../../include/pinot_host_c.h"
int main() {
  const char* qs[] = {
    "SELECT COUNT(*), SUM(column1), MAX(column3), MIN(column6), AVG(column7) FROM testTable",
    "select sum(a) as s from t where a > 1 and (b in (1, 2, -3) or not c between 5 and 9) and d <> 'x''y' group by k1, k2",
    "SET enableNullHandling = true; SELECT COUNT(a), SUM(a) FROM t WHERE a IS NOT NULL AND (b IS NULL OR NOT c > 3)",
    "SELECT SUM(a) FILTER (WHERE b > 3 AND c = 'x'), COUNT(*) FILTER(WHERE b > 3 AND c = 'x'), MAX(a) FROM t WHERE d < 5 GROUP BY e",
    "SELECT a FROM t", "SELECT SUM(a + 1) FROM t", "SELECT SUM(*) FROM t", "SELECT SUM(a) FROM", "SELECT SUM(a) FROM t WHERE a >",
    "SELECT SUM(a) FILTER (b > 3) FROM t", "SELECT COUNT(*) FROM t WHERE a IS 3", "SET useStarTree = true; SELECT COUNT(*) FROM t",
    "SET numGroupsLimit = 5; SELECT COUNT(*) FROM t GROUP BY a", "SELECT COUNT(*) FROM t WHERE a IN ()", "", "SELECT", "SELECT COUNT(*) FROM t WHERE ((((a = 1",
    "SELECT COUNT(*) FROM t WHERE a = 1.5e3 AND b = -0.0 AND c = '\xff\xfe'"};
  for (const char* q : qs) {
    int32_t st = 0;
    char* r = ph_parse_sql(q, &st);
    printf("%d %s\n", st, r ? "ok" : ph_last_error());
    if (r) ph_free(r);
  }
  std::vector<uint8_t> dict;
  int vals[] = {-50, -3, 0, 7, 8, 100, 2147483647};
  for (int v : vals) { dict.push_back((uint8_t)(v >> 24)); dict.push_back((uint8_t)(v >> 16)); dict.push_back((uint8_t)(v >> 8)); dict.push_back((uint8_t)v); }
  const char* ps[] = {"c BETWEEN 0 AND 8", "c > 0", "c < -50", "c > 100", "c = 7", "c != 9", "c IN (7, 8, 1000)", "c NOT IN (7)", "c = 'abc'", "c > 99999999999", "c IS NULL"};
  for (const char* p : ps) {
    int32_t st = 0;
    char* r = ph_lower_predicate(p, dict.data(), 7, &st);
    printf("%d %s\n", st, r ? r : ph_last_error());
    if (r) ph_free(r);
  }
  return 0;
}
