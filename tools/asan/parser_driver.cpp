#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include <string>
#include <vector>
#include "../../include/pinot_host_c.h"
// PINOT_ASAN_SQL_FILE: one query per line, parsed (and, for group-by queries, combined over a tiny synthetic block) under the sanitizers
static void run_sql_file(const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) return;
  char line[4096];
  int n = 0, ok = 0;
  // 3 rows, room for 8 key columns and 16 functions per row
  const int64_t rows = 3;
  int32_t kt[8]; int64_t kl[24], counts[48]; double kd[24], vals[48]; const char* ks[24]; uint8_t kn[24], nulls[48];
  for (int i = 0; i < 8; ++i) kt[i] = i % 3 == 0 ? 4 : (i % 3 == 1 ? 0 : 3);      // STRING, INT, DOUBLE, ...
  for (int i = 0; i < 24; ++i) { kl[i] = i % 5; kd[i] = (i % 4) * 0.5; ks[i] = (i / 8) % 2 ? "b" : "a"; kn[i] = i == 9; }
  for (int i = 0; i < 48; ++i) { counts[i] = 1 + i % 7; vals[i] = (i * 37 % 11) - 3.5; nulls[i] = i % 13 == 5; }
  while (fgets(line, sizeof(line), f)) {
    size_t len = strlen(line);
    while (len && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = 0;
    int32_t st = 0;
    char* r = ph_parse_sql(line, &st);
    ++n;
    if (r) { ++ok; ph_free(r); }
    // the combine entry reads (rows x keys) and (rows x functions) cells: only hand it queries whose shape fits the arrays above
    int parens = 0, commas = 0;
    for (const char* c = line; *c; ++c) { parens += *c == '('; commas += *c == ','; }
    if (st == 0 && strstr(line, "GROUP BY k1, k2") && !strstr(line, "FILTER") && parens <= 12 && commas <= 7) {
      r = ph_group_by_combine(line, 1, &rows, kt, kl, kd, ks, kn, counts, vals, vals, vals, nulls, &st);
      if (r) ph_free(r);
    }
  }
  fclose(f);
  printf("sql file: %d queries, %d parsed\n", n, ok);
}

int main() {
  if (const char* sql_file = getenv("PINOT_ASAN_SQL_FILE")) run_sql_file(sql_file);
  const char* qs[] = {
    "SELECT COUNT(*), SUM(column1), MAX(column3), MIN(column6), AVG(column7) FROM testTable",
    "select sum(a) as s from t where a > 1 and (b in (1, 2, -3) or not c between 5 and 9) and d <> 'x''y' group by k1, k2",
    "SET enableNullHandling = true; SELECT COUNT(a), SUM(a) FROM t WHERE a IS NOT NULL AND (b IS NULL OR NOT c > 3)",
    "SELECT SUM(a) FILTER (WHERE b > 3 AND c = 'x'), COUNT(*) FILTER(WHERE b > 3 AND c = 'x'), MAX(a) FROM t WHERE d < 5 GROUP BY e",
    "SELECT a FROM t", "SELECT SUM(a + 1) FROM t", "SELECT SUM(*) FROM t", "SELECT SUM(a) FROM", "SELECT SUM(a) FROM t WHERE a >",
    "SELECT SUM(a) FILTER (b > 3) FROM t", "SELECT COUNT(*) FROM t WHERE a IS 3", "SET useStarTree = true; SELECT COUNT(*) FROM t",
    "SET numGroupsLimit = 5; SELECT COUNT(*) FROM t GROUP BY a", "SELECT COUNT(*) FROM t WHERE a IN ()", "", "SELECT", "SELECT COUNT(*) FROM t WHERE ((((a = 1",
    "SELECT COUNT(*) FROM t WHERE a = 1.5e3 AND b = -0.0 AND c = '\xff\xfe'",
    "SET minServerGroupTrimSize = 3; SELECT SUM(a), COUNT(*) FROM t GROUP BY k1, k2 ORDER BY COUNT(*) DESC NULLS LAST, k2, sum(a) ASC NULLS FIRST LIMIT 5",
    "SELECT SUM(a) FROM t GROUP BY k ORDER BY", "SELECT SUM(a) FROM t GROUP BY k ORDER BY MAX(", "SELECT SUM(a) FROM t GROUP BY k LIMIT", "SELECT SUM(a) FROM t GROUP BY k LIMIT 99999999999",
    "SELECT SUM(a) FROM t GROUP BY k ORDER BY k NULLS"};
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
  // writers: every bit width with extreme values, all three RoaringBitmap container kinds, raw chunks
  for (int bits = 1; bits <= 31; ++bits) {
    const int n = 1000;
    std::vector<int32_t> ids(n);
    const uint32_t max = bits == 31 ? 0x7FFFFFFFu : ((1u << bits) - 1u);
    for (int i = 0; i < n; ++i) ids[i] = (int32_t)((i % 3 == 0) ? max : ((uint32_t)i * 2654435761u) & max);
    std::vector<uint8_t> out((size_t)ph_fixedbit_size(n, bits));
    ph_fixedbit_pack(ids.data(), n, bits, out.data(), 3);
    std::vector<uint8_t> gen((size_t)ph_fixedbit_size(n, bits));
    ph_generate_packed_uniform(7, n, (int32_t)std::min<uint32_t>(max, 100000u) + 1, bits, gen.data(), 2);
  }
  {
    const int n = 300000;
    std::vector<int32_t> ids(n);
    for (int i = 0; i < n; ++i) ids[i] = i < 100000 ? 0 : (i % 7 == 0 ? 1 : (i % 1000 == 1 ? 2 : 3));
    const int64_t size = ph_inverted_build(ids.data(), n, 4, 1, nullptr);
    std::vector<uint8_t> inv((size_t)size);
    ph_inverted_build(ids.data(), n, 4, 1, inv.data());
    std::vector<int32_t> docs;
    for (int i = 0; i < n; i += 3) docs.push_back(i);
    std::vector<uint8_t> rb((size_t)ph_roaring_serialize(docs.data(), (int64_t)docs.size(), 1, nullptr));
    ph_roaring_serialize(docs.data(), (int64_t)docs.size(), 1, rb.data());
    std::vector<int64_t> longs(5003, INT64_MIN);
    std::vector<uint8_t> raw((size_t)ph_raw_size_fixed_v2((int32_t)longs.size(), 1000, 8));
    ph_raw_write_fixed_v2(longs.data(), (int32_t)longs.size(), 1000, 8, raw.data());
    printf("writers ok: inverted %lld bytes, roaring %zu bytes, raw %zu bytes\n", (long long)size, rb.size(), raw.size());
  }
  // the combine operator's table and the reducer: trims while upserting, NULL keys, null results, every key type
  {
    const int nb = 3, rows_per = 40, nk = 3, nf = 3;
    std::vector<int64_t> block_rows(nb, rows_per), kl((size_t)nb * rows_per * nk), counts((size_t)nb * rows_per * nf);
    std::vector<double> kd(kl.size()), sums(counts.size()), mins(counts.size()), maxs(counts.size());
    std::vector<const char*> ks(kl.size(), "");
    std::vector<uint8_t> kn(kl.size(), 0), nulls(counts.size(), 0);
    std::vector<std::string> names;
    for (int i = 0; i < 17; ++i) names.push_back("key" + std::to_string(i));
    const int32_t kt[3] = {4, 0, 3};      // STRING, INT, DOUBLE
    for (int r = 0; r < nb * rows_per; ++r) {
      ks[(size_t)r * nk] = names[(size_t)(r * 7 % 17)].c_str();
      kl[(size_t)r * nk + 1] = r % 5;
      kn[(size_t)r * nk + 1] = r % 11 == 0;
      kd[(size_t)r * nk + 2] = (r % 3) * 0.5;
      for (int f = 0; f < nf; ++f) { counts[(size_t)r * nf + f] = 1 + r % 4; sums[(size_t)r * nf + f] = r * 1.5; maxs[(size_t)r * nf + f] = r; nulls[(size_t)r * nf + f] = (r + f) % 13 == 0; }
    }
    const char* sqls[] = {
      "SET enableNullHandling = true; SET minServerGroupTrimSize = 4; SET groupTrimThreshold = 10; SET minSegmentGroupTrimSize = 6; "
      "SELECT SUM(a), MAX(b), AVG(c) FROM t GROUP BY k1, k2, k3 ORDER BY AVG(c) DESC, k2 NULLS FIRST, k1 LIMIT 2",
      "SELECT SUM(a), MAX(b), AVG(c) FROM t GROUP BY k1, k2, k3 LIMIT 7", "SELECT SUM(a), MAX(b), AVG(c) FROM t GROUP BY k1, k2, k3 ORDER BY k3, SUM(a) LIMIT 0"};
    for (const char* q : sqls) {
      int32_t st = 0;
      char* r = ph_group_by_combine(q, nb, block_rows.data(), kt, kl.data(), kd.data(), ks.data(), kn.data(), counts.data(), sums.data(), mins.data(), maxs.data(), nulls.data(), &st);
      printf("combine %d %zu bytes %s\n", st, r ? strlen(r) : (size_t)0, r ? "" : ph_last_error());
      if (r) ph_free(r);
    }
  }
  return 0;
}
