"""GPU tests through the C++ host mirror: SQL text -> GpuPlanMaker -> C ABI -> kernels -> results blocks -> combine.
They read like the reference's own query tests (InnerSegment/InterSegmentAggregationSingleValueQueriesTest)."""
import numpy as np
import pytest

from pinot_amd import host
import helpers as H

pytestmark = pytest.mark.gpu

QUERY = "SELECT COUNT(*), SUM(column1), MAX(column3), MIN(column6), AVG(column7) FROM testTable"
FILTER = (" WHERE column1 > 100000000 AND column3 BETWEEN 20000000 AND 1000000000 AND column5 = 'gFuH'"
          " AND (column6 < 500000000 OR column11 NOT IN ('t', 'P')) AND daysSinceEpoch = 126164076")


@pytest.fixture(scope="module")
def golden_segments():
    import torch  # noqa: F401
    host.init_plan_maker(device=0, time_kernels=True)
    data = H.golden_segment()
    segs = [host.HostSegment(data, string_dicts=data.string_dicts) for _ in range(4)]
    yield data, segs
    for s in segs:
        s.destroy()


def test_inner_segment_aggregation_only(golden_segments):
    _, segs = golden_segments
    g = H.load_golden_queries()["inner_segment"]
    for sql, want in ((QUERY, g["unfiltered"]), (QUERY + FILTER, g["filtered"])):
        block = host.execute_sql(segs[:1], sql)["segments"][0]
        assert block["intermediate"] == [want["count"], float(want["sum_column1"]), float(want["max_column3"]),
                                         float(want["min_column6"]), [float(want["avg_column7"][0]), want["avg_column7"][1]]]
        st = block["stats"]
        # all four, numEntriesScannedInFilter = 63064 included: the host mirror builds the reference's operator tree from the SQL
        # (FilterOperatorUtils pruning and priorities), the engine accounts for its iterators
        assert [st["numDocsScanned"], st["numEntriesScannedInFilter"], st["numEntriesScannedPostFilter"], st["numTotalDocs"]] == want["stats"]


def test_inner_segment_small_group_by(golden_segments):
    _, segs = golden_segments
    g = H.load_golden_queries()["inner_segment_group_by_column9"]
    for sql, want in ((QUERY + " GROUP BY column9", g["unfiltered"]), (QUERY + FILTER + " GROUP BY column9", g["filtered"])):
        block = host.execute_sql(segs[:1], sql)["segments"][0]
        row = [r for r in block["groups"] if r["key"] == [want["key"]]]
        assert len(row) == 1
        assert row[0]["intermediate"] == [want["count"], float(want["sum_column1"]), float(want["max_column3"]),
                                          float(want["min_column6"]), [float(want["avg_column7"][0]), want["avg_column7"][1]]]


def test_inter_segment_results_through_combine(golden_segments):
    _, segs = golden_segments
    g = H.load_golden_queries()["inter_segment_x4"]
    for key, flt in (("unfiltered", ""), ("filtered", FILTER)):
        combined = host.execute_sql(segs, "SELECT COUNT(*), SUM(column1), SUM(column3) FROM testTable" + flt, max_execution_threads=4)["combined"]
        assert combined["final"] == [float(g["count"][key]), g["sum_column1"][key], g["sum_column3"][key]]
        assert combined["stats"]["numTotalDocs"] == 120000
    # GROUP BY column9 ORDER BY COUNT(*) DESC LIMIT 1 -> 64420 / 17080 (InterSegment...testCount)
    for want, flt in ((64420, ""), (17080, FILTER)):
        combined = host.execute_sql(segs, "SELECT COUNT(*) FROM testTable" + flt + " GROUP BY column9 LIMIT 100000")["combined"]      # (no LIMIT = 10 groups)
        assert max(r["final"][0] for r in combined["groups"]) == float(want)
    # string group key comes back as dictionary VALUES
    combined = host.execute_sql(segs[:2], "SELECT COUNT(*), MAX(column1) FROM testTable GROUP BY column11, column12")["combined"]
    d = H.load_golden_columns()
    c11 = d["column11__dict"][d["column11__ids"]]
    c12 = d["column12__dict"][d["column12__ids"]]
    key0 = combined["groups"][0]["key"]
    m = (c11 == key0[0]) & (c12 == key0[1])
    assert combined["groups"][0]["final"][0] == 2.0 * m.sum()
    assert combined["groups"][0]["final"][1] == float(d["column1"][m].max())


def test_inter_segment_group_by_order_by_limit_goldens(golden_segments):
    """InterSegmentAggregationSingleValueQueriesTest.java:36-203 with the reference's own SQL: GROUP BY column9 ORDER BY ... LIMIT 1 over the
    four segments, through the combine operator's IndexedTable and the broker's reduce (host/indexed_table.cpp); all four statistics."""
    _, segs = golden_segments
    g = H.load_golden_queries()["inter_segment_x4"]
    desc, asc = " GROUP BY column9 ORDER BY v1 DESC, v2 DESC LIMIT 1", " GROUP BY column9 ORDER BY v1, v2 LIMIT 1"

    def run(sql):
        out = host.execute_sql(segs, sql, max_execution_threads=4)
        st = out["combined"]["stats"]
        return out["reduced"], [st["numDocsScanned"], st["numEntriesScannedInFilter"], st["numEntriesScannedPostFilter"], st["numTotalDocs"]]

    for flt, want, stats in (("", 64420, [120000, 0, 120000, 120000]), (FILTER, 17080, [24516, 252256, 24516, 120000])):
        for sql in ("SELECT COUNT(*) FROM testTable" + flt + " GROUP BY column9 ORDER BY COUNT(*) DESC LIMIT 1",
                    "SELECT COUNT(*) AS v1 FROM testTable" + flt + " GROUP BY column9 ORDER BY v1 DESC LIMIT 1"):
            rows, st = run(sql)
            assert len(rows) == 1 and rows[0][1] == want and st == stats
    mx, mn = g["max_column1_column3"], g["min_column1_column3"]
    for sql, want in (("SELECT MAX(column1) AS v1, MAX(column3) AS v2 FROM testTable" + desc, mx["group_by_top_desc"]),
                      ("SELECT MAX(column1) AS v1, MAX(column3) AS v2 FROM testTable" + FILTER + desc, mx["filtered_group_by_top_desc"]),
                      ("SELECT MIN(column1) AS v1, MIN(column3) AS v2 FROM testTable" + asc, mn["group_by_top_asc"]),
                      ("SELECT MIN(column1) AS v1, MIN(column3) AS v2 FROM testTable" + FILTER + asc, mn["filtered_group_by_top_asc"])):
        rows, st = run(sql)
        assert len(rows) == 1 and rows[0][1:] == want["values"] and st == want["stats"], sql
    for flt, key, stats in (("", "unfiltered", [120000, 0, 360000, 120000]), (FILTER, "filtered", [24516, 252256, 73548, 120000])):
        rows, st = run("SELECT SUM(column1) AS v1, SUM(column3) AS v2 FROM testTable" + flt + desc)
        assert rows[0][1:] == g["sum_group_by_top_desc"][key] and st == stats
    rows, st = run("SELECT AVG(column1) AS v1, AVG(column3) AS v2 FROM testTable" + desc)
    assert rows[0][1:] == g["avg_column1_column3"]["group_by_top_desc"]["values"] and st == [120000, 0, 360000, 120000]
    rows, _ = run("SELECT AVG(column1) AS v1, AVG(column3) AS v2 FROM testTable" + FILTER + desc)
    assert rows[0][1:] == [2142595699.0, 334963174.0]
    # the server's combined block keeps max(5 * LIMIT, 5000) groups for the broker; without ORDER BY only LIMIT keys are admitted
    out = host.execute_sql(segs, "SELECT COUNT(*) FROM testTable GROUP BY column9 LIMIT 7", max_execution_threads=1)
    assert len(out["combined"]["groups"]) == 7 and len(out["reduced"]) == 7
    first7 = list(dict.fromkeys(sorted(H.load_golden_columns()["column9"].tolist())))[:7]      # blocks list groups by ascending raw key = ascending value
    assert [r[0] for r in out["reduced"]] == first7
    # in-segment trim: ORDER BY + minSegmentGroupTrimSize keeps max(5 * LIMIT, size) groups of every segment block
    out = host.execute_sql(segs[:1], "SET minSegmentGroupTrimSize = 20; SELECT COUNT(*) FROM testTable GROUP BY column9 ORDER BY COUNT(*) DESC, column9 LIMIT 3")
    assert len(out["segments"][0]["groups"]) == 20 and out["reduced"][0][1] == 64420 // 4
    counts = {}
    for v in H.load_golden_columns()["column9"].tolist():
        counts[v] = counts.get(v, 0) + 1
    assert out["reduced"] == [[k, c] for k, c in sorted(counts.items(), key=lambda kv: (-kv[1], kv[0]))[:3]]


# InterSegmentGroupBySingleValueQueriesTest.groupByOrderByDataProvider (:61-327): the reference's SQL, its expected
# numEntriesScannedPostFilter and its expected result tables (the DISTINCTCOUNT / PERCENTILE / transform-function entries are not on this path)
_C1112 = ["column11", "column12", "sum(column1)"]
GROUP_BY_ORDER_BY_GOLDENS = [
    ("SELECT column11, SUM(column1) FROM testTable GROUP BY column11 ORDER BY column11", 240000, ["column11", "sum(column1)"],
     [["", 5935285005452.0], ["P", 88832999206836.0], ["gFuH", 63202785888.0], ["o", 18105331533948.0], ["t", 16331923219264.0]]),
    ("SELECT column11, sum(column1) FROM testTable GROUP BY column11 ORDER BY column11 DESC", 240000, ["column11", "sum(column1)"],
     [["t", 16331923219264.0], ["o", 18105331533948.0], ["gFuH", 63202785888.0], ["P", 88832999206836.0], ["", 5935285005452.0]]),
    ("SELECT column11, Sum(column1) FROM testTable GROUP BY column11 ORDER BY column11 LIMIT 3", 240000, ["column11", "sum(column1)"],
     [["", 5935285005452.0], ["P", 88832999206836.0], ["gFuH", 63202785888.0]]),
    ("SELECT column11, column12, SUM(column1) FROM testTable GROUP BY column11, column12 ORDER BY column11, column12", 360000, _C1112,
     [["", "HEuxNvH", 3789390396216.0], ["", "KrNxpdycSiwoRohEiTIlLqDHnx", 733802350944.0], ["", "MaztCmmxxgguBUxPti", 1333941430664.0], ["", "dJWwFk", 55470665124.0],
      ["", "oZgnrlDEtjjVpUoFLol", 22680162504.0], ["P", "HEuxNvH", 21998672845052.0], ["P", "KrNxpdycSiwoRohEiTIlLqDHnx", 18069909216728.0],
      ["P", "MaztCmmxxgguBUxPti", 27177029040008.0], ["P", "TTltMtFiRqUjvOG", 4462670055540.0], ["P", "XcBNHe", 120021767504.0]]),
    ("SELECT column11, column12, SUM(column1) FROM testTable GROUP BY column11, column12 ORDER BY column11, column12 LIMIT 15", 360000, _C1112,
     [["", "HEuxNvH", 3789390396216.0], ["", "KrNxpdycSiwoRohEiTIlLqDHnx", 733802350944.0], ["", "MaztCmmxxgguBUxPti", 1333941430664.0], ["", "dJWwFk", 55470665124.0],
      ["", "oZgnrlDEtjjVpUoFLol", 22680162504.0], ["P", "HEuxNvH", 21998672845052.0], ["P", "KrNxpdycSiwoRohEiTIlLqDHnx", 18069909216728.0],
      ["P", "MaztCmmxxgguBUxPti", 27177029040008.0], ["P", "TTltMtFiRqUjvOG", 4462670055540.0], ["P", "XcBNHe", 120021767504.0],
      ["P", "dJWwFk", 6224665921376.0], ["P", "fykKFqiw", 1574451324140.0], ["P", "gFuH", 860077643636.0], ["P", "oZgnrlDEtjjVpUoFLol", 8345501392852.0],
      ["gFuH", "HEuxNvH", 29872400856.0]]),
    ("SELECT column11, column12, SUM(column1) FROM testTable GROUP BY column11, column12 ORDER BY column11, column12 DESC", 360000, _C1112,
     [["", "oZgnrlDEtjjVpUoFLol", 22680162504.0], ["", "dJWwFk", 55470665124.0], ["", "MaztCmmxxgguBUxPti", 1333941430664.0], ["", "KrNxpdycSiwoRohEiTIlLqDHnx", 733802350944.0],
      ["", "HEuxNvH", 3789390396216.0], ["P", "oZgnrlDEtjjVpUoFLol", 8345501392852.0], ["P", "gFuH", 860077643636.0], ["P", "fykKFqiw", 1574451324140.0],
      ["P", "dJWwFk", 6224665921376.0], ["P", "XcBNHe", 120021767504.0]]),
    ("SELECT column11, column12, SUM(column1) FROM testTable GROUP BY column11, column12 ORDER BY column11, sum(column1)", 360000, _C1112,
     [["", "oZgnrlDEtjjVpUoFLol", 22680162504.0], ["", "dJWwFk", 55470665124.0], ["", "KrNxpdycSiwoRohEiTIlLqDHnx", 733802350944.0], ["", "MaztCmmxxgguBUxPti", 1333941430664.0],
      ["", "HEuxNvH", 3789390396216.0], ["P", "XcBNHe", 120021767504.0], ["P", "gFuH", 860077643636.0], ["P", "fykKFqiw", 1574451324140.0],
      ["P", "TTltMtFiRqUjvOG", 4462670055540.0], ["P", "dJWwFk", 6224665921376.0]]),
    ("SELECT column11, column12, SUM(column1) FROM testTable GROUP BY column11, column12 ORDER BY SUM(column1) DESC LIMIT 50", 360000, _C1112,
     [["P", "MaztCmmxxgguBUxPti", 27177029040008.0], ["P", "HEuxNvH", 21998672845052.0], ["P", "KrNxpdycSiwoRohEiTIlLqDHnx", 18069909216728.0],
      ["P", "oZgnrlDEtjjVpUoFLol", 8345501392852.0], ["o", "MaztCmmxxgguBUxPti", 6905624581072.0], ["P", "dJWwFk", 6224665921376.0], ["o", "HEuxNvH", 5026384681784.0],
      ["t", "MaztCmmxxgguBUxPti", 4492405624940.0], ["P", "TTltMtFiRqUjvOG", 4462670055540.0], ["t", "HEuxNvH", 4424489490364.0],
      ["o", "KrNxpdycSiwoRohEiTIlLqDHnx", 4051812250524.0], ["", "HEuxNvH", 3789390396216.0], ["t", "KrNxpdycSiwoRohEiTIlLqDHnx", 3529048341192.0],
      ["P", "fykKFqiw", 1574451324140.0], ["t", "dJWwFk", 1349058948804.0], ["", "MaztCmmxxgguBUxPti", 1333941430664.0], ["o", "dJWwFk", 1152689463360.0],
      ["t", "oZgnrlDEtjjVpUoFLol", 1039101333316.0], ["P", "gFuH", 860077643636.0], ["", "KrNxpdycSiwoRohEiTIlLqDHnx", 733802350944.0],
      ["o", "oZgnrlDEtjjVpUoFLol", 699381633640.0], ["t", "TTltMtFiRqUjvOG", 675238030848.0], ["t", "fykKFqiw", 480973878052.0], ["t", "gFuH", 330331507792.0],
      ["o", "TTltMtFiRqUjvOG", 203835153352.0], ["P", "XcBNHe", 120021767504.0], ["o", "fykKFqiw", 62975165296.0], ["", "dJWwFk", 55470665124.0],
      ["gFuH", "HEuxNvH", 29872400856.0], ["gFuH", "MaztCmmxxgguBUxPti", 29170832184.0], ["", "oZgnrlDEtjjVpUoFLol", 22680162504.0], ["t", "XcBNHe", 11276063956.0],
      ["gFuH", "KrNxpdycSiwoRohEiTIlLqDHnx", 4159552848.0], ["o", "gFuH", 2628604920.0]]),
    ("SELECT sum(column1), MIN(column6) FROM testTable GROUP BY column11 ORDER BY column11", 360000, ["sum(column1)", "min(column6)"],
     [[5935285005452.0, 2.96467636E8], [88832999206836.0, 1689277.0], [63202785888.0, 2.96467636E8], [18105331533948.0, 2.96467636E8], [16331923219264.0, 1980174.0]]),
    ("SELECT column11, column12, SUM(column1) FROM testTable GROUP BY column11, column12 ORDER BY SUM  (\tcolumn1) DESC LIMIT 3", 360000, _C1112,
     [["P", "MaztCmmxxgguBUxPti", 27177029040008.0], ["P", "HEuxNvH", 21998672845052.0], ["P", "KrNxpdycSiwoRohEiTIlLqDHnx", 18069909216728.0]]),
    ("SELECT column12, MIN(column6) FROM testTable GROUP BY column12 ORDER BY Min(column6) DESC, column12", 240000, ["column12", "min(column6)"],
     [["XcBNHe", 329467557.0], ["fykKFqiw", 296467636.0], ["gFuH", 296467636.0], ["HEuxNvH", 6043515.0], ["MaztCmmxxgguBUxPti", 6043515.0], ["dJWwFk", 6043515.0],
      ["KrNxpdycSiwoRohEiTIlLqDHnx", 1980174.0], ["TTltMtFiRqUjvOG", 1980174.0], ["oZgnrlDEtjjVpUoFLol", 1689277.0]]),
    ("SELECT column12 FROM testTable GROUP BY column12 ORDER BY Min(column6) DESC, column12", 240000, ["column12"],
     [["XcBNHe"], ["fykKFqiw"], ["gFuH"], ["HEuxNvH"], ["MaztCmmxxgguBUxPti"], ["dJWwFk"], ["KrNxpdycSiwoRohEiTIlLqDHnx"], ["TTltMtFiRqUjvOG"], ["oZgnrlDEtjjVpUoFLol"]]),
    ("SELECT column12 FROM testTable GROUP BY column12 ORDER BY Min(column6) DESC, SUM(column1) LIMIT 3", 360000, ["column12"], [["XcBNHe"], ["gFuH"], ["fykKFqiw"]]),
    ("SELECT column12, MIN(column6) FROM testTable GROUP BY column12 ORDER BY Min(column6) DESC, SUM(column1) LIMIT 3", 360000, ["column12", "min(column6)"],
     [["XcBNHe", 329467557.0], ["gFuH", 296467636.0], ["fykKFqiw", 296467636.0]]),
    ("select column17, count(*) from testTable group by column17 order by column17 limit 15", 120000, ["column17", "count(*)"],
     [[83386499, 2924], [217787432, 3892], [227908817, 6564], [402773817, 7304], [423049234, 6556], [561673250, 7420], [635942547, 3308], [638936844, 3816],
      [939479517, 3116], [984091268, 3824], [1230252339, 5620], [1284373442, 7428], [1555255521, 2900], [1618904660, 2744], [1670085862, 3388]]),
    ("SELECT column11, AVG(column6) FROM testTable GROUP BY column11  ORDER BY column11", 240000, ["column11", "avg(column6)"],
     [["", 296467636.0], ["P", 909380310.3521485], ["gFuH", 296467636.0], ["o", 296467636.0], ["t", 526245333.3900426]]),
    ("SELECT column11, AVG(column6) FROM testTable GROUP BY column11 ORDER BY AVG(column6), column11 DESC", 240000, ["column11", "avg(column6)"],
     [["o", 296467636.0], ["gFuH", 296467636.0], ["", 296467636.0], ["t", 526245333.3900426], ["P", 909380310.3521485]]),
]


@pytest.mark.parametrize("trim", [False, True])
def test_group_by_order_by_goldens_with_the_reference_sql(golden_segments, trim):
    """testGroupByOrderBy / testGroupByOrderByWithTrim: the result table of every query and (120000, 0, numEntriesScannedPostFilter, 120000);
    the trim variant runs with minSegmentGroupTrimSize = 1 like TRIM_ENABLED_PLAN_MAKER (:38-41)."""
    _, segs = golden_segments
    for sql, post_filter, columns, rows in GROUP_BY_ORDER_BY_GOLDENS:
        out = host.execute_sql(segs, ("SET minSegmentGroupTrimSize = 1; " if trim else "") + sql, max_execution_threads=4)
        st = out["combined"]["stats"]
        assert [st["numDocsScanned"], st["numEntriesScannedInFilter"], st["numEntriesScannedPostFilter"], st["numTotalDocs"]] == [120000, 0, post_filter, 120000], sql
        assert out["resultTable"]["columns"] == columns, sql
        got = out["resultTable"]["rows"]
        assert len(got) == len(rows), sql
        for g, w in zip(got, rows):
            assert all((a == pytest.approx(b, rel=1e-12) if isinstance(b, float) else a == b) for a, b in zip(g, w)) and len(g) == len(w), (sql, g, w)


def test_filtered_aggregations_run_as_swim_lanes(golden_segments):
    """FILTER (WHERE ...) aggregations (FilteredAggregationOperator.java:68-110; the reference's FilteredAggregationsTest compares a
    filtered-aggregation query with the equivalent separately filtered queries -- so does this)."""
    _, segs = golden_segments
    main = " WHERE column3 BETWEEN 20000000 AND 1000000000"
    f1 = "column1 > 100000000 AND column11 NOT IN ('t', 'P')"
    f2 = "column6 < 500000000 OR column5 = 'gFuH'"
    for where in ("", main):
        sql = (f"SELECT SUM(column1) FILTER (WHERE {f1}), COUNT(*), MAX(column3) FILTER (WHERE {f2}), "
               f"AVG(column7) FILTER (WHERE {f1}), MIN(column6) FROM testTable{where}")
        got = host.execute_sql(segs[:1], sql)["segments"][0]
        glue = " AND " if where else " WHERE "
        lane0 = host.execute_sql(segs[:1], "SELECT COUNT(*), MIN(column6) FROM testTable" + where)["segments"][0]
        lane1 = host.execute_sql(segs[:1], f"SELECT SUM(column1), AVG(column7) FROM testTable{where}{glue}({f1})")["segments"][0]
        lane2 = host.execute_sql(segs[:1], f"SELECT MAX(column3) FROM testTable{where}{glue}({f2})")["segments"][0]
        assert got["intermediate"] == [lane1["intermediate"][0], lane0["intermediate"][0], lane2["intermediate"][0],
                                       lane1["intermediate"][1], lane0["intermediate"][1]]
        for k in ("numDocsScanned", "numEntriesScannedInFilter", "numEntriesScannedPostFilter"):
            assert got["stats"][k] == lane0["stats"][k] + lane1["stats"][k] + lane2["stats"][k], k
        assert got["stats"]["numTotalDocs"] == 30000
        # and through the combine over 4 copies of the segment
        combined = host.execute_sql(segs, sql, max_execution_threads=4)["combined"]
        assert combined["final"][1] == 4.0 * lane0["intermediate"][0] and combined["final"][0] == 4.0 * lane1["intermediate"][0]
    # under GROUP BY (FilteredGroupByOperator): the union of the lanes' groups; a function whose lane never saw a group keeps its default
    sql = f"SELECT SUM(column1) FILTER (WHERE {f1}), COUNT(*), MAX(column3) FILTER (WHERE {f2}), AVG(column7) FILTER (WHERE {f1}) FROM testTable{main} GROUP BY column9"
    got = host.execute_sql(segs[:1], sql)["segments"][0]
    lane0 = {tuple(r["key"]): r["intermediate"] for r in host.execute_sql(segs[:1], f"SELECT COUNT(*) FROM testTable{main} GROUP BY column9")["segments"][0]["groups"]}
    lane1 = {tuple(r["key"]): r["intermediate"] for r in host.execute_sql(segs[:1], f"SELECT SUM(column1), AVG(column7) FROM testTable{main} AND ({f1}) GROUP BY column9")["segments"][0]["groups"]}
    lane2 = {tuple(r["key"]): r["intermediate"] for r in host.execute_sql(segs[:1], f"SELECT MAX(column3) FROM testTable{main} AND ({f2}) GROUP BY column9")["segments"][0]["groups"]}
    rows = {tuple(r["key"]): r["intermediate"] for r in got["groups"]}
    assert set(rows) == set(lane0) | set(lane1) | set(lane2) and len(lane1) < len(lane0)
    for key, row in rows.items():
        s1, a7 = lane1.get(key, [0.0, [0.0, 0]])
        assert row == [s1, lane0.get(key, [0])[0], lane2.get(key, ["-Infinity"])[0], a7], key
    assert [r["key"] for r in got["groups"]] == sorted(r["key"] for r in got["groups"])


def test_medium_group_by_golden_and_num_groups_limit_through_sql(golden_segments):
    """InnerSegmentAggregationSingleValueQueriesTest.testMediumAggregationGroupBy :114-132 (78 165 raw keys: the reference's INT_MAP_BASED
    holder) as SQL, and the numGroupsLimit option."""
    _, segs = golden_segments
    g = H.load_golden_queries()["inner_segment_group_by_medium"]
    for sql, want in ((QUERY + " GROUP BY column9, column11, column12", g["unfiltered"]), (QUERY + FILTER + " GROUP BY column9, column11, column12", g["filtered"])):
        block = host.execute_sql(segs[:1], sql)["segments"][0]
        row = [r for r in block["groups"] if r["key"] == want["key"]]
        assert len(row) == 1 and not block["numGroupsLimitReached"]
        assert row[0]["intermediate"] == [want["count"], float(want["sum_column1"]), float(want["max_column3"]),
                                          float(want["min_column6"]), [float(want["avg_column7"][0]), want["avg_column7"][1]]]
        st = block["stats"]
        assert (st["numDocsScanned"], st["numEntriesScannedPostFilter"], st["numTotalDocs"]) == (want["stats"][0], want["stats"][2], want["stats"][3])
    d = H.load_golden_columns()
    all_groups = host.execute_sql(segs[:1], "SELECT COUNT(*) FROM testTable GROUP BY column9, column11, column12")["segments"][0]
    limited = host.execute_sql(segs[:1], "SET numGroupsLimit = 50; SELECT COUNT(*) FROM testTable GROUP BY column9, column11, column12")["segments"][0]
    assert len(all_groups["groups"]) > 50 and len(limited["groups"]) == 50 and limited["numGroupsLimitReached"]
    # the survivors are the first 50 distinct keys in docId order (IntGroupIdMap admits keys in order of first appearance)
    keys = list(zip(d["column9"].tolist(), d["column11__dict"][d["column11__ids"]].tolist(), d["column12__dict"][d["column12__ids"]].tolist()))
    first50 = list(dict.fromkeys(keys))[:50]
    assert sorted(tuple(r["key"]) for r in limited["groups"]) == sorted(first50)
    assert limited["stats"]["numDocsScanned"] == 30000


def test_large_and_very_large_group_by_goldens_through_sql(golden_segments):
    """testLargeAggregationGroupBy / testVeryLargeAggregationGroupBy (InnerSegmentAggregationSingleValueQueriesTest.java:134-176) as the
    reference's own SQL through the C++ plan maker: five and nine group-by columns, the Long / ArrayMap holders, keys back as VALUES."""
    _, segs = golden_segments
    g = H.load_golden_queries()
    for row in ("inner_segment_group_by_large", "inner_segment_group_by_very_large"):
        group_by = " GROUP BY " + ", ".join(g[row]["group_by"])
        for sql, want in ((QUERY + group_by, g[row]["unfiltered"]), (QUERY + FILTER + group_by, g[row]["filtered"])):
            block = host.execute_sql(segs[:1], sql)["segments"][0]
            hit = [r for r in block["groups"] if r["key"] == want["key"]]
            assert len(hit) == 1 and not block["numGroupsLimitReached"]
            assert hit[0]["intermediate"] == [want["count"], float(want["sum_column1"]), float(want["max_column3"]),
                                              float(want["min_column6"]), [float(want["avg_column7"][0]), want["avg_column7"][1]]]
            st = block["stats"]
            assert (st["numDocsScanned"], st["numEntriesScannedInFilter"], st["numEntriesScannedPostFilter"], st["numTotalDocs"]) == tuple(want["stats"])


def test_group_by_over_144_million_raw_keys(golden_segments):
    """GROUP BY column1, column3: 6582 * 21910 raw keys, the upper IntMapBasedHolder range, one direct-indexed table in HBM."""
    d, segs = golden_segments
    d = H.load_golden_columns()
    got = host.execute_sql(segs[:1], "SELECT COUNT(*), SUM(column6) FROM testTable GROUP BY column1, column3")["segments"][0]
    want = {}
    for a, b, c in zip(d["column1"].tolist(), d["column3"].tolist(), d["column6"].tolist()):
        e = want.setdefault((a, b), [0, 0.0])
        e[0] += 1
        e[1] += float(c)
    assert {tuple(r["key"]): r["intermediate"] for r in got["groups"]} == want
    assert got["stats"]["numDocsScanned"] == 30000 and not got["numGroupsLimitReached"]


def test_plan_time_rejection_and_errors(golden_segments):
    _, segs = golden_segments
    for sql, status in (("SELECT column1 FROM testTable", 2),
                        ("SELECT SUM(nope) FROM testTable", 1),
                        ("SELECT SUM(column11) FROM testTable", 1),
                        ("SELECT SUM(column1) FROM testTable WHERE column1 = 'abc'", 1)):
        with pytest.raises(host.HostError) as e:
            host.execute_sql(segs[:1], sql)
        assert e.value.status == status, sql


def test_group_by_key_space_beyond_an_int_through_sql(golden_segments):
    """GROUP BY column1, column3, column6: 6582 * 21910 * 608 raw keys are not an int -- the reference's LongMapBasedHolder, the device's
    hashed table; the plan maker hands the query over like any other and the keys come back as values."""
    _, segs = golden_segments
    d = H.load_golden_columns()
    out = host.execute_sql(segs[:1], "SELECT column1, column3, column6, SUM(column1), COUNT(*) FROM testTable GROUP BY column1, column3, column6 LIMIT 100000")["segments"][0]
    keys = np.stack([d["column1"], d["column3"], d["column6"]], axis=1)
    uniq, counts = np.unique(keys, axis=0, return_counts=True)
    got = {tuple(g["key"]): g["intermediate"] for g in out["groups"]}
    assert len(got) == len(uniq)
    for row, c in list(zip(uniq.tolist(), counts.tolist()))[:: max(1, len(uniq) // 500)]:
        assert got[tuple(row)] == [float(row[0]) * c, c]


def test_sql_over_the_segment_written_by_the_reference():
    """paddingOld (pinot-core/src/test/resources/data/paddingOld.tar.gz): INT `age`, FLOAT `percent`, LONG `outgoingName1`,
    STRING `name` -- the bytes the reference's Java writers produced, queried with SQL through the C++ plan maker."""
    import json
    import os
    from pinot_amd import _abi
    from pinot_amd import segment as S
    import torch  # noqa: F401
    host.init_plan_maker(device=0, time_kernels=False)
    p = json.load(open(os.path.join(H.GOLDEN_DIR, "pinot_v1_segment_paddingOld.json")))
    types = {"age": _abi.PG_TYPE_INT, "percent": _abi.PG_TYPE_FLOAT, "outgoingName1": _abi.PG_TYPE_LONG, "name": _abi.PG_TYPE_INT}
    cols, raw = [], {}
    for name, st in types.items():
        c = p["columns"][name]
        raw[name] = bytes.fromhex(c["dict_hex"])
        dict_bytes = np.frombuffer(raw[name], dtype=np.uint8).copy()
        if name == "name":      # STRING dictionary: fixed-length padded entries; the device only sees dictIds
            dict_bytes = np.arange(c["cardinality"], dtype=">i4").view(np.uint8).copy()
        cols.append(S.Column(name, _abi.PG_FWD_FIXED_BIT_DICT, c["bitsPerElement"], c["cardinality"],
                             np.frombuffer(bytes.fromhex(c["fwd_hex"]), dtype=np.uint8).copy(), dict_bytes, stored_type=st))
    width = len(raw["name"]) // p["columns"]["name"]["cardinality"]
    names = [raw["name"][i * width:(i + 1) * width].decode().rstrip("%\0") for i in range(p["columns"]["name"]["cardinality"])]
    data = S.SegmentData("paddingOld", p["total_docs"], cols)
    seg = host.HostSegment(data, string_dicts={"name": names})
    try:
        ages = np.frombuffer(raw["age"], dtype=">i4").astype(np.int64)
        longs = np.frombuffer(raw["outgoingName1"], dtype=">i8").astype(np.int64)
        floats = np.frombuffer(raw["percent"], dtype=">f4").astype(np.float32)
        b = host.execute_sql([seg], "SELECT COUNT(*), SUM(age), SUM(outgoingName1), MAX(outgoingName1), MIN(percent), SUM(percent) FROM t")["segments"][0]
        got = b["intermediate"]
        assert got[:5] == [5, float(ages.sum()), float(longs.sum()), float(longs.max()), float(floats.min())]
        assert abs(got[5] - float(floats.astype(np.float64).sum())) <= 1e-11 * abs(got[5])
        # predicates on the LONG and FLOAT dictionaries (LongDictionary / FloatDictionary.insertionIndexOf)
        thr = int(longs[2])
        b = host.execute_sql([seg], "SELECT COUNT(*), SUM(outgoingName1) FROM t WHERE outgoingName1 >= %d" % thr)["segments"][0]
        assert b["intermediate"] == [int((longs >= thr).sum()), float(longs[longs >= thr].sum())]
        fthr = float(floats[1])
        b = host.execute_sql([seg], "SELECT COUNT(*) FROM t WHERE percent > %r" % fthr)["segments"][0]
        assert b["intermediate"] == [int((floats > np.float32(fthr)).sum())]
        # group by the STRING column: keys come back as dictionary values
        g = host.execute_sql([seg], "SELECT COUNT(*), SUM(outgoingName1) FROM t GROUP BY name")["segments"][0]
        assert sorted(r["key"][0] for r in g["groups"]) == sorted(names)
        assert sum(r["intermediate"][0] for r in g["groups"]) == 5 and sum(r["intermediate"][1] for r in g["groups"]) == float(longs.sum())
        # group by the LONG column
        g = host.execute_sql([seg], "SELECT COUNT(*) FROM t GROUP BY outgoingName1")["segments"][0]
        assert sorted(r["key"][0] for r in g["groups"]) == sorted(int(x) for x in longs)
    finally:
        seg.destroy()


def test_string_key_group_by_goldens_through_sql(golden_segments):
    """InterSegmentGroupBySingleValueQueriesTest.java:66-107 through SQL -> plan maker -> device -> value-keyed combine of 4 segments."""
    _, segs = golden_segments
    g = H.load_golden_queries()["inter_segment_group_by_x4"]
    combined = host.execute_sql(segs, "SELECT SUM(column1) FROM testTable GROUP BY column11", max_execution_threads=4)["combined"]
    assert sorted([r["key"][0], r["final"][0]] for r in combined["groups"]) == g["sum_column1_by_column11"]
    combined = host.execute_sql(segs, "SELECT SUM(column1) FROM testTable GROUP BY column11, column12 LIMIT 1000", max_execution_threads=2)["combined"]
    assert sorted([r["key"][0], r["key"][1], r["final"][0]] for r in combined["groups"])[:15] == g["sum_column1_by_column11_column12_first15"]


def test_datatable_v4_of_the_golden_queries(golden_segments):
    """The bytes the server would send the broker (DataTableImplV4, intermediate results): decoded with the reader of tests/datatable_v4.py
    and compared with the reference's golden values; and equal to that file's own encoding of the same values."""
    import datatable_v4 as D
    _, segs = golden_segments
    g = H.load_golden_queries()
    want = g["inner_segment"]["filtered"]
    data = host.execute_sql_datatable(segs[:1], QUERY + FILTER)
    back = D.decode(data)
    assert back["names"] == ["count(*)", "sum(column1)", "max(column3)", "min(column6)", "avg(column7)"] and back["types"] == ["LONG", "DOUBLE", "DOUBLE", "DOUBLE", "OBJECT"]
    assert back["rows"] == [[want["count"], float(want["sum_column1"]), float(want["max_column3"]), float(want["min_column6"]), (float(want["avg_column7"][0]), want["avg_column7"][1])]]
    assert [back["metadata"][k] for k in ("numDocsScanned", "numEntriesScannedInFilter", "numEntriesScannedPostFilter", "totalDocs")] == want["stats"]
    import json, os
    fixture = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "datatable_v4_golden.json")))
    assert data.hex() == fixture["inner_segment_filtered_aggregation"]["hex"]
    # four segments through the combine: the x4 goldens, and a STRING-keyed group-by whose keys travel through the string dictionary
    x4 = g["inter_segment_x4"]
    back = D.decode(host.execute_sql_datatable(segs, "SELECT COUNT(*), SUM(column1), SUM(column3) FROM testTable" + FILTER, max_execution_threads=4))
    assert back["rows"] == [[x4["count"]["filtered"], x4["sum_column1"]["filtered"], x4["sum_column3"]["filtered"]]]
    assert back["metadata"]["numSegmentsProcessed"] == 4 and back["metadata"]["numSegmentsMatched"] == 4 and back["metadata"]["totalDocs"] == 120000
    combined = host.execute_sql(segs, "SELECT SUM(column1) FROM testTable GROUP BY column11")["combined"]
    back = D.decode(host.execute_sql_datatable(segs, "SELECT SUM(column1) FROM testTable GROUP BY column11"))
    assert back["names"] == ["column11", "sum(column1)"] and back["types"] == ["STRING", "DOUBLE"]
    assert sorted(back["rows"]) == sorted([r["key"][0], r["intermediate"][0]] for r in combined["groups"])
    assert sorted(back["rows"]) == sorted(g["inter_segment_group_by_x4"]["sum_column1_by_column11"])


def test_execute_combined_over_segments_on_different_devices():
    """CombinePlanNode over segments that live on different GPUs of the node (one segment per device, host-side merge: SURVEY 8(e)).
    On a one-GPU box device ids 0 and 1 alias the one chip (PINOT_GPU_ALIAS_DEVICES=2, read by pg_init): the same code runs."""
    import os
    import torch
    aliased = torch.cuda.device_count() < 2
    if aliased:
        os.environ["PINOT_GPU_ALIAS_DEVICES"] = "2"
    host.init_plan_maker(device=0, time_kernels=True)
    data = H.golden_segment()
    segs = []
    try:
        segs = [host.HostSegment(data, string_dicts=data.string_dicts, device=d) for d in (0, 1, 0, 1)]
        g = H.load_golden_queries()["inter_segment_x4"]
        for key, flt in (("unfiltered", ""), ("filtered", FILTER)):
            combined = host.execute_sql(segs, "SELECT COUNT(*), SUM(column1), SUM(column3) FROM testTable" + flt, max_execution_threads=4)["combined"]
            assert combined["final"] == [float(g["count"][key]), g["sum_column1"][key], g["sum_column3"][key]]
        rows = host.execute_sql(segs, "SELECT SUM(column1) FROM testTable GROUP BY column11")["combined"]["groups"]
        assert sorted([r["key"][0], r["final"][0]] for r in rows) == sorted(H.load_golden_queries()["inter_segment_group_by_x4"]["sum_column1_by_column11"])
    finally:
        for s in segs:
            s.destroy()
        if aliased:
            os.environ.pop("PINOT_GPU_ALIAS_DEVICES", None)
            host.init_plan_maker(device=0, time_kernels=True)


def test_fast_filtered_count_cases_of_the_reference_test():
    """FastFilteredCountTest.java:103-318 with its own SQL (the TEXT_MATCH / JSON_MATCH entries are not on this path): 1000 records,
    `sorted` = i (sorted column, inverted index), `class` = i % 8 (inverted index), `intRangeCol` = 1000 - i (the reference gives it a range
    index, here it is scanned).  COUNT(*) over filters the indexes answer alone never scans an entry."""
    from pinot_amd import segment as S
    n, bucket = 1000, 8
    i = np.arange(n, dtype=np.int32)
    cols = [S.Column.dict_encoded("sorted", i, with_inverted=True), S.Column.dict_encoded("class", i % bucket, with_inverted=True),
            S.Column.dict_encoded("intRangeCol", n - i)]
    seg = host.HostSegment(S.SegmentData("testSegment", n, cols))
    bc, bcc, lo, hi = n // bucket, n - n // bucket, 20, n - 20
    all_buckets, two = "(0, 1, 2, 3, 4, 5, 6, 7)", "(0, 7)"
    t = "select count(*) from testTable"
    cases = [(t, n), (t + " where class = 1", bc), (t + " where sorted = 1", 1), (t + " where sorted between %d and %d" % (lo, hi), hi - lo + 1),
             (t + " where sorted not between %d and %d" % (lo, hi), n - (hi - lo + 1)), (t + " where sorted in " + all_buckets, bucket),
             (t + " where sorted in " + all_buckets + " and class in " + all_buckets, bucket), (t + " where class <> 1", bcc),
             (t + " where class in " + two, 2 * bc), (t + " where class not in " + two, n - 2 * bc),
             (t + " where class in " + two + " and sorted < %d" % (n // 2), bc), (t + " where sorted = 1 and class = 1", 1),
             (t + " where sorted = 1 and class <> 1", 0), (t + " where sorted = 1 and class <> 0", 1),
             (t + " where sorted <> 1 and class = 1", bc - 1), (t + " where sorted >= 0 and class = 1", bc), (t + " where sorted > 1 and class = 1", bc - 1),
             (t + " where sorted >= 0 and class <> 1", bcc), (t + " where sorted >= 0 or class <> 0", n),
             (t + " where sorted < %d and class <> 0" % bc, bc - bc // bucket - 1), (t + " where sorted >= %d and class <> 0" % bc, bcc - bcc // bucket),
             (t + " where sorted < %d and class = %d" % (bucket - 1, bucket - 1), 0), (t + " where sorted >= %d and class = %d" % (bucket - 2, bucket - 2), bc),
             (t + " where sorted >= %d and sorted < %d and class = 0" % (lo, hi), bc - (lo + n - hi) // bucket),
             (t + " where intRangeCol >= %d and intRangeCol < %d" % (lo, hi), hi - lo), (t + " where intRangeCol < %d" % hi, hi - 1),
             (t + " where intRangeCol not between %d and %d" % (lo, hi), n - hi + lo - 1),
             (t + " where intRangeCol between %d and %d and class = 0" % (lo, hi), bc - (lo + n - hi) // bucket),
             (t + " where intRangeCol not between %d and %d and class = 0" % (lo, hi), (lo + n - hi) // bucket)]
    try:
        for sql, want in cases:
            b = host.execute_sql([seg], sql)["segments"][0]
            assert b["intermediate"] == [want], sql
            assert b["stats"]["numDocsScanned"] == want and b["stats"]["numTotalDocs"] == n, sql
            if "intRangeCol" not in sql:
                assert b["stats"]["numEntriesScannedInFilter"] == 0, sql          # sorted / inverted indexes only: nothing is scanned
        two_segments = host.execute_sql([seg, seg], t + " where class in " + two + " and sorted < 500")["combined"]       # getIndexSegments(): the segment twice
        assert two_segments["final"] == [2.0 * bc]
    finally:
        seg.destroy()


def test_group_by_raw_key_columns_through_sql():
    """GROUP BY over no-dictionary INT / LONG columns (DefaultGroupByExecutor.java:106-121: the no-dictionary key generators) through
    the plan maker: the device's entries are offsets from the column's smallest value, the block's keys are the values; two segments
    with different value ranges merge on the values."""
    from pinot_amd import segment as S
    rng = np.random.default_rng(21)
    segs, values = [], []
    try:
        for s, (lo, span) in enumerate(((-40, 90), (10, 200))):
            n = 20_000 + 7 * s
            k = rng.integers(lo, lo + span, n).astype(np.int32)
            big = (rng.integers(0, 50, n).astype(np.int64) * 1_000_003 + 10 ** 11)
            d = rng.integers(0, 9, n).astype(np.int32)
            v = rng.integers(0, 1000, n).astype(np.int32)
            segs.append(host.HostSegment(S.SegmentData("rawkeys%d" % s, n, [S.Column.raw("k", k), S.Column.raw_typed("big", big), S.Column.dict_encoded("d", d), S.Column.dict_encoded("v", v)])))
            values.append((k, big, d, v))
        for key_cols, sql in ((("k",), "SELECT k, SUM(v), COUNT(*) FROM testTable GROUP BY k LIMIT 100000"),
                              (("big", "d"), "SELECT big, d, SUM(v), COUNT(*) FROM testTable WHERE v < 700 GROUP BY big, d LIMIT 100000")):
            out = host.execute_sql(segs, sql)
            want = {}
            for k, big, d, v in values:
                cols = {"k": k, "big": big, "d": d}
                mask = v < 700 if "WHERE" in sql else np.ones(len(v), bool)
                for row, val in zip(zip(*[cols[c][mask].tolist() for c in key_cols]), v[mask].tolist()):
                    acc = want.setdefault(tuple(row), [0.0, 0])
                    acc[0] += val; acc[1] += 1
            got = {tuple(g["key"]): g["final"] for g in out["combined"]["groups"]}
            assert got == {k: [float(s), c] for k, (s, c) in want.items()}, sql
    finally:
        for seg in segs:
            seg.destroy()


def test_group_by_raw_double_and_wide_long_key_columns_through_sql():
    """Round 5: GROUP BY over a no-dictionary DOUBLE column and a LONG column spanning more than an int (NoDictionarySingleColumnGroupKey
    Generator.java:100-135 keys them by value; the device through a dictionary it builds from each segment's column, pg_group_key_values):
    the blocks' keys are the values, two segments with different value sets merge on the values."""
    from pinot_amd import segment as S
    rng = np.random.default_rng(23)
    segs, values = [], []
    try:
        for s in range(2):
            n = 20_000 + 11 * s
            price = np.round(rng.normal(100.0 + 5 * s, 20.0, 300), 2)[rng.integers(0, 300, n)]
            wide = (rng.integers(-40, 40, 70).astype(np.int64) * (2 ** 40 + 17 + s))[rng.integers(0, 70, n)]
            d = rng.integers(0, 9, n).astype(np.int32)
            v = rng.integers(0, 1000, n).astype(np.int32)
            segs.append(host.HostSegment(S.SegmentData("rankkeys%d" % s, n, [S.Column.raw_typed("price", price.astype(np.float64)), S.Column.raw_typed("wide", wide),
                                                                           S.Column.dict_encoded("d", d), S.Column.dict_encoded("v", v)])))
            values.append((price, wide, d, v))
        for key_cols, sql in ((("price",), "SELECT price, SUM(v), COUNT(*) FROM testTable GROUP BY price LIMIT 100000"),
                              (("wide", "d"), "SELECT wide, d, SUM(v), COUNT(*) FROM testTable WHERE v < 700 GROUP BY wide, d LIMIT 100000")):
            out = host.execute_sql(segs, sql)
            want = {}
            for price, wide, d, v in values:
                cols = {"price": price, "wide": wide, "d": d}
                mask = v < 700 if "WHERE" in sql else np.ones(len(v), bool)
                for row, val in zip(zip(*[cols[c][mask].tolist() for c in key_cols]), v[mask].tolist()):
                    acc = want.setdefault(tuple(row), [0.0, 0])
                    acc[0] += val; acc[1] += 1
            got = {tuple(g["key"]): g["final"] for g in out["combined"]["groups"]}
            assert got == {k: [float(s), c] for k, (s, c) in want.items()}, sql
    finally:
        for seg in segs:
            seg.destroy()


def test_combine_through_one_batch_equals_one_execute_per_segment(golden_segments, monkeypatch):
    """executeCombined registers the segment operators of a query with ONE pg_execute_batch (GpuBatch in host/plan_maker.cpp: the first
    BaseCombineOperator task to ask for its block runs the batch, BaseCombineOperator.java:85-142); with
    pinot.server.query.executor.gpu.batch = false every operator runs its own pg_execute.  Same blocks either way -- aggregations,
    filters, group-by (int and string keys), FILTER (WHERE) swim lanes, fewer tasks than segments."""
    _, segs = golden_segments
    f1 = "column1 > 100000000 AND column11 NOT IN ('t', 'P')"
    sqls = ["SELECT COUNT(*), SUM(column1), MAX(column3), MIN(column6), AVG(column7) FROM testTable",
            "SELECT COUNT(*), SUM(column1), SUM(column3) FROM testTable" + FILTER,
            "SELECT COUNT(*), SUM(column1) FROM testTable WHERE column1 > 100000000 AND column3 < 1000000000",
            "SELECT SUM(column1), MAX(column3) FROM testTable GROUP BY column9 LIMIT 100000",
            "SELECT COUNT(*), MAX(column1) FROM testTable" + FILTER + " GROUP BY column11, column12 LIMIT 100000",
            f"SELECT SUM(column1) FILTER (WHERE {f1}), COUNT(*), AVG(column7) FILTER (WHERE {f1}) FROM testTable WHERE column3 BETWEEN 20000000 AND 1000000000"]

    def strip(block):
        return {k: v for k, v in block.items() if k not in ("deviceMs", "kernelMs")}

    def run_all():
        return [[strip(host.execute_sql(segs, sql, max_execution_threads=t)["combined"]) for t in (4, 2, 1)] for sql in sqls]

    try:
        monkeypatch.setenv("PINOT_GPU_HOST_BATCH", "0")
        host.init_plan_maker(device=0, time_kernels=True)
        one_by_one = run_all()
        monkeypatch.delenv("PINOT_GPU_HOST_BATCH")
        host.init_plan_maker(device=0, time_kernels=True)
        batched = run_all()
    finally:
        monkeypatch.delenv("PINOT_GPU_HOST_BATCH", raising=False)
        host.init_plan_maker(device=0, time_kernels=True)
    assert batched == one_by_one
    for per_threads in batched:
        assert per_threads[0] == per_threads[1] == per_threads[2]
    assert batched[0][0]["stats"]["numTotalDocs"] == 120000
    # a query the device declines at plan time is still declined for every segment before anything runs
    with pytest.raises(host.HostError):
        host.execute_sql(segs, "SELECT column1 FROM testTable LIMIT 5")
