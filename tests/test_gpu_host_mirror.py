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
        assert (st["numDocsScanned"], st["numEntriesScannedPostFilter"], st["numTotalDocs"]) == (want["stats"][0], want["stats"][2], want["stats"][3])


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
        combined = host.execute_sql(segs, "SELECT COUNT(*) FROM testTable" + flt + " GROUP BY column9")["combined"]
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


def test_plan_time_rejection_and_errors(golden_segments):
    _, segs = golden_segments
    for sql, status in (("SELECT column1 FROM testTable", 2),
                        ("SELECT SUM(column1) FROM testTable GROUP BY column1, column3", 2),     # 6582 * 21910 > array-based threshold
                        ("SELECT SUM(nope) FROM testTable", 1),
                        ("SELECT SUM(column11) FROM testTable", 1),
                        ("SELECT SUM(column1) FROM testTable WHERE column1 = 'abc'", 1)):
        with pytest.raises(host.HostError) as e:
            host.execute_sql(segs[:1], sql)
        assert e.value.status == status, sql
