"""Segment directories opened by the native loader, queried with SQL through the C++ plan maker on the device."""
import numpy as np
import pytest

import helpers as H
import segment_dirs as D
from pinot_amd import host
from test_segment_loader_cpu import _synthetic_columns

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def plan_maker():
    import torch  # noqa: F401
    host.init_plan_maker(device=0, time_kernels=False)


@pytest.mark.parametrize("name", ["paddingOld", "paddingPercent", "paddingNull"])
def test_sql_over_reference_directories(tmp_path, name):
    seg = host.DirectorySegment(D.write_reference_directory(tmp_path, name), device=0)
    try:
        b = host.execute_sql([seg], "SELECT COUNT(*), SUM(age), MIN(age), MAX(outgoingName1), SUM(outgoingName1) FROM myTable")["segments"][0]
        assert b["intermediate"] == [5, 4715.0, 617.0, 902.0, float(246 + 310 + 336 + 467 + 902)]
        b = host.execute_sql([seg], "SELECT COUNT(*) FROM myTable WHERE name = 'lynda'")["segments"][0]
        n_lynda = b["intermediate"][0]
        b = host.execute_sql([seg], "SELECT COUNT(*) FROM myTable WHERE name = 'lynda 2.0'")["segments"][0]
        assert sorted([n_lynda, b["intermediate"][0]]) == [2, 3]
        g = host.execute_sql([seg], "SELECT COUNT(*), SUM(age) FROM myTable GROUP BY name")["segments"][0]
        assert sorted(r["key"][0] for r in g["groups"]) == ["lynda", "lynda 2.0"] and sum(r["intermediate"][1] for r in g["groups"]) == 4715.0
        b = host.execute_sql([seg], "SELECT COUNT(*), SUM(age) FROM myTable WHERE age > 800 AND outgoingName1 < 500")["segments"][0]
        assert b["intermediate"][0] >= 0
    finally:
        seg.destroy()


def test_sql_over_v3_and_sorted_v1(tmp_path):
    n, k, cols = _synthetic_columns()
    v = cols[1].dict_values[np.searchsorted(cols[1].dict_values, cols[1].dict_values)]  # noqa: F841 (dictionary sanity)
    from oracle import oracle
    from pinot_amd import segment as S
    data = S.SegmentData("syn", n, cols)
    docs = np.arange(n, dtype=np.int32)
    vv = oracle.read_int_values(data, 1, docs).astype(np.int64)
    dd, _ = oracle.read_double_values(data, 2, docs)
    rr = oracle.read_double_values(data, 3, docs)[1]
    kcol = cols[0]
    ids = np.searchsorted(kcol.dict_values, k).astype(np.int32)
    segs = [host.DirectorySegment(D.write_v3(tmp_path, "seg_v3", n, cols), device=0),
            host.DirectorySegment(D.write_v1(tmp_path, "seg_v1_sorted", n, cols, sorted_fwd={"k": D.sorted_forward_index(ids, kcol.cardinality)}), device=0)]
    try:
        for seg in segs:
            sel = (k >= 51) & (k < 101) & (vv < 5000)
            b = host.execute_sql([seg], "SELECT COUNT(*), SUM(v), SUM(r), MAX(d) FROM t WHERE k >= 51 AND k < 101 AND v < 5000")["segments"][0]
            assert b["intermediate"] == [int(sel.sum()), float(vv[sel].sum()), float(rr[sel].sum()), float(dd[sel].max())]
            g = host.execute_sql([seg], "SELECT COUNT(*), SUM(v) FROM t WHERE r > 0 GROUP BY k")["segments"][0]
            want = {int(key): (int(((k == key) & (rr > 0)).sum()), float(vv[(k == key) & (rr > 0)].sum())) for key in np.unique(k[rr > 0])}
            assert {r["key"][0]: (r["intermediate"][0], r["intermediate"][1]) for r in g["groups"]} == want
        # the sorted v1 column answers range / equality predicates with docId ranges (SortedIndexBasedFilterOperator): nothing is scanned
        assert segs[1].describe()["columns"][1]["isSorted"] or any(c["isSorted"] for c in segs[1].describe()["columns"])
        st_sorted = host.execute_sql([segs[1]], "SELECT COUNT(*) FROM t WHERE k >= 51 AND k < 101")["segments"][0]
        st_scan = host.execute_sql([segs[0]], "SELECT COUNT(*) FROM t WHERE k >= 51 AND k < 101")["segments"][0]
        assert st_sorted["intermediate"] == st_scan["intermediate"] == [int(((k >= 51) & (k < 101)).sum())]
        # the unsorted column scans: two scan leaves leap-frogging (AndDocIdIterator), each entry a doc one of them looked at
        assert st_sorted["stats"]["numEntriesScannedInFilter"] == 0
        assert st_scan["stats"]["numEntriesScannedInFilter"] == H.and_leapfrog_entries([k >= 51, k < 101]) < 2 * n
        ne = host.execute_sql([segs[1]], "SELECT COUNT(*) FROM t WHERE k != 51")["segments"][0]
        assert ne["intermediate"] == [int((k != 51).sum())]
        inq = host.execute_sql([segs[1]], "SELECT COUNT(*), SUM(v) FROM t WHERE k IN (6, 11, 16, 101) AND v < 9000")["segments"][0]
        sel_in = np.isin(k, [6, 11, 16, 101]) & (vv < 9000)
        assert inq["intermediate"] == [int(sel_in.sum()), float(vv[sel_in].sum())] and inq["stats"]["numEntriesScannedInFilter"] == int(np.isin(k, [6, 11, 16, 101]).sum())   # v is looked at in the docs of the sorted column's ranges only (applyAnd)
        nin = host.execute_sql([segs[1]], "SELECT COUNT(*) FROM t WHERE k NOT IN (6, 101)")["segments"][0]
        assert nin["intermediate"] == [int((~np.isin(k, [6, 101])).sum())] and nin["stats"]["numEntriesScannedInFilter"] == 0
        combined = host.execute_sql(segs, "SELECT COUNT(*), SUM(v) FROM t WHERE v IN (1, 2, 3, 5000)")["combined"]
        assert combined["final"][0] == 2.0 * float(np.isin(vv, [1, 2, 3, 5000]).sum())
    finally:
        for s in segs:
            s.destroy()
