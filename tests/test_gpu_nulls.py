"""GPU parity tests of null handling (query option enableNullHandling, null value vectors, IS NULL / IS NOT NULL) through the C ABI and
the host mirror: the reference's own known-answer tables (tests/golden/null_handling_kats.json), the oracle on random trees and
columns, and a per-doc restatement of the three-set filter rules."""
import numpy as np
import pytest

from oracle import oracle
from pinot_amd import _abi
from pinot_amd import host
from pinot_amd import query as Q
from pinot_amd import segment as S
import helpers as H
import null_cases as NC
import segment_dirs as D
from test_oracle_nulls import kat_filter_segment, random_nullable_segment, random_tree, run_aggregation_kat

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("data_type", ["INT", "LONG", "FLOAT", "DOUBLE"])
@pytest.mark.parametrize("raw", [False, True])
def test_aggregation_kats_of_the_reference(engine, data_type, raw):
    def execute(seg_data, spec):
        with engine.open(seg_data) as seg:
            return seg.execute(spec)
    for case in NC.load_kats()["aggregation"]:
        run_aggregation_kat(execute, case, data_type, raw)


@pytest.mark.parametrize("raw", [False, True])
def test_filter_kats_of_the_reference(engine, raw):
    for case in NC.load_kats()["filter"]:
        data = kat_filter_segment(case, raw)
        seg = engine.open(data)
        try:
            spec = Q.QuerySpec([(Q.COUNT, -1)], filter=NC.tree_for(data, case["filter"]), null_handling=True)
            assert seg.execute(spec).aggregations[0].count == case["expected_count"], case["ref"]
            words, card = seg.filter_bitmap(spec)
            assert card == case["expected_count"]
            if "expected_rows" in case:
                assert [d for d in range(data.num_docs) if (int(words[d >> 6]) >> (d & 63)) & 1] == case["expected_rows"], case["ref"]
        finally:
            seg.close()


@pytest.mark.parametrize("raw_second", [False, True])
def test_random_filter_trees_against_the_oracle_and_the_per_doc_rules(engine, raw_second):
    rng = np.random.default_rng(99 + int(raw_second))
    data, values, nulls = random_nullable_segment(rng, 70000, raw_second)     # 35 tiles, two roaring containers
    raw_columns = ("c2",) if raw_second else ()
    seg = engine.open(data)
    ran = rejected = 0
    try:
        for i in range(40):
            tree = random_tree(rng, depth=1)
            inverted = bool(i & 1)
            spec = Q.QuerySpec([(Q.COUNT, -1)], filter=NC.tree_for(data, tree, inverted=inverted), null_handling=True)
            want_words, want_card = oracle.filter_bitmap(data, spec)
            ref = NC.reference_trues(tree, values, nulls, data.num_docs, raw_columns)
            assert want_card == int(ref.sum())
            try:
                words, card = seg.filter_bitmap(spec)
            except _abi.PinotGpuError as e:       # plan-time rejection: the rewritten tree outgrew the kernel's node / leaf tables
                assert e.status == _abi.PG_ERR_UNSUPPORTED, tree
                rejected += 1
                continue
            assert card == want_card and np.array_equal(words[:len(want_words)], want_words), tree
            ran += 1
    finally:
        seg.close()
    assert ran >= 25, (ran, rejected)


@pytest.mark.parametrize("num_docs", [3000, 200000])
def test_aggregation_lanes_skip_the_nulls_of_their_own_column(engine, num_docs):
    rng = np.random.default_rng(5 + num_docs)
    data, values, nulls = random_nullable_segment(rng, num_docs)
    # a DOUBLE metric with a dense null vector (bitmap containers) and a raw LONG one with a run of nulls (run container)
    dv = rng.normal(0, 1000, num_docs)
    dm = rng.random(num_docs) < 0.6
    dv[dm] = 0.0
    lv = rng.integers(-2 ** 40, 2 ** 40, num_docs)
    lm = np.zeros(num_docs, bool)
    lm[num_docs // 4: num_docs // 2] = True
    lv[lm] = 0
    cols = list(data.columns) + [S.Column.dict_encoded_typed("d", dv).with_nulls(dm), S.Column.raw_typed("l", lv).with_nulls(lm)]
    data = S.SegmentData("nullable", num_docs, cols)
    seg = engine.open(data)
    try:
        trees = [None, ["GT", "c3", 0], ["OR", ["GT", "c3", 5], ["NOT", ["LT", "c2", 5]]], ["AND", ["IS_NOT_NULL", "l"], ["NOT", ["EQ", "c1", 3]]]]
        agg_lists = [[(Q.COUNT, -1), (Q.COUNT, 0), (Q.SUM, 0), (Q.MIN, 1), (Q.MAX, 0), (Q.AVG, 1), (Q.SUM, 2)],
                     [(Q.SUM, 3), (Q.SUM, 2), (Q.MIN, 3), (Q.AVG, 4), (Q.COUNT, 3), (Q.MAX, 4)]]
        for tree in trees:
            flt = NC.tree_for(data, tree) if tree else None
            for nh in (True, False):
                for aggs in agg_lists:
                    spec = Q.QuerySpec(aggs, filter=flt, null_handling=nh)
                    want = oracle.execute(data, spec)
                    got = seg.execute(spec)
                    H.assert_results_equal(got, want)
        # every doc of the aggregated column is null under the filter -> count 0 (the reference's holder stays null)
        spec = Q.QuerySpec([(Q.SUM, 4), (Q.COUNT, 4), (Q.COUNT, -1)], filter=Q.leaf(Q.Pred.is_null(4)), null_handling=True)
        got, want = seg.execute(spec), oracle.execute(data, spec)
        H.assert_results_equal(got, want)
        assert got.aggregations[0].count == 0 and got.aggregations[1].count == 0 and got.aggregations[2].count == int(lm.sum())
        # metadata-only answers stay available when no aggregated column has nulls, and are dropped when one has
        fast = seg.execute(Q.QuerySpec([(Q.COUNT, -1), (Q.MAX, 2)], null_handling=True))
        slow = seg.execute(Q.QuerySpec([(Q.COUNT, -1), (Q.MAX, 0)], null_handling=True))
        assert fast.stats[2] == 0 and slow.stats[2] == num_docs and slow.aggregations[1].count == int((~nulls["c1"]).sum())
        H.assert_results_equal(slow, oracle.execute(data, Q.QuerySpec([(Q.COUNT, -1), (Q.MAX, 0)], null_handling=True)))
        # GROUP BY under null handling: NULL is a key of its own (the key column's null-key image), every function skips the nulls of
        # its own column (one lane per nullable aggregated column, merged by group id); numGroupsLimit binds at any key-space size
        for keys in ([2], [0], [1, 0], [0, 2, 1], [3]):
            for tree in (None, ["NOT", ["GT", "c1", 0]], ["OR", ["GT", "c3", 5], ["NOT", ["LT", "c2", 5]]]):
                for aggs in ([(Q.SUM, 2), (Q.COUNT, -1)], [(Q.COUNT, -1), (Q.SUM, 0), (Q.COUNT, 1), (Q.MIN, 1), (Q.MAX, 0), (Q.AVG, 2), (Q.SUM, 2)], [(Q.SUM, 3), (Q.AVG, 3), (Q.MAX, 0)]):
                    spec = Q.QuerySpec(aggs, filter=NC.tree_for(data, tree) if tree else None, group_by=keys, null_handling=True)
                    H.assert_results_equal(seg.execute(spec), oracle.execute(data, spec))
        for limit in (3, 5, 41):
            spec = Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, 0)], group_by=[0], null_handling=True, num_groups_limit=limit)
            got, want = seg.execute(spec), oracle.execute(data, spec)
            H.assert_results_equal(got, want)
            assert got.num_groups_limit_reached == want.num_groups_limit_reached
        # a raw nullable column cannot be a key on either side (group-by keys are dictionary columns)
        with pytest.raises(_abi.PinotGpuError) as e:
            seg.execute(Q.QuerySpec([(Q.COUNT, -1)], group_by=[4], null_handling=True))
        assert e.value.status in (_abi.PG_ERR_UNSUPPORTED, _abi.PG_ERR_INVALID_ARGUMENT)
    finally:
        seg.close()


def test_sql_null_handling_through_the_host_mirror(tmp_path):
    import torch  # noqa: F401
    host.init_plan_maker(device=0, time_kernels=True)
    kats = NC.load_kats()
    # the reference's two-instance aggregation tables through SQL + combine: the final result is null when every value is null
    for case in kats["aggregation"]:
        segs = [host.HostSegment(S.SegmentData("testTable", len(rows), [NC.nullable_column("myField", rows, "INT", case["field_type"])]))
                for rows in case["segments"]]
        try:
            prefix = "SET enableNullHandling = true; " if case["null_handling"] else ""
            out = host.execute_sql(segs, prefix + "SELECT %s(myField) FROM testTable" % case["function"])["combined"]["final"][0]
            want = case["expected"]
            if want == "DEFAULT":
                want = float(NC.DEFAULT_NULL[(case["field_type"], "INT")])
            assert out == want, (case["ref"], out, want)
        finally:
            for s in segs:
                s.destroy()
    for case in kats["filter"]:
        data = kat_filter_segment(case)
        seg = host.HostSegment(data)
        try:
            def sql(s):
                if s[0] in ("AND", "OR"):
                    return "(" + (" %s " % s[0]).join(sql(c) for c in s[1:]) + ")"
                if s[0] == "NOT":
                    return "NOT " + sql(s[1])
                return "%s %s %d" % (s[1], {"LT": "<", "GT": ">", "EQ": "="}[s[0]], s[2])
            block = host.execute_sql([seg], "SET enableNullHandling = true; SELECT COUNT(*) FROM testTable WHERE " + sql(case["filter"]))["segments"][0]
            assert block["intermediate"] == [case["expected_count"]], case["ref"]
        finally:
            seg.destroy()
    # IS NULL / IS NOT NULL, COUNT(col) and FILTER lanes over a directory whose columns carry .bitmap.nullvalue files (v1) / nullvalue_vector (v3)
    rng = np.random.default_rng(3)
    data, values, nulls = random_nullable_segment(rng, 20000)
    for writer in (D.write_v1, D.write_v3):
        seg = host.DirectorySegment(writer(tmp_path, "nullable_" + writer.__name__, data.num_docs, data.columns), device=0)
        try:
            d = {c["name"]: c for c in seg.describe()["columns"]}
            assert d["c1"]["hasNullValueVector"] and d["c2"]["hasNullValueVector"] and not d["c3"]["hasNullValueVector"]
            b = host.execute_sql([seg], "SELECT COUNT(*) FROM nullable WHERE c1 IS NULL AND c2 IS NOT NULL")["segments"][0]
            assert b["intermediate"] == [int((nulls["c1"] & ~nulls["c2"]).sum())] and b["stats"]["numEntriesScannedInFilter"] == 0
            q = ("SET enableNullHandling = true; SELECT COUNT(*), COUNT(c1), SUM(c1), MIN(c2) FILTER (WHERE c3 > 0), AVG(c1) FILTER (WHERE c1 IS NULL) "
                 "FROM nullable WHERE NOT c2 < 5")
            b = host.execute_sql([seg], q)["segments"][0]
            f = ~(values["c2"] < 5) & ~nulls["c2"]
            m1 = f & ~nulls["c1"]
            lane = f & (values["c3"] > 0) & ~nulls["c2"]
            assert b["intermediate"] == [int(f.sum()), int(m1.sum()), float(values["c1"][m1].astype(np.int64).sum()), float(values["c2"][lane].min()), None]
            assert b["final"][4] is None
            # without the option nulls are ordinary default values
            b = host.execute_sql([seg], "SELECT COUNT(c1), MIN(c1) FROM nullable WHERE NOT c2 < 5")["segments"][0]
            assert b["intermediate"] == [int((~(values["c2"] < 5)).sum()), float(-2 ** 31)]
        finally:
            seg.destroy()
