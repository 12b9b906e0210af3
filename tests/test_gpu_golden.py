"""GPU tests against the reference's own golden vectors (test_data-sv.avro fixture) through the C ABI."""
import numpy as np
import pytest

from oracle import oracle
from pinot_amd import query as Q
import helpers as H

pytestmark = pytest.mark.gpu


def _check_inner(vals, want):
    count, s1, mx3, mn6, avg7 = vals
    assert count.intermediate(Q.COUNT) == want["count"]
    assert s1.intermediate(Q.SUM) == float(want["sum_column1"]) and s1.sum_i64 == want["sum_column1"]
    assert mx3.intermediate(Q.MAX) == float(want["max_column3"])
    assert mn6.intermediate(Q.MIN) == float(want["min_column6"])
    assert avg7.intermediate(Q.AVG) == (float(want["avg_column7"][0]), want["avg_column7"][1])


def test_num_entries_scanned_in_filter_goldens(engine):
    """InnerSegmentAggregationSingleValueQueriesTest :56,108,127,148,169: all four ExecutionStatistics, numEntriesScannedInFilter = 63064
    included -- the reference's operator tree (sorted docId range, OR of a scan and a posting, two scan leaves) whose OR leap-frogs with
    the merged bitmap, so the count comes from the iterator replay; plus the shapes the kernels count themselves."""
    g = H.load_golden_queries()
    seg = H.golden_segment()
    flt = H.golden_filter_physical(seg)
    with engine.open(seg) as gseg:
        for group_by, key in ((None, "inner_segment"), ([seg.column_index("column9")], "inner_segment_group_by_column9")):
            spec = Q.QuerySpec(H.golden_aggregations(seg), filter=flt, group_by=group_by or ())
            res = gseg.execute(spec)
            assert res.filter_entries_exact and list(res.stats) == g[key]["filtered"]["stats"]
            H.assert_results_equal(res, oracle.execute(seg, spec))
        mw = g["inner_segment_group_by_medium"]["filtered"]
        cols, _ = H.golden_medium_group(seg, mw)
        mres = gseg.execute(Q.QuerySpec(H.golden_aggregations(seg), filter=flt, group_by=cols))
        assert list(mres.stats) == mw["stats"]
        # root AND of plain leaves behind an index-based child: counted by the scan kernel itself (ScanBasedDocIdIterator.applyAnd)
        c1 = Q.leaf(H.range_pred(seg, "column1", lower=100000000, lower_inclusive=False))
        c3 = Q.leaf(H.range_pred(seg, "column3", lower=20000000, upper=1000000000))
        days = Q.leaf(Q.Pred.doc_range(0, 24718))
        c11 = Q.leaf(H.string_in_pred(seg, "column11", ["t", "P"], exclusive=True, inverted=True))
        for chain in (Q.and_(days, c1, c3), Q.and_(c1, days, c3), Q.and_(c11, c3, c1), Q.and_(c3, c11, days, c1)):
            for group_by in (None, [seg.column_index("column9")]):
                spec = Q.QuerySpec(H.golden_aggregations(seg), filter=chain, group_by=group_by or ())
                res, want = gseg.execute(spec), oracle.execute(seg, spec)
                assert res.filter_entries_exact and res.stats == want.stats and 0 < res.stats[1] < 2 * 30000, (res.stats, want.stats)
        # scan leaves only (AndDocIdIterator leap-frogs them), OR / NOT above scans, nested AND: the replay
        for other in (Q.and_(c1, c3), Q.or_(c1, c3), Q.not_(c1), Q.and_(c1, Q.or_(c3, c11)), Q.or_(Q.and_(c1, c3), days), Q.not_(Q.and_(c1, Q.not_(c3)))):
            spec = Q.QuerySpec(H.golden_aggregations(seg), filter=other)
            res, want = gseg.execute(spec), oracle.execute(seg, spec)
            assert res.filter_entries_exact and res.stats == want.stats, (res.stats, want.stats)


def test_inner_segment_goldens(engine):
    g = H.load_golden_queries()
    seg = H.golden_segment()
    c9 = seg.column("column9")
    with engine.open(seg) as gseg:
        for key, inverted in (("unfiltered", False), ("filtered", False), ("filtered", True)):
            flt = None if key == "unfiltered" else H.golden_filter(seg, inverted)
            spec = Q.QuerySpec(H.golden_aggregations(seg), filter=flt)
            res = gseg.execute(spec)
            want = g["inner_segment"][key]
            _check_inner(res.aggregations, want)
            assert (res.stats[0], res.stats[2], res.stats[3]) == (want["stats"][0], want["stats"][2], want["stats"][3])
            H.assert_results_equal(res, oracle.execute(seg, spec))
            gspec = Q.QuerySpec(H.golden_aggregations(seg), filter=flt, group_by=[seg.column_index("column9")])
            gres = gseg.execute(gspec)
            gw = g["inner_segment_group_by_column9"][key]
            gid = int(np.searchsorted(c9.dict_values, gw["key"]))
            _check_inner(gres.groups[gid], gw)
            assert (gres.stats[0], gres.stats[2], gres.stats[3]) == (gw["stats"][0], gw["stats"][2], gw["stats"][3])
            H.assert_results_equal(gres, oracle.execute(seg, gspec))
            # testMediumAggregationGroupBy :114-132: 78 165 raw keys, the reference's INT_MAP_BASED holder -> HBM table + device compaction
            mw = g["inner_segment_group_by_medium"][key]
            cols, raw = H.golden_medium_group(seg, mw)
            mspec = Q.QuerySpec(H.golden_aggregations(seg), filter=flt, group_by=cols)
            mres = gseg.execute(mspec)
            _check_inner(mres.groups[raw], mw)
            assert (mres.stats[0], mres.stats[2], mres.stats[3]) == (mw["stats"][0], mw["stats"][2], mw["stats"][3])
            assert mres.group_id_upper_bound == 1737 * 5 * 9 and not mres.num_groups_limit_reached
            H.assert_results_equal(mres, oracle.execute(seg, mspec))


def test_large_and_very_large_group_by_goldens_long_and_array_map_holders(engine):
    """InnerSegmentAggregationSingleValueQueriesTest.testLargeAggregationGroupBy :134-153 (five key columns, LONG_MAP_BASED) and
    testVeryLargeAggregationGroupBy :155-176 (nine, ARRAY_MAP_BASED: the hashed table behind two chained first tables), both filter
    variants -- the reference's key tuples, values and all four statistics -- and every row against the oracle."""
    from test_oracle_golden import LARGE_GOLDENS, check_large_group_by_goldens
    g = H.load_golden_queries()
    seg = H.golden_segment()
    with engine.open(seg) as gseg:
        check_large_group_by_goldens(gseg.execute, seg)
        for row, kind in LARGE_GOLDENS:
            cols = [seg.column_index(c) for c in g[row]["group_by"]]
            for flt in (None, H.golden_filter_physical(seg), H.golden_filter(seg)):
                spec = Q.QuerySpec(H.golden_aggregations(seg), filter=flt, group_by=cols)
                got, want = gseg.execute(spec), oracle.execute(seg, spec)
                H.assert_results_equal(got, want)
                assert got.group_key_kind == want.group_key_kind == kind
                assert got.group_keys == want.group_keys and got.group_ids64 == want.group_ids64
            # numGroupsLimit binds: the first keys in docId order survive (LongGroupIdMap / ArrayGroupIdMap hand out ids by first appearance)
            spec = Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, seg.column_index("column1"))], group_by=cols, num_groups_limit=1000)
            got, want = gseg.execute(spec), oracle.execute(seg, spec)
            H.assert_results_equal(got, want)
            assert got.num_groups_limit_reached and len(got.groups) == 1000 and got.group_keys == want.group_keys


def test_large_group_by_goldens_with_raw_key_columns(engine):
    """The same goldens with column1 / column3 / column9 stored without a dictionary: NoDictionaryMultiColumnGroupKeyGenerator's keys by
    value (through pg_group_key_info), raw range leaves, raw SUM / MAX inputs -- the reference's results do not depend on the encoding."""
    from test_oracle_golden import RAW_KEY_COLUMNS, check_large_group_by_goldens
    g = H.load_golden_queries()
    seg = H.golden_segment(raw_columns=RAW_KEY_COLUMNS)
    with engine.open(seg) as gseg:
        def base_of(c):
            base, is_offset, _ = gseg.group_key_info(c)
            assert is_offset
            return base
        check_large_group_by_goldens(gseg.execute, seg, base_of=base_of, check_kind=False)
        for key, flt in (("unfiltered", None), ("filtered", H.golden_filter_physical(seg))):
            spec = Q.QuerySpec(H.golden_aggregations(seg), filter=flt)
            res = gseg.execute(spec)
            _check_inner(res.aggregations, g["inner_segment"][key])
            assert list(res.stats) == g["inner_segment"][key]["stats"]
            for row in ("inner_segment_group_by_large", "inner_segment_group_by_very_large"):
                gspec = Q.QuerySpec(H.golden_aggregations(seg), filter=flt, group_by=[seg.column_index(c) for c in g[row]["group_by"]])
                H.assert_results_equal(gseg.execute(gspec), oracle.execute(seg, gspec))


def test_inter_segment_goldens(engine):
    g = H.load_golden_queries()["inter_segment_x4"]
    seg = H.golden_segment()
    ci = seg.column_index
    aggs = [(Q.COUNT, -1), (Q.SUM, ci("column1")), (Q.SUM, ci("column3"))]
    segments = [engine.open(seg) for _ in range(4)]
    try:
        for key in ("unfiltered", "filtered"):
            flt = None if key == "unfiltered" else H.golden_filter(seg)
            parts = [s.execute(Q.QuerySpec(aggs, filter=flt)) for s in segments]
            count = sum(p.aggregations[0].intermediate(Q.COUNT) for p in parts)
            s1 = s3 = 0.0
            for p in parts:   # SumAggregationFunction.merge: double '+'
                s1 = s1 + p.aggregations[1].intermediate(Q.SUM)
                s3 = s3 + p.aggregations[2].intermediate(Q.SUM)
            assert count == g["count"][key]
            assert s1 == g["sum_column1"][key] and s3 == g["sum_column3"][key]
    finally:
        for s in segments:
            s.close()


def test_segment_written_by_the_reference_java_writers(engine):
    """The `age` column of pinot-core/src/test/resources/data/paddingOld.tar.gz (bytes produced by the reference's Java
    writers, stored in tests/golden/pinot_v1_segment_paddingOld.json) opened as-is through the C ABI."""
    import json
    import os
    from pinot_amd import _abi
    from pinot_amd import segment as S
    g = json.load(open(os.path.join(H.GOLDEN_DIR, "pinot_v1_segment_paddingOld.json")))
    age = g["columns"]["age"]
    col = S.Column("age", _abi.PG_FWD_FIXED_BIT_DICT, age["bitsPerElement"], age["cardinality"],
                   np.frombuffer(bytes.fromhex(age["fwd_hex"]), dtype=np.uint8).copy(),
                   np.frombuffer(bytes.fromhex(age["dict_hex"]), dtype=np.uint8).copy())
    seg = S.SegmentData("paddingOld", g["total_docs"], [col])
    ages = [1228, 837, 1209, 617, 824]     # dictIds [4, 2, 3, 0, 1] over the dictionary [617, 824, 837, 1209, 1228]
    with engine.open(seg) as gseg:
        assert gseg.read_int_values(0, np.arange(5, dtype=np.int32)).tolist() == ages
        assert gseg.read_dict_ids(0, np.arange(5, dtype=np.int32)).tolist() == [4, 2, 3, 0, 1]
        r = gseg.execute(Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, 0), (Q.MIN, 0), (Q.MAX, 0), (Q.AVG, 0)]))
        assert r.intermediates() == [5, float(sum(ages)), 617.0, 1228.0, (float(sum(ages)), 5)]
        s, e = oracle.lower_range(col.dictionary, 5, lower=800, lower_inclusive=False)      # age > 800
        r = gseg.execute(Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, 0)], filter=Q.leaf(Q.Pred.dict_range(0, s, e))))
        assert r.intermediates() == [4, float(1228 + 837 + 1209 + 824)]
        H.assert_results_equal(r, oracle.execute(seg, Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, 0)], filter=Q.leaf(Q.Pred.dict_range(0, s, e)))))


def test_non_scan_based_aggregation_plan(engine):
    """COUNT / dictionary-based MIN, MAX without a filter are answered from the metadata (NonScanBasedAggregationOperator): same
    values and the reference's statistics (totalDocs, 0, 0, totalDocs), no kernel launch."""
    seg = H.golden_segment(use_inverted=False)
    ci = seg.column_index
    with engine.open(seg) as g:
        spec = Q.QuerySpec([(Q.COUNT, -1), (Q.MAX, ci("column3")), (Q.MIN, ci("column6"))])
        r = g.execute(spec)
        assert r.intermediates() == [30000, 2147419555.0, 1689277.0] and r.stats == (30000, 0, 0, 30000)
        H.assert_results_equal(r, oracle.execute(seg, spec))
        spec = Q.QuerySpec([(Q.COUNT, -1)], filter=Q.leaf(Q.Pred.match_all()))
        assert g.execute(spec).stats == (30000, 0, 0, 30000)


def test_inter_segment_max_min_avg_goldens(engine):
    """InterSegmentAggregationSingleValueQueriesTest.testMax / testMin / testSum / testAvg through the C ABI (values and the
    statistics that show the non-scan plan for the unfiltered MAX / MIN)."""
    from test_oracle_golden import check_inter_segment_max_min_avg
    seg = H.golden_segment()
    with engine.open(seg) as g:
        check_inter_segment_max_min_avg(g.execute, seg)


def test_string_key_group_by_goldens(engine):
    from test_oracle_golden import check_string_key_group_by
    seg = H.golden_segment()
    with engine.open(seg) as g:
        check_string_key_group_by(g.execute, seg)
