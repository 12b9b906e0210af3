"""CPU tests: pin the oracle against the reference's own golden vectors (test_data-sv.avro fixture)."""
import numpy as np

from oracle import oracle
from pinot_amd import query as Q
import helpers as H


def _check_inner(res, want):
    count, s1, mx3, mn6, avg7 = res.aggregations
    assert count.intermediate(Q.COUNT) == want["count"]
    assert s1.intermediate(Q.SUM) == float(want["sum_column1"]) and s1.sum_i64 == want["sum_column1"]
    assert mx3.intermediate(Q.MAX) == float(want["max_column3"])
    assert mn6.intermediate(Q.MIN) == float(want["min_column6"])
    assert avg7.intermediate(Q.AVG) == (float(want["avg_column7"][0]), want["avg_column7"][1])
    docs, in_filter, post_filter, total = want["stats"]
    assert res.stats[0] == docs and res.stats[2] == post_filter and res.stats[3] == total


def test_inner_segment_aggregation_goldens():
    # InnerSegmentAggregationSingleValueQueriesTest.testAggregationOnly :44-61
    g = H.load_golden_queries()["inner_segment"]
    seg = H.golden_segment()
    _check_inner(oracle.execute(seg, Q.QuerySpec(H.golden_aggregations(seg))), g["unfiltered"])
    for inverted in (False, True):
        res = oracle.execute(seg, Q.QuerySpec(H.golden_aggregations(seg), filter=H.golden_filter(seg, inverted)))
        _check_inner(res, g["filtered"])


def test_num_entries_scanned_in_filter_golden():
    # InnerSegmentAggregationSingleValueQueriesTest :56,108,127: (6129, 63064, 24516, 30000) -- the iterator accounting of
    # AndDocIdSet.iterator / SVScanDocIdIterator / OrDocIdIterator on the operator tree FilterOperatorUtils builds
    g = H.load_golden_queries()
    seg = H.golden_segment()
    res = oracle.execute(seg, Q.QuerySpec(H.golden_aggregations(seg), filter=H.golden_filter_physical(seg)))
    _check_inner(res, g["inner_segment"]["filtered"])
    assert list(res.stats) == g["inner_segment"]["filtered"]["stats"] and res.filter_entries_exact
    gres = oracle.execute(seg, Q.QuerySpec(H.golden_aggregations(seg), filter=H.golden_filter_physical(seg), group_by=[seg.column_index("column9")]))
    assert list(gres.stats) == g["inner_segment_group_by_column9"]["filtered"]["stats"]
    # unfiltered: nothing scanned; one scan leaf alone: every doc once (SVScanDocIdIterator.next over the whole column)
    assert oracle.execute(seg, Q.QuerySpec(H.golden_aggregations(seg))).stats[1] == 0
    one = oracle.execute(seg, Q.QuerySpec(H.golden_aggregations(seg), filter=Q.leaf(H.range_pred(seg, "column1", lower=100000000, lower_inclusive=False))))
    assert one.stats[1] == 30000


def test_inner_segment_group_by_goldens():
    # testSmallAggregationGroupBy :96-112 (GROUP BY column9, ARRAY_BASED holder)
    g = H.load_golden_queries()["inner_segment_group_by_column9"]
    seg = H.golden_segment()
    c9 = seg.column("column9")
    for key, flt in (("unfiltered", None), ("filtered", H.golden_filter(seg))):
        res = oracle.execute(seg, Q.QuerySpec(H.golden_aggregations(seg), filter=flt, group_by=[seg.column_index("column9")]))
        want = g[key]
        gid = int(np.searchsorted(c9.dict_values, want["key"]))
        assert c9.value_of(gid) == want["key"]
        count, s1, mx3, mn6, avg7 = res.groups[gid]
        assert count.intermediate(Q.COUNT) == want["count"]
        assert s1.intermediate(Q.SUM) == float(want["sum_column1"])
        assert mx3.intermediate(Q.MAX) == float(want["max_column3"])
        assert mn6.intermediate(Q.MIN) == float(want["min_column6"])
        assert avg7.intermediate(Q.AVG) == (float(want["avg_column7"][0]), want["avg_column7"][1])
        assert res.stats[0] == want["stats"][0] and res.stats[2] == want["stats"][2] and res.stats[3] == want["stats"][3]


def test_inner_segment_medium_group_by_goldens_int_map_holder():
    # testMediumAggregationGroupBy :114-132 (GROUP BY column9, column11, column12 -> 78 165 raw keys, INT_MAP_BASED holder)
    g = H.load_golden_queries()["inner_segment_group_by_medium"]
    seg = H.golden_segment()
    for key, flt in (("unfiltered", None), ("filtered", H.golden_filter(seg))):
        want = g[key]
        cols, raw = H.golden_medium_group(seg, want)
        res = oracle.execute(seg, Q.QuerySpec(H.golden_aggregations(seg), filter=flt, group_by=cols))
        assert res.group_id_upper_bound == 1737 * 5 * 9 and not res.num_groups_limit_reached
        count, s1, mx3, mn6, avg7 = res.groups[raw]
        assert count.intermediate(Q.COUNT) == want["count"]
        assert s1.intermediate(Q.SUM) == float(want["sum_column1"])
        assert mx3.intermediate(Q.MAX) == float(want["max_column3"])
        assert mn6.intermediate(Q.MIN) == float(want["min_column6"])
        assert avg7.intermediate(Q.AVG) == (float(want["avg_column7"][0]), want["avg_column7"][1])
        assert res.stats[0] == want["stats"][0] and res.stats[2] == want["stats"][2] and res.stats[3] == want["stats"][3]
        assert sum(v[0].count for v in res.groups.values()) == want["stats"][0]


LARGE_GOLDENS = (("inner_segment_group_by_large", 1), ("inner_segment_group_by_very_large", 2))      # (fixture row, pg_result.group_key_kind)
RAW_KEY_COLUMNS = ("column1", "column3", "column9")


def check_large_group_by_goldens(execute, seg, base_of=None, exact_filter_stats=True, check_kind=True):
    """testLargeAggregationGroupBy :134-153 (LONG_MAP_BASED holder) and testVeryLargeAggregationGroupBy :155-176 (ARRAY_MAP_BASED), both
    filter variants: the key tuple, the five values and all four ExecutionStatistics, with `execute(spec) -> Result` for one segment.
    `base_of(column index)`: value of digit 0 of a raw (no-dictionary) key column (pg_group_key_info), None when every key column has a
    dictionary.  `check_kind`: the holder kind is a property of the dictionary-encoded segment the reference's test builds (a raw key
    column spans max - min + 1 digits on the ABI's raw-key scale, so the same query lands in a wider holder)."""
    g = H.load_golden_queries()
    for row, kind in LARGE_GOLDENS:
        for key, flt in (("unfiltered", None), ("filtered", H.golden_filter_physical(seg))):
            want = g[row][key]
            cols, tup = H.golden_group_key(seg, g[row]["group_by"], want["key"])
            res = execute(Q.QuerySpec(H.golden_aggregations(seg), filter=flt, group_by=cols))
            assert not check_kind or res.group_key_kind == kind, (row, res.group_key_kind)
            assert not res.num_groups_limit_reached
            groups = res.groups
            if base_of is not None:
                bases = [base_of(c) if seg.columns[c].dictionary is None else 0 for c in cols]
                groups = {tuple(int(d) + b for d, b in zip(t, bases)): v for t, v in res.groups.items()}
            H.check_golden_row(groups[tup], want)
            if exact_filter_stats:
                assert res.filter_entries_exact and list(res.stats) == want["stats"], (row, key, res.stats)
            else:
                assert (res.stats[0], res.stats[2], res.stats[3]) == (want["stats"][0], want["stats"][2], want["stats"][3])
            assert sum(v[0].count for v in res.groups.values()) == want["stats"][0]


def test_inner_segment_large_and_very_large_group_by_goldens_long_and_array_map_holders():
    seg = H.golden_segment()
    check_large_group_by_goldens(lambda spec: oracle.execute(seg, spec), seg)


def test_large_group_by_goldens_with_raw_key_columns():
    """The same four goldens with column1 / column3 / column9 stored WITHOUT a dictionary: the reference's results do not depend on the
    encoding, so they also pin NoDictionaryMultiColumnGroupKeyGenerator's restatement (keys by value, raw range leaves, raw SUM / MAX)."""
    seg = H.golden_segment(raw_columns=RAW_KEY_COLUMNS)
    d = H.load_golden_columns()
    check_large_group_by_goldens(lambda spec: oracle.execute(seg, spec), seg, base_of=lambda c: int(d[seg.columns[c].name].min()), check_kind=False)
    # and the aggregation-only / small / medium goldens over the raw columns
    g = H.load_golden_queries()
    for key, flt in (("unfiltered", None), ("filtered", H.golden_filter_physical(seg))):
        res = oracle.execute(seg, Q.QuerySpec(H.golden_aggregations(seg), filter=flt))
        H.check_golden_row(res.aggregations, g["inner_segment"][key])
        assert list(res.stats) == g["inner_segment"][key]["stats"]


def test_inter_segment_goldens_by_merging_four_copies():
    # InterSegmentAggregationSingleValueQueriesTest: 4 identical segments through combine + reduce;
    # merge rule = AggregationFunction.merge (SUM '+', COUNT '+'), AggregationResultsBlockMerger.java:34-44
    g = H.load_golden_queries()["inter_segment_x4"]
    seg = H.golden_segment()
    ci = seg.column_index
    aggs = [(Q.COUNT, -1), (Q.SUM, ci("column1")), (Q.SUM, ci("column3"))]
    for key, flt in (("unfiltered", None), ("filtered", H.golden_filter(seg))):
        r = oracle.execute(seg, Q.QuerySpec(aggs, filter=flt))
        count = sum(r.aggregations[0].intermediate(Q.COUNT) for _ in range(4))
        s1 = 0.0
        s3 = 0.0
        for _ in range(4):
            s1 = s1 + r.aggregations[1].intermediate(Q.SUM)
            s3 = s3 + r.aggregations[2].intermediate(Q.SUM)
        assert count == g["count"][key]
        assert s1 == g["sum_column1"][key] and s3 == g["sum_column3"][key]
    # GROUP BY column9 ORDER BY COUNT(*) DESC LIMIT 1 -> 64420 / 17080 (InterSegment...testCount :60-67)
    for want, flt in ((64420, None), (17080, H.golden_filter(seg))):
        r = oracle.execute(seg, Q.QuerySpec([(Q.COUNT, -1)], filter=flt, group_by=[ci("column9")]))
        assert 4 * max(v[0].count for v in r.groups.values()) == want


def test_oracle_matches_numpy_on_fixture_filters():
    d = H.load_golden_columns()
    seg = H.golden_segment()
    c1 = d["column1"].astype(np.int64)
    m = (d["column17"] == d["column17"][0]) & (d["column18"] > np.median(d["column18"]))
    flt = Q.and_(Q.leaf(H.eq_pred(seg, "column17", int(d["column17"][0]), inverted=True)),
                 Q.leaf(H.range_pred(seg, "column18", lower=int(np.median(d["column18"])), lower_inclusive=False)))
    r = oracle.execute(seg, Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, seg.column_index("column1"))], filter=flt))
    assert r.aggregations[0].count == int(m.sum())
    assert r.aggregations[1].sum_i64 == int(c1[m].sum())
    words, card = oracle.filter_bitmap(seg, Q.QuerySpec([], filter=flt))
    assert card == int(m.sum())
    bits = np.unpackbits(words.view(np.uint8), bitorder="little")[:30000].astype(bool)
    assert (bits == m).all()


def test_non_scan_based_aggregation_plan():
    """AggregationPlanNode.java:98-115: no filter + COUNT / dictionary-based MIN, MAX -> NonScanBasedAggregationOperator: the answer
    comes from the metadata and the dictionary ends, statistics (totalDocs, 0, 0, totalDocs); one SUM in the list and the scan is back."""
    seg = H.golden_segment(use_inverted=False)
    ci = seg.column_index
    r = oracle.execute(seg, Q.QuerySpec([(Q.COUNT, -1), (Q.MAX, ci("column3")), (Q.MIN, ci("column6"))]))
    assert r.intermediates() == [30000, 2147419555.0, 1689277.0]
    assert r.stats == (30000, 0, 0, 30000)
    r = oracle.execute(seg, Q.QuerySpec([(Q.COUNT, -1), (Q.MAX, ci("column3")), (Q.SUM, ci("column1"))]))
    assert r.stats == (30000, 0, 60000, 30000)


def _merged_stats(r):
    """Four identical segments: every ExecutionStatistics field is summed by the combine operator."""
    return [4 * r.stats[0], 4 * r.stats[1], 4 * r.stats[2], 4 * r.stats[3]]


def check_inter_segment_max_min_avg(execute, seg):
    """InterSegmentAggregationSingleValueQueriesTest.testMax / testMin / testSum / testAvg (:91-203) with `execute(spec) -> Result`
    for one segment; the x4 merge (MAX max, MIN min, SUM +, AVG pairwise) and ORDER BY ... LIMIT 1 are done here."""
    g = H.load_golden_queries()["inter_segment_x4"]
    ci = seg.column_index
    c1, c3, c9 = ci("column1"), ci("column3"), ci("column9")
    flt = H.golden_filter(seg)
    for func, key, pick in ((Q.MAX, "max_column1_column3", "max"), (Q.MIN, "min_column1_column3", "min")):
        gg = g[key]
        for name, f in (("unfiltered", None), ("filtered", flt)):
            r = execute(Q.QuerySpec([(func, c1), (func, c3)], filter=f))
            assert [getattr(r.aggregations[0], pick), getattr(r.aggregations[1], pick)] == gg[name]["values"]
            st, want = _merged_stats(r), gg[name]["stats"]
            # numEntriesScannedInFilter (index 1) is the documented deviation; the unfiltered case must show the non-scan plan: 0 entries
            assert (st[0], st[2], st[3]) == (want[0], want[2], want[3]), (name, st, want)
            if f is None:
                assert st[1] == 0
        desc = func == Q.MAX
        for name, f in (("group_by_top_desc" if desc else "group_by_top_asc", None), ("filtered_group_by_top_desc" if desc else "filtered_group_by_top_asc", flt)):
            r = execute(Q.QuerySpec([(func, c1), (func, c3)], filter=f, group_by=[c9]))
            rows = sorted(([getattr(v[0], pick), getattr(v[1], pick)] for v in r.groups.values()), reverse=desc)
            assert rows[0] == gg[name]["values"]
            st, want = _merged_stats(r), gg[name]["stats"]
            assert (st[0], st[2], st[3]) == (want[0], want[2], want[3])
    for name, f in (("unfiltered", None), ("filtered", flt)):
        r = execute(Q.QuerySpec([(Q.SUM, c1), (Q.SUM, c3)], filter=f, group_by=[c9]))
        rows = sorted(([4 * v[0].sum, 4 * v[1].sum] for v in r.groups.values()), reverse=True)
        assert rows[0] == g["sum_group_by_top_desc"][name]
        ga = g["avg_column1_column3"][name]
        r = execute(Q.QuerySpec([(Q.AVG, c1), (Q.AVG, c3)], filter=f))
        for i in range(2):
            avg = (4 * r.aggregations[i].sum) / (4 * r.aggregations[i].count)
            assert abs(avg - ga["values"][i]) <= ga["tolerance"] * max(1.0, abs(ga["values"][i])) or abs(avg - ga["values"][i]) < 1e-3
        st = _merged_stats(r)
        assert (st[0], st[2], st[3]) == (ga["stats"][0], ga["stats"][2], ga["stats"][3])
    r = execute(Q.QuerySpec([(Q.AVG, c1), (Q.AVG, c3)], group_by=[c9]))
    rows = sorted(([v[0].sum / v[0].count, v[1].sum / v[1].count] for v in r.groups.values()), reverse=True)
    assert rows[0] == g["avg_column1_column3"]["group_by_top_desc"]["values"]


def test_inter_segment_max_min_avg_goldens():
    seg = H.golden_segment()
    check_inter_segment_max_min_avg(lambda spec: oracle.execute(seg, spec), seg)


def check_string_key_group_by(execute, seg):
    """InterSegmentGroupBySingleValueQueriesTest.java:66-107: GROUP BY on STRING dictionary columns (one and two keys); group ids are
    turned back into dictionary values (mixed radix, first key fastest) and the four identical segments are merged by '+'."""
    g = H.load_golden_queries()["inter_segment_group_by_x4"]
    ci = seg.column_index
    c1, c11, c12 = ci("column1"), ci("column11"), ci("column12")
    d11, d12 = seg.string_dicts["column11"], seg.string_dicts["column12"]
    r = execute(Q.QuerySpec([(Q.SUM, c1)], group_by=[c11]))
    rows = sorted([d11[gid], 4 * v[0].sum] for gid, v in r.groups.items())
    assert rows == g["sum_column1_by_column11"]
    assert 4 * r.stats[2] == g["numEntriesScannedPostFilter"][0]
    r = execute(Q.QuerySpec([(Q.SUM, c1)], group_by=[c11, c12]))
    card11 = len(d11)
    rows = sorted([d11[gid % card11], d12[gid // card11], 4 * v[0].sum] for gid, v in r.groups.items())
    assert rows[:15] == g["sum_column1_by_column11_column12_first15"]
    assert 4 * r.stats[2] == g["numEntriesScannedPostFilter"][1]


def test_string_key_group_by_goldens():
    seg = H.golden_segment()
    check_string_key_group_by(lambda spec: oracle.execute(seg, spec), seg)
