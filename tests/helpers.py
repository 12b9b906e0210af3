"""Shared test helpers: golden fixture segment, query lowering through the oracle's dictionary search, comparisons."""
import json
import math
import os

import numpy as np

from oracle import oracle
from pinot_amd import query as Q
from pinot_amd import segment as S

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# BaseSingleValueQueriesTest.java:84 setInvertedIndexColumns
GOLDEN_INVERTED = {"column6", "column7", "column11", "column17", "column18"}
GOLDEN_INT = ["column1", "column3", "column6", "column7", "column9", "column17", "column18", "daysSinceEpoch"]
GOLDEN_STR = ["column5", "column11", "column12"]


def load_golden_columns():
    return np.load(os.path.join(GOLDEN_DIR, "test_data_sv.npz"))


def load_golden_queries():
    with open(os.path.join(GOLDEN_DIR, "golden_queries.json")) as f:
        return json.load(f)


def golden_segment(use_inverted=True, raw_columns=()):
    """The segment BaseSingleValueQueriesTest builds from test_data-sv.avro (every column dictionary encoded).
    String columns carry their dictIds with a placeholder int dictionary: string values never reach the device.
    `raw_columns`: INT columns stored WITHOUT a dictionary (raw PASS_THROUGH forward index) instead -- query results do not depend on
    the encoding, so the reference's goldens also pin the no-dictionary readers and key generators."""
    d = load_golden_columns()
    cols = []
    for name in GOLDEN_INT:
        if name in raw_columns:
            cols.append(S.Column.raw(name, d[name]))
            continue
        cols.append(S.Column.dict_encoded(name, d[name], with_inverted=use_inverted and name in GOLDEN_INVERTED))
    for name in GOLDEN_STR:
        ids = d[name + "__ids"]
        card = int(d[name + "__dict"].shape[0])
        cols.append(S.Column.from_dict_ids(name, np.arange(card, dtype=np.int32), ids,
                                           with_inverted=use_inverted and name in GOLDEN_INVERTED))
    seg = S.SegmentData("testTable_126164076_167572854", 30000, cols)
    seg.string_dicts = {name: [str(x) for x in d[name + "__dict"]] for name in GOLDEN_STR}
    return seg


def range_pred(seg, name, lower=None, upper=None, lower_inclusive=True, upper_inclusive=True, inverted=False):
    """RANGE predicate on a dictionary column lowered like SortedDictionaryBasedRangePredicateEvaluator
    (RangePredicateEvaluatorFactory.java:126-169), with the alwaysTrue / alwaysFalse short-circuits (:163-168)."""
    ci = seg.column_index(name)
    col = seg.columns[ci]
    if col.dictionary is None:
        # raw INT column: IntRawValueBasedRangePredicateEvaluator (RangePredicateEvaluatorFactory.java:331-366), inclusive bounds
        lo = -2 ** 31 if lower is None else (lower if lower_inclusive else lower + 1)
        hi = 2 ** 31 - 1 if upper is None else (upper if upper_inclusive else upper - 1)
        return Q.Pred.match_none() if lo > hi else Q.Pred.raw_range(ci, lo, hi)
    s, e = oracle.lower_range(col.dictionary, col.cardinality, lower, upper, lower_inclusive, upper_inclusive)
    n = max(e - s, 0)
    if n == 0:
        return Q.Pred.match_none()
    if n == col.cardinality:
        return Q.Pred.match_all()
    return Q.Pred.dict_range(ci, s, e, inverted=inverted)


def eq_pred(seg, name, value, exclusive=False, inverted=False):
    """EQ / NOT_EQ (EqualsPredicateEvaluatorFactory.java:92-124)."""
    ci = seg.column_index(name)
    col = seg.columns[ci]
    d = oracle.index_of(col.dictionary, col.cardinality, value)
    if d < 0:
        return Q.Pred.match_all() if exclusive else Q.Pred.match_none()
    if col.cardinality == 1:
        return Q.Pred.match_none() if exclusive else Q.Pred.match_all()
    return Q.Pred.dict_range(ci, d, d + 1, exclusive=exclusive, inverted=inverted)


def in_pred(seg, name, values, exclusive=False, inverted=False):
    """IN / NOT_IN (InPredicateEvaluatorFactory.java:161-188)."""
    ci = seg.column_index(name)
    col = seg.columns[ci]
    ids = sorted({d for d in (oracle.index_of(col.dictionary, col.cardinality, v) for v in values) if d >= 0})
    if not ids:
        return Q.Pred.match_all() if exclusive else Q.Pred.match_none()
    if len(ids) == col.cardinality:
        return Q.Pred.match_none() if exclusive else Q.Pred.match_all()
    return Q.Pred.dict_set(ci, ids, col.cardinality, exclusive=exclusive, inverted=inverted)


def string_in_pred(seg, name, strings, exclusive=False, inverted=False):
    ci = seg.column_index(name)
    col = seg.columns[ci]
    values = seg.string_dicts[name]
    ids = sorted(values.index(s) for s in strings if s in values)
    if not ids:
        return Q.Pred.match_all() if exclusive else Q.Pred.match_none()
    if len(ids) == col.cardinality:
        return Q.Pred.match_none() if exclusive else Q.Pred.match_all()
    return Q.Pred.dict_set(ci, ids, col.cardinality, exclusive=exclusive, inverted=inverted)


def golden_filter(seg, inverted=False):
    """BaseSingleValueQueriesTest.FILTER (:101-106).  With inverted=True the leaves the reference would serve from
    an inverted index (non-range predicates on column6/7/11/17/18, FilterOperatorUtils.java:123-126) use postings."""
    c5_values = seg.string_dicts["column5"]
    c5 = Q.Pred.match_all() if c5_values == ["gFuH"] else string_in_pred(seg, "column5", ["gFuH"])
    return Q.and_(
        Q.leaf(range_pred(seg, "column1", lower=100000000, lower_inclusive=False)),
        Q.leaf(range_pred(seg, "column3", lower=20000000, upper=1000000000)),
        Q.leaf(c5),
        Q.or_(Q.leaf(range_pred(seg, "column6", upper=500000000, upper_inclusive=False)),
              Q.leaf(string_in_pred(seg, "column11", ["t", "P"], exclusive=True, inverted=inverted))),
        Q.leaf(eq_pred(seg, "daysSinceEpoch", 126164076)))


def and_leapfrog_entries(matches):
    """numEntriesScannedInFilter of an AND whose children are all scan leaves (no index-based child: AndDocIdIterator.java:41-74 leap-frogs
    SVScanDocIdIterator.advance, one entry per doc a leaf looks at), written as the state machine the loop amounts to: exactly one leaf
    is scanning at any doc; when it hits a match the others are asked about that doc in order until one of them says no (that one scans
    on), and when all say yes the doc is a result and the first leaf scans on from the next doc."""
    n, k = len(matches[0]), len(matches)
    entries, scanning = 0, 0
    for d in range(n):
        entries += 1
        if not matches[scanning][d]:
            continue
        nxt = 0
        for i in range(k):
            if i == scanning:
                continue
            entries += 1
            if not matches[i][d]:
                nxt = i
                break
        scanning = nxt
    return entries


def golden_filter_physical(seg, inverted=True):
    """The operator tree the reference builds for BaseSingleValueQueriesTest.FILTER (FilterPlanNode + FilterOperatorUtils):
    column5 = 'gFuH' is always true on a one-value dictionary and drops out (getLeafFilterOperator :72-88, getAndFilterOperator :140-144);
    daysSinceEpoch is sorted -> SortedIndexBasedFilterOperator, i.e. a docId range (priority 0); the OR (priority 400) comes before the
    scan leaves (500), which keep their order in the query (reorderAndFilterChildOperators :196-245)."""
    days = load_golden_columns()["daysSinceEpoch"]
    hit = np.flatnonzero(days == 126164076)
    assert hit[-1] - hit[0] + 1 == hit.shape[0]            # the column is sorted: one docId range
    return Q.and_(
        Q.leaf(Q.Pred.doc_range(int(hit[0]), int(hit[-1]))),
        Q.or_(Q.leaf(range_pred(seg, "column6", upper=500000000, upper_inclusive=False)),
              Q.leaf(string_in_pred(seg, "column11", ["t", "P"], exclusive=True, inverted=inverted))),
        Q.leaf(range_pred(seg, "column1", lower=100000000, lower_inclusive=False)),
        Q.leaf(range_pred(seg, "column3", lower=20000000, upper=1000000000)))


def golden_medium_group(seg, want):
    """(group-by column indexes, raw key) of a testMediumAggregationGroupBy golden row: raw key = sum dictId_j * prod_{k<j} cardinality_k
    (DictionaryBasedGroupKeyGenerator.java:437-445)."""
    cols, raw, mult = [], 0, 1
    for name, value in zip(("column9", "column11", "column12"), want["key"]):
        ci = seg.column_index(name)
        c = seg.columns[ci]
        d = seg.string_dicts[name].index(value) if name in seg.string_dicts else int(np.searchsorted(c.dict_values, value))
        raw += d * mult
        mult *= c.cardinality
        cols.append(ci)
    return cols, raw


def golden_group_key(seg, names, key, base_of=None):
    """(group-by column indexes, key tuple as the result rows carry it) of a golden row with any number of group-by columns: the dictId
    of every dictionary column (strings through the fixture's string dictionaries); for a raw (no-dictionary) column the VALUE, which
    NoDictionary*GroupKeyGenerator keys by -- callers turn the result's digits into values with pg_group_key_info's base."""
    cols, tup = [], []
    for name, value in zip(names, key):
        ci = seg.column_index(name)
        c = seg.columns[ci]
        cols.append(ci)
        if name in getattr(seg, "string_dicts", {}):
            tup.append(seg.string_dicts[name].index(value))
        elif c.dictionary is None:
            tup.append(int(value))
        else:
            d = int(np.searchsorted(c.dict_values, value))
            assert c.value_of(d) == value
            tup.append(d)
    return cols, tuple(tup)


def check_golden_row(vals, want):
    """One group (or the aggregation-only row) of SELECT COUNT(*), SUM(column1), MAX(column3), MIN(column6), AVG(column7) against the
    literals of QueriesTestUtils.testInnerSegmentAggregation[GroupBy]Result."""
    count, s1, mx3, mn6, avg7 = vals
    assert count.intermediate(Q.COUNT) == want["count"]
    assert s1.intermediate(Q.SUM) == float(want["sum_column1"]) and s1.sum_i64 == want["sum_column1"]
    assert mx3.intermediate(Q.MAX) == float(want["max_column3"])
    assert mn6.intermediate(Q.MIN) == float(want["min_column6"])
    assert avg7.intermediate(Q.AVG) == (float(want["avg_column7"][0]), want["avg_column7"][1])


def golden_aggregations(seg):
    """SELECT COUNT(*), SUM(column1), MAX(column3), MIN(column6), AVG(column7)"""
    ci = seg.column_index
    return [(Q.COUNT, -1), (Q.SUM, ci("column1")), (Q.MAX, ci("column3")), (Q.MIN, ci("column6")), (Q.AVG, ci("column7"))]


# SUM / AVG over FLOAT / DOUBLE values: |device - reference| <= 1e-11 * |reference| + count * 1e-15 * (largest |value|, 1e6 unless the
# test says otherwise): the additions run in tile / atomic order instead of doc order, each one off by at most an ulp of the running
# sum, and sums that cancel have no meaningful relative error.  (north_star allows 1e-6 relative.)  Everything else is bit exact.
FP_SUM_RTOL = 1e-11
FP_SUM_ATOL = 1e-9
FP_VALUE_SCALE = 1e6


def assert_agg_equal(a, b, function, where=""):
    """Bit-exact comparison of two AggValue for the fields `function` defines."""
    assert a.count == b.count, "%s count %d != %d" % (where, a.count, b.count)
    if function in (Q.SUM, Q.AVG) and not b.sum_exact:
        # FLOAT / DOUBLE columns: the reference adds doubles in doc order, the device in tile order; tolerance 1e-11 relative
        assert not a.sum_exact, "%s: a floating-point sum must not be flagged exact" % where
        assert (math.isnan(a.sum) and math.isnan(b.sum)) or \
            math.isclose(a.sum, b.sum, rel_tol=FP_SUM_RTOL, abs_tol=max(FP_SUM_ATOL, max(b.count, 1) * FP_VALUE_SCALE * 1e-15)), \
            "%s double sum %r != %r" % (where, a.sum, b.sum)
    elif function in (Q.SUM, Q.AVG):
        assert a.sum_i64 == b.sum_i64, "%s exact sum %d != %d" % (where, a.sum_i64, b.sum_i64)
        assert a.sum == b.sum or abs(a.sum - b.sum) <= 1e-6 * abs(b.sum), "%s sum %r != %r" % (where, a.sum, b.sum)
        if abs(b.sum_i64) < 2 ** 53:
            assert a.sum == b.sum, "%s double sum %r != %r (below 2^53 must be bit exact)" % (where, a.sum, b.sum)
    if function == Q.MIN:
        assert a.min == b.min or (math.isnan(a.min) and math.isnan(b.min)), "%s min %r != %r" % (where, a.min, b.min)
    if function == Q.MAX:
        assert a.max == b.max or (math.isnan(a.max) and math.isnan(b.max)), "%s max %r != %r" % (where, a.max, b.max)


def assert_results_equal(got, want, check_stats=True):
    assert len(got.aggregations) == len(want.aggregations)
    for i, f in enumerate(want.functions):
        if want.aggregations:
            assert_agg_equal(got.aggregations[i], want.aggregations[i], f, "agg %d" % i)
    assert sorted(got.groups) == sorted(want.groups), "group ids differ"
    for gid, vals in want.groups.items():
        for i, f in enumerate(want.functions):
            assert_agg_equal(got.groups[gid][i], vals[i], f, "group %r agg %d" % (gid, i))
    if check_stats:
        assert got.stats[0] == want.stats[0], "numDocsScanned %r != %r" % (got.stats, want.stats)
        # numEntriesScannedInFilter: the reference's iterator accounting on both sides, unless one of them declares an upper bound
        # (enableNullHandling; leap-frogging filters on segments above PINOT_GPU_EXACT_FILTER_STATS_DOCS)
        if getattr(got, "filter_entries_exact", False) and getattr(want, "filter_entries_exact", False):
            assert got.stats[1] == want.stats[1], "numEntriesScannedInFilter %r != %r" % (got.stats, want.stats)
        assert got.stats[2] == want.stats[2], "numEntriesScannedPostFilter %r != %r" % (got.stats, want.stats)
        assert got.stats[3] == want.stats[3]


def random_dict_column(rng, name, num_docs, cardinality, value_stride=7, with_inverted=False, run_optimize=True, sorted_runs=False):
    """Column whose dictionary really has `cardinality` entries (ids cover 0..cardinality-1)."""
    dict_values = (np.arange(cardinality, dtype=np.int64) * value_stride + 3 - (cardinality // 2) * (value_stride // 2)).astype(np.int32)
    ids = rng.integers(0, cardinality, num_docs).astype(np.int32)
    if sorted_runs:
        ids = np.sort(ids)
    if num_docs >= cardinality:
        ids[rng.permutation(num_docs)[:cardinality]] = np.arange(cardinality, dtype=np.int32)  # every dictId present
    return S.Column.from_dict_ids(name, dict_values, ids, with_inverted=with_inverted, run_optimize=run_optimize), ids, dict_values
