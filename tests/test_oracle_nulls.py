"""CPU tests of the oracle's null-handling restatement against the reference's own known-answer tests
(tests/golden/null_handling_kats.json: {Sum,Min,Max,Avg}AggregationFunctionTest and NullHandlingEnabledQueriesTest literal tables)
and against a per-doc numpy restatement of getTrues / getNulls / getFalses on random trees."""
import math

import numpy as np
import pytest

from oracle import oracle
from pinot_amd import query as Q
from pinot_amd import segment as S
import helpers as H
import null_cases as NC


def run_aggregation_kat(execute, case, data_type, raw):
    fn = NC.FUNCTIONS[case["function"]]
    partials = []
    for rows in case["segments"]:
        col = NC.nullable_column("myField", rows, data_type, case["field_type"], raw=raw)
        seg = S.SegmentData("testTable", len(rows), [col])
        res = execute(seg, Q.QuerySpec([(fn, 0)], null_handling=case["null_handling"]))
        partials.append(res.aggregations[0])
    got = NC.reduce_partials(fn, partials, case["null_handling"])
    want = case["expected"]
    if want == "DEFAULT":
        want = float(NC.DEFAULT_NULL[(case["field_type"], data_type)])
    assert got == want, (case["ref"], data_type, raw, got, want)


@pytest.mark.parametrize("data_type", ["INT", "LONG", "FLOAT", "DOUBLE"])
@pytest.mark.parametrize("raw", [False, True])
def test_aggregation_kats_of_the_reference(data_type, raw):
    for case in NC.load_kats()["aggregation"]:
        run_aggregation_kat(oracle.execute, case, data_type, raw)


def kat_filter_segment(case, raw=False):
    rows = case["rows"]
    names = ["c1", "c2"][:len(rows[0])]
    cols = [NC.nullable_column(n, [r[i] for r in rows], "INT", "DIMENSION", raw=raw) for i, n in enumerate(names)]
    return S.SegmentData("testTable", len(rows), cols)


@pytest.mark.parametrize("raw", [False, True])
def test_filter_kats_of_the_reference(raw):
    for case in NC.load_kats()["filter"]:
        seg = kat_filter_segment(case, raw)
        spec = Q.QuerySpec([(Q.COUNT, -1)], filter=NC.tree_for(seg, case["filter"]), null_handling=True)
        res = oracle.execute(seg, spec)
        assert res.aggregations[0].count == case["expected_count"], case["ref"]
        words, card = oracle.filter_bitmap(seg, spec)
        assert card == case["expected_count"]
        if "expected_rows" in case:
            assert [d for d in range(seg.num_docs) if (int(words[d >> 6]) >> (d & 63)) & 1] == case["expected_rows"], case["ref"]


def random_nullable_segment(rng, num_docs, raw_second=False):
    """Three INT dimension columns: c1 with many nulls, c2 with few, c3 with none."""
    cols, values, nulls = [], {}, {}
    for i, (name, null_rate) in enumerate((("c1", 0.3), ("c2", 0.02), ("c3", 0.0))):
        v = rng.integers(-20, 21, num_docs).astype(np.int32)
        m = rng.random(num_docs) < null_rate
        v[m] = -2 ** 31
        col = S.Column.raw_typed(name, v) if (raw_second and i == 1) else S.Column.dict_encoded_typed(name, v, with_inverted=(i == 0))
        cols.append(col.with_nulls(m))
        values[name], nulls[name] = v, m
    return S.SegmentData("nullable", num_docs, cols), values, nulls


def random_tree(rng, depth=0):
    r = rng.random()
    if depth >= 3 or r < 0.35:
        name = ["c1", "c2", "c3"][int(rng.integers(0, 3))]
        op = ["LT", "LE", "GT", "GE", "EQ", "IS_NULL", "IS_NOT_NULL"][int(rng.integers(0, 7))]
        if op.startswith("IS_"):
            return [op, name]
        return [op, name, int(rng.integers(-25, 26))]
    if r < 0.55:
        return ["NOT", random_tree(rng, depth + 1)]
    return [["AND", "OR"][int(rng.integers(0, 2))]] + [random_tree(rng, depth + 1) for _ in range(int(rng.integers(2, 4)))]


@pytest.mark.parametrize("raw_second", [False, True])
def test_random_filter_trees_against_the_per_doc_rules(raw_second):
    rng = np.random.default_rng(20240917)
    seg, values, nulls = random_nullable_segment(rng, 5000, raw_second)
    raw_columns = ("c2",) if raw_second else ()
    for _ in range(60):
        tree = random_tree(rng)
        want = NC.reference_trues(tree, values, nulls, seg.num_docs, raw_columns)
        spec = Q.QuerySpec([(Q.COUNT, -1)], filter=NC.tree_for(seg, tree), null_handling=True)
        words, card = oracle.filter_bitmap(seg, spec)
        got = np.unpackbits(words.view(np.uint8), bitorder="little")[:seg.num_docs].astype(bool)
        assert card == int(want.sum()) and np.array_equal(got, want), tree


def test_aggregations_skip_nulls_per_column_and_count_column():
    rng = np.random.default_rng(7)
    seg, values, nulls = random_nullable_segment(rng, 30000)
    tree = ["OR", ["GT", "c3", 0], ["NOT", ["LT", "c2", 5]]]
    flt = NC.reference_trues(tree, values, nulls, seg.num_docs)
    aggs = [(Q.COUNT, -1), (Q.COUNT, 0), (Q.SUM, 0), (Q.MIN, 1), (Q.MAX, 0), (Q.AVG, 1), (Q.SUM, 2)]
    res = oracle.execute(seg, Q.QuerySpec(aggs, filter=NC.tree_for(seg, tree), null_handling=True))
    m1, m2 = flt & ~nulls["c1"], flt & ~nulls["c2"]
    a = res.aggregations
    assert a[0].count == flt.sum() and a[1].count == m1.sum()
    assert a[2].sum_i64 == int(values["c1"][m1].astype(np.int64).sum()) and a[2].count == m1.sum()
    assert a[3].min == float(values["c2"][m2].min()) and a[4].max == float(values["c1"][m1].max())
    assert a[5].count == m2.sum() and a[5].sum_i64 == int(values["c2"][m2].astype(np.int64).sum())
    assert a[6].sum_i64 == int(values["c3"][flt].astype(np.int64).sum())
    assert res.stats[0] == flt.sum() and res.stats[2] == 3 * flt.sum() and res.stats[3] == seg.num_docs
    # without the option the stored default null values are aggregated like any other value
    plain = oracle.execute(seg, Q.QuerySpec([(Q.MIN, 0), (Q.COUNT, 0)]))
    assert plain.aggregations[0].min == float(-2 ** 31) and plain.aggregations[1].count == seg.num_docs
    ok = oracle.execute(seg, Q.QuerySpec([(Q.SUM, 2)], group_by=[2], null_handling=True))
    assert sum(v[0].count for v in ok.groups.values()) == seg.num_docs


def brute_force_groups(seg, values, nulls, keys, aggs, flt):
    """GROUP BY under enableNullHandling, doc by doc: NULL is a key value of its own (digit = cardinality on the ABI's raw-key scale),
    every function skips the null docs of its own column (DefaultGroupByExecutor.java:106-121, NullableSingleInputAggregationFunction)."""
    names = [c.name for c in seg.columns]
    digits, radix = [], []
    for k in keys:
        col = seg.columns[k]
        has_nulls = bool(nulls[names[k]].any())
        d = np.searchsorted(col.dict_values, values[names[k]]).astype(np.int64)
        if has_nulls:
            d[nulls[names[k]]] = col.cardinality
        digits.append(d)
        radix.append(col.cardinality + (1 if has_nulls else 0))
    raw, mult = np.zeros(seg.num_docs, dtype=np.int64), 1
    for d, r in zip(digits, radix):
        raw += d * mult
        mult *= r
    out = {}
    for gid in np.unique(raw[flt]):
        m = flt & (raw == gid)
        row = []
        for f, c in aggs:
            if f == Q.COUNT and c < 0:
                row.append((int(m.sum()), None))
                continue
            mm = m & ~nulls[names[c]]
            v = values[names[c]][mm].astype(np.int64)
            row.append((int(mm.sum()), {Q.COUNT: None, Q.SUM: int(v.sum()), Q.AVG: int(v.sum()), Q.MIN: float(v.min()) if len(v) else np.inf,
                                        Q.MAX: float(v.max()) if len(v) else -np.inf}[f]))
        out[int(gid)] = row
    return out


def check_groups(res, want, aggs):
    assert sorted(res.groups) == sorted(want)
    for gid, row in want.items():
        for (f, _), (count, value), got in zip(aggs, row, res.groups[gid]):
            assert got.count == count, (gid, f, got.count, count)
            if f in (Q.SUM, Q.AVG):
                assert got.sum_i64 == value
            if f == Q.MIN:
                assert got.min == value
            if f == Q.MAX:
                assert got.max == value


def test_group_by_under_null_handling_against_a_per_doc_restatement():
    rng = np.random.default_rng(11)
    seg, values, nulls = random_nullable_segment(rng, 20000)
    tree = ["OR", ["GT", "c3", 0], ["NOT", ["LT", "c2", 5]]]
    for flt_tree in (None, tree):
        flt = NC.reference_trues(flt_tree, values, nulls, seg.num_docs) if flt_tree else np.ones(seg.num_docs, bool)
        for keys in ([0], [2], [1, 0], [0, 2, 1]):
            aggs = [(Q.COUNT, -1), (Q.SUM, 0), (Q.COUNT, 1), (Q.MIN, 1), (Q.MAX, 0), (Q.AVG, 2), (Q.SUM, 2)]
            spec = Q.QuerySpec(aggs, filter=NC.tree_for(seg, flt_tree) if flt_tree else None, group_by=keys, null_handling=True)
            res = oracle.execute(seg, spec)
            check_groups(res, brute_force_groups(seg, values, nulls, keys, aggs, flt), aggs)
            assert res.stats[0] == int(flt.sum())
    # numGroupsLimit binds at any key-space size (the no-dictionary generators): the first `limit` keys in docId order survive
    spec = Q.QuerySpec([(Q.COUNT, -1)], group_by=[0], null_handling=True, num_groups_limit=5)
    res = oracle.execute(seg, spec)
    d = np.searchsorted(seg.columns[0].dict_values, values["c1"]).astype(np.int64)
    d[nulls["c1"]] = seg.columns[0].cardinality
    first = []
    for x in d:
        if x not in first:
            first.append(int(x))
        if len(first) == 5:
            break
    assert sorted(res.groups) == sorted(first) and res.num_groups_limit_reached
