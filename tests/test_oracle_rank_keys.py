"""CPU test: the oracle's restatement of GROUP BY over raw FLOAT / DOUBLE columns and raw INT / LONG columns spanning more than an int
(NoDictionarySingleColumnGroupKeyGenerator.java:100-135, NoDictionaryMultiColumnGroupKeyGenerator: keys by value, ids by first appearance up
to numGroupsLimit) against a per-doc numpy / dict restatement keyed by the values' identities (Double.doubleToLongBits / the long)."""
import numpy as np
import pytest

from oracle import oracle
from pinot_amd import query as Q
import rank_key_cases as KC


def check_against_numpy(seg, identities, spec, got, values_of, base_of):
    mask = None
    if spec.filter is not None:
        words, _ = oracle.filter_bitmap(seg, Q.QuerySpec([], filter=spec.filter))
        mask = np.unpackbits(words.view(np.uint8), bitorder="little")[: seg.num_docs].astype(bool)
    want, scanned = KC.numpy_groups(identities, spec, mask, seg.num_docs)
    assert got.stats[0] == scanned
    rows = KC.key_tuples(got, seg, spec, values_of, base_of)
    assert sorted(rows) == sorted(want)
    limit = spec.num_groups_limit if spec.num_groups_limit > 0 else 100000
    assert got.num_groups_limit_reached == (len(want) >= limit)
    for key, docs in want.items():
        docs = np.asarray(docs)
        for a, (fn, col) in enumerate(spec.aggregations):
            v = rows[key][a]
            if fn == Q.COUNT:
                assert v.count == len(docs)
                continue
            vals = oracle.read_int_values(seg, col, docs.astype(np.int32)).astype(np.int64)
            if fn in (Q.SUM, Q.AVG):
                assert v.sum_i64 == int(vals.sum())
            if fn == Q.MIN:
                assert v.min == float(vals.min())
            if fn == Q.MAX:
                assert v.max == float(vals.max())


def oracle_base(seg, c):
    docs = np.arange(seg.num_docs, dtype=np.int32)
    return int(oracle.read_int_values(seg, c, docs).min())


@pytest.mark.parametrize("case", KC.cases(), ids=lambda c: c[0])
def test_oracle_groups_raw_float_double_and_wide_columns_by_value(case):
    seg, identities, specs = KC.build(case)
    for spec in specs:
        got = oracle.execute(seg, spec)
        check_against_numpy(seg, identities, spec, got, lambda c: KC.rank_values(seg, c), lambda c: oracle_base(seg, c))


def test_the_order_of_the_rank_scale():
    """-inf < -max < -1 < -denormal < -0.0 < 0.0 < denormal < 1 < max < +inf < NaN: Double.compare's order, one NaN."""
    seg, identities, specs = KC.build(("order", 5_003, [("double", 11)]))
    ranks = KC.rank_values(seg, 0).view(np.float64)
    assert np.isnan(ranks[-1]) and np.isinf(ranks[0]) and ranks[0] < 0
    finite = ranks[:-1]
    assert all(a <= b for a, b in zip(finite, finite[1:]))
    zeros = [i for i, x in enumerate(ranks) if x == 0.0]
    assert len(zeros) == 2 and np.signbit(ranks[zeros[0]]) and not np.signbit(ranks[zeros[1]])
