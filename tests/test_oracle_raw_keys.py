"""CPU test: the oracle's restatement of GROUP BY over raw (no-dictionary) INT / LONG key columns -- NoDictionarySingleColumnGroupKeyGenerator /
NoDictionaryMultiColumnGroupKeyGenerator: keys by value, ids by first appearance up to numGroupsLimit -- against a per-doc numpy / dict
restatement of the same rules keyed by the true values."""
import numpy as np
import pytest

from oracle import oracle
from pinot_amd import _abi
from pinot_amd import query as Q
from pinot_amd import segment as S
import raw_key_cases as RC


def _check_against_numpy(seg, key_values, spec, got, base_of):
    mask = None
    if spec.filter is not None:
        words, _ = oracle.filter_bitmap(seg, Q.QuerySpec([], filter=spec.filter))
        mask = np.unpackbits(words.view(np.uint8), bitorder="little")[: seg.num_docs].astype(bool)
    want, scanned = RC.numpy_groups(key_values, spec, mask, seg.num_docs)
    assert got.stats[0] == scanned
    rows = RC.key_tuples(got, seg, spec, base_of)
    assert sorted(rows) == sorted(want)
    limit = spec.num_groups_limit if spec.num_groups_limit > 0 else 100000
    assert got.num_groups_limit_reached == (len(want) >= limit)
    assert got.group_id_upper_bound == limit                   # NoDictionarySingleColumnGroupKeyGenerator.java:73-79
    for key, docs in want.items():
        docs = np.asarray(docs)
        for a, (fn, col) in enumerate(spec.aggregations):
            v = rows[key][a]
            if fn == Q.COUNT:
                assert v.count == len(docs)
                continue
            vals = oracle.read_int_values(seg, col, docs.astype(np.int32)).astype(np.int64)
            if fn in (Q.SUM, Q.AVG):
                assert v.sum_i64 == int(vals.sum()) and v.sum == float(vals.sum())
            if fn == Q.AVG:
                assert v.count == len(docs)
            if fn == Q.MIN:
                assert v.min == float(vals.min())
            if fn == Q.MAX:
                assert v.max == float(vals.max())


@pytest.mark.parametrize("case", RC.cases(), ids=[c[0] for c in RC.cases()])
def test_oracle_no_dictionary_group_key_generators(case):
    seg, key_values, specs = RC.build(case)
    base_of = lambda c: int(key_values[c].min())
    for spec in specs:
        got = oracle.execute(seg, spec)
        assert got.group_key_kind == case[3]
        _check_against_numpy(seg, key_values, spec, got, base_of)


def test_oracle_keys_columns_outside_the_key_image_by_rank():
    """Round 5: a raw INT column whose range is beyond an int and a raw DOUBLE column are no longer declined -- they are keyed by value
    on the rank scale (tests/test_oracle_rank_keys.py holds that restatement against numpy); under null handling they still are."""
    n = 1000
    rng = np.random.default_rng(3)
    wide = np.array([-(2 ** 31), 2 ** 31 - 1] + list(rng.integers(-1000, 1000, n - 2)), dtype=np.int32)
    v = S.Column.synthetic_uniform("v", n, np.arange(50, dtype=np.int32), seed=1)
    seg = S.SegmentData("wide", n, [S.Column.raw("k", wide), S.Column.raw_typed("d", np.round(rng.random(n), 2)), v])
    for col, values in ((0, wide.astype(np.int64)), (1, None)):
        got = oracle.execute(seg, Q.QuerySpec([(Q.COUNT, -1)], group_by=[col]))
        distinct = len(np.unique(values)) if values is not None else len(got.groups)
        assert len(got.groups) == distinct and sum(vals[0].count for vals in got.groups.values()) == n
        assert sorted(t[0] for t in got.group_keys) == list(range(distinct))              # every rank occurs once
        with pytest.raises(Exception):
            oracle.execute(seg, Q.QuerySpec([(Q.COUNT, -1)], group_by=[col], null_handling=True))
