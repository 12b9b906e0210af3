"""GPU tests of GROUP BY over raw (no-dictionary) INT / LONG columns -- the reference's NoDictionarySingleColumnGroupKeyGenerator /
NoDictionaryMultiColumnGroupKeyGenerator (core/query/aggregation/groupby/DefaultGroupByExecutor.java:106-121) -- run through the column's
KEY IMAGE (the fixed-bit stream of value - min built on first use: pg_kernels.h build_raw_key_image_kernel), keys back as min + digit
(pg_group_key_info), numGroupsLimit honoured in docId order.  Against the oracle, which tests/test_oracle_raw_keys.py holds against a
per-doc restatement keyed by the true values."""
import numpy as np
import pytest

from oracle import oracle
from pinot_amd import _abi
from pinot_amd import query as Q
from pinot_amd import segment as S
import helpers as H
import raw_key_cases as RC

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", RC.cases(), ids=[c[0] for c in RC.cases()])
def test_no_dictionary_group_key_generators(engine, case):
    seg, key_values, specs = RC.build(case)
    with engine.open(seg) as g:
        before = g.device_bytes()
        for spec in specs:
            assert g.check(spec) == _abi.PG_OK
            got = g.execute(spec)
            want = oracle.execute(seg, spec)
            H.assert_results_equal(got, want)
            assert got.group_key_kind == want.group_key_kind == case[3]
            assert got.group_keys == want.group_keys                       # same rows, same (ascending raw key) order
            assert got.group_ids64 == want.group_ids64
            assert got.num_groups_limit_reached == want.num_groups_limit_reached
            assert got.group_id_upper_bound == want.group_id_upper_bound
            # and keyed by the true values: the device's base (the column's smallest value) + digit
            mask = None
            if spec.filter is not None:
                words, _ = oracle.filter_bitmap(seg, Q.QuerySpec([], filter=spec.filter))
                mask = np.unpackbits(words.view(np.uint8), bitorder="little")[: seg.num_docs].astype(bool)
            rows, _ = RC.numpy_groups(key_values, spec, mask, seg.num_docs)
            assert sorted(RC.key_tuples(got, seg, spec, lambda c: g.group_key_info(c)[0])) == sorted(rows)
        for c in range(len(case[2])):
            base, is_offset, null_entry = g.group_key_info(c)
            assert is_offset == RC.is_raw(seg, c)
            assert base == (int(key_values[c].min()) if is_offset else 0)
            assert null_entry == (int(key_values[c].max() - key_values[c].min()) + 1 if is_offset else seg.columns[c].cardinality)
        assert g.device_bytes() > before                                   # the key images are resident now (and counted)


def test_key_columns_outside_the_key_image_are_keyed_through_a_rank_image(engine):
    """Round 5: a raw INT column whose value range is beyond an int and a raw DOUBLE column were declined as group keys (PG_ERR_UNSUPPORTED);
    they are keyed by value now, through a dictionary the device builds from the column (tests/test_gpu_rank_keys.py).  pg_query_check and
    pg_execute still agree, pg_group_key_info says is_offset = 2.  Under null handling such a key keeps the CPU plan."""
    n = 5000
    rng = np.random.default_rng(3)
    wide = np.array([-(2 ** 31), 2 ** 31 - 1] + list(rng.integers(-1000, 1000, n - 2)), dtype=np.int32)
    v = S.Column.synthetic_uniform("v", n, np.arange(50, dtype=np.int32), seed=1)
    seg = S.SegmentData("wide", n, [S.Column.raw("k", wide), S.Column.raw_typed("d", np.round(rng.random(n), 2)), v])
    with engine.open(seg) as g:
        for col in (0, 1):
            spec = Q.QuerySpec([(Q.COUNT, -1), (Q.MAX, 2)], group_by=[col])
            assert g.check(spec) == _abi.PG_OK
            got, want = g.execute(spec), oracle.execute(seg, spec)
            H.assert_results_equal(got, want)
            assert got.group_keys == want.group_keys
            assert g.group_key_info(col)[1] == 2
            nulls = Q.QuerySpec([(Q.COUNT, -1)], group_by=[col], null_handling=True)
            assert g.check(nulls) == _abi.PG_OK                       # (no null vector in this segment: null handling changes nothing)
        got = g.execute(Q.QuerySpec([(Q.MAX, 0), (Q.COUNT, -1)], group_by=[2]))
        H.assert_results_equal(got, oracle.execute(seg, Q.QuerySpec([(Q.MAX, 0), (Q.COUNT, -1)], group_by=[2])))


def test_nullable_raw_key_under_null_handling(engine):
    """enableNullHandling: NULL is a key of its own (NoDictionarySingleColumnGroupKeyGenerator.java:150-175); the nullable raw column is read
    through the null-key image of its key image (digit max - min + 1 = NULL)."""
    n = 40_013
    rng = np.random.default_rng(11)
    values = rng.integers(-50, 450, n).astype(np.int32)
    nulls = rng.random(n) < 0.07
    values[nulls] = 0                                                      # the default null value of an INT metric sits in the forward index
    k = S.Column.raw("k", values).with_nulls(nulls)
    v = S.Column.synthetic_uniform("v", n, (np.arange(900, dtype=np.int64) * 3 + 1).astype(np.int32), seed=5)
    f = S.Column.synthetic_uniform("f", n, np.arange(100, dtype=np.int32), seed=6)
    seg = S.SegmentData("nullable_raw_key", n, [k, v, f])
    with engine.open(seg) as g:
        for spec in (Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, 1)], group_by=[0], null_handling=True),
                     Q.QuerySpec([(Q.MAX, 1)], filter=Q.leaf(Q.Pred.dict_range(2, 10, 60)), group_by=[0], null_handling=True, num_groups_limit=12)):
            assert g.check(spec) == _abi.PG_OK
            want = oracle.execute(seg, spec)
            got = g.execute(spec)
            assert g.group_key_info(0) == (-50, True, 500)
            assert any(t[0] == 500 for t in got.group_keys)               # the NULL key: digit max - min + 1 of the value range [-50, 449]
            H.assert_results_equal(got, want, check_stats=False)
            assert got.group_keys == want.group_keys
            assert got.stats[0] == want.stats[0]
