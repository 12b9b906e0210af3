"""GPU tests of group-by key spaces beyond an int: the reference's LongMapBasedHolder and ArrayMapBasedHolder
(core/query/aggregation/groupby/DictionaryBasedGroupKeyGenerator.java:150-184, 628-700, 808+) as a hashed table in HBM
(group_private_kernel<.., kHash>: 64-bit keys, open addressing; two chained tables when the key does not fit a long), keys back as long
raw keys / dictId tuples, numGroupsLimit honoured in docId order.  Against the oracle, which tests/test_oracle_hash_holders.py holds
against a per-doc restatement."""
import numpy as np
import pytest

from oracle import oracle
from pinot_amd import _abi
from pinot_amd import query as Q
import hash_holder_cases as HC
import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", HC.cases(), ids=[c[0] for c in HC.cases()])
def test_long_and_array_map_holders(engine, case):
    seg, ids, specs = HC.build(case)
    with engine.open(seg) as g:
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


def test_int_holders_also_return_the_dict_id_tuples(engine):
    rng = np.random.default_rng(5)
    n = 30_011
    a, ia, _ = H.random_dict_column(rng, "a", n, 13)
    b, ib, _ = H.random_dict_column(rng, "b", n, 700)
    from pinot_amd import segment as S
    seg = S.SegmentData("tuples", n, [a, b])
    for group_by in ([0], [0, 1], [1, 0]):
        spec = Q.QuerySpec([(Q.COUNT, -1)], group_by=group_by)
        with engine.open(seg) as g:
            got = g.execute(spec)
        want = oracle.execute(seg, spec)
        H.assert_results_equal(got, want)
        assert got.group_key_kind == 0 and got.group_keys == want.group_keys
        cards = [seg.columns[c].cardinality for c in group_by]
        for gid, tup in zip(sorted(got.groups), got.group_keys):
            raw, mult = 0, 1
            for d, c in zip(tup, cards):
                raw += d * mult; mult *= c
            assert raw == gid


@pytest.mark.parametrize("case", [HC.cases()[0], HC.cases()[2], HC.cases()[3]], ids=["long-2cols", "array-3cols", "long-dense"])
def test_hashed_holders_over_raw_eight_byte_inputs(engine, case):
    """Round 6b: SUM / MIN / MAX / AVG of RAW LONG, DOUBLE and FLOAT columns under keys beyond an int -- group_typed_direct_kernel<.., kHash>:
    the slots from the hashed table, the accumulators as in the direct-indexed form (SumAggregationFunction.aggregateGroupBySV over a
    no-dictionary column, DictionaryBasedGroupKeyGenerator.java:628-806 for the keys).  With a filter, and with numGroupsLimit binding."""
    from pinot_amd import segment as S
    seg, ids, specs = HC.build(case)
    n, nk = seg.num_docs, len(case[2])
    rng = np.random.default_rng(11)
    lv = rng.integers(-(1 << 40), 1 << 40, n).astype(np.int64)
    dv = (rng.standard_normal(n) * 1e6).round(3)
    fv = rng.integers(-5000, 5000, n).astype(np.float32)
    first = len(seg.columns)
    seg2 = S.SegmentData(seg.name + "_raw8", n, list(seg.columns) + [S.Column.raw_typed("l", lv), S.Column.raw_typed("d", dv), S.Column.raw_typed("f", fv)])
    keys = list(range(nk))
    flt = Q.leaf(Q.Pred.dict_range(nk + 2, 0, 37))
    specs2 = [Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, first), (Q.MAX, first + 1), (Q.MIN, first)], group_by=keys),
              Q.QuerySpec([(Q.SUM, first + 1), (Q.AVG, first), (Q.MIN, first + 2), (Q.SUM, nk)], filter=flt, group_by=keys),
              Q.QuerySpec([(Q.MAX, first), (Q.SUM, first + 2)], group_by=keys, num_groups_limit=50),
              Q.QuerySpec([(Q.AVG, first + 1)], filter=flt, group_by=keys, num_groups_limit=7)]
    with engine.open(seg2) as g:
        for spec in specs2:
            assert g.check(spec) == _abi.PG_OK
            got = g.execute(spec)
            want = oracle.execute(seg2, spec)
            H.assert_results_equal(got, want)
            assert got.group_key_kind == want.group_key_kind == case[3]
            assert got.group_keys == want.group_keys and got.group_ids64 == want.group_ids64
            assert got.num_groups_limit_reached == want.num_groups_limit_reached


@pytest.mark.parametrize("case", [HC.cases()[0], HC.cases()[2]], ids=["long-2cols", "array-3cols"])
def test_hashed_holders_summing_eight_byte_dictionary_columns(engine, case):
    """SUM / AVG of DICTIONARY columns whose values are LONG beyond an int, DOUBLE or FLOAT under keys beyond an int: the values are gathered
    from the 8-byte device dictionary inside group_typed_direct_kernel<.., kHash> (SumAggregationFunction.aggregateGroupBySV over
    getDoubleValuesSV of a dictionary column; DictionaryBasedGroupKeyGenerator.java:628-806 for the keys).  Beside 32-bit-domain inputs,
    with a filter, and with numGroupsLimit binding."""
    from pinot_amd import segment as S
    seg, ids, specs = HC.build(case)
    n, nk = seg.num_docs, len(case[2])
    rng = np.random.default_rng(23)
    lv = (rng.integers(0, 977, n).astype(np.int64) - 400) * (1 << 37)
    dv = (rng.integers(0, 1500, n) * 0.125 - 77.5).astype(np.float64)
    fv = (rng.integers(0, 300, n) * 0.5 - 31.0).astype(np.float32)
    first = len(seg.columns)
    seg2 = S.SegmentData(seg.name + "_dict8", n, list(seg.columns) + [S.Column.dict_encoded_typed("l", lv), S.Column.dict_encoded_typed("d", dv), S.Column.dict_encoded_typed("f", fv)])
    keys = list(range(nk))
    flt = Q.leaf(Q.Pred.dict_range(nk + 2, 0, 37))
    specs2 = [Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, first), (Q.MAX, first + 1), (Q.MIN, first)], group_by=keys),
              Q.QuerySpec([(Q.SUM, first + 1), (Q.AVG, first), (Q.AVG, first + 2), (Q.SUM, nk)], filter=flt, group_by=keys),
              Q.QuerySpec([(Q.MAX, first), (Q.SUM, first + 2)], group_by=keys, num_groups_limit=50),
              Q.QuerySpec([(Q.AVG, first + 1)], filter=flt, group_by=keys, num_groups_limit=7)]
    with engine.open(seg2) as g:
        for spec in specs2:
            assert g.check(spec) == _abi.PG_OK
            got = g.execute(spec)
            want = oracle.execute(seg2, spec)
            H.assert_results_equal(got, want)
            assert got.group_key_kind == want.group_key_kind == case[3]
            assert got.group_keys == want.group_keys and got.group_ids64 == want.group_ids64
            assert got.num_groups_limit_reached == want.num_groups_limit_reached


def test_plan_time_limits_of_the_hashed_holders(engine):
    """pg_query_check on the hashed holders: the SUM of a DICTIONARY column with 8-byte values is accepted (it runs: the test above) unless
    numDocs x max|value| could overflow the table's one int64 slot; MIN / MAX run on dictIds either way."""
    from pinot_amd import segment as S
    case = HC.cases()[0]
    seg, ids, specs = HC.build(case)
    wide = S.Column.dict_encoded_typed("big", (np.arange(seg.num_docs, dtype=np.int64) % 977) * (1 << 40))
    fits = S.Column.dict_encoded_typed("fits", (np.arange(seg.num_docs, dtype=np.int64) % 977) * (1 << 33))
    first = len(seg.columns)
    seg2 = S.SegmentData("hash_wide", seg.num_docs, list(seg.columns) + [wide, fits])
    with engine.open(seg2) as g:
        assert g.check(Q.QuerySpec([(Q.SUM, first)], group_by=[0, 1])) == _abi.PG_ERR_UNSUPPORTED      # could overflow
        assert g.check(Q.QuerySpec([(Q.SUM, first + 1)], group_by=[0, 1])) == _abi.PG_OK
        assert g.check(Q.QuerySpec([(Q.MAX, first)], group_by=[0, 1])) == _abi.PG_OK      # MIN / MAX run on dictIds
        # a table beyond PINOT_GPU_GROUP_TABLE_BYTES is declined at plan time (the CPU plan keeps the query), and accepted again without the bound
        engine.reinit(PINOT_GPU_GROUP_TABLE_BYTES=4096)
        try:
            assert g.check(Q.QuerySpec([(Q.SUM, first + 1)], group_by=[0, 1])) == _abi.PG_ERR_UNSUPPORTED
            assert g.check(Q.QuerySpec([(Q.COUNT, -1)], group_by=[0, 1])) == _abi.PG_ERR_UNSUPPORTED
        finally:
            engine.reinit(PINOT_GPU_GROUP_TABLE_BYTES=None)
        assert g.check(Q.QuerySpec([(Q.SUM, first + 1)], group_by=[0, 1])) == _abi.PG_OK
