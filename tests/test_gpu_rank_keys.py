"""GPU parity: GROUP BY over raw FLOAT / DOUBLE columns and raw INT / LONG columns spanning more than an int, keyed by VALUE through the
dictionary + rank image the device builds from the column (pinot_amd/csrc/pg_rank_image.h; NoDictionarySingleColumnGroupKeyGenerator.java:
100-135, NoDictionaryMultiColumnGroupKeyGenerator).  Against the oracle (which tests/test_oracle_rank_keys.py holds against a numpy
restatement keyed by the values), digit for digit, and the values behind the digits through pg_group_key_values."""
import numpy as np
import pytest

import helpers as H
from oracle import oracle
from pinot_amd import _abi
from pinot_amd import query as Q
from pinot_amd import segment as S
import rank_key_cases as KC
from test_oracle_rank_keys import check_against_numpy

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", KC.cases(), ids=lambda c: c[0])
def test_group_by_raw_float_double_and_wide_columns(engine, case):
    seg, identities, specs = KC.build(case)
    with engine.open(seg) as g:
        before = g.device_bytes()
        for spec in specs:
            assert g.check(spec) == _abi.PG_OK
            got, want = g.execute(spec), oracle.execute(seg, spec)
            H.assert_results_equal(got, want, check_stats=False)
            assert got.group_keys == want.group_keys and got.stats[0] == want.stats[0]
            assert got.num_groups_limit_reached == want.num_groups_limit_reached
        # the values behind the digits: the column's distinct values in Double.compare's / Long.compare's order
        values = {}
        for c in range(len(case[2])):
            base, is_offset, null_entry = g.group_key_info(c)
            if KC.is_rank_keyed(seg, c):
                assert is_offset == 2
                vals = g.group_key_values(c)
                assert np.array_equal(vals, KC.rank_values(seg, c)) and null_entry == len(vals)
                values[c] = vals
        assert values and g.device_bytes() > before               # the dictionaries and rank images are resident now (and counted)
        check_against_numpy(seg, identities, specs[0], g.execute(specs[0]), lambda c: values[c], lambda c: g.group_key_info(c)[0])


def test_rank_keyed_group_bys_in_a_batch_and_on_a_tiny_grid(engine, monkeypatch):
    """The rank image is an ordinary dictionary-encoded key downstream: the batch's shared group-by launch takes it, and so does every
    grid (PINOT_GPU_TEST_CUS=1: many tiles per wave)."""
    monkeypatch.setenv("PINOT_GPU_TEST_CUS", "1")
    cases = [c for c in KC.cases() if c[0] in ("single-double", "single-float", "wide-int", "double-and-dict")]
    built = [KC.build(c, seed=5) for c in cases]
    opened = [engine.open(b[0]) for b in built]
    try:
        specs = [b[2][0] for b in built]
        for rep in range(2):
            for (status, res), b, spec in zip(engine.execute_batch(opened, specs), built, specs):
                assert status == _abi.PG_OK
                want = oracle.execute(b[0], spec)
                H.assert_results_equal(res, want, check_stats=False)
                assert res.group_keys == want.group_keys
    finally:
        [g.close() for g in opened]


@pytest.mark.parametrize("slots_log2", ["6", "12"])
def test_the_dictionary_hash_table_is_rebuilt_larger_when_it_fills(engine, monkeypatch, slots_log2):
    """The device builds a rank-keyed column's dictionary through an open-addressing table that starts at 2^20 slots and is rebuilt 16x
    larger when probes grow long or more than half of it fills (pg_unit_rank_image.hip).  PINOT_GPU_RANK_SLOTS_LOG2 starts it at 64 / 4096
    slots under 40 000 distinct doubles (and the specials, among them the image that equals the table's empty marker: none here has it --
    the LONG case below does): two and one rebuilds, the same dictionary and ranks."""
    monkeypatch.setenv("PINOT_GPU_RANK_SLOTS_LOG2", slots_log2)
    case = [c for c in KC.cases() if c[0] == "many-doubles"][0]
    seg, identities, specs = KC.build(case, seed=9)
    with engine.open(seg) as g:
        got, want = g.execute(specs[0]), oracle.execute(seg, specs[0])
        H.assert_results_equal(got, want, check_stats=False)
        assert got.group_keys == want.group_keys
        assert np.array_equal(g.group_key_values(0), KC.rank_values(seg, 0))
    # a LONG column that holds Long.MAX_VALUE: its order image is all ones -- the table's empty marker; it travels in a flag and ranks last
    n = 20_011
    rng = np.random.default_rng(3)
    pool = np.unique(np.concatenate([np.array([np.iinfo(np.int64).max, np.iinfo(np.int64).min, -1, 0, 1], dtype=np.int64), rng.integers(-(2 ** 62), 2 ** 62, 300, dtype=np.int64)]))
    key = S.Column.raw_typed("k", pool[rng.integers(0, len(pool), n)].astype(np.int64))
    v = S.Column.synthetic_uniform("v", n, (np.arange(500, dtype=np.int64) * 7 + 3).astype(np.int32), seed=4)
    seg2 = S.SegmentData("rank_max", n, [key, v])
    spec = Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, 1)], group_by=[0])
    with engine.open(seg2) as g:
        got, want = g.execute(spec), oracle.execute(seg2, spec)
        H.assert_results_equal(got, want, check_stats=False)
        assert got.group_keys == want.group_keys
        vals = g.group_key_values(0)
        assert vals[-1] == np.iinfo(np.int64).max and np.array_equal(vals, KC.rank_values(seg2, 0))
