"""GPU parity of the histogram SUM path (pinot_amd/csrc/pg_scan_hist.h): SUM(col) = sum_d matches[d] * dictionary[d].

The dictionaries here are IRREGULAR (sorted distinct values drawn from the whole int32 range, or from a 2^20 window), so neither
the arithmetic-progression shortcut nor a narrow value plane applies: what is tested is Dictionary.readIntValues + SumAggregationFunction
(sspi/index/reader/Dictionary.java:207-211, core/query/aggregation/function/SumAggregationFunction.java:69-157) through the
LDS histogram, against the oracle's per-doc dictionary lookups, bit for bit.
"""
import os

import numpy as np
import pytest

from oracle import oracle
from pinot_amd import query as Q
from pinot_amd import segment as S
import helpers as H

pytestmark = pytest.mark.gpu

ALL_AGGS = lambda c: [(Q.COUNT, -1), (Q.SUM, c), (Q.MIN, c), (Q.MAX, c), (Q.AVG, c)]


def irregular_dictionary(cardinality, seed, lo=-2 ** 31, hi=2 ** 31 - 1):
    rng = np.random.default_rng(seed)
    vals = np.unique(rng.integers(lo, hi, int(cardinality * 1.05) + 16, dtype=np.int64))
    while vals.shape[0] < cardinality:
        vals = np.unique(np.concatenate([vals, rng.integers(lo, hi, cardinality, dtype=np.int64)]))
    pick = np.sort(rng.permutation(vals.shape[0])[:cardinality])
    return vals[pick].astype(np.int32)


def segment(n, cardinality, seed, ids=None, value_range=None):
    rng = np.random.default_rng(seed)
    dv = irregular_dictionary(cardinality, seed + 1, *(value_range or (-2 ** 31, 2 ** 31 - 1)))
    if ids is None:
        ids = rng.integers(0, cardinality, n).astype(np.int32)
    v = S.Column.from_dict_ids("v", dv, ids)
    fids = rng.integers(0, 1000, n).astype(np.int32)
    f = S.Column.from_dict_ids("f", np.arange(1000, dtype=np.int32) * 3 - 7, fids)
    return S.SegmentData("hist_%d_%d" % (cardinality, n), n, [v, f]), ids, fids, dv


def check(engine, seg, spec, expect_kernel="scan_hist_kernel"):
    with engine.open(seg) as g:
        got = g.execute(spec)
        again = g.execute(spec)         # the counters are zeroed per launch: a second run must agree
    want = oracle.execute(seg, spec)
    H.assert_results_equal(got, want)
    H.assert_results_equal(again, want)
    if expect_kernel is not None:
        assert got.dominant_kernel == expect_kernel, got.dominant_kernel
    return got


@pytest.mark.parametrize("cardinality,bits", [(3, 32), (1000, 32), (38912, 32), (38913, 16), (77824, 16), (77825, 8), (100000, 8), (155648, 8)])
def test_filtered_sum_over_an_irregular_dictionary(engine, cardinality, bits):
    """C2b shape at every counter width the histogram uses (32 / 16 / 8 bits per dictId, by cardinality)."""
    n = 300007
    seg, ids, fids, dv = segment(n, cardinality, 1000 + cardinality)
    got = check(engine, seg, Q.QuerySpec([(Q.SUM, 0)], filter=Q.leaf(Q.Pred.dict_range(1, 0, 100))))
    m = fids < 100
    assert got.aggregations[0].sum_i64 == int(dv[ids[m]].astype(np.int64).sum())
    assert got.aggregations[0].count == int(m.sum())
    check(engine, seg, Q.QuerySpec(ALL_AGGS(0), filter=Q.leaf(Q.Pred.dict_range(1, 0, 500))))
    check(engine, seg, Q.QuerySpec(ALL_AGGS(0)))
    # C2a shape: the predicate on the summed column itself
    check(engine, seg, Q.QuerySpec([(Q.SUM, 0), (Q.COUNT, -1)], filter=Q.leaf(Q.Pred.dict_range(0, cardinality // 3, max(cardinality // 3 + 1, 2 * cardinality // 3)))))
    # nothing matches
    check(engine, seg, Q.QuerySpec(ALL_AGGS(0), filter=Q.leaf(Q.Pred.dict_range(1, 5, 5))), expect_kernel=None)


def test_cardinality_above_the_lds_histogram_uses_the_other_paths(engine):
    seg, ids, fids, dv = segment(200003, 155649, 7)
    got = check(engine, seg, Q.QuerySpec([(Q.SUM, 0)], filter=Q.leaf(Q.Pred.dict_range(1, 0, 100))), expect_kernel=None)
    assert got.dominant_kernel != "scan_hist_kernel"


@pytest.mark.parametrize("n", [1, 31, 63, 64, 2047, 2048, 2049, 4097, 65536 + 5])
def test_ragged_sizes(engine, n):
    seg, ids, fids, dv = segment(n, 100000, 50 + n)
    check(engine, seg, Q.QuerySpec(ALL_AGGS(0), filter=Q.leaf(Q.Pred.dict_range(1, 0, 700))), expect_kernel=None)
    check(engine, seg, Q.QuerySpec([(Q.SUM, 0)]), expect_kernel=None)


def test_narrow_value_window_and_filter_trees(engine):
    """Values from a 2^20 window (a 20-bit plane would be the alternative); OR / NOT / IN leaves feed the same histogram."""
    n = 250000
    seg, ids, fids, dv = segment(n, 100000, 77, value_range=(5_000_000, 5_000_000 + 2 ** 20))
    flt = Q.or_(Q.and_(Q.leaf(Q.Pred.dict_range(1, 0, 300)), Q.not_(Q.leaf(Q.Pred.dict_range(0, 0, 20000)))),
                Q.leaf(Q.Pred.dict_set(1, [5, 17, 900, 999], 1000)))
    check(engine, seg, Q.QuerySpec(ALL_AGGS(0), filter=flt))
    check(engine, seg, Q.QuerySpec([(Q.AVG, 0)], filter=Q.leaf(Q.Pred.dict_range(1, 10, 20))))


def test_long_dictionary_in_the_int32_domain(engine):
    """A LONG dictionary whose range fits 31 bits is an offset dictionary on the device: the histogram sums (value - min)."""
    n = 120000
    rng = np.random.default_rng(5)
    vals = rng.integers(0, 2 ** 31 - 2, n, dtype=np.int64) + 1_600_000_000_000
    col = S.Column.dict_encoded_typed("t", vals)
    f = S.Column.from_dict_ids("f", np.arange(100, dtype=np.int32), rng.integers(0, 100, n).astype(np.int32))
    seg = S.SegmentData("hist_long", n, [col, f])
    check(engine, seg, Q.QuerySpec(ALL_AGGS(0), filter=Q.leaf(Q.Pred.dict_range(1, 0, 50))))


def _engine_with_env(env):
    import torch  # noqa: F401
    from pinot_amd.engine import Engine
    os.environ.update(env)
    try:
        yield Engine(device_id=0, time_kernels=True)
    finally:
        for k in env:
            del os.environ[k]
        Engine(device_id=0, time_kernels=True)       # pg_init re-reads the environment


@pytest.fixture()
def narrow_counter_engine():
    """PINOT_GPU_HIST_BITS forces 8-bit counters whatever the cardinality, so that small test segments make counters wrap."""
    yield from _engine_with_env({"PINOT_GPU_HIST_BITS": "8"})


@pytest.fixture()
def guarded_engine():
    """Every column starts in the guarded tier (returning adds + guard-bit claims), with 8-bit counters."""
    yield from _engine_with_env({"PINOT_GPU_HIST_BITS": "8", "PINOT_GPU_HIST_GUARD": "1"})


def skewed_segment(hot_fraction, n=1_500_000, cardinality=5000):
    rng = np.random.default_rng(int(hot_fraction * 100))
    ids = rng.integers(0, cardinality, n).astype(np.int32)
    hot = rng.random(n) < hot_fraction
    ids[hot] = 4321
    return segment(n, cardinality, 99, ids=ids)


@pytest.mark.parametrize("hot_fraction", [0.0, 0.02, 0.3, 1.0])
def test_wrapped_counters_are_detected_and_the_next_tier_answers(narrow_counter_engine, hot_fraction):
    """8-bit counters over 1.5 M docs and 5000 dictIds wrap (about 300 docs per counter and segment, many more for the hot dictId):
    the checksum `sum of counters == matches` fails, the column moves to guarded counters (which count the hot dictId in claims of 128)
    and, under extreme skew, on to the other paths.  Every answer is exact, including the repeats that start in the later tier."""
    seg, ids, fids, dv = skewed_segment(hot_fraction)
    for spec in (Q.QuerySpec([(Q.SUM, 0)]), Q.QuerySpec(ALL_AGGS(0), filter=Q.leaf(Q.Pred.dict_range(1, 0, 900)))):
        with narrow_counter_engine.open(seg) as g:
            for _ in range(3):
                got = g.execute(spec)
                H.assert_results_equal(got, oracle.execute(seg, spec))
    assert oracle.execute(seg, Q.QuerySpec([(Q.SUM, 0)])).aggregations[0].sum_i64 == int(dv[ids].astype(np.int64).sum())


@pytest.fixture()
def guarded_few_blocks_engine():
    """Guarded 8-bit counters and only 8 workgroups: a 3 M-doc segment then puts ~190 docs on every counter of a workgroup, one at a
    time (about one per tile), which is the regime the guard-bit claims are for."""
    yield from _engine_with_env({"PINOT_GPU_HIST_BITS": "8", "PINOT_GPU_HIST_GUARD": "1", "PINOT_GPU_HIST_BLOCKS": "8"})


def test_guard_bit_claims_keep_slowly_filling_counters_exact(guarded_few_blocks_engine):
    n, cardinality = 3_000_000, 2000
    seg, ids, fids, dv = segment(n, cardinality, 4242)
    for spec in (Q.QuerySpec([(Q.SUM, 0)]), Q.QuerySpec(ALL_AGGS(0), filter=Q.leaf(Q.Pred.dict_range(1, 0, 800)))):
        with guarded_few_blocks_engine.open(seg) as g:
            got = g.execute(spec)
        H.assert_results_equal(got, oracle.execute(seg, spec))
        assert got.dominant_kernel == "scan_hist_kernel"       # every counter passed 128 at least once and none ran away: no fallback


@pytest.mark.parametrize("hot_fraction", [0.0, 0.02, 0.3])
def test_guarded_counters_under_bursts(guarded_engine, hot_fraction):
    """A dictId hit by sixteen wavefronts at once can outrun the claims: the alarm sends the query to the other paths.  Exact either way."""
    seg, ids, fids, dv = skewed_segment(hot_fraction, n=400_000)
    for spec in (Q.QuerySpec([(Q.SUM, 0)]), Q.QuerySpec(ALL_AGGS(0), filter=Q.leaf(Q.Pred.dict_range(1, 0, 500)))):
        with guarded_engine.open(seg) as g:
            got = g.execute(spec)
            H.assert_results_equal(got, oracle.execute(seg, spec))
            if hot_fraction == 0.0:
                assert got.dominant_kernel == "scan_hist_kernel"


def test_a_small_sparse_filter_keeps_the_plain_tier(narrow_counter_engine):
    """Few matches per counter: the plain 8-bit counters do not wrap and the checksum passes (no rerun: same kernel, one launch)."""
    seg, ids, fids, dv = skewed_segment(0.0, n=200_000, cardinality=5000)
    spec = Q.QuerySpec([(Q.SUM, 0), (Q.COUNT, -1)], filter=Q.leaf(Q.Pred.dict_range(1, 0, 100)))
    check(narrow_counter_engine, seg, spec)
