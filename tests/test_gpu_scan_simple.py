"""GPU tests of scan_simple_kernel (pinot_amd/csrc/pg_scan_simple.h): one dictionary-range leaf (or no filter) in front of at most one
aggregated packed column, both of at most 20 bits -- the shape of BASELINE.json configs[1].  The planner sends that shape there and
everything else to scan_private_kernel; both must give the oracle's answer bit for bit, so every case runs in both kernels
(PINOT_GPU_SCAN_SIMPLE=0 keeps the general kernel) -- which also keeps the general kernel's coverage of the simple shapes now that
they no longer reach it by default.  What the reference does for the same query: DocIdSetOperator.java:59-86 over ONE
SVScanDocIdIterator (:76-142) -> ProjectionOperator -> ONE AggregationFunction.aggregate (FixedBitIntReaderTest.java:52-84 is the
width sweep these cases restate)."""
import numpy as np
import pytest

from oracle import oracle
from pinot_amd import query as Q
from pinot_amd import segment as S
import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture
def both_kernels(engine):
    def run(seg, spec, expect_simple=True):
        want = oracle.execute(seg, spec)
        with engine.open(seg) as g:
            try:
                got = g.execute(spec)
                assert (got.dominant_kernel == "scan_simple_kernel") == expect_simple, got.dominant_kernel
                H.assert_results_equal(got, want, True)
                engine.reinit(PINOT_GPU_SCAN_SIMPLE="0")
                general = g.execute(spec)
                assert general.dominant_kernel != "scan_simple_kernel"
                H.assert_results_equal(general, want, True)
            finally:
                engine.reinit(PINOT_GPU_SCAN_SIMPLE=None)
        return got
    return run


@pytest.mark.parametrize("bits", list(range(1, 21)))
def test_every_width_as_filter_and_as_aggregated_column(both_kernels, bits):
    """A b-bit filter column in front of a (21 - b)-bit aggregated column (affine dictionary: the dictId stream is its own value
    plane), so that every width 1 .. 20 is decoded once by the range leaf and once by the aggregation; ragged sizes."""
    rng = np.random.default_rng(500 + bits)
    n = 6151 + 37 * bits                                      # three tiles and a ragged one
    cf = 2 ** bits - (1 if bits > 1 and bits % 3 == 0 else 0)
    bv = 21 - bits
    cv = 2 ** bv - (1 if bv > 1 and bv % 2 == 0 else 0)
    f = S.Column.synthetic_uniform("f", n, np.arange(cf, dtype=np.int32) * 2 - 9, seed=bits)
    v = S.Column.synthetic_uniform("v", n, (np.arange(cv, dtype=np.int64) * 7 + 3).astype(np.int32), seed=100 + bits)
    assert f.bits == bits and v.bits == bv
    seg = S.SegmentData("s%d" % bits, n, [f, v])
    lo, hi = cf // 4, max(cf // 4 + 1, (3 * cf) // 4)
    both_kernels(seg, Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, 1), (Q.MIN, 1), (Q.MAX, 1), (Q.AVG, 1)], filter=Q.leaf(Q.Pred.dict_range(0, lo, hi))))
    both_kernels(seg, Q.QuerySpec([(Q.SUM, 1)], filter=Q.leaf(Q.Pred.dict_range(0, 0, max(1, cf // 3)))))                      # lo == 0: the kLoZero decode
    both_kernels(seg, Q.QuerySpec([(Q.COUNT, -1), (Q.MAX, 1)], filter=Q.leaf(Q.Pred.dict_range(0, lo, hi, exclusive=True))))     # NOT_IN / NEQ form
    del rng


def test_shapes_at_the_edges(both_kernels):
    rng = np.random.default_rng(77)
    for n in (1, 63, 64, 2047, 2048, 2049, 10_000, 70_001):
        f, fid, _ = H.random_dict_column(rng, "f", n, 700)
        v = S.Column.synthetic_uniform("v", n, (np.arange(30000, dtype=np.int64) * 5 + 11).astype(np.int32), seed=n)
        seg = S.SegmentData("e%d" % n, n, [f, v])
        both_kernels(seg, Q.QuerySpec([(Q.SUM, 1), (Q.COUNT, -1)], filter=Q.leaf(Q.Pred.dict_range(0, 100, 400))))
        both_kernels(seg, Q.QuerySpec([(Q.MIN, 1), (Q.MAX, 1)], filter=Q.leaf(Q.Pred.dict_range(0, 0, 699))))                 # nearly every doc matches
        both_kernels(seg, Q.QuerySpec([(Q.SUM, 1)], filter=Q.leaf(Q.Pred.dict_range(0, 699, 700))))                            # ~1 in 700: the sparse walk
        got = both_kernels(seg, Q.QuerySpec([(Q.AVG, 1)], filter=Q.leaf(Q.Pred.dict_range(0, 5, 5))), expect_simple=False)     # empty range: folded away before any kernel
        assert got.aggregations[0].count == 0
        both_kernels(seg, Q.QuerySpec([(Q.SUM, 1), (Q.MAX, 1)]))                                                               # no filter at all


def test_what_stays_in_the_general_kernel(both_kernels):
    """Two leaves, two aggregated columns, a column wider than 20 bits, the one-stream shape (its fused decode): scan_private_kernel."""
    rng = np.random.default_rng(5)
    n = 50_021
    f, _, _ = H.random_dict_column(rng, "f", n, 1000)
    g, _, _ = H.random_dict_column(rng, "g", n, 50)
    v = S.Column.synthetic_uniform("v", n, (np.arange(100000, dtype=np.int64) * 7 + 3).astype(np.int32), seed=1)
    w = S.Column.synthetic_uniform("w", n, np.arange(3_000_000, dtype=np.int32), seed=2)                 # 22 bits
    seg = S.SegmentData("general", n, [f, g, v, w])
    fl = Q.leaf(Q.Pred.dict_range(0, 0, 100))
    both_kernels(seg, Q.QuerySpec([(Q.SUM, 2)], filter=Q.and_(fl, Q.leaf(Q.Pred.dict_range(1, 3, 30)))), expect_simple=False)
    both_kernels(seg, Q.QuerySpec([(Q.SUM, 2), (Q.MAX, 1)], filter=fl), expect_simple=False)
    both_kernels(seg, Q.QuerySpec([(Q.SUM, 3)], filter=fl), expect_simple=False)
    both_kernels(seg, Q.QuerySpec([(Q.SUM, 2)], filter=Q.leaf(Q.Pred.dict_range(2, 40000, 60000))), expect_simple=False)
    both_kernels(seg, Q.QuerySpec([(Q.SUM, 2)], filter=fl), expect_simple=True)
