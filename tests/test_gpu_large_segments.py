"""GPU parity on segments large enough that a wavefront takes SEVERAL tiles (12 M docs: 5 860 tiles for at most ~5 000 resident waves).

Round 4 found group_private_kernel's direct-HBM-table forms wrong from the second tile of a wave on -- built for 1024-thread workgroups
they spilled ~40 registers inside the masked tile loop, and what came back under a partial exec mask was not what had been stored: a
filtered GROUP BY over more than 2 M raw keys on a segment above ~8 M docs lost most of its docs, an unfiltered one counted the docs past
numDocs of the last tile.  Every test of that path ran on segments of at most 4.5 M docs -- one tile per wave.  These cases hold the
kernels whose tile loops carry the most state to the oracle at a size where the loop iterates: the direct group-by (four accumulators keep
it off the partitioned path), the four-slot typed scan, and the LDS-staged group-by with a global table."""
import numpy as np
import pytest

from oracle import oracle
from pinot_amd import query as Q
from pinot_amd import segment as S
import helpers as H

pytestmark = pytest.mark.gpu

N = 12_000_017


@pytest.fixture(scope="module")
def big_segment():
    rng = np.random.default_rng(12)
    v = S.Column.synthetic_uniform("v", N, (np.arange(100000, dtype=np.int64) * 7 + 3).astype(np.int32), seed=1)
    f = S.Column.synthetic_uniform("f", N, np.arange(1000, dtype=np.int32), seed=2)
    k = S.Column.synthetic_uniform("k", N, np.arange(1000, dtype=np.int32) * 3, seed=3)
    b = S.Column.synthetic_uniform("b", N, np.arange(65536, dtype=np.int32) * 2, seed=5)
    lraw = S.Column.raw_typed("lraw", rng.integers(-10 ** 12, 10 ** 12, N).astype(np.int64))
    draw = S.Column.raw_typed("draw", rng.normal(0.0, 1e4, N))
    return S.SegmentData("big", N, [v, f, k, b, lraw, draw])


def test_direct_group_by_over_a_hundred_million_keys(engine, big_segment):
    seg = big_segment
    flt = Q.leaf(Q.Pred.dict_range(1, 0, 100))
    with engine.open(seg) as g:
        exact = g.execute(Q.QuerySpec([(Q.COUNT, -1)], filter=flt)).aggregations[0].count
        # (SUM f, MIN f, MAX f, MAX b: four accumulators -- more than the partitioned path carries -- over v x k = 10^8 raw keys)
        aggs = [(Q.SUM, 1), (Q.MIN, 1), (Q.MAX, 1), (Q.MAX, 3), (Q.COUNT, -1)]
        for f_, want_docs in ((None, N), (flt, exact)):
            spec = Q.QuerySpec(aggs, filter=f_, group_by=[0, 2], num_groups_limit=500)
            got, want = g.execute(spec), oracle.execute(seg, spec)
            assert got.dominant_kernel == "group_private_kernel"
            assert got.stats[0] == want_docs == want.stats[0]
            H.assert_results_equal(got, want)
        # the same key space through the partitioned path (two levels), at a size where its workgroups take several rounds
        spec = Q.QuerySpec([(Q.SUM, 1), (Q.COUNT, -1)], filter=flt, group_by=[0, 2], num_groups_limit=500)
        got, want = g.execute(spec), oracle.execute(seg, spec)
        assert got.stats[0] == exact
        H.assert_results_equal(got, want)


def test_four_slot_typed_scan(engine, big_segment):
    seg = big_segment
    flt = Q.leaf(Q.Pred.dict_range(1, 0, 100))
    with engine.open(seg) as g:
        for f_ in (None, flt):
            spec = Q.QuerySpec([(Q.SUM, 4), (Q.MAX, 4), (Q.SUM, 5), (Q.MIN, 5), (Q.SUM, 0), (Q.MAX, 3), (Q.COUNT, -1)], filter=f_)
            got, want = g.execute(spec), oracle.execute(seg, spec)
            H.assert_results_equal(got, want)


def test_lds_staged_group_by_with_a_global_table(monkeypatch, big_segment):
    """PINOT_GPU_GROUP_PRIVATE=0: scan_group_kernel's global-table forms (38-40 registers spilled in its tile loop)."""
    import torch  # noqa: F401
    from pinot_amd.engine import Engine
    monkeypatch.setenv("PINOT_GPU_GROUP_PRIVATE", "0")
    eng = Engine(device_id=0, time_kernels=True)
    try:
        seg = big_segment
        flt = Q.leaf(Q.Pred.dict_range(1, 0, 100))
        with eng.open(seg) as g:
            exact = g.execute(Q.QuerySpec([(Q.COUNT, -1)], filter=flt)).aggregations[0].count
            for f_, want_docs in ((None, N), (flt, exact)):
                spec = Q.QuerySpec([(Q.SUM, 1), (Q.MAX, 3), (Q.COUNT, -1)], filter=f_, group_by=[0, 2], num_groups_limit=500)
                got, want = g.execute(spec), oracle.execute(seg, spec)
                assert got.stats[0] == want_docs
                H.assert_results_equal(got, want)
    finally:
        monkeypatch.delenv("PINOT_GPU_GROUP_PRIVATE")
        Engine(device_id=0, time_kernels=True)
