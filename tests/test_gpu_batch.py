"""GPU tests of pg_execute_batch: one query over many resident segments in one call (what BaseCombineOperator does with a thread pool,
core/operator/combine/BaseCombineOperator.java:85-142).  Items whose device work is one launch of the lane-private scan kernel share ONE
launch (scan_private_batch_kernel: every item folds and publishes its own record); every other item runs as a pg_execute of its own on
the library's worker threads.  Item by item the results must be what pg_execute returns -- and what the oracle says."""
import numpy as np
import pytest

from oracle import oracle
from pinot_amd import _abi
from pinot_amd import query as Q
from pinot_amd import segment as S
import helpers as H

pytestmark = pytest.mark.gpu

SIZES = [1, 2047, 2049, 70001, 333337, 1000003, 64, 4096, 250000, 1500007, 99999, 800001]


def _segments():
    """Segments of different sizes whose dictionaries differ (the same VALUE predicate lowers to other dictIds in every segment)."""
    segs = []
    for s, n in enumerate(SIZES):
        rng = np.random.default_rng(1000 + s)
        card = 50 + 37 * s
        v = S.Column.synthetic_uniform("v", n, (np.arange(card, dtype=np.int64) * (3 + s) + s).astype(np.int32), seed=2 * s + 1)
        f, _, _ = H.random_dict_column(rng, "f", n, 20 + s)
        k = S.Column.synthetic_uniform("k", n, np.arange(7, dtype=np.int32), seed=99 + s)
        w, _, _ = H.random_dict_column(rng, "w", n, 30, with_inverted=True)
        segs.append(S.SegmentData("b%d" % s, n, [v, f, k, w]))
    return segs


def _spec(seg, s, shape):
    card_f = seg.columns[1].cardinality
    flt = Q.leaf(Q.Pred.dict_range(1, s % 5, min(card_f, s % 5 + 6)))
    if shape == "sum":
        return Q.QuerySpec([(Q.SUM, 0), (Q.COUNT, -1)], filter=flt)
    if shape == "minmax":
        return Q.QuerySpec([(Q.MIN, 0), (Q.MAX, 1), (Q.AVG, 0), (Q.SUM, 1)], filter=Q.or_(flt, Q.leaf(Q.Pred.dict_range(0, 0, 5))))
    if shape == "nofilter":
        return Q.QuerySpec([(Q.SUM, 0), (Q.MAX, 0)])
    if shape == "group":
        return Q.QuerySpec([(Q.SUM, 0), (Q.COUNT, -1)], filter=flt, group_by=[2])
    if shape == "inverted":
        return Q.QuerySpec([(Q.SUM, 0)], filter=Q.and_(Q.leaf(Q.Pred.dict_range(3, 3, 9, inverted=True)), flt))
    if shape == "and2":
        return Q.QuerySpec([(Q.COUNT, -1)], filter=Q.and_(flt, Q.leaf(Q.Pred.dict_range(0, 0, 20))))
    raise ValueError(shape)


@pytest.mark.parametrize("shapes", [["sum"], ["minmax"], ["nofilter"], ["sum", "group", "inverted", "minmax", "and2", "nofilter"]])
def test_batch_equals_execute_and_the_oracle(engine, shapes):
    segs = _segments()
    opened = [engine.open(seg) for seg in segs]
    try:
        specs = [_spec(seg, s, shapes[s % len(shapes)]) for s, seg in enumerate(segs)]
        for rep in range(3):                                     # the batch context (records, counters, staging) is reused from call to call
            got = engine.execute_batch(opened, specs)
            for s, (status, res) in enumerate(got):
                assert status == _abi.PG_OK, (s, shapes[s % len(shapes)])
                H.assert_results_equal(res, oracle.execute(segs[s], specs[s]))
                single = opened[s].execute(specs[s])
                assert res.stats == single.stats and res.filter_entries_exact == single.filter_entries_exact
                assert [a.sum_i64 for a in res.aggregations] == [a.sum_i64 for a in single.aggregations]
    finally:
        [g.close() for g in opened]


def test_a_failing_item_does_not_stop_the_others(engine):
    segs = _segments()[:4]
    opened = [engine.open(seg) for seg in segs]
    try:
        specs = [_spec(seg, s, "sum") for s, seg in enumerate(segs)]
        specs[2] = Q.QuerySpec([(Q.SUM, 17)])                    # no such column
        got = engine.execute_batch(opened, specs)
        assert [st for st, _ in got] == [_abi.PG_OK, _abi.PG_OK, _abi.PG_ERR_INVALID_ARGUMENT, _abi.PG_OK]
        assert b"batch item 2" in engine.lib.pg_last_error()
        for s in (0, 1, 3):
            H.assert_results_equal(got[s][1], oracle.execute(segs[s], specs[s]))
        assert engine.execute_batch([], []) == []
    finally:
        [g.close() for g in opened]


def test_batch_without_the_shared_launch(monkeypatch):
    """PINOT_GPU_BATCH_LAUNCH=0: every item as a pg_execute of its own on the worker threads -- same answers."""
    import torch  # noqa: F401
    from pinot_amd.engine import Engine
    monkeypatch.setenv("PINOT_GPU_BATCH_LAUNCH", "0")
    eng = Engine(device_id=0, time_kernels=True)
    try:
        segs = _segments()
        opened = [eng.open(seg) for seg in segs]
        specs = [_spec(seg, s, "sum") for s, seg in enumerate(segs)]
        for s, (status, res) in enumerate(eng.execute_batch(opened, specs)):
            assert status == _abi.PG_OK
            H.assert_results_equal(res, oracle.execute(segs[s], specs[s]))
        [g.close() for g in opened]
    finally:
        monkeypatch.delenv("PINOT_GPU_BATCH_LAUNCH")
        Engine(device_id=0, time_kernels=True)
