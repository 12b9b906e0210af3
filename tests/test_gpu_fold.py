"""GPU tests of the in-kernel fold ("last block done", publish_block_partial in pg_kernels.h): the workgroup whose arrival completes the
counter folds every workgroup's record inside the scan kernel -- an inter-workgroup hand-off across XCDs that an idle chip or an
L1-cold reader would hide if it were wrong (MI355X_MICROARCH.md).  So: many back-to-back launches of different grid sizes on ONE
context (the folding workgroup re-reads a record buffer its CU may still hold stale lines of), uneven work per workgroup (a filter that
empties most tiles), every kernel family that publishes records, and several contexts at once.  Every result bit exact vs the oracle."""
import threading

import numpy as np
import pytest

from oracle import oracle
from pinot_amd import query as Q
from pinot_amd import segment as S
import helpers as H

pytestmark = pytest.mark.gpu


def _segment(n, seed):
    rng = np.random.default_rng(seed)
    v, _, _ = H.random_dict_column(rng, "v", n, 60000)                    # irregular dictionary: scan_hist_kernel sums it
    a = S.Column.synthetic_uniform("a", n, (np.arange(3000, dtype=np.int64) * 11 + 5).astype(np.int32), seed=seed + 1)   # affine: scan_private_kernel
    ids = (np.arange(n, dtype=np.int64) * 7919 % 1000).astype(np.int32)
    ids[: n // 2] = 999                                                    # the first half of the segment matches nothing below: uneven workgroups
    f = S.Column.from_dict_ids("f", np.arange(1000, dtype=np.int32), ids)
    p = S.Column.synthetic_uniform("p", n, np.arange(16, dtype=np.int32), seed=seed + 2)      # 4 bits: scan_narrow_kernel
    raw = S.Column.raw("r", S.synthetic_dict_ids(seed + 3, 0, n, 1 << 20))
    rawd = S.Column.raw_typed("d", np.arange(n, dtype=np.float64) * 0.5 - 7.0)
    return S.SegmentData("fold_%d" % n, n, [v, a, f, p, raw, rawd])


def _specs(seg):
    fl = lambda t: Q.leaf(Q.Pred.dict_range(2, 0, t))
    specs = [
        Q.QuerySpec([(Q.SUM, 1), (Q.COUNT, -1)], filter=fl(100)),                                   # scan_private_kernel
        Q.QuerySpec([(Q.SUM, 1), (Q.MIN, 1), (Q.MAX, 0), (Q.AVG, 1)], filter=fl(7)),                # two aggregated columns, sparse matches
        Q.QuerySpec([(Q.SUM, 0), (Q.COUNT, -1)], filter=fl(300)),                                   # scan_hist_kernel
        Q.QuerySpec([(Q.COUNT, -1)], filter=Q.leaf(Q.Pred.dict_range(3, 3, 4))),                    # scan_narrow_single_kernel
        Q.QuerySpec([(Q.COUNT, -1)], filter=Q.and_(Q.leaf(Q.Pred.dict_range(3, 2, 9)), Q.not_(Q.leaf(Q.Pred.dict_range(3, 4, 6))))),   # scan_narrow_kernel
        Q.QuerySpec([(Q.SUM, 4), (Q.MIN, 4), (Q.MAX, 4)], filter=Q.leaf(Q.Pred.raw_range(4, 1000, 900000))),      # raw INT: typed / staged kernel
        Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, 1)]),                                                   # no filter
    ]
    specs.append(Q.QuerySpec([(Q.SUM, 5), (Q.MIN, 5), (Q.MAX, 5)], filter=fl(500)))                 # raw DOUBLE: scan_private_typed_kernel, double sums
    return specs


@pytest.mark.parametrize("n", [1, 2047, 2049, 70001, 1000003, 6000011])
def test_fold_matches_the_oracle_at_every_grid_size(engine, n):
    seg = _segment(n, 100 + n % 97)
    specs = _specs(seg)
    want = [oracle.execute(seg, s) for s in specs]
    with engine.open(seg) as g:
        for rep in range(6):                      # back to back on one context: the record buffer and the counter are reused
            for s, w in zip(specs, want):
                H.assert_results_equal(g.execute(s), w)


def test_fold_on_many_contexts_at_once(engine):
    """Sixteen threads, two segments of different sizes: every pg_execute gets its own context (stream, records, counter); launches of
    different grids overlap on the device while their folding workgroups publish to different host records."""
    segs = [_segment(3000017, 5), _segment(400009, 6)]
    plans = [(seg, _specs(seg)) for seg in segs]
    wants = [[oracle.execute(seg, s) for s in specs] for seg, specs in plans]
    errors = []
    opened = [engine.open(seg) for seg in segs]
    try:
        def worker(t):
            try:
                for rep in range(12):
                    k = (t + rep) % 2
                    specs = plans[k][1]
                    i = (3 * t + rep) % len(specs)
                    H.assert_results_equal(opened[k].execute(specs[i]), wants[k][i])
            except Exception as ex:  # noqa: BLE001
                errors.append(ex)
        threads = [threading.Thread(target=worker, args=(t,)) for t in range(16)]
        [t.start() for t in threads]
        [t.join() for t in threads]
    finally:
        [g.close() for g in opened]
    assert not errors, errors[:3]
