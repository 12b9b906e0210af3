"""GPU tests of numEntriesScannedInFilter for filters that leap-frog (AndDocIdIterator over scan-based children).

`a AND b` with two scan leaves is counted ON THE DEVICE (Plan::kLeap2: leapfrog2_tile in the lane-private kernels + leapfrog2_chain_kernel),
at any segment size and with no host pass; the host replay (pg_filter_stats.h) stays for every other leap-frogging shape up to
PINOT_GPU_EXACT_FILTER_STATS_DOCS docs.  The first test turns the replay OFF, so an exact count can only have come from the device."""
import numpy as np
import pytest

from oracle import oracle
from pinot_amd import _abi
from pinot_amd import query as Q
from pinot_amd import segment as S
import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture
def engine_without_replay(monkeypatch):
    """pg_init re-reads the environment: the same process-wide engine, with the host replay of leap-frogging filters disabled."""
    import torch  # noqa: F401
    from pinot_amd.engine import Engine
    monkeypatch.setenv("PINOT_GPU_EXACT_FILTER_STATS_DOCS", "0")
    eng = Engine(device_id=0, time_kernels=True)
    yield eng
    monkeypatch.delenv("PINOT_GPU_EXACT_FILTER_STATS_DOCS")
    Engine(device_id=0, time_kernels=True)


@pytest.mark.parametrize("n", [1, 33, 2047, 2049, 70001, 1000003, 5000011])
def test_two_scan_leaves_are_counted_on_the_device(engine_without_replay, n):
    rng = np.random.default_rng(n)
    ca, ia, _ = H.random_dict_column(rng, "a", n, 100)
    cb, ib, _ = H.random_dict_column(rng, "b", n, 10)
    if n > 5000:                                  # a long stretch in which only one leaf ever matches, then one in which neither does
        ib[n // 5:n // 2] = 9
        ia[n // 2:(3 * n) // 4] = 99
        cb = S.Column.from_dict_ids("b", np.arange(10, dtype=np.int32) * 3, ib)
        ca = S.Column.from_dict_ids("a", np.arange(100, dtype=np.int32) * 5 - 7, ia)
    v, _, _ = H.random_dict_column(rng, "v", n, 40000)                                           # irregular: SUM runs in scan_hist_kernel
    w = S.Column.synthetic_uniform("w", n, (np.arange(2000, dtype=np.int64) * 3 + 1).astype(np.int32), seed=7)   # affine: scan_private_kernel
    k = S.Column.synthetic_uniform("k", n, np.arange(50, dtype=np.int32), seed=8)
    seg = S.SegmentData("leap2_%d" % n, n, [ca, cb, v, w, k])
    filters = [Q.and_(Q.leaf(Q.Pred.dict_range(0, 0, 10)), Q.leaf(Q.Pred.dict_range(1, 0, 3))),                 # 10 % AND 30 %
               Q.and_(Q.leaf(Q.Pred.dict_range(0, 5, 95)), Q.leaf(Q.Pred.dict_range(1, 8, 9, exclusive=True))),  # 90 % AND NOT 10 %
               Q.and_(Q.leaf(Q.Pred.dict_range(1, 9, 10)), Q.leaf(Q.Pred.dict_set(0, [1, 50, 99], 100)))]        # EQ AND IN
    with engine_without_replay.open(seg) as g:
        for flt in filters:
            for aggs, group_by in (([(Q.COUNT, -1)], []), ([(Q.SUM, 3), (Q.MAX, 3)], []), ([(Q.SUM, 2)], []), ([(Q.SUM, 3), (Q.COUNT, -1)], [4])):
                spec = Q.QuerySpec(aggs, filter=flt, group_by=group_by)
                got = g.execute(spec)
                want = oracle.execute(seg, spec)
                H.assert_results_equal(got, want)
                assert got.filter_entries_exact, (n, aggs, group_by)
                assert got.stats[1] == want.stats[1]


def test_random_filter_trees_on_the_device(engine):
    """The trees of tests/test_filter_stats_cpu.py on the device: whatever plan the engine picks (closed forms, applyAnd counting in the
    kernels, the two-leaf carry chain, the host replay), an exact count equals the oracle's iterator tree."""
    rng = np.random.default_rng(20260921)
    n = 20_011
    cols = [H.random_dict_column(rng, "a", n, 50)[0], H.random_dict_column(rng, "b", n, 7, with_inverted=True)[0],
            H.random_dict_column(rng, "c", n, 300, with_inverted=True)[0], H.random_dict_column(rng, "d", n, 3)[0]]
    seg = S.SegmentData("fs", n, cols)

    def leaf():
        k = int(rng.integers(0, 7))
        if k == 0:
            lo = int(rng.integers(0, 40)); return Q.leaf(Q.Pred.dict_range(0, lo, lo + int(rng.integers(1, 12)), exclusive=bool(rng.integers(0, 2))))
        if k == 1:
            return Q.leaf(Q.Pred.dict_range(1, int(rng.integers(0, 6)), 7, inverted=True, exclusive=bool(rng.integers(0, 2))))
        if k == 2:
            return Q.leaf(Q.Pred.dict_set(2, sorted(set(int(x) for x in rng.integers(0, 300, size=40))), 300, inverted=bool(rng.integers(0, 2))))
        if k == 3:
            lo = int(rng.integers(0, n)); return Q.leaf(Q.Pred.doc_range(lo, min(n - 1, lo + int(rng.integers(0, n))), exclusive=bool(rng.integers(0, 4) == 0)))
        if k == 4:
            return Q.leaf(Q.Pred.dict_range(3, int(rng.integers(0, 2)), 3))
        if k == 5:
            return Q.leaf(Q.Pred.dict_set(0, sorted(set(int(x) for x in rng.integers(0, 50, size=5))), 50, exclusive=bool(rng.integers(0, 2))))
        return Q.leaf(Q.Pred.dict_range(2, 0, int(rng.integers(1, 300))))

    def tree(depth):
        k = int(rng.integers(0, 10))
        if depth == 0 or k < 3:
            return leaf()
        if k < 6:
            return Q.and_(*[tree(depth - 1) for _ in range(int(rng.integers(2, 4)))])
        if k < 9:
            return Q.or_(*[tree(depth - 1) for _ in range(int(rng.integers(2, 4)))])
        return Q.not_(tree(depth - 1))

    ran = exact = 0
    with engine.open(seg) as g:
        for _ in range(300):
            spec = Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, 0)], filter=tree(3))
            if spec.c.num_filter_nodes > 24 or g.check(spec) != 0:
                continue
            got = g.execute(spec)
            want = oracle.execute(seg, spec)
            H.assert_results_equal(got, want)
            ran += 1
            if got.filter_entries_exact:
                exact += 1
                assert got.stats[1] == want.stats[1]
    assert ran >= 150 and exact >= ran * 0.9, (ran, exact)


def _and_tree_segment(rng, n):
    cols = [H.random_dict_column(rng, "a", n, 50)[0], H.random_dict_column(rng, "b", n, 7, with_inverted=True)[0],
            H.random_dict_column(rng, "c", n, 300, with_inverted=True)[0], H.random_dict_column(rng, "d", n, 3)[0],
            H.random_dict_column(rng, "e", n, 11)[0]]
    return S.SegmentData("fsm_%d" % n, n, cols)


def _random_and_tree(rng, n):
    def scan_leaf():
        k = int(rng.integers(0, 4))
        if k == 0:
            lo = int(rng.integers(0, 40)); return Q.leaf(Q.Pred.dict_range(0, lo, lo + int(rng.integers(1, 25)), exclusive=bool(rng.integers(0, 2))))
        if k == 1:
            return Q.leaf(Q.Pred.dict_range(3, int(rng.integers(0, 2)), int(rng.integers(2, 4))))
        if k == 2:
            lo = int(rng.integers(0, 9)); return Q.leaf(Q.Pred.dict_range(4, lo, lo + int(rng.integers(1, 6))))
        return Q.leaf(Q.Pred.dict_set(0, sorted(set(int(x) for x in rng.integers(0, 50, size=12))), 50, exclusive=bool(rng.integers(0, 2))))

    def index_leaf():
        k = int(rng.integers(0, 3))
        if k == 0:
            return Q.leaf(Q.Pred.dict_range(1, int(rng.integers(0, 5)), 7, inverted=True, exclusive=bool(rng.integers(0, 2))))
        if k == 1:
            return Q.leaf(Q.Pred.dict_set(2, sorted(set(int(x) for x in rng.integers(0, 300, size=60))), 300, inverted=True))
        lo = int(rng.integers(0, n)); return Q.leaf(Q.Pred.doc_range(lo, min(n - 1, lo + int(rng.integers(0, n)))))

    kids = []
    for _c in range(int(rng.integers(2, 5))):
        r = int(rng.integers(0, 10))
        if r < 5:
            kids.append(scan_leaf())
        elif r < 7:
            kids.append(index_leaf())
        else:
            kids.append(Q.or_(*[scan_leaf() if rng.integers(0, 3) else index_leaf() for _m in range(int(rng.integers(2, 4)))]))
    return Q.and_(*kids)


@pytest.mark.parametrize("n", [1, 33, 2049, 70_003, 2_200_013])
def test_leapfrogging_ands_are_counted_on_the_device_by_the_transducer(engine_without_replay, n):
    """Root ANDs of scan leaves, index-based leaves and ORs of leaves (a AND b AND c, a AND (b OR c), the golden filter's shape) with the
    host replay switched OFF: an exact numEntriesScannedInFilter can only have come from fsm_tiles_kernel / fsm_chain_kernel / fsm_finish_kernel
    (pg_filter_fsm.h compiles the reference's iterator tree into a finite-state walk) -- equal to the oracle's iterator tree."""
    rng = np.random.default_rng(1000 + n)
    seg = _and_tree_segment(rng, n)
    exact = ran = 0
    with engine_without_replay.open(seg) as g:
        for _ in range(12 if n > 1_000_000 else 40):
            spec = Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, 0)], filter=_random_and_tree(rng, n))
            if len(spec.predicates) > 8 or spec.c.num_filter_nodes > 24 or g.check(spec) != 0:
                continue
            got = g.execute(spec)
            want = oracle.execute(seg, spec)
            H.assert_results_equal(got, want)
            ran += 1
            if got.filter_entries_exact:
                exact += 1
                assert got.stats[1] == want.stats[1]
        # three scan leaves, and a scan leaf AND an OR of two: the two shapes named by the review
        a, d, e = Q.leaf(Q.Pred.dict_range(0, 0, 10)), Q.leaf(Q.Pred.dict_range(3, 0, 1)), Q.leaf(Q.Pred.dict_range(4, 2, 6))
        for flt in (Q.and_(a, d, e), Q.and_(a, Q.or_(d, e)), Q.and_(Q.or_(d, e), a), Q.and_(a, d, e, Q.leaf(Q.Pred.dict_range(0, 5, 40)))):
            for group_by in ([], [3]):
                spec = Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, 0)], filter=flt, group_by=group_by)
                got, want = g.execute(spec), oracle.execute(seg, spec)
                H.assert_results_equal(got, want)
                assert got.filter_entries_exact and got.stats[1] == want.stats[1], (n, got.stats, want.stats)
    assert ran >= 8 and exact >= ran * 0.7, (n, ran, exact)


def test_the_golden_filter_is_counted_on_the_device(engine_without_replay):
    """InnerSegmentAggregationSingleValueQueriesTest.java:56: 63064, with the host replay off."""
    seg = H.golden_segment()
    spec = Q.QuerySpec(H.golden_aggregations(seg), filter=H.golden_filter_physical(seg))
    with engine_without_replay.open(seg) as g:
        got = g.execute(spec)
    assert got.filter_entries_exact and list(got.stats) == [6129, 63064, 24516, 30000]


def _not_tree_segment(rng, n):
    cols = [H.random_dict_column(rng, "a", n, 50)[0], H.random_dict_column(rng, "b", n, 7, with_inverted=True)[0],
            H.random_dict_column(rng, "c", n, 300, with_inverted=True)[0], H.random_dict_column(rng, "d", n, 3)[0],
            H.random_dict_column(rng, "e", n, 11)[0], H.random_dict_column(rng, "f", n, 2000)[0]]
    return S.SegmentData("fsm_not_%d" % n, n, cols)


def _random_and_tree_with_not(rng, n):
    def scan_leaf():
        k = int(rng.integers(0, 6))
        if k == 0:
            lo = int(rng.integers(0, 40)); return Q.leaf(Q.Pred.dict_range(0, lo, lo + int(rng.integers(1, 25)), exclusive=bool(rng.integers(0, 2))))
        if k == 1:
            return Q.leaf(Q.Pred.dict_range(3, int(rng.integers(0, 2)), int(rng.integers(2, 4))))
        if k == 2:
            lo = int(rng.integers(0, 9)); return Q.leaf(Q.Pred.dict_range(4, lo, lo + int(rng.integers(1, 6))))
        if k == 3:
            lo = int(rng.integers(0, 1990)); return Q.leaf(Q.Pred.dict_range(5, lo, lo + int(rng.integers(1, 8))))                # rare: episodes of many batches
        if k == 4:
            lo = int(rng.integers(0, 1000)); return Q.leaf(Q.Pred.dict_range(5, lo, lo + int(rng.integers(900, 1000)), exclusive=bool(rng.integers(0, 2))))
        return Q.leaf(Q.Pred.dict_set(0, sorted(set(int(x) for x in rng.integers(0, 50, size=12))), 50, exclusive=bool(rng.integers(0, 2))))

    def index_leaf():
        k = int(rng.integers(0, 3))
        if k == 0:
            return Q.leaf(Q.Pred.dict_range(1, int(rng.integers(0, 5)), 7, inverted=True, exclusive=bool(rng.integers(0, 2))))
        if k == 1:
            return Q.leaf(Q.Pred.dict_set(2, sorted(set(int(x) for x in rng.integers(0, 300, size=60))), 300, inverted=True))
        lo = int(rng.integers(0, n)); return Q.leaf(Q.Pred.doc_range(lo, min(n - 1, lo + int(rng.integers(0, n)))))

    kids, nots = [], 0
    for _c in range(int(rng.integers(2, 5))):
        r = int(rng.integers(0, 12))
        if r < 4:
            kids.append(scan_leaf())
        elif r < 6:
            kids.append(index_leaf())
        elif r < 8:
            kids.append(Q.or_(*[scan_leaf() if rng.integers(0, 3) else index_leaf() for _m in range(int(rng.integers(2, 4)))]))
        elif r < 10 and nots < 2:
            kids.append(Q.not_(scan_leaf())); nots += 1
        elif r < 11 and nots < 2:
            # NOT over an OR of leaves: an episode stream per scan member of the OR (OrFilterOperator.getFalses; round 6c)
            kids.append(Q.not_(Q.or_(*[scan_leaf() if rng.integers(0, 3) else index_leaf() for _m in range(int(rng.integers(2, 4)))]))); nots += 1
        else:
            kids.append(Q.not_(index_leaf()))
    if nots == 0:
        kids.append(Q.not_(scan_leaf()))
    return Q.and_(*kids)


@pytest.mark.parametrize("n", [1, 33, 257, 2049, 70_003, 2_200_013])
def test_not_children_are_counted_on_the_device(engine_without_replay, n):
    """`a AND NOT b`: NotDocIdIterator pulls its scan leaf with next() -- whole 256-doc batches from wherever the last advance() left it
    (NotDocIdIterator.java:45-76, SVScanDocIdIterator.java:76-112).  The transducer marks where an episode of batches opens and closes
    (pg_filter_fsm.h "NOT children"), fsm_chunk_states / fsm_tile_states / fsm_episode_tiles / fsm_episode_finish pair them.  Host replay
    OFF: an exact numEntriesScannedInFilter can only have come from those kernels; equal to the oracle's iterator objects."""
    rng = np.random.default_rng(2000 + n)
    seg = _not_tree_segment(rng, n)
    exact = ran = 0
    with engine_without_replay.open(seg) as g:
        for _ in range(10 if n > 1_000_000 else 40):
            spec = Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, 0)], filter=_random_and_tree_with_not(rng, n))
            if len(spec.predicates) > 8 or spec.c.num_filter_nodes > 24 or g.check(spec) != 0:
                continue
            got = g.execute(spec)
            want = oracle.execute(seg, spec)
            H.assert_results_equal(got, want)
            ran += 1
            if got.filter_entries_exact:
                exact += 1
                assert got.stats[1] == want.stats[1], (n, got.stats, want.stats)
        a, d, e = Q.leaf(Q.Pred.dict_range(0, 0, 10)), Q.leaf(Q.Pred.dict_range(3, 0, 1)), Q.leaf(Q.Pred.dict_range(4, 2, 6))
        rare = Q.leaf(Q.Pred.dict_range(5, 100, 102))
        posting = Q.leaf(Q.Pred.dict_range(1, 0, 3, inverted=True))
        for flt in (Q.and_(a, Q.not_(d)), Q.and_(Q.not_(d), a), Q.and_(a, Q.not_(rare)), Q.and_(Q.not_(rare), e), Q.and_(a, Q.not_(d), e), Q.and_(posting, Q.not_(rare)),
                    Q.and_(posting, a, Q.not_(e)), Q.and_(a, Q.not_(posting)), Q.and_(Q.or_(a, d), Q.not_(rare)), Q.and_(a, Q.not_(rare), Q.not_(posting)),
                    # two NOT children over scan leaves: an episode stream each (7 states: fsm_episode_ranges_kernel twice; 15 states: fsm_episode_tiles_kernel twice)
                    Q.and_(Q.not_(a), Q.not_(rare)), Q.and_(Q.not_(d), Q.not_(e)), Q.and_(a, Q.not_(d), Q.not_(rare)), Q.and_(posting, Q.not_(rare), Q.not_(e)),
                    Q.and_(posting, a, Q.not_(d), Q.not_(rare)),
                    # NOT over an OR of leaves (round 6c): every scan member of the OR is an episode stream of its own
                    Q.and_(a, Q.not_(Q.or_(d, rare))), Q.and_(Q.not_(Q.or_(rare, e)), a), Q.and_(posting, Q.not_(Q.or_(d, rare))), Q.and_(a, Q.not_(Q.or_(posting, rare))),
                    Q.and_(e, Q.not_(Q.or_(posting, d))), Q.and_(Q.not_(Q.or_(posting, rare)), Q.not_(e))):
            # (three scan members under the NOT beside a scan leaf, or two beside another NOT child, reach more than 16 states: the upper bound here)
            for group_by in ([], [3]):
                spec = Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, 0)], filter=flt, group_by=group_by)
                got, want = g.execute(spec), oracle.execute(seg, spec)
                H.assert_results_equal(got, want)
                assert got.filter_entries_exact and got.stats[1] == want.stats[1], (n, got.stats, want.stats)
    # (an OR beside the NOT, or two NOTs beside two more children, often need more than 16 states: those keep the upper bound here)
    assert ran >= 8 and exact >= ran * 0.3, (n, ran, exact)


def test_a_and_not_b_above_the_replay_cap(engine):
    """70 M docs (the host replay stops at 64 Mi: round 4 answered this query with an upper bound and filter_entries_exact = 0): `a AND NOT b`
    with a rarely matching b (episodes of many batches) and with a dense one, exact and equal to the oracle."""
    n = 70_000_001
    rng = np.random.default_rng(5)
    seg = S.SegmentData("not_70m", n, [S.Column.dict_encoded("a", rng.integers(0, 50, n).astype(np.int32)),
                                        S.Column.dict_encoded("f", rng.integers(0, 2000, n).astype(np.int32))])
    a = Q.leaf(Q.Pred.dict_range(0, 0, 10))
    with engine.open(seg) as g:
        for b in (Q.leaf(Q.Pred.dict_range(1, 100, 102)), Q.leaf(Q.Pred.dict_range(1, 0, 1200))):
            spec = Q.QuerySpec([(Q.COUNT, -1)], filter=Q.and_(a, Q.not_(b)))
            got, want = g.execute(spec), oracle.execute(seg, spec)
            H.assert_results_equal(got, want)
            assert got.filter_entries_exact and got.stats[1] == want.stats[1], (got.stats, want.stats)


def test_the_caller_can_take_the_upper_bound(engine):
    """PG_QUERY_STATS_UPPER_BOUND_OK (include/pinot_gpu.h): a leap-frogging filter runs nothing but the query -- same answer, same other
    statistics, numEntriesScannedInFilter = numDocs x scan leaves with filter_entries_exact = 0; filters whose count costs nothing stay exact."""
    n = 250_003
    rng = np.random.default_rng(5)
    ca, _, _ = H.random_dict_column(rng, "a", n, 100)
    cb, _, _ = H.random_dict_column(rng, "b", n, 10)
    cc, _, _ = H.random_dict_column(rng, "c", n, 40)
    v = S.Column.synthetic_uniform("v", n, (np.arange(2000, dtype=np.int64) * 3 + 1).astype(np.int32), seed=7)
    k = S.Column.synthetic_uniform("k", n, np.arange(50, dtype=np.int32), seed=8)
    seg = S.SegmentData("bound_ok", n, [ca, cb, cc, v, k])
    a, b, c = Q.leaf(Q.Pred.dict_range(0, 0, 30)), Q.leaf(Q.Pred.dict_range(1, 0, 5)), Q.leaf(Q.Pred.dict_range(2, 0, 20))
    leapfrogging = [(Q.and_(a, b), 2), (Q.and_(a, b, c), 3), (Q.and_(a, Q.or_(b, c)), 3), (Q.and_(a, Q.not_(b)), 2), (Q.and_(Q.not_(a), Q.not_(b)), 2)]
    with engine.open(seg) as g:
        for flt, leaves in leapfrogging:
            for aggs, group_by in (([(Q.COUNT, -1), (Q.SUM, 3)], []), ([(Q.SUM, 3), (Q.MAX, 3)], [4])):
                exact = g.execute(Q.QuerySpec(aggs, filter=flt, group_by=group_by))
                bound = g.execute(Q.QuerySpec(aggs, filter=flt, group_by=group_by, stats_upper_bound_ok=True))
                want = oracle.execute(seg, Q.QuerySpec(aggs, filter=flt, group_by=group_by))
                H.assert_results_equal(exact, want)
                assert not bound.filter_entries_exact and bound.stats[1] == leaves * n
                assert (bound.stats[0], bound.stats[2], bound.stats[3]) == (want.stats[0], want.stats[2], want.stats[3])
                assert [(x.count, x.sum_i64, x.min, x.max) for x in bound.aggregations] == [(x.count, x.sum_i64, x.min, x.max) for x in exact.aggregations]
                assert sorted(bound.groups) == sorted(exact.groups)
                for gid in exact.groups:
                    assert [(x.count, x.sum_i64, x.max) for x in bound.groups[gid]] == [(x.count, x.sum_i64, x.max) for x in exact.groups[gid]]
                if exact.filter_entries_exact:
                    assert exact.stats[1] == want.stats[1]
                    # (numDocs x scan leaves bounds AND / OR trees; a NOT child re-scans 256-doc batches and can pass it: 21.8 G entries over
                    #  1 G docs in bench.py's AND-NOT-scan)
                    has_not = any(ch.op == _abi.PG_FILTER_NOT for ch in flt.children)
                    assert has_not or want.stats[1] <= bound.stats[1]
        # nothing to skip: one scan leaf (numDocs), an OR of scan leaves (numDocs each) -- exact with or without the flag
        for flt in (a, Q.or_(a, b)):
            r = g.execute(Q.QuerySpec([(Q.COUNT, -1)], filter=flt, stats_upper_bound_ok=True))
            w = oracle.execute(seg, Q.QuerySpec([(Q.COUNT, -1)], filter=flt))
            assert r.filter_entries_exact and r.stats[1] == w.stats[1]


def test_the_upper_bound_flag_removes_every_statistics_kernel(tmp_path):
    """The same four leap-frogging queries under rocprofv3 --kernel-trace with and without the flag: without it the trace holds the
    transducer / chain kernels, with it none of them -- and the answers are the same."""
    import csv
    import glob
    import json
    import os
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    stats_kernels = ("fsm_", "leapfrog2_chain_kernel", "scan_private_fsm_kernel")
    seen, answers = {}, {}
    for mode in ("exact", "bound"):
        out_dir = str(tmp_path / mode)
        env = dict(os.environ, TMPDIR="/tmp", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        for name in list(env):
            if name.startswith("PINOT_GPU_") and name != "PINOT_GPU_LIB":
                del env[name]
        proc = subprocess.run([rocprof, "--kernel-trace", "--output-format", "csv", "-d", out_dir, "-o", "t", "--", sys.executable, os.path.join(root, "tools", "stats_flag_probe.py"), mode],
                              cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        lines = [ln for ln in proc.stdout.decode().splitlines() if ln.startswith("{")]
        assert proc.returncode == 0 and lines, proc.stderr.decode()[-2000:]
        answers[mode] = json.loads(lines[-1])
        names = set()
        for path in glob.glob(os.path.join(out_dir, "**", "*kernel_trace.csv"), recursive=True):
            with open(path, newline="") as f:
                names.update(row["Kernel_Name"] for row in csv.DictReader(f))
        assert names, "rocprofv3 left no kernel trace"
        seen[mode] = sorted(nm for nm in names if any(s in nm for s in stats_kernels))
    assert seen["exact"], "without the flag the statistic's kernels run"
    assert any("fsm_episode" in nm for nm in seen["exact"]) and any("leapfrog2_chain" in nm for nm in seen["exact"])
    assert seen["bound"] == [], seen["bound"]
    for name, e in answers["exact"].items():
        bnd = answers["bound"][name]
        assert e["exact"] and not bnd["exact"] and bnd["entries"] == bnd["scan_leaves_x_docs"]
        assert "NOT" in name or bnd["entries"] >= e["entries"]
        assert {k: e[k] for k in ("count", "sum", "docs_scanned", "post", "total")} == {k: bnd[k] for k in ("count", "sum", "docs_scanned", "post", "total")}


def test_two_leapfrogging_queries_on_one_segment_overlap(engine):
    """The transducer pass runs on the query's own context and stream (round 6): two threads with `a AND NOT b` on ONE segment are in
    flight together.  Rounds 4-5 held a segment-wide mutex across the query and its pass: two threads took twice one thread's time.
    The segment is small, so a query is launch and hand-off latency (a dozen kernels, three waits) -- what overlaps when nothing serialises."""
    import threading
    import time
    n = 1_000_003
    rng = np.random.default_rng(11)
    ca, _, _ = H.random_dict_column(rng, "a", n, 100)
    cb, _, _ = H.random_dict_column(rng, "b", n, 10)
    v = S.Column.synthetic_uniform("v", n, (np.arange(2000, dtype=np.int64) * 3 + 1).astype(np.int32), seed=7)
    seg = S.SegmentData("overlap", n, [ca, cb, v])
    spec = Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, 2)], filter=Q.and_(Q.leaf(Q.Pred.dict_range(0, 0, 30)), Q.not_(Q.leaf(Q.Pred.dict_range(1, 0, 5)))))
    want = oracle.execute(seg, spec)
    rounds = 40
    with engine.open(seg) as g:
        first = g.execute(spec)
        H.assert_results_equal(first, want)
        assert first.filter_entries_exact and first.stats[1] == want.stats[1]
        res = _abi.pg_result()

        def work(out):
            import ctypes as C
            r = _abi.pg_result()
            entries = set()
            for _ in range(rounds):
                assert g.execute_raw(spec, r) == _abi.PG_OK      # (ctypes releases the GIL inside the call)
                entries.add((int(r.stats.num_entries_scanned_in_filter), int(r.filter_entries_exact)))
                engine.lib.pg_result_free(C.byref(r))
            out.append(entries)

        def timed(threads):
            outs = []
            ts = [threading.Thread(target=work, args=(outs,)) for _ in range(threads)]
            t0 = time.perf_counter()
            [t.start() for t in ts]
            [t.join() for t in ts]
            return time.perf_counter() - t0, outs

        timed(1)                                                   # warm: contexts, scratch, clocks
        timed(2)
        one = min(timed(1)[0] for _ in range(3))
        two, outs = min((timed(2) for _ in range(3)), key=lambda x: x[0])
        for entries in outs:
            assert entries == {(want.stats[1], 1)}                 # every concurrent query counted exactly, on its own scratch
        # serialised, two threads take 2.0x one thread; in flight together they share the launch and hand-off latencies
        assert two < 1.7 * one, (one, two)
        del res
