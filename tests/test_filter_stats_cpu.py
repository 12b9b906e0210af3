"""CPU test of the engine's numEntriesScannedInFilter replay (pinot_amd/csrc/pg_filter_stats.h, host C++): built stand-alone through
tools/fstats/fstats_driver.cpp and compared with the oracle's restatement of the reference's iterator tree (AndDocIdSet.iterator,
AndDocIdIterator, OrDocIdIterator, NotDocIdIterator, SVScanDocIdIterator) on the golden filter and on random trees.  Two independent
implementations of the same accounting: the oracle's is pinned to the reference's 63064 in test_oracle_golden.py."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import oracle
from pinot_amd import _abi
from pinot_amd import query as Q
from pinot_amd import segment as S
import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLAN_ZERO, PLAN_PER_LEAF, PLAN_CHAIN, PLAN_REPLAY, PLAN_LEAP2 = 0, 1, 2, 3, 4


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("needs g++")
    out = str(tmp_path_factory.mktemp("fstats") / "libfstats_driver.so")
    src = os.path.join(ROOT, "tools", "fstats", "fstats_driver.cpp")
    # (-Bsymbolic: the header's inline functions are weak symbols; libpinot_gpu.so, which another test of the process may have loaded, holds
    #  copies of the same names -- this library must run ITS OWN build of the header)
    base = ["g++", "-std=c++17", "-O1", "-g", "-shared", "-fPIC", "-pthread", "-Wl,-Bsymbolic", "-Wall", "-Werror", "-o", out, src]
    subprocess.run(base, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    lib = C.CDLL(out)
    lib.fstats_replay.restype = C.c_int64
    lib.fstats_replay.argtypes = [C.POINTER(_abi.pg_query), C.c_int32, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.fstats_replay_mode.restype = C.c_int64
    lib.fstats_replay_mode.argtypes = [C.POINTER(_abi.pg_query), C.c_int32, C.POINTER(C.POINTER(C.c_uint64)), C.c_int32, C.c_int32, C.c_int32]
    lib.fstats_leap2.restype = C.c_int64
    lib.fstats_leap2.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int32]
    lib.fstats_fsm.restype = C.c_int64
    lib.fstats_fsm.argtypes = [C.POINTER(_abi.pg_query), C.c_int32, C.POINTER(C.POINTER(C.c_uint64)), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    return lib


def leaf_bitmaps(seg, spec):
    preds = spec.predicates
    keep, ptrs = [], (C.POINTER(C.c_uint64) * max(len(preds), 1))()
    for i, p in enumerate(preds):
        if p.kind in (_abi.PG_PRED_MATCH_ALL, _abi.PG_PRED_MATCH_NONE):
            continue
        words, _ = oracle.filter_bitmap(seg, Q.QuerySpec([], filter=Q.leaf(p)))
        words = np.ascontiguousarray(np.concatenate([words, np.zeros(1, dtype=np.uint64)]))
        keep.append(words)
        ptrs[i] = words.ctypes.data_as(C.POINTER(C.c_uint64))
    return keep, ptrs


def replay(lib, seg, spec):
    """(entries, plan, scan leaves): leaf docId sets from the oracle's single-leaf filter bitmaps, the walk from the engine's header."""
    preds = spec.predicates
    keep, ptrs = [], (C.POINTER(C.c_uint64) * max(len(preds), 1))()
    for i, p in enumerate(preds):
        if p.kind in (_abi.PG_PRED_MATCH_ALL, _abi.PG_PRED_MATCH_NONE):
            continue
        words, _ = oracle.filter_bitmap(seg, Q.QuerySpec([], filter=Q.leaf(p)))
        words = np.ascontiguousarray(np.concatenate([words, np.zeros(1, dtype=np.uint64)]))
        keep.append(words)
        ptrs[i] = words.ctypes.data_as(C.POINTER(C.c_uint64))
    plan, leaves = C.c_int32(), C.c_int32()
    entries = lib.fstats_replay(C.byref(spec.c), seg.num_docs, ptrs, C.byref(plan), C.byref(leaves))
    return int(entries), int(plan.value), int(leaves.value)


def test_replay_reproduces_the_reference_golden(driver):
    seg = H.golden_segment()
    spec = Q.QuerySpec(H.golden_aggregations(seg), filter=H.golden_filter_physical(seg))
    entries, plan, leaves = replay(driver, seg, spec)
    assert (entries, plan, leaves) == (63064, PLAN_REPLAY, 3)       # InnerSegmentAggregationSingleValueQueriesTest.java:56
    assert oracle.execute(seg, spec).stats[1] == 63064


def test_replay_matches_the_oracle_on_random_trees(driver):
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

    plans = set()
    for _ in range(300):
        spec = Q.QuerySpec([(Q.COUNT, -1)], filter=tree(3))
        if spec.c.num_filter_nodes > 24:
            continue
        entries, plan, leaves = replay(driver, seg, spec)
        want = oracle.execute(seg, spec)
        plans.add(plan)
        assert entries == want.stats[1], (plan, entries, want.stats)
        if plan == PLAN_PER_LEAF:
            assert entries == leaves * n            # no AND above a scan leaf: every scan leaf looks at every doc
        if plan == PLAN_ZERO:
            assert entries == 0
    assert plans >= {PLAN_ZERO, PLAN_PER_LEAF, PLAN_CHAIN, PLAN_REPLAY} and plans <= {PLAN_ZERO, PLAN_PER_LEAF, PLAN_CHAIN, PLAN_REPLAY, PLAN_LEAP2}


def test_and_of_scan_leaves_three_ways(driver):
    """The commonest replayed shape, an AND of k scan leaves: the iterator objects, the word-level loop and the parallel state machine
    (chunks of 997 docs on 4 threads, so that every chunk boundary falls somewhere else) agree with each other, with the oracle and with
    the state machine as tests/helpers.py states it."""
    rng = np.random.default_rng(99)
    for n in (1, 63, 64, 65, 1000, 20_011, 131_075):
        k = int(rng.integers(2, 5))
        cols, masks = [], []
        for c in range(k):
            card = int(rng.choice([2, 3, 10, 50]))
            col, ids, _ = H.random_dict_column(rng, "c%d" % c, n, card)
            lo = int(rng.integers(0, card)); hi = int(rng.integers(lo + 1, card + 1))
            cols.append(col); masks.append(((ids >= lo) & (ids < hi), lo, hi))
        seg = S.SegmentData("and", n, cols)
        spec = Q.QuerySpec([(Q.COUNT, -1)], filter=Q.and_(*[Q.leaf(Q.Pred.dict_range(c, lo, hi)) for c, (_, lo, hi) in enumerate(masks)]))
        if any(m.all() or not m.any() for m, _, _ in masks):
            continue                                            # a constant leaf folds away: not this shape
        keep, ptrs = leaf_bitmaps(seg, spec)
        want = H.and_leapfrog_entries([m for m, _, _ in masks])
        generic = driver.fstats_replay_mode(C.byref(spec.c), n, ptrs, 0, 0, 0)
        sequential = driver.fstats_replay_mode(C.byref(spec.c), n, ptrs, 1, 0, 1)
        parallel = driver.fstats_replay_mode(C.byref(spec.c), n, ptrs, 1, 997, 4)
        assert generic == sequential == parallel == want == oracle.execute(seg, spec).stats[1], (n, k)


def test_two_scan_leaves_as_a_carry_chain(driver):
    """Plan::kLeap2 -- `a AND b`, both scan leaves -- as the device counts it (leapfrog2_tile / leapfrog2_chain_kernel in pg_kernels.h, restated
    on the host in pg_filter_stats.h): 32-doc lanes, 2048-doc tiles summarised for either entry state, summaries chained.  Against the
    iterator walk, the oracle and the state machine of tests/helpers.py, at sizes around every boundary and at selectivities from
    nothing to everything (long runs without an event, and events in every doc)."""
    rng = np.random.default_rng(2026)
    sizes = [1, 31, 32, 33, 63, 64, 65, 2047, 2048, 2049, 4096, 70_001, 131_072, 300_007]
    for n in sizes:
        for pa, pb in ((0.5, 0.5), (0.1, 0.1), (0.9, 0.02), (0.001, 0.7), (0.02, 0.02), (1.0, 0.3), (0.3, 0.0)):
            a = rng.random(n) < pa
            b = rng.random(n) < pb
            if n > 5000:                                      # long stretches without any event, and a state handed across many tiles
                a[1000:n - 1000] &= rng.random(n - 2000) < 0.05
                b[3000:n // 2] = False
            pack = lambda m: np.ascontiguousarray(np.concatenate([np.packbits(m.astype(np.uint8), bitorder="little"), np.zeros(16, dtype=np.uint8)])[: 8 * ((n + 63) // 64 + 1)].view(np.uint64))
            wa, wb = pack(a), pack(b)
            got = driver.fstats_leap2(wa.ctypes.data_as(C.POINTER(C.c_uint64)), wb.ctypes.data_as(C.POINTER(C.c_uint64)), n)
            assert got == H.and_leapfrog_entries([a, b]), (n, pa, pb)
    # and through the query path: the plan is kLeap2, the count equals the oracle's iterator tree
    n = 70_001
    ca, ia, _ = H.random_dict_column(rng, "a", n, 20)
    cb, ib, _ = H.random_dict_column(rng, "b", n, 7)
    seg = S.SegmentData("leap2", n, [ca, cb])
    spec = Q.QuerySpec([(Q.COUNT, -1)], filter=Q.and_(Q.leaf(Q.Pred.dict_range(0, 3, 9)), Q.leaf(Q.Pred.dict_range(1, 2, 5, exclusive=True))))
    entries, plan, leaves = replay(driver, seg, spec)
    assert plan == PLAN_LEAP2 and leaves == 2
    keep, ptrs = leaf_bitmaps(seg, spec)
    assert driver.fstats_leap2(ptrs[0], ptrs[1], n) == entries == oracle.execute(seg, spec).stats[1]


def fsm(lib, seg, spec, mode):
    keep, ptrs = leaf_bitmaps(seg, spec)
    states, inputs = C.c_int32(), C.c_int32()
    return int(lib.fstats_fsm(C.byref(spec.c), seg.num_docs, ptrs, mode, C.byref(states), C.byref(inputs))), int(states.value), int(inputs.value)


def test_the_golden_filter_as_a_transducer(driver):
    """The reference's own filter -- sorted docId range AND (scan OR posting) AND scan AND scan -- compiled into the finite-state walk the
    device runs (pg_filter_fsm.h): 63064, InnerSegmentAggregationSingleValueQueriesTest.java:56, doc by doc and in the tiled form."""
    seg = H.golden_segment()
    spec = Q.QuerySpec(H.golden_aggregations(seg), filter=H.golden_filter_physical(seg))
    for mode in (0, 1):
        entries, states, inputs = fsm(driver, seg, spec, mode)
        assert entries == 63064 and inputs == 5 and 2 <= states <= 16, (entries, states, inputs)


def test_transducer_matches_the_iterator_replay_on_random_and_trees(driver):
    """Root ANDs of scan leaves, index-based leaves and ORs of such leaves over random columns: the transducer (both forms) against the
    replay of the reference's iterator objects and against the oracle.  Sizes around the lane / tile boundaries."""
    rng = np.random.default_rng(4)
    shapes_seen, compiled, small, medium = set(), 0, 0, 0
    or_of_sorted_and_scan = 0
    for n in (1, 31, 33, 2047, 2049, 4100, 20_011, 70_003):
        cols = [H.random_dict_column(rng, "a", n, 50)[0], H.random_dict_column(rng, "b", n, 7, with_inverted=True)[0],
                H.random_dict_column(rng, "c", n, 300, with_inverted=True)[0], H.random_dict_column(rng, "d", n, 3)[0],
                H.random_dict_column(rng, "e", n, 11)[0]]
        seg = S.SegmentData("fsm", n, cols)

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

        for _ in range(60):
            kids = []
            for _c in range(int(rng.integers(2, 5))):
                r = int(rng.integers(0, 10))
                if r < 5:
                    kids.append(scan_leaf())
                elif r < 7:
                    kids.append(index_leaf())
                elif r < 9:
                    kids.append(Q.or_(*[scan_leaf() if rng.integers(0, 3) else index_leaf() for _m in range(int(rng.integers(2, 4)))]))
                else:
                    # two (or three) SORTED members beside a scan member, sometimes a posting too: OrDocIdSet.iterator() merges the sorted ones
                    # into a bitmap iterator that leads the OrDocIdIterator (the advisor's shape, round 5)
                    sorted_members = []
                    for _m in range(int(rng.integers(2, 4))):
                        lo = int(rng.integers(0, n)); sorted_members.append(Q.leaf(Q.Pred.doc_range(lo, min(n - 1, lo + int(rng.integers(0, max(1, n // 3)))))))
                    extra = [Q.leaf(Q.Pred.dict_range(1, int(rng.integers(0, 5)), 7, inverted=True))] if rng.integers(0, 2) else []
                    kids.append(Q.or_(*(sorted_members + [scan_leaf()] + extra)))
                    or_of_sorted_and_scan += 1
            spec = Q.QuerySpec([(Q.COUNT, -1)], filter=Q.and_(*kids))
            if len(spec.predicates) > 8 or any(p.kind in (_abi.PG_PRED_MATCH_ALL, _abi.PG_PRED_MATCH_NONE) for p in spec.predicates):
                continue
            want = oracle.execute(seg, spec).stats[1]
            keep, ptrs = leaf_bitmaps(seg, spec)
            assert driver.fstats_replay_mode(C.byref(spec.c), n, ptrs, 0, 0, 0) == want
            seq, states, inputs = fsm(driver, seg, spec, 0)
            if seq < 0:
                continue                                         # more than 16 states (or an OR of sorted members only): the replay keeps it
            compiled += 1
            shapes_seen.add((len(kids), states))
            tiled, _, _ = fsm(driver, seg, spec, 1)
            assert seq == tiled == want, (n, states, inputs, seq, tiled, want)
            # the byte-function form (fsm_tiles_perm_kernel's arithmetic): machines of at most four states and four inputs
            perm, _, _ = fsm(driver, seg, spec, 2)
            if states <= 4 and inputs <= 4:
                small += 1
                assert perm == want, (n, states, inputs, perm, want)
            else:
                assert perm == -1
            perm8, _, _ = fsm(driver, seg, spec, 3)          # the eight-state form takes the small machines too
            if states <= 8 and inputs <= 4:
                medium += 1 if states > 4 else 0
                assert perm8 == want, (n, states, inputs, perm8, want)
            else:
                assert perm8 == -1
    assert compiled > 200 and len(shapes_seen) > 8 and small > 60 and medium > 20, (compiled, shapes_seen, small, medium)
    assert or_of_sorted_and_scan > 30


def test_an_or_of_index_based_members_only_is_not_a_leap_frogging_child(driver):
    """`posting AND scan AND (posting OR sorted OR sorted)`: OrDocIdSet.iterator() merges the two sorted members (and, in the oracle's
    reading, the bitmap member beside them) into ONE bitmap iterator (OrDocIdSet.java:62-126) -- an index-based child of the AND, and-ed
    into its bitmap before the scan leaf sees a doc.  Round 4's transducer walked it as a leap-frogging OR (it bailed out only when every
    member was sorted) and reported 27 559 entries with filter_entries_exact = 1 where the iterators count 15 468; found by round 5's
    kernel-coverage table (tools/kernel_coverage.py, tree 313).  The shape stays with the replay; shared predicates (one predicate
    behind several leaves) compile and agree."""
    rng = np.random.default_rng(12)
    n = 50_021
    cols = [H.random_dict_column(rng, "a", n, 200)[0], H.random_dict_column(rng, "x", n, 50, with_inverted=True)[0],
            H.random_dict_column(rng, "y", n, 40, with_inverted=True)[0]]
    seg = S.SegmentData("fsm_or", n, cols)
    a = Q.leaf(Q.Pred.dict_range(0, 12, 60))
    y = Q.leaf(Q.Pred.dict_set(2, [1, 4, 9, 16, 25, 36], 40, inverted=True))
    x = Q.leaf(Q.Pred.dict_range(1, 0, 7, inverted=True))
    lo, hi = Q.leaf(Q.Pred.doc_range(4808, 30822)), Q.leaf(Q.Pred.doc_range(35000, n - 1))
    for flt, compiles in ((Q.and_(y, a, Q.or_(x, hi, lo)), False), (Q.and_(y, a, Q.or_(hi, lo)), False), (Q.and_(y, a, Q.or_(x, lo)), True),
                          (Q.and_(y, a, Q.or_(x, hi, lo, Q.leaf(Q.Pred.dict_range(0, 100, 150)))), True),
                          (Q.and_(a, Q.or_(x, a), Q.or_(a, lo)), True)):
        spec = Q.QuerySpec([(Q.COUNT, -1)], filter=flt)
        want = oracle.execute(seg, spec).stats[1]
        entries, plan, _ = replay(driver, seg, spec)
        assert entries == want and plan == PLAN_REPLAY
        got, states, inputs = fsm(driver, seg, spec, 0)
        assert (got >= 0) == compiles, (got, states, inputs)
        if compiles:
            assert got == want == fsm(driver, seg, spec, 1)[0]


def test_not_children_are_episodes_of_the_transducer(driver):
    """`a AND NOT b` and its relatives: a NOT child over a scan leaf pulls its leaf with next() -- whole 256-doc batches from wherever the
    last advance() left it (NotDocIdIterator.java:45-76, SVScanDocIdIterator.java:76-112).  The batch phase does not become state: the walk
    marks where an episode of batches opens and where it closes, an episode costs episode_entries(origin, close) (pg_filter_fsm.h "NOT
    children").  The doc-by-doc walk and the device's tile structure (fsm_episode_entries_tiled: the twin of fsm_episode_tiles_kernel /
    fsm_episode_finish_kernel) against the replay of the iterator objects and the oracle, sizes around the batch, lane and tile edges;
    leaves with rare matches (episodes of many batches) and with dense ones."""
    rng = np.random.default_rng(7)
    compiled, with_not, shapes, not_ors, not_ors_compiled = 0, 0, set(), 0, 0
    for n in (1, 31, 33, 255, 257, 513, 2047, 2049, 4100, 20_011, 70_003):
        cols = [H.random_dict_column(rng, "a", n, 50)[0], H.random_dict_column(rng, "b", n, 7, with_inverted=True)[0],
                H.random_dict_column(rng, "c", n, 300, with_inverted=True)[0], H.random_dict_column(rng, "d", n, 3)[0],
                H.random_dict_column(rng, "e", n, 11)[0], H.random_dict_column(rng, "f", n, 2000)[0]]
        seg = S.SegmentData("fsm_not", n, cols)

        def scan_leaf():
            k = int(rng.integers(0, 6))
            if k == 0:
                lo = int(rng.integers(0, 40)); return Q.leaf(Q.Pred.dict_range(0, lo, lo + int(rng.integers(1, 25)), exclusive=bool(rng.integers(0, 2))))
            if k == 1:
                return Q.leaf(Q.Pred.dict_range(3, int(rng.integers(0, 2)), int(rng.integers(2, 4))))
            if k == 2:
                lo = int(rng.integers(0, 9)); return Q.leaf(Q.Pred.dict_range(4, lo, lo + int(rng.integers(1, 6))))
            if k == 3:
                lo = int(rng.integers(0, 1990)); return Q.leaf(Q.Pred.dict_range(5, lo, lo + int(rng.integers(1, 8))))          # rare: episodes of many batches
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

        for _ in range(40):
            kids, nots = [], 0
            not_ors_before = not_ors
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
                    # NOT over an OR of leaves: an episode stream per scan member (OrFilterOperator.getFalses)
                    members = [scan_leaf() if rng.integers(0, 3) else index_leaf() for _m in range(int(rng.integers(2, 4)))]
                    kids.append(Q.not_(Q.or_(*members))); nots += 1; not_ors += 1
                else:
                    kids.append(Q.not_(index_leaf()))
            spec = Q.QuerySpec([(Q.COUNT, -1)], filter=Q.and_(*kids))
            if len(spec.predicates) > 8 or any(p.kind in (_abi.PG_PRED_MATCH_ALL, _abi.PG_PRED_MATCH_NONE) for p in spec.predicates):
                continue
            want = oracle.execute(seg, spec).stats[1]
            keep, ptrs = leaf_bitmaps(seg, spec)
            assert driver.fstats_replay_mode(C.byref(spec.c), n, ptrs, 0, 0, 0) == want
            seq, states, inputs = fsm(driver, seg, spec, 0)
            if seq < 0:
                continue
            compiled += 1
            with_not += nots
            not_ors_compiled += not_ors - not_ors_before
            shapes.add((len(kids), nots, states))
            tiled, _, _ = fsm(driver, seg, spec, 1)
            assert seq == tiled == want, (n, states, inputs, seq, tiled, want)
    assert compiled > 250 and with_not > 120 and len(shapes) > 20 and not_ors_compiled > 12, (compiled, with_not, len(shapes), not_ors, not_ors_compiled)
    # many tiles (the finish kernel's sixteen ranges of 64-tile groups): 3 M docs = 1 465 tiles, episodes from one batch to thousands of docs long
    n = 3_000_017
    seg = S.SegmentData("fsm_not_big", n, [H.random_dict_column(rng, "a", n, 50)[0], H.random_dict_column(rng, "f", n, 2000)[0], H.random_dict_column(rng, "d", n, 3)[0]])
    a, d = Q.leaf(Q.Pred.dict_range(0, 3, 20)), Q.leaf(Q.Pred.dict_range(2, 1, 2))
    for b in (Q.leaf(Q.Pred.dict_range(1, 100, 101)), Q.leaf(Q.Pred.dict_range(1, 100, 130)), Q.leaf(Q.Pred.dict_range(1, 0, 1500))):
        for flt in (Q.and_(a, Q.not_(b)), Q.and_(Q.not_(b), d), Q.and_(d, Q.not_(b), a)):
            spec = Q.QuerySpec([(Q.COUNT, -1)], filter=flt)
            want = oracle.execute(seg, spec).stats[1]
            assert fsm(driver, seg, spec, 0)[0] == want == fsm(driver, seg, spec, 1)[0]
    # two NOT children over scan leaves are two episode streams of one machine (7 states for the pair alone, 15 beside a third child);
    # three of them beside a scan leaf (24-30 states after minimisation) stay with the replay; NOT over an OR of leaves is an episode stream
    # per scan member of the OR (round 6c); more than three streams and NOT over an AND stay with the replay
    n = 5000
    seg = S.SegmentData("fsm_not2", n, [H.random_dict_column(rng, "a", n, 50)[0], H.random_dict_column(rng, "d", n, 3)[0], H.random_dict_column(rng, "f", n, 2000)[0]])
    a, d, f = Q.leaf(Q.Pred.dict_range(0, 3, 20)), Q.leaf(Q.Pred.dict_range(1, 1, 2)), Q.leaf(Q.Pred.dict_range(2, 100, 130))
    for flt, compiles in ((Q.and_(Q.not_(a), Q.not_(d)), 7), (Q.and_(a, Q.not_(d), Q.not_(f)), 15), (Q.and_(Q.not_(a), Q.not_(d), Q.not_(f)), 0), (Q.and_(a, Q.not_(Q.or_(a, d))), 16),
                          (Q.and_(f, Q.not_(Q.or_(a, d))), 16), (Q.and_(Q.not_(Q.or_(f, d)), a), 16), (Q.and_(a, Q.not_(Q.or_(a, d, f))), 16), (Q.and_(a, Q.not_(Q.or_(a, d, f)), Q.not_(f)), 0),
                          (Q.and_(a, Q.not_(Q.and_(f, d))), 0),
                          (Q.and_(a, Q.not_(d)), 8), (Q.and_(Q.not_(d), a), 8)):
        spec = Q.QuerySpec([(Q.COUNT, -1)], filter=flt)
        got, states, _ = fsm(driver, seg, spec, 0)
        assert (got >= 0) == (compiles > 0)
        if compiles:
            assert states <= compiles, (states, compiles)
            assert got == oracle.execute(seg, spec).stats[1] == fsm(driver, seg, spec, 1)[0]


def test_not_over_an_or_is_an_episode_stream_per_scan_member(driver):
    """`a AND NOT (b OR c)`: OrFilterOperator.getFalses is a NotDocIdSet over the OrDocIdSet of the members' trues (OrFilterOperator.java:61-88);
    the NotDocIdIterator knows the smallest of the members' look-aheads (NotDocIdIterator.java:35-76 over OrDocIdIterator.java:52-108).
    OrDocIdIterator.advance() advances exactly the members behind the target, next() pulls exactly the members standing at the doc handed
    on -- every scan member is the three-state machine of a NOT child over a scan leaf, with an episode stream of its own.  The machine
    (doc by doc and in the device's tile structure) against the replay of the iterator objects and the oracle: members that match rarely
    (episodes of many batches), densely, index-based members beside them, the NOT child leading and following, sizes around the batch,
    lane and tile edges."""
    rng = np.random.default_rng(61)
    compiled = 0
    for n in (1, 2, 255, 256, 257, 511, 2047, 2048, 2049, 4097, 30_011, 131_075):
        cols = [H.random_dict_column(rng, "a", n, 50)[0], H.random_dict_column(rng, "b", n, 7, with_inverted=True)[0],
                H.random_dict_column(rng, "d", n, 3)[0], H.random_dict_column(rng, "f", n, 2000)[0], H.random_dict_column(rng, "g", n, 400)[0]]
        seg = S.SegmentData("fsm_not_or", n, cols)

        def scan_leaf():
            k = int(rng.integers(0, 5))
            if k == 0:
                lo = int(rng.integers(0, 40)); return Q.leaf(Q.Pred.dict_range(0, lo, lo + int(rng.integers(1, 25)), exclusive=bool(rng.integers(0, 2))))
            if k == 1:
                return Q.leaf(Q.Pred.dict_range(2, int(rng.integers(0, 2)), int(rng.integers(2, 4))))
            if k == 2:
                lo = int(rng.integers(0, 1990)); return Q.leaf(Q.Pred.dict_range(3, lo, lo + int(rng.integers(1, 8))))          # rare: episodes of many batches
            if k == 3:
                lo = int(rng.integers(0, 390)); return Q.leaf(Q.Pred.dict_range(4, lo, lo + int(rng.integers(1, 6))))
            return Q.leaf(Q.Pred.dict_set(0, sorted(set(int(x) for x in rng.integers(0, 50, size=9))), 50, exclusive=bool(rng.integers(0, 2))))

        def index_leaf():
            if rng.integers(0, 2):
                return Q.leaf(Q.Pred.dict_range(1, int(rng.integers(0, 5)), 7, inverted=True, exclusive=bool(rng.integers(0, 2))))
            lo = int(rng.integers(0, n)); return Q.leaf(Q.Pred.doc_range(lo, min(n - 1, lo + int(rng.integers(0, n)))))

        for _ in range(30):
            members = [scan_leaf() for _m in range(int(rng.integers(1, 4)))] + [index_leaf() for _m in range(int(rng.integers(0, 2)))]
            if len(members) < 2:
                members.append(scan_leaf())
            order = rng.permutation(len(members))
            not_or = Q.not_(Q.or_(*[members[i] for i in order]))
            others = [scan_leaf() if rng.integers(0, 4) else index_leaf() for _c in range(int(rng.integers(1, 3)))]
            kids = others + [not_or]
            kids = [kids[i] for i in rng.permutation(len(kids))]
            spec = Q.QuerySpec([(Q.COUNT, -1)], filter=Q.and_(*kids))
            if len(spec.predicates) > 8 or any(p.kind in (_abi.PG_PRED_MATCH_ALL, _abi.PG_PRED_MATCH_NONE) for p in spec.predicates):
                continue
            want = oracle.execute(seg, spec).stats[1]
            keep, ptrs = leaf_bitmaps(seg, spec)
            assert driver.fstats_replay_mode(C.byref(spec.c), n, ptrs, 0, 0, 0) == want
            seq, states, inputs = fsm(driver, seg, spec, 0)
            if seq < 0:
                continue
            compiled += 1
            tiled, _, _ = fsm(driver, seg, spec, 1)
            assert seq == tiled == want, (n, states, inputs, seq, tiled, want)
    assert compiled > 120, compiled
