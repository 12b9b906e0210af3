"""GPU tests of pg_execute_batch: one query over many resident segments in one call (what BaseCombineOperator does with a thread pool,
core/operator/combine/BaseCombineOperator.java:85-142).  Items whose device work is one launch of the lane-private scan kernel share ONE
launch (scan_private_batch_kernel: every item folds and publishes its own record); every other item runs as a pg_execute of its own on
the library's worker threads.  Item by item the results must be what pg_execute returns -- and what the oracle says."""
import os

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


def _heavy_segment(n=48_000_017):
    rng = np.random.default_rng(77)
    v = S.Column.synthetic_uniform("v", n, (np.arange(5000, dtype=np.int64) * 7 + 3).astype(np.int32), seed=5)
    k1 = S.Column.synthetic_uniform("k1", n, np.arange(700, dtype=np.int32) * 3 - 50, seed=8)
    f = S.Column.synthetic_uniform("f", n, np.arange(100, dtype=np.int32), seed=6)
    return S.SegmentData("heavy", n, [v, k1, f])


def test_concurrent_batches_do_not_serialise(engine):
    """Two queries of one server = two threads inside pg_execute_batch (SURVEY.md 8b Threading).  Thread A keeps the library busy with
    batches of 32 group-bys over a 48 M-row segment (items that run whole kernels inside their claim: milliseconds per call, all of it
    inside the native call -- the raw ctypes entry point, no Python-side conversion holding the GIL); thread B's small batch must come
    back in its own time, not after A's call (it did, behind a process-wide mutex), and both get the oracle's answers."""
    import ctypes as C
    import threading
    import time
    heavy, small = _heavy_segment(), _segments()[:6]
    gh, gs = engine.open(heavy), [engine.open(s) for s in small]
    try:
        # (round 5: these group-bys share ONE launch now -- 48 of them keep the call in the milliseconds the comparison below needs)
        hspecs = [Q.QuerySpec([(Q.SUM, 0), (Q.COUNT, -1)], filter=Q.leaf(Q.Pred.dict_range(2, 0, 40 + i)), group_by=[1]) for i in range(48)]
        sspecs = [_spec(seg, s, "sum") for s, seg in enumerate(small)]
        want_h = oracle.execute(heavy, hspecs[3])
        want_s = [oracle.execute(seg, sp) for seg, sp in zip(small, sspecs)]
        nh = len(hspecs)
        handles = (C.c_void_p * nh)(*[gh.handle] * nh)
        queries = (C.POINTER(_abi.pg_query) * nh)(*[C.pointer(sp.c) for sp in hspecs])

        def run_heavy(results, statuses):
            t = time.perf_counter()
            assert engine.execute_batch_raw(handles, queries, nh, results, statuses) == _abi.PG_OK
            dt = time.perf_counter() - t
            assert all(statuses[i] == _abi.PG_OK for i in range(nh))
            from pinot_amd.query import Result
            got = Result(results[3], hspecs[3])
            for i in range(nh):
                engine.lib.pg_result_free(C.byref(results[i]))
            return dt, got

        def run_small():
            t = time.perf_counter()
            got = engine.execute_batch(gs, sspecs)
            return time.perf_counter() - t, got

        for _ in range(3):
            run_small()
        solo = sorted(run_small()[0] for _ in range(15))[7]
        res0, st0 = (_abi.pg_result * nh)(), (C.c_int * nh)()
        run_heavy(res0, st0)
        heavy_alone, got_h = run_heavy(res0, st0)
        H.assert_results_equal(got_h, want_h)

        stop, errors, heavy_calls = threading.Event(), [], []

        def keep_busy():
            try:
                res, st = (_abi.pg_result * nh)(), (C.c_int * nh)()
                while not stop.is_set():
                    dt, got = run_heavy(res, st)
                    heavy_calls.append(dt)
                    H.assert_results_equal(got, want_h)
            except Exception as e:      # noqa: BLE001
                errors.append(e)

        th = threading.Thread(target=keep_busy)
        th.start()
        try:
            time.sleep(0.05)
            lat = []
            for _ in range(40):
                dt, got = run_small()
                lat.append(dt)
                for s, (status, res) in enumerate(got):
                    assert status == _abi.PG_OK
                    H.assert_results_equal(res, want_s[s])
        finally:
            stop.set()
            th.join()
        assert not errors, errors
        med = sorted(lat)[len(lat) // 2]
        busy_call = sorted(heavy_calls)[len(heavy_calls) // 2]
        print("small batch solo %.3f ms, beside the heavy batches %.3f ms; heavy call %.3f ms (alone %.3f ms), %d heavy calls meanwhile"
              % (solo * 1e3, med * 1e3, busy_call * 1e3, heavy_alone * 1e3, len(heavy_calls)))
        # behind a process-wide lock the small batch waited for the rest of a heavy call: half of one on average
        assert busy_call > 8 * solo, "the heavy batch is not heavy enough to tell"
        assert med < 0.3 * busy_call, "the small batch waited for the other query's batch: %.3f ms vs %.3f ms" % (med * 1e3, busy_call * 1e3)
    finally:
        gh.close()
        [g.close() for g in gs]


def test_batch_spanning_two_devices(engine):
    """Segment s on device s mod N inside ONE process (the deployment shape: a Pinot server is one JVM): one pg_execute_batch
    whose items live on two devices -- one launch per device, both in flight before either is waited for.
    On a one-GPU box the two device ids alias the one chip (PINOT_GPU_ALIAS_DEVICES=2: own batch contexts, streams and launches per id),
    so the per-device grouping, enqueue_deferred / finish_deferred per device and the placement all execute here too."""
    aliased = engine.device_count()[0] < 2
    if aliased:
        engine.reinit(PINOT_GPU_ALIAS_DEVICES=2)
    assert engine.device_count()[0] >= 2
    segs = _segments()
    for s, seg in enumerate(segs):
        seg.desc.device_id = s % 2
    opened = []
    try:
        opened = [engine.open(seg) for seg in segs]
        for shapes in (["sum"], ["sum", "group", "inverted", "minmax", "and2", "nofilter"]):
            specs = [_spec(seg, s, shapes[s % len(shapes)]) for s, seg in enumerate(segs)]
            for rep in range(2):
                for s, (status, res) in enumerate(engine.execute_batch(opened, specs)):
                    assert status == _abi.PG_OK
                    H.assert_results_equal(res, oracle.execute(segs[s], specs[s]))
        # a device id past the accepted ones is refused at open, not mapped somewhere
        bad = _segments()[0]
        bad.desc.device_id = engine.device_count()[0]
        with pytest.raises(_abi.PinotGpuError):
            engine.open(bad)
    finally:
        [g.close() for g in opened]
        for seg in segs:
            seg.desc.device_id = -1
        if aliased:
            engine.reinit(PINOT_GPU_ALIAS_DEVICES=None)


def _cache_specs(segs, variant):
    """New QuerySpec objects on every call (new addresses for the same -- or almost the same -- content)."""
    out = []
    for s, seg in enumerate(segs):
        card_f = seg.columns[1].cardinality
        if variant == "range-a":
            flt = Q.leaf(Q.Pred.dict_range(1, 1, min(card_f, 7)))
        elif variant == "range-b":                      # same shape, other bounds
            flt = Q.leaf(Q.Pred.dict_range(1, 2, min(card_f, 9)))
        elif variant == "set-a":
            flt = Q.leaf(Q.Pred.dict_set(1, [0, 3, 5, 11], card_f))
        elif variant == "set-b":                        # same number of set words, other bits
            flt = Q.leaf(Q.Pred.dict_set(1, [1, 3, 6, 10], card_f))
        elif variant == "sum-w":                        # other aggregated column
            out.append(Q.QuerySpec([(Q.SUM, 3), (Q.COUNT, -1)], filter=Q.leaf(Q.Pred.dict_range(1, 1, min(card_f, 7)))))
            continue
        else:
            raise ValueError(variant)
        out.append(Q.QuerySpec([(Q.SUM, 0), (Q.COUNT, -1)], filter=flt))
    return out


def test_the_plan_cache_goes_by_the_content_of_the_query(engine):
    """pg_execute_batch keeps the last few lowered queries of a segment (pg_segment.plan_cache): the same query again skips the lowering, a
    query that differs in one bound / one set bit / one column must not be served from it, and the caller's pg_query may be gone by then."""
    import gc
    segs = _segments()[:6]
    opened = [engine.open(seg) for seg in segs]
    try:
        order = ["range-a", "range-a", "range-b", "range-a", "set-a", "set-b", "set-a", "sum-w", "range-b", "range-a", "set-b", "sum-w", "sum-w", "range-a"]
        for step, variant in enumerate(order):
            specs = _cache_specs(segs, variant)
            got = engine.execute_batch(opened, specs)
            for s, (status, res) in enumerate(got):
                assert status == _abi.PG_OK, (step, variant, s)
                H.assert_results_equal(res, oracle.execute(segs[s], specs[s]))
            del specs, got
            gc.collect()
        # other engine settings: what was lowered before them is not reused (the general kernel's body instead of the lean one here)
        engine.reinit(PINOT_GPU_SCAN_SIMPLE=0)
        try:
            for variant in ("range-a", "set-a", "range-a"):
                specs = _cache_specs(segs, variant)
                for s, (status, res) in enumerate(engine.execute_batch(opened, specs)):
                    assert status == _abi.PG_OK
                    H.assert_results_equal(res, oracle.execute(segs[s], specs[s]))
                    if s == 5 and variant == "range-a":
                        assert res.dominant_kernel == "scan_private_kernel", res.dominant_kernel
        finally:
            engine.reinit(PINOT_GPU_SCAN_SIMPLE=None)
        specs = _cache_specs(segs, "range-a")
        for s, (status, res) in enumerate(engine.execute_batch(opened, specs)):
            assert status == _abi.PG_OK and (s != 5 or res.dominant_kernel == "scan_simple_kernel"), res.dominant_kernel
            H.assert_results_equal(res, oracle.execute(segs[s], specs[s]))
    finally:
        [g.close() for g in opened]


def test_the_plan_cache_follows_the_value_planes(engine):
    """A cached item reads a value plane by address: when planes are built, evicted and built again between batches the item is lowered anew."""
    import ctypes as C
    import time
    rng = np.random.default_rng(404)
    n = 400_003
    segs = []
    for s in range(3):
        irregular = lambda card: np.sort(rng.choice(np.arange(-2_000_000, 2_000_000, dtype=np.int64), size=card, replace=False)).astype(np.int32)
        x = S.Column.synthetic_uniform("x", n, irregular(12_000), seed=31 + s)              # no structure: the plane is materialised
        y = S.Column.synthetic_uniform("y", n, irregular(9_000), seed=41 + s)
        f = H.random_dict_column(rng, "f", n, 25)[0]
        segs.append(S.SegmentData("pc%d" % s, n, [x, y, f]))
    mk = lambda c: [Q.QuerySpec([(Q.SUM, c), (Q.COUNT, -1)], filter=Q.leaf(Q.Pred.dict_range(2, 3, 17))) for _ in segs]
    wants = {c: [oracle.execute(seg, sp) for seg, sp in zip(segs, mk(c))] for c in (0, 1)}
    lib = engine.lib
    previous = C.c_uint64()
    engine.reinit(PINOT_GPU_HIST=0)                          # (12 000 values fit the LDS histogram, which needs no plane and does not share launches)
    opened = [engine.open(seg) for seg in segs]
    try:
        def batch(c):
            for s, (status, res) in enumerate(engine.execute_batch(opened, mk(c))):
                assert status == _abi.PG_OK
                H.assert_results_equal(res, wants[c][s])
        for _ in range(8):                                   # before, while and after x's planes are built
            batch(0)
            time.sleep(0.02)
        one = max(g.plane_bytes() for g in opened)
        assert one > 0
        # room for the planes of ONE column of the three segments: x and y take turns, every turn evicts what the other's items read
        assert lib.pg_set_plane_budget(int(3.5 * one), C.byref(previous)) == 0
        try:
            for turn in range(4):
                for c in (1, 0):
                    for _ in range(6):
                        batch(c)
                        time.sleep(0.02)
            assert lib.pg_set_plane_budget(0, None) == 0     # and with no planes at all
            for c in (0, 1, 0):
                batch(c)
        finally:
            assert lib.pg_set_plane_budget(previous.value, None) == 0
    finally:
        [g.close() for g in opened]
        engine.reinit(PINOT_GPU_HIST=None)


def test_the_plan_cache_under_concurrent_batches(engine):
    """Several threads inside pg_execute_batch over the SAME segments, each cycling through its own order of five queries (more shapes
    than a segment remembers: entries are inserted, moved to the front and pushed out all the time while other threads hold them);
    every item of every call must carry its own query's answer."""
    import ctypes as C
    import threading
    segs = _segments()[4:10]
    opened = [engine.open(seg) for seg in segs]
    try:
        variants = ["range-a", "range-b", "set-a", "set-b", "sum-w"]
        specs = {v: _cache_specs(segs, v) for v in variants}
        want = {}
        for v in variants:
            want[v] = []
            for g, seg, sp in zip(opened, segs, specs[v]):
                single = g.execute(sp)
                H.assert_results_equal(single, oracle.execute(seg, sp))
                want[v].append((int(single.aggregations[0].sum_i64), int(single.stats[0])))
        n = len(segs)
        handles = (C.c_void_p * n)(*[g.handle for g in opened])
        errors = []

        def worker(t):
            try:
                results = (_abi.pg_result * n)()
                statuses = (C.c_int * n)()
                order = variants[t % len(variants):] + variants[:t % len(variants)]
                for it in range(int(os.environ.get("PINOT_STRESS_ITERS", "60"))):      # (a soak: PINOT_STRESS_ITERS=2000 PINOT_STRESS_THREADS=16)
                    v = order[(it * (t + 1)) % len(order)]
                    queries = (C.POINTER(_abi.pg_query) * n)(*[C.pointer(sp.c) for sp in specs[v]])
                    assert engine.execute_batch_raw(handles, queries, n, results, statuses) == _abi.PG_OK
                    for i in range(n):
                        assert statuses[i] == _abi.PG_OK, (t, it, v, i)
                        got = (int(results[i].aggregations[0].sum_i64), int(results[i].stats.num_docs_scanned))
                        engine.lib.pg_result_free(C.byref(results[i]))
                        assert got == want[v][i], (t, it, v, i, got, want[v][i])
            except BaseException as e:      # noqa: BLE001 -- reported on the main thread
                errors.append(e)

        threads = [threading.Thread(target=worker, args=(t,)) for t in range(int(os.environ.get("PINOT_STRESS_THREADS", "6")))]
        [t.start() for t in threads]
        [t.join() for t in threads]
        assert not errors, errors[:2]
    finally:
        [g.close() for g in opened]


# ---- items of scan_hist_kernel's shape share a launch too (round 5): SUM over dictionaries without structure ----
def _hist_segments(cards, n_of=lambda s: 150_000 + 37_001 * s, skew=None):
    from test_gpu_hist import irregular_dictionary
    segs = []
    for s, card in enumerate(cards):
        n = n_of(s)
        rng = np.random.default_rng(4000 + s)
        ids = rng.integers(0, card, n).astype(np.int32)
        if skew is not None and s in skew:
            ids[: n * 3 // 4] = skew[s]                  # one hot dictId: an 8- / 16-bit counter of a workgroup wraps
        v = S.Column.from_dict_ids("v", irregular_dictionary(card, 5000 + s), ids)
        f = S.Column.from_dict_ids("f", np.arange(1000, dtype=np.int32) * 3 - 7, rng.integers(0, 1000, n).astype(np.int32))
        segs.append(S.SegmentData("hb%d" % s, n, [v, f]))
    return segs


def test_irregular_dictionary_sums_share_one_launch(engine):
    """`SUM(v) WHERE f < x` over segments whose dictionaries have no structure: every item is scan_hist_kernel-shaped (the normal case of a
    real Pinot dictionary).  Items of one counter width share a launch (scan_hist_batch_kernel<8 | 16 | 32>), each with its own dictionary,
    histogram size and record; the answers are the oracle's and pg_execute's, and the library reports the body that ran."""
    cards = [100000, 90000, 155648, 100000, 50000, 77824, 1000, 38912, 3, 100000, 120000, 60000]
    segs = _hist_segments(cards)
    opened = [engine.open(seg) for seg in segs]
    try:
        for variant in range(3):
            if variant == 0:
                specs = [Q.QuerySpec([(Q.SUM, 0)], filter=Q.leaf(Q.Pred.dict_range(1, 0, 100 + 10 * s))) for s in range(len(segs))]
            elif variant == 1:
                specs = [Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, 0), (Q.MIN, 0), (Q.MAX, 0), (Q.AVG, 0)], filter=Q.leaf(Q.Pred.dict_range(1, 5, 600))) for s in range(len(segs))]
            else:       # the predicate on the summed column itself (C2a: the fused decode), and no filter at all
                specs = [Q.QuerySpec([(Q.SUM, 0), (Q.COUNT, -1)], filter=Q.leaf(Q.Pred.dict_range(0, cards[s] // 3, max(cards[s] // 3 + 1, 2 * cards[s] // 3)))) if s % 2 else
                         Q.QuerySpec([(Q.SUM, 0)]) for s in range(len(segs))]
            for rep in range(2):            # (the second call takes the items from the plan cache)
                got = engine.execute_batch(opened, specs)
                for s, (status, res) in enumerate(got):
                    assert status == _abi.PG_OK, s
                    H.assert_results_equal(res, oracle.execute(segs[s], specs[s]))
                    assert res.dominant_kernel == "scan_hist_kernel", (s, res.dominant_kernel)
                    single = opened[s].execute(specs[s])
                    assert res.stats == single.stats and [a.sum_i64 for a in res.aggregations] == [a.sum_i64 for a in single.aggregations]
    finally:
        [g.close() for g in opened]


def test_a_wrapped_counter_in_the_shared_launch_is_answered_again(engine):
    """Skewed dictIds: a plain 8- or 16-bit counter of the shared launch wraps, the item's checksum says so, and the library answers that
    item through pg_execute (guarded tier from then on) -- the other items of the launch are untouched; the next batch is exact as well."""
    cards = [100000, 100000, 50000, 100000]
    segs = _hist_segments(cards, n_of=lambda s: 400_000, skew={1: 777, 2: 5})
    opened = [engine.open(seg) for seg in segs]
    try:
        specs = [Q.QuerySpec([(Q.SUM, 0), (Q.COUNT, -1)]) for _ in segs]
        for rep in range(3):
            for s, (status, res) in enumerate(engine.execute_batch(opened, specs)):
                assert status == _abi.PG_OK, (rep, s)
                H.assert_results_equal(res, oracle.execute(segs[s], specs[s]))
    finally:
        [g.close() for g in opened]


# ---- group-bys of the LDS-table form share a launch (round 5): GroupByCombineOperator's one task per segment ----
def _group_specs(segs, variant):
    out = []
    for s, seg in enumerate(segs):
        card_f = seg.columns[1].cardinality
        flt = Q.leaf(Q.Pred.dict_range(1, s % 5, min(card_f, s % 5 + 9)))
        if variant == "sum-max":
            out.append(Q.QuerySpec([(Q.SUM, 0), (Q.MAX, 1)], group_by=[2]))
        elif variant == "filtered":
            out.append(Q.QuerySpec([(Q.SUM, 0), (Q.COUNT, -1), (Q.MIN, 0), (Q.MAX, 0), (Q.AVG, 1)], filter=flt, group_by=[2]))
        elif variant == "two-keys":
            out.append(Q.QuerySpec([(Q.COUNT, -1), (Q.MIN, 1), (Q.SUM, 1)], filter=Q.or_(flt, Q.leaf(Q.Pred.dict_range(0, 0, 7))), group_by=[2, 3]))
        elif variant == "count-only":
            out.append(Q.QuerySpec([(Q.COUNT, -1)], group_by=[3]))
        elif variant == "nothing-matches":
            out.append(Q.QuerySpec([(Q.SUM, 0)], filter=Q.leaf(Q.Pred.dict_range(1, 3, 3)), group_by=[2]))
        else:
            raise ValueError(variant)
    return out


@pytest.mark.parametrize("variant", ["sum-max", "filtered", "two-keys", "count-only", "nothing-matches"])
def test_group_bys_share_one_launch(engine, variant):
    """Every item a group-by whose table fits the LDS: one launch of group_lds_batch_kernel for the batch, each item with its own slice of
    the batch's table (all-zero before the launch, MIN / MAX as zero-identity keys), the groups kept on the host from the slots whose count
    is not zero.  Twice per variant: the second call takes the items from the plan cache and finds the table zeroed again."""
    segs = _segments()
    opened = [engine.open(seg) for seg in segs]
    try:
        for rep in range(2):
            specs = _group_specs(segs, variant)
            got = engine.execute_batch(opened, specs)
            for s, (status, res) in enumerate(got):
                assert status == _abi.PG_OK, (variant, s)
                H.assert_results_equal(res, oracle.execute(segs[s], specs[s]))
                single = opened[s].execute(specs[s])
                assert res.stats == single.stats and res.filter_entries_exact == single.filter_entries_exact
                assert sorted(res.groups) == sorted(single.groups)
    finally:
        [g.close() for g in opened]


def test_group_bys_beside_other_shapes_and_with_a_groups_limit(engine):
    """A batch that mixes group-bys with scans (a launch per kind), and a group-by whose numGroupsLimit is below its key space (that item
    keeps its own launches: the limit's first-appearance order is not the shared launch's business)."""
    segs = _segments()
    opened = [engine.open(seg) for seg in segs]
    try:
        specs = []
        for s, seg in enumerate(segs):
            if s % 3 == 0:
                specs.append(Q.QuerySpec([(Q.SUM, 0), (Q.MAX, 1)], group_by=[2]))
            elif s % 3 == 1:
                specs.append(_spec(seg, s, "sum"))
            else:
                specs.append(Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, 0)], group_by=[2, 3], num_groups_limit=5))
        for rep in range(2):
            for s, (status, res) in enumerate(engine.execute_batch(opened, specs)):
                assert status == _abi.PG_OK, s
                single = opened[s].execute(specs[s])
                assert res.stats == single.stats and sorted(res.groups) == sorted(single.groups)
                if s % 3 != 2:
                    H.assert_results_equal(res, oracle.execute(segs[s], specs[s]))
    finally:
        [g.close() for g in opened]


def test_group_bys_publish_themselves_under_concurrency(engine):
    """Round 6b: the last workgroup of a group-by item writes the item's table slice to the pinned image, zeroes it and stores the item's
    sequence number (GroupParams.host_table) -- no copy, no memset behind the launch.  Six threads, each with batches over all segments and
    single pg_execute calls (the one-item form) in turn, every thread on batch contexts of its own: every answer against the oracle, and the
    tables must come back zeroed for whoever takes the context next."""
    import threading
    segs = _segments()
    opened = [engine.open(seg) for seg in segs]
    try:
        variants = ["sum-max", "filtered", "two-keys", "count-only"]
        want = {v: [oracle.execute(seg, sp) for seg, sp in zip(segs, _group_specs(segs, v))] for v in variants}
        errors = []

        def work(t):
            try:
                for rep in range(6):
                    v = variants[(t + rep) % len(variants)]
                    specs = _group_specs(segs, v)
                    for s, (status, res) in enumerate(engine.execute_batch(opened, specs)):
                        assert status == _abi.PG_OK, (t, rep, v, s)
                        H.assert_results_equal(res, want[v][s])
                    s = (t * 7 + rep) % len(segs)
                    H.assert_results_equal(opened[s].execute(specs[s]), want[v][s])
            except Exception as e:          # noqa: BLE001 -- reported by the main thread
                errors.append(repr(e))

        threads = [threading.Thread(target=work, args=(t,)) for t in range(6)]
        [t.start() for t in threads]
        [t.join() for t in threads]
        assert errors == [], errors[:3]
    finally:
        [g.close() for g in opened]


# ---- narrow-filter COUNTs and typed aggregations share launches as well (round 5) ----
def _narrow_typed_segments(count=9):
    segs = []
    for s in range(count):
        n = 90_000 + 41_003 * s
        rng = np.random.default_rng(7000 + s)
        a = S.Column.dict_encoded("a", rng.integers(0, 200, n).astype(np.int32))          # 8 bits
        b = S.Column.dict_encoded("b", rng.integers(0, 13, n).astype(np.int32) * 5)       # 4 bits
        c = S.Column.dict_encoded("c", rng.integers(0, 3, n).astype(np.int32))            # 2 bits
        raw_l = S.Column.raw_typed("rl", rng.integers(-2 ** 40, 2 ** 40, n).astype(np.int64))
        raw_d = S.Column.raw_typed("rd", rng.normal(0, 1e6, n).astype(np.float64))
        dl = S.Column.dict_encoded_typed("dl", (rng.integers(0, 500, n).astype(np.int64) - 250) * (2 ** 33 + 7))
        segs.append(S.SegmentData("nt%d" % s, n, [a, b, c, raw_l, raw_d, dl]))
    return segs


def _narrow_typed_spec(seg, s, shape):
    ca, cb = seg.columns[0].cardinality, seg.columns[1].cardinality
    if shape == "narrow-single":
        return Q.QuerySpec([(Q.COUNT, -1)], filter=Q.leaf(Q.Pred.dict_range(0, s % 7, min(ca, 90 + s))))
    if shape == "narrow-tree":
        return Q.QuerySpec([(Q.COUNT, -1)], filter=Q.and_(Q.leaf(Q.Pred.dict_range(0, 10, min(ca, 150))),
                                                         Q.or_(Q.leaf(Q.Pred.dict_range(1, 0, min(cb, 4 + s % 3))), Q.not_(Q.leaf(Q.Pred.dict_range(2, 0, 1))))))
    flt = Q.leaf(Q.Pred.dict_range(1, s % 3, min(cb, s % 3 + 6)))
    if shape == "typed-1":
        return Q.QuerySpec([(Q.SUM, 3), (Q.MIN, 3), (Q.MAX, 3), (Q.COUNT, -1)], filter=flt)
    if shape == "typed-2":
        return Q.QuerySpec([(Q.SUM, 4), (Q.MAX, 5), (Q.AVG, 5)], filter=flt)
    if shape == "typed-3":
        return Q.QuerySpec([(Q.SUM, 3), (Q.MIN, 4), (Q.SUM, 5), (Q.MAX, 3)])
    raise ValueError(shape)


@pytest.mark.parametrize("shapes", [["narrow-single"], ["narrow-tree"], ["typed-1"], ["typed-2"], ["typed-3"],
                                    ["narrow-single", "typed-1", "narrow-tree", "typed-3", "typed-2"]])
def test_narrow_and_typed_items_share_launches(engine, shapes):
    """Items of scan_narrow_kernel's, scan_narrow_single_kernel's and scan_private_typed_kernel's shape: a launch per kind (the mixed batch
    is five launches), same answers as the oracle and as pg_execute, twice (the second call through the plan cache)."""
    segs = _narrow_typed_segments()
    if not hasattr(Q, "not_"):
        pytest.skip("no NOT constructor")
    opened = [engine.open(seg) for seg in segs]
    try:
        specs = [_narrow_typed_spec(seg, s, shapes[s % len(shapes)]) for s, seg in enumerate(segs)]
        for rep in range(2):
            for s, (status, res) in enumerate(engine.execute_batch(opened, specs)):
                assert status == _abi.PG_OK, (shapes[s % len(shapes)], s)
                H.assert_results_equal(res, oracle.execute(segs[s], specs[s]))
                single = opened[s].execute(specs[s])
                assert res.stats == single.stats and res.dominant_kernel == single.dominant_kernel
    finally:
        [g.close() for g in opened]


def _index_segments():
    """Segments with three inverted-index columns (C = 8 / 16 / 40: array, bitset and run containers turn up) beside two value columns."""
    segs = []
    for s, n in enumerate([70_001, 1, 200_003, 65_536, 1_000_003, 131_073, 333_337, 2049]):
        rng = np.random.default_rng(4000 + s)
        cols = [H.random_dict_column(rng, "p", n, 8, with_inverted=True)[0], H.random_dict_column(rng, "q", n, 16, with_inverted=True)[0],
                H.random_dict_column(rng, "r", n, 40, with_inverted=True)[0],
                S.Column.synthetic_uniform("v", n, (np.arange(3000, dtype=np.int64) * 7 + 3 + s).astype(np.int32), seed=5 * s + 1),
                S.Column.synthetic_uniform("f", n, np.arange(900, dtype=np.int32), seed=5 * s + 2)]
        segs.append(S.SegmentData("ix%d" % s, n, cols))
    return segs


@pytest.mark.parametrize("shapes", [["count"], ["gather"], ["count", "gather", "gather2", "not-eq", "dense-beside", "scan-beside"]])
def test_index_led_items_share_one_launch(engine, shapes):
    """COUNT(*) over an index-only filter and the gathered aggregation are index_and_kernel publishing the query's record: in a batch they share
    ONE launch of index_and_batch_kernel (lean_kind 12), every item with its own postings, records and pinned result -- same answers as
    pg_execute and as the oracle, twice (the second call through the plan cache); shapes that keep a launch of their own ride along."""
    inv = lambda c, d, **kw: Q.leaf(Q.Pred.dict_range(c, d, d + 1, inverted=True, **kw))

    def spec(seg, s, shape):
        if shape == "count":
            return Q.QuerySpec([(Q.COUNT, -1)], filter=Q.and_(inv(0, s % 8), inv(1, s % 16)))
        if shape == "gather":
            return Q.QuerySpec([(Q.SUM, 3)], filter=Q.and_(inv(0, 3), inv(1, 5), inv(2, 7)))
        if shape == "gather2":
            return Q.QuerySpec([(Q.SUM, 3), (Q.MAX, 4), (Q.MIN, 3), (Q.COUNT, -1)], filter=Q.and_(inv(0, 1), inv(1, 2), inv(2, s % 40)))
        if shape == "not-eq":
            return Q.QuerySpec([(Q.COUNT, -1)], filter=Q.and_(inv(0, 2), inv(1, 3, exclusive=True)))
        if shape == "dense-beside":                                   # denser than a handful per window: scan_sparse_kernel behind the AND, a launch of its own
            return Q.QuerySpec([(Q.SUM, 3)], filter=Q.and_(inv(0, 3), inv(1, 5)))
        return Q.QuerySpec([(Q.SUM, 3)], filter=Q.leaf(Q.Pred.dict_range(4, 0, 90)))
    segs = _index_segments()
    opened = [engine.open(seg) for seg in segs]
    try:
        specs = [spec(seg, s, shapes[s % len(shapes)]) for s, seg in enumerate(segs)]
        for rep in range(2):
            for s, (status, res) in enumerate(engine.execute_batch(opened, specs)):
                assert status == _abi.PG_OK, (shapes[s % len(shapes)], s)
                H.assert_results_equal(res, oracle.execute(segs[s], specs[s]))
                single = opened[s].execute(specs[s])
                assert res.stats == single.stats and res.filter_entries_exact == single.filter_entries_exact
                assert [(a.count, a.sum_i64, a.min, a.max) for a in res.aggregations] == [(a.count, a.sum_i64, a.min, a.max) for a in single.aggregations]
    finally:
        [g.close() for g in opened]


def test_items_with_dictid_sets_share_the_launch_of_their_kind():
    """IN lists ride in the batch's blob (round 6): items whose filters carry dictId-set leaves share the launch of their kind instead of
    each running on a context of its own.  tools/batch_set_probe.py holds every answer against the oracle and pg_execute (twice: the second
    call through the plan cache); PINOT_GPU_BATCH_TRACE says how many items every shared launch carried."""
    import json
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PINOT_GPU_BATCH_TRACE="1")
    proc = subprocess.run([sys.executable, os.path.join(root, "tools", "batch_set_probe.py")], capture_output=True, text=True, env=env, timeout=900)
    assert proc.returncode == 0, proc.stderr[-2000:]
    report = json.loads(proc.stdout.strip().splitlines()[-1])
    assert report["failed"] == [], report["failed"][:10]
    # per shape: the largest shared launch of each of the two calls carried (nearly) all of the items -- a 1-doc or 2049-doc segment may
    # be answered by a plan of its own
    shape, largest = None, {}
    for line in proc.stderr.splitlines():
        if line.startswith("== shape "):
            shape = line[len("== shape "):]
        m = re.search(r"deferred (?:group-by )?launch on device \d+: (\d+) items", line)
        if m and shape:
            largest[shape] = max(largest.get(shape, 0), int(m.group(1)))
    for name, items in report["shapes"].items():
        assert largest.get(name, 0) >= items - 2, (name, largest, proc.stderr[-1500:])


@pytest.mark.parametrize("bits", [1, 2, 5, 6, 9, 10, 13, 16])
def test_one_in_list_in_front_of_one_column_alone_and_in_a_batch(engine, bits):
    """`SUM(v) ... WHERE f IN (...)` / `NOT IN (...)` has scan_simple_kernel's shape: alone it takes scan_simple_set_kernel, as items of a batch
    scan_lean_batch_kernel<13> (ScanParams.lean_kind 13) -- the set's words in LDS, from the query's own buffer or from the batch's blob.
    Filter columns of 1 .. 16 bits (InPredicateEvaluatorFactory.java:158-230, DictionaryBasedInPredicateEvaluator, is what the sets restate), ragged last tiles, IN and NOT IN."""
    card = 2 if bits == 1 else (1 << bits) - (1 << (bits - 2)) + 1          # needs exactly `bits` bits
    segs, specs = [], []
    for s, n in enumerate([2047, 70001, 333337, 1000003, 4096, 1]):
        rng = np.random.default_rng(7000 + 31 * bits + s)
        v = S.Column.synthetic_uniform("v", n, (np.arange(900 + s, dtype=np.int64) * 11 - 4000).astype(np.int32), seed=5 * s + bits)
        f = S.Column.synthetic_uniform("f", n, np.arange(card, dtype=np.int32) * 3, seed=77 * s + bits)
        assert f.bits == bits
        segs.append(S.SegmentData("in%d_%d" % (bits, s), n, [v, f]))
        members = sorted(set(int(x) for x in rng.integers(0, card, max(1, card // 3))))
        flt = Q.leaf(Q.Pred.dict_set(1, members, card, exclusive=(s % 2 == 1)))
        specs.append(Q.QuerySpec([(Q.SUM, 0), (Q.COUNT, -1)] if s % 3 else [(Q.MIN, 0), (Q.MAX, 0), (Q.AVG, 0)], filter=flt))
    opened = [engine.open(seg) for seg in segs]
    try:
        wants = [oracle.execute(seg, spec) for seg, spec in zip(segs, specs)]
        for g, spec, want in zip(opened, specs, wants):
            H.assert_results_equal(g.execute(spec), want)
        for rep in range(2):
            for s, (status, res) in enumerate(engine.execute_batch(opened, specs)):
                assert status == _abi.PG_OK, (bits, s)
                H.assert_results_equal(res, wants[s])
    finally:
        [g.close() for g in opened]
