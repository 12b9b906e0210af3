#!/usr/bin/env python3
"""Host phases of pg_execute_batch over 64 resident 10 M-row segments (the C1x64 shapes of tools/bench_variants.py), with the
library's own trace (PINOT_GPU_BATCH_TRACE=1: lowering, enqueue, wait, kernel) beside the wall clock of the call, kernel timing on and off.

    PINOT_GPU_BATCH_TRACE=1 python tools/batch_trace_probe.py [--segments 64] [--rows 10000000] 2> trace.err
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--segments", type=int, default=64)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--shapes", default="", help="comma-separated shape names (default: all)")
    args = ap.parse_args()
    import numpy as np
    from pinot_amd import _abi
    from pinot_amd import query as Q
    from pinot_amd import segment as S
    from pinot_amd.engine import Engine

    n, nseg = args.rows, args.segments
    engine = Engine(device_id=0, time_kernels=False)
    lib = engine.lib
    segs = []
    for s in range(nseg):
        raw = S.Column.raw("raw_i32", S.synthetic_dict_ids(4200 + s, 0, n, 1_000_000))
        fcol = S.Column.synthetic_uniform("f", n, np.arange(1000, dtype=np.int32), seed=7000 + s)
        vcol = S.Column.synthetic_uniform("v", n, (np.arange(100000, dtype=np.int64) * 7 + 3 + s).astype(np.int32), seed=8000 + s)
        kcol = S.Column.synthetic_uniform("k", n, np.arange(1000, dtype=np.int32) * 3, seed=9500 + s)
        segs.append(S.SegmentData("c1_%d" % s, n, [raw, fcol, vcol, kcol]))
    opened = [engine.open(sd) for sd in segs]
    B = lambda c: int(np.asarray(c.fwd).nbytes)
    shapes = [("dict-sum", lambda: Q.QuerySpec([(Q.SUM, 2)], filter=Q.leaf(Q.Pred.dict_range(1, 0, 100))), lambda sd: B(sd.columns[1]) + B(sd.columns[2])),
              ("group-by", lambda: Q.QuerySpec([(Q.SUM, 2), (Q.MAX, 1)], group_by=[3]), lambda sd: B(sd.columns[1]) + B(sd.columns[2]) + B(sd.columns[3])),
              ("raw-count-range", lambda: Q.QuerySpec([(Q.COUNT, -1)], filter=Q.leaf(Q.Pred.raw_range(0, 1, 10))), lambda sd: B(sd.columns[0]))]
    for name, mk, nb in shapes:
        if args.shapes and name not in args.shapes.split(","):
            continue
        specs = [mk() for _ in segs]
        nbytes = sum(nb(sd) for sd in segs)
        handles = (C.c_void_p * nseg)(*[g.handle for g in opened])
        queries = (C.POINTER(_abi.pg_query) * nseg)(*[C.pointer(sp.c) for sp in specs])
        results = (_abi.pg_result * nseg)()
        statuses = (C.c_int * nseg)()
        for timed in (False, True):
            cfg = _abi.pg_config(_abi.PG_ABI_VERSION, 0, 0, _abi.PG_CFG_TIME_KERNELS if timed else 0)
            _abi.check(lib, lib.pg_init(C.byref(cfg)))
            walls, dev = [], []
            for step in range(5 + args.steps):
                sys.stderr.write("# %s timed=%d step %d\n" % (name, timed, step))
                t0 = time.perf_counter()
                st = lib.pg_execute_batch(handles, queries, nseg, results, statuses)
                w = (time.perf_counter() - t0) * 1e3
                assert st == _abi.PG_OK
                t1 = time.perf_counter()
                d = 0.0
                for i in range(nseg):
                    assert statuses[i] == _abi.PG_OK
                    d += results[i].device_ms
                    lib.pg_result_free(C.byref(results[i]))
                if step >= 5:
                    walls.append(w); dev.append(d)
            walls.sort()
            print(json.dumps({"shape": name, "timed": timed, "segments": nseg, "rows": n, "algorithmic_bytes": nbytes, "call_ms_mean": sum(walls) / len(walls), "call_ms_min": walls[0],
                              "call_ms_median": walls[len(walls) // 2], "kernel_ms": sum(dev) / len(dev) if timed else None,
                              "frac_call_mean": nbytes / (sum(walls) / len(walls)) / 1e6 / 8000.0, "frac_call_min": nbytes / walls[0] / 1e6 / 8000.0}), flush=True)
    for g in opened:
        g.close()


if __name__ == "__main__":
    main()
