#!/usr/bin/env python3
"""Where a batch launch's time goes: the headline query (SUM(v) WHERE f < 100) on ONE resident 1 B-row segment, run (a) alone through
pg_execute and (b) as k items of ONE pg_execute_batch launch (the same segment k times) for k = 1, 2, 4, 8, at several block budgets.

    python tools/batch_probe.py [--rows N]
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    import numpy as np
    from bench import v_dictionary
    from pinot_amd import _abi
    from pinot_amd import query as Q
    from pinot_amd import segment as S
    from pinot_amd.engine import Engine

    n = args.rows
    engine = Engine(device_id=0, time_kernels=True)
    lib = engine.lib
    v = S.Column.synthetic_uniform("v", n, v_dictionary("affine"), seed=1)
    f = S.Column.synthetic_uniform("f", n, np.arange(1000, dtype=np.int32), seed=2)
    seg = engine.open(S.SegmentData("probe", n, [v, f]))
    spec = Q.QuerySpec([(Q.SUM, 0)], filter=Q.leaf(Q.Pred.dict_range(1, 0, 100)))
    res = _abi.pg_result()

    def single():
        _abi.check(lib, lib.pg_execute(seg.handle, C.byref(spec.c), C.byref(res)))
        ms = res.dominant_kernel_ms
        lib.pg_result_free(C.byref(res))
        return ms

    for _ in range(5):
        single()
    print("single launch: kernel %.4f ms" % (sum(single() for _ in range(args.steps)) / args.steps), flush=True)
    for bbc in (None, "2", "8", "16"):
        engine.reinit(PINOT_GPU_BATCH_BLOCKS_PER_CU=bbc)
        for k in (1, 2, 4, 8):
            handles = (C.c_void_p * k)(*[seg.handle] * k)
            queries = (C.POINTER(_abi.pg_query) * k)(*[C.pointer(spec.c) for _ in range(k)])
            bres = (_abi.pg_result * k)()
            bst = (C.c_int * k)()

            def batch():
                if engine.execute_batch_raw(handles, queries, k, bres, bst) != _abi.PG_OK:
                    raise RuntimeError(lib.pg_last_error().decode())
                ms = bres[0].device_ms
                for i in range(k):
                    lib.pg_result_free(C.byref(bres[i]))
                return ms

            for _ in range(3):
                batch()
            ms = sum(batch() for _ in range(args.steps)) / args.steps
            print("batch blocks/CU %-4s items %d: kernel %.4f ms = %.4f ms per item" % (bbc or "4", k, ms, ms / k), flush=True)
    engine.reinit(PINOT_GPU_BATCH_BLOCKS_PER_CU=None)


if __name__ == "__main__":
    main()
