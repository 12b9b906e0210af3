#!/usr/bin/env python3
"""The small-segment regime (BASELINE.json configs[0]): the C1 scan pair and the dictionary SUM on resident segments of several sizes,
so that a query's fixed cost (launch, ramp, fold, record) and its per-row cost can be read apart.

    python tools/c1_probe.py [--sizes 2048,1000000,10000000] [--settings default,bpc4] > gpurun_out/r3/c1_probe.jsonl

One JSON line per (setting, size, query): kernel_ms / all_kernels_ms (HIP events), host wall per pg_execute untimed, checked against
the oracle for sizes up to 10 M rows.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SETTINGS = {
    "default": {},
    "bpc1": {"PINOT_GPU_BLOCKS_PER_CU": "1"},
    "bpc2": {"PINOT_GPU_BLOCKS_PER_CU": "2"},
    "bpc4": {"PINOT_GPU_BLOCKS_PER_CU": "4"},
    "bpc8": {"PINOT_GPU_BLOCKS_PER_CU": "8"},
    "fold0": {"PINOT_GPU_FOLD_FINALIZE": "0"},
    "onecounter": {"PINOT_GPU_FOLD_ONE_COUNTER": "1"},
}
KNOBS = sorted({k for s in SETTINGS.values() for k in s})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="2048,200000,1000000,10000000,40000000")
    ap.add_argument("--settings", default="default")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=50)
    args = ap.parse_args()
    import numpy as np
    from bench import v_dictionary
    from oracle import oracle
    from pinot_amd import _abi
    from pinot_amd import query as Q
    from pinot_amd import segment as S
    from pinot_amd.engine import Engine

    engine = Engine(device_id=0, time_kernels=True)
    lib = engine.lib

    def reinit(env, timed):
        for k in KNOBS:
            os.environ.pop(k, None)
        os.environ.update(env)
        cfg = _abi.pg_config(_abi.PG_ABI_VERSION, 0, 0, _abi.PG_CFG_TIME_KERNELS if timed else 0)
        _abi.check(lib, lib.pg_init(C.byref(cfg)))

    res = _abi.pg_result()
    for n in [int(x) for x in args.sizes.split(",")]:
        raw = S.Column.raw("raw_i32", S.synthetic_dict_ids(42, 0, n, 1_000_000))
        f = S.Column.synthetic_uniform("f", n, np.arange(1000, dtype=np.int32), seed=2)
        v = S.Column.synthetic_uniform("v", n, v_dictionary("affine"), seed=1)
        sd = S.SegmentData("c1", n, [raw, f, v])
        queries = [
            ("count-range", Q.QuerySpec([(Q.COUNT, -1)], filter=Q.leaf(Q.Pred.raw_range(0, 1, 10))), 4 * n),
            ("sum-raw", Q.QuerySpec([(Q.SUM, 0)]), 4 * n),
            ("dict-sum", Q.QuerySpec([(Q.SUM, 2)], filter=Q.leaf(Q.Pred.dict_range(1, 0, 100))), f.fwd.nbytes + v.fwd.nbytes),
            ("dict-count", Q.QuerySpec([(Q.COUNT, -1)], filter=Q.leaf(Q.Pred.dict_range(1, 0, 100))), f.fwd.nbytes),
        ]
        g = engine.open(sd)
        for sname in args.settings.split(","):
            for qname, spec, nbytes in queries:
                rec = {"setting": sname, "rows": n, "query": qname, "bytes": int(nbytes)}
                for timed in (True, False):
                    reinit(SETTINGS[sname], timed)
                    kernel, device, wall = [], [], []
                    for i in range(args.warmup + args.steps):
                        t0 = time.perf_counter()
                        st = g.execute_raw(spec, res)
                        t1 = time.perf_counter()
                        if st != _abi.PG_OK:
                            raise RuntimeError(lib.pg_last_error().decode())
                        if i >= args.warmup:
                            kernel.append(res.dominant_kernel_ms); device.append(res.device_ms); wall.append((t1 - t0) * 1e3)
                        kid = int(res.dominant_kernel)
                        lib.pg_result_free(C.byref(res))
                    mean = lambda x: sum(x) / len(x)
                    if timed:
                        rec.update({"kernel": _abi.KERNEL_NAMES.get(kid, ""), "kernel_us": round(mean(kernel) * 1e3, 2), "kernel_us_min": round(min(kernel) * 1e3, 2),
                                    "all_kernels_us": round(mean(device) * 1e3, 2)})
                    else:
                        rec.update({"wall_us": round(mean(wall) * 1e3, 2), "wall_us_min": round(min(wall) * 1e3, 2)})
                rec["GBps_kernel"] = round(nbytes / rec["kernel_us"] / 1e3, 1) if rec["kernel_us"] > 0 else None
                if n <= 10_000_000:
                    got = g.execute(spec)
                    want = oracle.execute_sliced(sd, spec)
                    rec["bit_exact_vs_oracle"] = bool(oracle.matches_sliced(got, want, [fn for fn, _ in spec.aggregations]) and got.stats[0] == want["docs_scanned"])
                print(json.dumps(rec), flush=True)
        g.close()


if __name__ == "__main__":
    main()
