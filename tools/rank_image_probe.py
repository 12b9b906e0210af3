"""What the first GROUP BY over a raw DOUBLE / wide LONG key column costs (the dictionary + rank image of DESIGN.md section 4.3g is built
inside it), what later ones cost, and what the segment's HBM accounting says before and after.  PINOT_GPU_RANK_TRACE=1 makes the library
print the build's phases and transient allocations on stderr.
  python tools/rank_image_probe.py [--rows N] [--distinct D]    (one JSON line per column kind)"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--distinct", type=int, default=100_000)
    ap.add_argument("--kinds", default="double,long40")
    ap.add_argument("--check-rows", type=int, default=20_000_000, help="the oracle checks the answer when rows <= this")
    args = ap.parse_args()
    os.environ.setdefault("PINOT_GPU_RANK_TRACE", "1")
    import torch
    from oracle import oracle
    from pinot_amd import query as Q
    from pinot_amd import segment as S
    from pinot_amd.engine import Engine

    n = args.rows
    engine = Engine(device_id=0, time_kernels=True)
    rng = np.random.default_rng(5)
    v = S.Column.synthetic_uniform("v", n, (np.arange(100000, dtype=np.int64) * 7 + 3).astype(np.int32), seed=1)
    for kind in args.kinds.split(","):
        t0 = time.time()
        pick = S.synthetic_dict_ids(77, 0, n, args.distinct)
        if kind == "double":
            pool = np.sort(rng.normal(0, 1e6, args.distinct))
            key = S.Column.raw_typed("k", pool[pick].astype(np.float64))
        else:
            pool = np.unique(rng.integers(-(2 ** 39), 2 ** 39, args.distinct, dtype=np.int64))      # a 40-bit range: beyond the int key image
            key = S.Column.raw_typed("k", pool[pick % len(pool)].astype(np.int64))
        del pick
        seg = S.SegmentData("rank_" + kind, n, [key, v])
        gen_s = time.time() - t0
        spec = Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, 1)], group_by=[0])
        with engine.open(seg) as g:
            free0, total = torch.cuda.mem_get_info()
            before = g.device_bytes()
            t1 = time.perf_counter()
            first = g.execute(spec)
            first_ms = (time.perf_counter() - t1) * 1e3
            after = g.device_bytes()
            free1, _ = torch.cuda.mem_get_info()
            later = []
            for _ in range(5):
                t1 = time.perf_counter()
                r = g.execute(spec)
                later.append((time.perf_counter() - t1) * 1e3)
            exact = None
            if n <= args.check_rows:
                w = oracle.execute(seg, spec)
                exact = bool(len(w.groups) == len(first.groups) and sorted(a[0].count for a in w.groups.values()) == sorted(a[0].count for a in first.groups.values())
                             and sum(a[1].sum_i64 for a in w.groups.values()) == sum(a[1].sum_i64 for a in first.groups.values()))
            print(json.dumps({"kind": kind, "rows": n, "distinct": args.distinct, "groups": len(first.groups), "column_bytes": int(key.fwd.nbytes), "host_generate_s": gen_s,
                              "first_query_ms_incl_build": first_ms, "later_query_ms": later, "later_kernel_ms": r.device_ms, "kernel": r.dominant_kernel,
                              "segment_device_bytes_before": before, "segment_device_bytes_after": after, "kept_by_the_build_bytes": after - before,
                              "hip_free_before": free0, "hip_free_after": free1, "hip_total": total, "stream_time_of_the_column_ms_at_6TBps": key.fwd.nbytes / 6e9,
                              "counts_and_sums_equal_the_oracle": exact}), flush=True)
        del seg, key


if __name__ == "__main__":
    main()
