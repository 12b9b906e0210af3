#!/usr/bin/env python3
"""Times every BASELINE.md configuration (C1, C2a, C2b, C3, C5) on one GPU and checks each result against the oracle.

Not the driver's bench (that is bench.py, C2b only): this fills the per-config table of DESIGN.md / profiles/.
Usage: python tools/bench_configs.py [--rows N] [--rows-c5 N] [--out profiles/xxx.jsonl]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402,F401

from oracle import oracle  # noqa: E402
from pinot_amd import _abi  # noqa: E402
from pinot_amd import query as Q  # noqa: E402
from pinot_amd import segment as S  # noqa: E402
from pinot_amd.engine import Engine  # noqa: E402


SETTLE = 40     # untimed launches that step through the GPU clock transient; --no-settle sets 2


def timed(gseg, spec, reps=8):
    res = _abi.pg_result()
    ms, dev, wall = [], [], []
    for i in range(reps + SETTLE):
        t0 = time.perf_counter()
        st = gseg.execute_raw(spec, res)
        t1 = time.perf_counter()
        if st != _abi.PG_OK:
            raise RuntimeError(gseg.lib.pg_last_error().decode())
        if i >= SETTLE:
            ms.append(res.dominant_kernel_ms)
            dev.append(res.device_ms)
            wall.append((t1 - t0) * 1e3)
            timed.cycles = [int(c) for c in res.profile_cycles] + [int(res.profile_waves)]
        gseg.lib.pg_result_free(C.byref(res))
    timed.step_ms = sum(wall) / len(wall)       # the whole pg_execute call on the host clock (every pass, copies and syncs)
    return sum(ms) / len(ms), min(ms), sum(dev) / len(dev)


MATCH = None


def report(out, name, n, nbytes, gseg, seg, spec, check=True):
    if MATCH is not None and not MATCH.search(name):
        return
    avg, best, dev = timed(gseg, spec)
    got = gseg.execute(spec)
    ok = None
    if check:
        t0 = time.time()
        want = oracle.execute(seg, spec)
        cpu_s = time.time() - t0
        ok = (got.stats[0] == want.stats[0] and
              all(a.sum_i64 == b.sum_i64 and a.count == b.count and a.min == b.min and a.max == b.max for a, b in zip(got.aggregations, want.aggregations)) and
              sorted(got.groups) == sorted(want.groups) and
              all(all(x.sum_i64 == y.sum_i64 and x.count == y.count and x.min == y.min and x.max == y.max for x, y in zip(got.groups[g], want.groups[g])) for g in want.groups))
    else:
        cpu_s = None
    safe = avg if avg > 0 else float("inf")      # metadata-only answers launch no kernel
    rec = {"config": name, "rows": n, "kernel_ms": avg, "kernel_ms_min": best, "device_ms_all_kernels": dev, "step_ms_host_clock": timed.step_ms, "rows_per_s": n / safe * 1e3, "algorithmic_GB": nbytes / 1e9,
           "GBps": nbytes / safe / 1e6, "frac_of_8TBps": nbytes / safe / 1e6 / 8000.0, "docs_matched": got.stats[0],
           "bit_exact_vs_oracle": ok, "oracle_rows_per_s_1core": (n / cpu_s if cpu_s else None)}
    cyc = getattr(timed, "cycles", None)
    if cyc and cyc[4] and cyc[3]:
        rec["wave_cycles"] = {"waves": cyc[4], "wait": cyc[0] // cyc[4], "filter": cyc[1] // cyc[4], "aggregate": cyc[2] // cyc[4], "loop": cyc[3] // cyc[4]}
    print(json.dumps(rec), flush=True)
    out.append(rec)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--rows-c5", type=int, default=250_000_000)
    ap.add_argument("--out", default=None)
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--no-settle", action="store_true")
    ap.add_argument("--profile-waves", action="store_true")
    ap.add_argument("--only", default="")
    ap.add_argument("--match", default="", help="regexp: run only the queries whose label matches (segments nobody needs are skipped)")
    args = ap.parse_args()
    import re
    global MATCH, SETTLE
    if args.no_settle:
        SETTLE = 2
    MATCH = re.compile(args.match) if args.match else None
    want = lambda prefix: MATCH is None or prefix in args.match
    engine = Engine(device_id=0, time_kernels=True, profile_waves=args.profile_waves)
    out = []
    check = not args.no_check
    B = lambda col: col.fwd.nbytes

    only = args.only
    # ---- C1: 10 M rows, raw int32 forward index ----
    n1 = 10_000_000 if want("C1") else 1000
    vals = S.synthetic_dict_ids(42, 0, n1, 1_000_000)
    raw = S.Column.raw("raw_i32", vals)
    seg1 = S.SegmentData("c1", n1, [raw])
    with engine.open(seg1) as g:
        report(out, "C1 COUNT(*) WHERE raw_i32 BETWEEN 1 AND 10 (10M rows, raw)", n1, 4 * n1, g, seg1, Q.QuerySpec([(Q.COUNT, -1)], filter=Q.leaf(Q.Pred.raw_range(0, 1, 10))), check)
        report(out, "C1 SUM(raw_i32) (10M rows, raw)", n1, 4 * n1, g, seg1, Q.QuerySpec([(Q.SUM, 0)]), check)
        report(out, "C1 COUNT(*) no filter (10M rows; O(1) in the reference, a24)", n1, 0, g, seg1, Q.QuerySpec([(Q.COUNT, -1)]), check)
    del seg1, raw, vals

    # ---- C2 / C3 share one segment: v (C=100000), f (C=1000), k (C=1000), b (C=65536) ----
    n = args.rows if only in ('', 'c23') else 1_000_000
    t0 = time.time()
    v = S.Column.synthetic_uniform("v", n, (np.arange(100000, dtype=np.int64) * 7 + 3).astype(np.int32), seed=1)
    f = S.Column.synthetic_uniform("f", n, np.arange(1000, dtype=np.int32), seed=2)
    k = S.Column.synthetic_uniform("k", n, np.arange(1000, dtype=np.int32) * 3, seed=3)
    a = S.Column.synthetic_uniform("a", n, (np.arange(100000, dtype=np.int64) * 5 + 1).astype(np.int32), seed=4)
    b = S.Column.synthetic_uniform("b", n, np.arange(65536, dtype=np.int32) * 2, seed=5)
    seg = S.SegmentData("c23", n, [v, f, k, a, b])
    print(json.dumps({"setup": "C2/C3 segment", "rows": n, "generate_s": time.time() - t0}), flush=True)
    with engine.open(seg) as g:
        for pct, lo, hi in ((10, 45000, 55000), (50, 25000, 75000), (90, 5000, 95000)):
            report(out, "C2a SUM(v) WHERE v BETWEEN (%d%%)" % pct, n, B(v), g, seg, Q.QuerySpec([(Q.SUM, 0)], filter=Q.leaf(Q.Pred.dict_range(0, lo, hi))), check)
        for pct, t in ((1, 10), (10, 100), (50, 500)):
            report(out, "C2b SUM(v) WHERE f < t (%d%%)" % pct, n, B(v) + B(f), g, seg, Q.QuerySpec([(Q.SUM, 0)], filter=Q.leaf(Q.Pred.dict_range(1, 0, t))), check)
        report(out, "C3 SUM(a), MAX(b) GROUP BY k", n, B(k) + B(a) + B(b), g, seg, Q.QuerySpec([(Q.SUM, 3), (Q.MAX, 4)], group_by=[2]), check)
        report(out, "C3 SUM(a), MAX(b) WHERE f < 100 GROUP BY k", n, B(k) + B(a) + B(b) + B(f), g, seg,
               Q.QuerySpec([(Q.SUM, 3), (Q.MAX, 4)], filter=Q.leaf(Q.Pred.dict_range(1, 0, 100)), group_by=[2]), check)
        # key spaces above the array-based threshold (the reference's IntMapBasedHolder range): HBM table, global atomics, device compaction
        report(out, "C6 COUNT(*) GROUP BY k, f (1M raw keys)", n, B(k) + B(f), g, seg, Q.QuerySpec([(Q.COUNT, -1)], group_by=[2, 1], num_groups_limit=2_000_000), check)
        report(out, "C6 SUM(a) GROUP BY k, f (1M raw keys)", n, B(k) + B(f) + B(a), g, seg, Q.QuerySpec([(Q.SUM, 3)], group_by=[2, 1], num_groups_limit=2_000_000), check)
        report(out, "C6 SUM(a), MAX(b) WHERE f < 100 GROUP BY k, f (100k of 1M raw keys present)", n, B(k) + B(f) + B(a) + B(b), g, seg,
               Q.QuerySpec([(Q.SUM, 3), (Q.MAX, 4)], filter=Q.leaf(Q.Pred.dict_range(1, 0, 100)), group_by=[2, 1], num_groups_limit=2_000_000), check)
        report(out, "C6 SUM(a) GROUP BY k, f LIMITED to 100000 groups (first-doc pass)", n, B(k) + B(f) + B(a), g, seg, Q.QuerySpec([(Q.SUM, 3)], group_by=[2, 1]), check)
        # 100 M raw keys (v x k): above what one scatter pass partitions -- two-level partitioning (PINOT_GPU_PARTITION_TWO_LEVEL=0: direct HBM atomics)
        report(out, "C8 COUNT(*) GROUP BY v, k (100M raw keys)", n, B(v) + B(k), g, seg, Q.QuerySpec([(Q.COUNT, -1)], group_by=[0, 2]), check)
        report(out, "C8 MAX(f) GROUP BY v, k (100M raw keys)", n, B(v) + B(k) + B(f), g, seg, Q.QuerySpec([(Q.MAX, 1)], group_by=[0, 2]), check)
        report(out, "C8 SUM(a), MAX(b) WHERE f < 100 GROUP BY v, k (100M raw keys)", n, B(v) + B(k) + B(a) + B(b) + B(f), g, seg,
               Q.QuerySpec([(Q.SUM, 3), (Q.MAX, 4)], filter=Q.leaf(Q.Pred.dict_range(1, 0, 100)), group_by=[0, 2]), check)
        report(out, "COUNT(*) WHERE f < 100", n, B(f), g, seg, Q.QuerySpec([(Q.COUNT, -1)], filter=Q.leaf(Q.Pred.dict_range(1, 0, 100))), check)
        report(out, "MIN(v), MAX(v), AVG(v) WHERE f < 100", n, B(v) + B(f), g, seg,
               Q.QuerySpec([(Q.MIN, 0), (Q.MAX, 0), (Q.AVG, 0)], filter=Q.leaf(Q.Pred.dict_range(1, 0, 100))), check)
    del seg, f, k, a, b

    # ---- C5: inverted-index AND of 3 postings -> docIds -> gather + SUM ----
    n5 = args.rows_c5 if want("C5") else 100_000
    t0 = time.time()
    cols = []
    for name, card, seed in (("p", 16, 11), ("q", 64, 12), ("r", 256, 13)):
        ids = S.synthetic_dict_ids(seed, 0, n5, card)
        cols.append(S.Column.from_dict_ids(name, np.arange(card, dtype=np.int32), ids, with_inverted=True))
    v5 = S.Column.synthetic_uniform("v", n5, (np.arange(100000, dtype=np.int64) * 7 + 3).astype(np.int32), seed=1)
    seg5 = S.SegmentData("c5", n5, cols + [v5])
    print(json.dumps({"setup": "C5 segment", "rows": n5, "generate_s": time.time() - t0,
                      "posting_bytes": [int(c.inverted.nbytes) for c in cols]}), flush=True)
    inv = lambda c, d: Q.leaf(Q.Pred.dict_range(c, d, d + 1, inverted=True))
    with engine.open(seg5) as g:
        # bytes: the three postings that are read (1/card of each index) + one docId bitmap written and read per posting
        post = sum(c.inverted.nbytes / c.cardinality for c in cols)
        bitmaps = 3 * 2 * ((n5 + 7) // 8)
        report(out, "C5 SUM(v) WHERE p=3 AND q=5 AND r=7 (inverted, sparse)", n5, post + bitmaps, g, seg5,
               Q.QuerySpec([(Q.SUM, 3)], filter=Q.and_(inv(0, 3), inv(1, 5), inv(2, 7))), check)
        report(out, "C5 COUNT(*) WHERE p=3 AND q=5 (inverted)", n5, post + bitmaps, g, seg5,
               Q.QuerySpec([(Q.COUNT, -1)], filter=Q.and_(inv(0, 3), inv(1, 5))), check)
        report(out, "C5 same filter evaluated by scanning p,q,r", n5, sum(B(c) for c in cols), g, seg5,
               Q.QuerySpec([(Q.SUM, 3)], filter=Q.and_(Q.leaf(Q.Pred.dict_range(0, 3, 4)), Q.leaf(Q.Pred.dict_range(1, 5, 6)), Q.leaf(Q.Pred.dict_range(2, 7, 8)))), check)
    # ---- C5 dense variant: C = 2 / 4 / 8 (bitmap containers; 1/64 of the docs survive the AND) ----
    if want("C5d"):
        n5d = args.rows_c5
        t0 = time.time()
        dcols = []
        for name, card, seed in (("p2", 2, 21), ("q4", 4, 22), ("r8", 8, 23)):
            ids = S.synthetic_dict_ids(seed, 0, n5d, card)
            dcols.append(S.Column.from_dict_ids(name, np.arange(card, dtype=np.int32), ids, with_inverted=True))
            del ids
        v5d = S.Column.synthetic_uniform("v", n5d, (np.arange(100000, dtype=np.int64) * 7 + 3).astype(np.int32), seed=1)
        seg5d = S.SegmentData("c5d", n5d, dcols + [v5d])
        print(json.dumps({"setup": "C5d segment", "rows": n5d, "generate_s": time.time() - t0, "posting_bytes": [int(c.inverted.nbytes) for c in dcols]}), flush=True)
        with engine.open(seg5d) as g:
            post = sum(c.inverted.nbytes / c.cardinality for c in dcols)
            bitmaps = 3 * 2 * ((n5d + 7) // 8)
            report(out, "C5d SUM(v) WHERE p2=1 AND q4=2 AND r8=5 (inverted, dense: 1/64 survive)", n5d, post + bitmaps + min(B(v5d), (n5d // 64) * 64), g, seg5d,
                   Q.QuerySpec([(Q.SUM, 3)], filter=Q.and_(inv(0, 1), inv(1, 2), inv(2, 5))), check)
            report(out, "C5d same filter evaluated by scanning p2,q4,r8", n5d, sum(B(c) for c in dcols) + B(v5d), g, seg5d,
                   Q.QuerySpec([(Q.SUM, 3)], filter=Q.and_(Q.leaf(Q.Pred.dict_range(0, 1, 2)), Q.leaf(Q.Pred.dict_range(1, 2, 3)), Q.leaf(Q.Pred.dict_range(2, 5, 6)))), check)
        del seg5d, dcols

    # ---- C7: LONG / DOUBLE metric columns (dictionary-encoded and raw), the typed paths of DESIGN.md section 4.5 ----
    if want("C7"):
        n7 = args.rows_c5
        t0 = time.time()
        lib = S.load_host_library()
        f7 = S.Column.synthetic_uniform("f", n7, np.arange(1000, dtype=np.int32), seed=2)
        base = S.Column.synthetic_uniform("x", n7, np.arange(100000, dtype=np.int32), seed=7)       # dictIds 0..99999, 17 bits

        def typed_dict(name, dict_values):
            dict_values = np.ascontiguousarray(dict_values)
            dictionary = np.zeros(dict_values.shape[0] * dict_values.dtype.itemsize, dtype=np.uint8)
            lib.ph_dict_write_fixed(dict_values.ctypes.data, int(dict_values.shape[0]), dict_values.dtype.itemsize, S._u8p(dictionary))
            return S.Column(name, _abi.PG_FWD_FIXED_BIT_DICT, base.bits, base.cardinality, base.fwd, dictionary, None, dict_values,
                            stored_type=S.stored_type_of(dict_values.dtype))
        d_dict = typed_dict("d_dict", np.sort(np.random.default_rng(1).normal(0, 1e6, 100000)))
        l_wide = typed_dict("l_wide", (np.arange(100000, dtype=np.int64) * 92233720368547 - 2 ** 62))      # needs the 8-byte dictionary
        l_narrow = typed_dict("l_narrow", (np.arange(100000, dtype=np.int64) * 1000 + 1_600_000_000_000))   # epoch millis: 31-bit range -> int32 domain
        rng7 = np.random.default_rng(3)
        d_raw = S.Column.raw_typed("d_raw", rng7.normal(0, 1e6, n7))
        l_raw = S.Column.raw_typed("l_raw", rng7.integers(-2 ** 40, 2 ** 40, n7))
        seg7 = S.SegmentData("c7", n7, [f7, d_dict, l_wide, l_narrow, d_raw, l_raw])
        print(json.dumps({"setup": "C7 segment", "rows": n7, "generate_s": time.time() - t0}), flush=True)
        flt = Q.leaf(Q.Pred.dict_range(0, 0, 100))
        with engine.open(seg7) as g:
            for ci, label in ((1, "DOUBLE dictionary (8-byte gather)"), (2, "LONG dictionary, wide (8-byte gather)"), (3, "LONG dictionary, 31-bit range (int32 domain)"),
                              (4, "raw DOUBLE"), (5, "raw LONG")):
                col = seg7.columns[ci]
                report(out, "C7 SUM(%s) WHERE f < 100 (10%%): %s" % (col.name, label), n7, B(col) + B(f7), g, seg7, Q.QuerySpec([(Q.SUM, ci)], filter=flt), check)
                report(out, "C7 SUM(%s) no filter: %s" % (col.name, label), n7, B(col), g, seg7, Q.QuerySpec([(Q.SUM, ci)]), check)
            report(out, "C7 MIN(d_raw), MAX(d_raw) WHERE f < 100: raw DOUBLE", n7, B(d_raw) + B(f7), g, seg7, Q.QuerySpec([(Q.MIN, 4), (Q.MAX, 4)], filter=flt), check)
        del seg7

    if args.out:
        with open(args.out, "w") as fh:
            for rec in out:
                fh.write(json.dumps(rec) + "\n")


if __name__ == "__main__":
    main()
