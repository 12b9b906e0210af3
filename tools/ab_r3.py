#!/usr/bin/env python3
"""A/B of the library's environment switches on RESIDENT segments, one process: pg_init re-reads the environment, so every setting is
timed against the same HBM-resident columns on the same box (a fresh box per gpurun call moves numbers by +-3 %).

    python tools/ab_r3.py [--rows N] [--match REGEX] [--settings NAME,NAME...] > gpurun_out/r3/ab.jsonl

One JSON line per (setting, query): kernel_ms / all_kernels_ms (HIP events), host wall per pg_execute with and without event timing,
bit-exact check of every setting's result against the first setting's (and against the oracle for the first, with --check).
"""
import argparse
import ctypes as C
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SETTINGS = {
    "default": {},
    "fold0": {"PINOT_GPU_FOLD_FINALIZE": "0"},
    "fold1": {"PINOT_GPU_FOLD_FINALIZE": "1"},
    "poll1": {"PINOT_GPU_POLL_RESULT": "1"},
    "fold0_poll1": {"PINOT_GPU_FOLD_FINALIZE": "0", "PINOT_GPU_POLL_RESULT": "1"},
    "laneskip0": {"PINOT_GPU_LANE_SKIP": "0", "PINOT_GPU_SPARSE_LANES": "0"},
    "leap0": {"PINOT_GPU_LEAP2": "0"},
    "simple0": {"PINOT_GPU_SCAN_SIMPLE": "0"},
    "scansparse0": {"PINOT_GPU_SCAN_SPARSE": "0"},
    "sparse0": {"PINOT_GPU_SPARSE_LANES": "0"},
    "sparse12": {"PINOT_GPU_SPARSE_LANES": "12"},
    "sparse32": {"PINOT_GPU_SPARSE_LANES": "32"},
    "sparse48": {"PINOT_GPU_SPARSE_LANES": "48"},
    "sparse64": {"PINOT_GPU_SPARSE_LANES": "64"},
    "rep0": {"PINOT_GPU_GROUP_REPLICAS": "0"},
    "rep1": {"PINOT_GPU_GROUP_REPLICAS": "1"},
    "rep2": {"PINOT_GPU_GROUP_REPLICAS": "2"},
    "rep3": {"PINOT_GPU_GROUP_REPLICAS": "3"},
    "rep4": {"PINOT_GPU_GROUP_REPLICAS": "4"},
    "bpc8": {"PINOT_GPU_BLOCKS_PER_CU": "8"},
    "bpc16": {"PINOT_GPU_BLOCKS_PER_CU": "16"},
    "bpc32": {"PINOT_GPU_BLOCKS_PER_CU": "32"},
}
KNOBS = sorted({k for s in SETTINGS.values() for k in s})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--match", default="")
    ap.add_argument("--settings", default=",".join(SETTINGS))
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--c3", action="store_true", help="also the C3 group-by queries (k, a, b columns: three more 1 B-row columns)")
    ap.add_argument("--c5", action="store_true", help="also the C5 dense / sparse index-led queries (12 s of host-side index generation each)")
    args = ap.parse_args()
    import numpy as np
    from bench import v_dictionary
    from pinot_amd import _abi
    from pinot_amd import query as Q
    from pinot_amd import segment as S
    from pinot_amd.engine import Engine
    from tools.bench_variants import _shared

    n = args.rows
    match = re.compile(args.match) if args.match else None
    engine = Engine(device_id=0, time_kernels=True)
    lib = engine.lib

    def reinit(env, timed):
        for k in KNOBS:
            os.environ.pop(k, None)
        os.environ.update(env)
        cfg = _abi.pg_config(_abi.PG_ABI_VERSION, 0, 0, _abi.PG_CFG_TIME_KERNELS if timed else 0)
        _abi.check(lib, lib.pg_init(C.byref(cfg)))

    v = S.Column.synthetic_uniform("v", n, v_dictionary("affine"), seed=1)
    f = S.Column.synthetic_uniform("f", n, np.arange(1000, dtype=np.int32), seed=2)
    v_irr = _shared(S, v, "v_irr", v_dictionary("irregular"))
    seg = S.SegmentData("ab", n, [v, f, v_irr])
    n1 = 10_000_000
    raw = S.Column.raw("raw_i32", S.synthetic_dict_ids(42, 0, n1, 1_000_000))
    f1 = S.Column.synthetic_uniform("f", n1, np.arange(1000, dtype=np.int32), seed=2)
    v1 = S.Column.synthetic_uniform("v", n1, v_dictionary("affine"), seed=1)
    seg1 = S.SegmentData("c1", n1, [raw, f1, v1])
    fl = lambda t: Q.leaf(Q.Pred.dict_range(1, 0, t))
    queries = [
        ("C2b-10pct", seg, Q.QuerySpec([(Q.SUM, 0)], filter=fl(100))),
        ("C2b-3pct", seg, Q.QuerySpec([(Q.SUM, 0)], filter=fl(30))),
        ("C2b-1pct", seg, Q.QuerySpec([(Q.SUM, 0)], filter=fl(10))),
        ("C2b-0.1pct", seg, Q.QuerySpec([(Q.SUM, 0)], filter=fl(1))),
        ("C2b-irr-10pct", seg, Q.QuerySpec([(Q.SUM, 2)], filter=fl(100))),
        ("C2b-irr-1pct", seg, Q.QuerySpec([(Q.SUM, 2)], filter=fl(10))),
        ("AND2-count", seg, Q.QuerySpec([(Q.COUNT, -1)], filter=Q.and_(fl(100), Q.leaf(Q.Pred.dict_range(0, 0, 30000))))),
        ("AND2-sum", seg, Q.QuerySpec([(Q.SUM, 0)], filter=Q.and_(fl(300), Q.leaf(Q.Pred.dict_range(2, 10000, 60000))))),
        ("TWOCOL-sum-max", seg, Q.QuerySpec([(Q.SUM, 0), (Q.MAX, 1)], filter=fl(100))),
        ("TWOCOL-sum-sum", seg, Q.QuerySpec([(Q.SUM, 0), (Q.SUM, 1)], filter=Q.leaf(Q.Pred.dict_range(0, 0, 50000)))),
        ("C2a", seg, Q.QuerySpec([(Q.SUM, 0)], filter=Q.leaf(Q.Pred.dict_range(0, 45000, 55000)))),
        ("COUNT-filter", seg, Q.QuerySpec([(Q.COUNT, -1)], filter=fl(100))),
        ("MINMAXAVG", seg, Q.QuerySpec([(Q.MIN, 0), (Q.MAX, 0), (Q.AVG, 0)], filter=fl(100))),
        ("C1-count-range", seg1, Q.QuerySpec([(Q.COUNT, -1)], filter=Q.leaf(Q.Pred.raw_range(0, 1, 10)))),
        ("C1-sum", seg1, Q.QuerySpec([(Q.SUM, 0)])),
        ("C1-dict-sum", seg1, Q.QuerySpec([(Q.SUM, 2)], filter=fl(100))),
    ]
    if args.c3:
        k = S.Column.synthetic_uniform("k", n, np.arange(1000, dtype=np.int32) * 3, seed=3)
        a = S.Column.synthetic_uniform("a", n, (np.arange(100000, dtype=np.int64) * 5 + 1).astype(np.int32), seed=4)
        b = S.Column.synthetic_uniform("b", n, np.arange(65536, dtype=np.int32) * 2, seed=5)
        k40 = S.Column.synthetic_uniform("k40", n, np.arange(40, dtype=np.int32), seed=6)
        seg3 = S.SegmentData("c3", n, [k, a, b, f, k40])
        queries += [
            ("C3", seg3, Q.QuerySpec([(Q.SUM, 1), (Q.MAX, 2)], group_by=[0])),
            ("C3-filter", seg3, Q.QuerySpec([(Q.SUM, 1), (Q.MAX, 2)], filter=Q.leaf(Q.Pred.dict_range(3, 0, 100)), group_by=[0])),
            ("C3-count", seg3, Q.QuerySpec([(Q.COUNT, -1)], group_by=[0])),
            ("C3-minmaxavg", seg3, Q.QuerySpec([(Q.MIN, 1), (Q.MAX, 1), (Q.AVG, 2)], group_by=[0])),
            ("C3-40groups", seg3, Q.QuerySpec([(Q.SUM, 1), (Q.MAX, 2)], group_by=[4])),
            ("C3-2keys", seg3, Q.QuerySpec([(Q.SUM, 1)], group_by=[4, 0])),
        ]
    opened = {}
    segs_c5 = []
    if args.c5:
        for vid, cards, seeds, picks in (("C5-sparse", (16, 64, 256), (11, 12, 13), (3, 5, 7)), ("C5-dense", (2, 4, 8), (21, 22, 23), (1, 2, 5))):
            cols = []
            for name, card, sd in zip("pqr", cards, seeds):
                ids = S.synthetic_dict_ids(sd, 0, n, card)
                cols.append(S.Column.from_dict_ids(name, np.arange(card, dtype=np.int32), ids, with_inverted=True))
                del ids
            s5 = S.SegmentData(vid, n, cols + [v])
            inv = lambda c, d: Q.leaf(Q.Pred.dict_range(c, d, d + 1, inverted=True))
            scan = lambda c, d: Q.leaf(Q.Pred.dict_range(c, d, d + 1))
            queries.append((vid, s5, Q.QuerySpec([(Q.SUM, 3)], filter=Q.and_(inv(0, picks[0]), inv(1, picks[1]), inv(2, picks[2]))),
                            Q.QuerySpec([(Q.SUM, 3)], filter=Q.and_(scan(0, picks[0]), scan(1, picks[1]), scan(2, picks[2])))))
            queries.append((vid + "-count", s5, Q.QuerySpec([(Q.COUNT, -1)], filter=Q.and_(inv(0, picks[0]), inv(1, picks[1]))),
                            Q.QuerySpec([(Q.COUNT, -1)], filter=Q.and_(scan(0, picks[0]), scan(1, picks[1])))))
            segs_c5.append(s5)
    queries = [q for q in queries if match is None or match.search(q[0])]
    for q in queries:
        sd = q[1]
        if id(sd) not in opened:
            opened[id(sd)] = engine.open(sd)
    res = _abi.pg_result()
    first = {}
    for sname in args.settings.split(","):
        env = SETTINGS[sname]
        for q in queries:
            qname, sd, spec = q[0], q[1], q[2]
            ospec = q[3] if len(q) > 3 else spec
            g = opened[id(sd)]
            rec = {"setting": sname, "query": qname, "rows": sd.num_docs}
            for timed in (True, False):
                reinit(env, timed)
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
                    rec.update({"kernel": _abi.KERNEL_NAMES.get(kid, ""), "kernel_ms": round(mean(kernel), 5), "all_kernels_ms": round(mean(device), 5),
                                "wall_ms_timed": round(mean(wall), 5), "wall_ms_timed_min": round(min(wall), 5)})
                else:
                    rec.update({"wall_ms_untimed": round(mean(wall), 5), "wall_ms_untimed_min": round(min(wall), 5)})
            got = g.execute(spec)
            key = repr([(a.count, a.sum_i64, a.sum, a.min, a.max) for a in got.aggregations]) + repr(got.stats) + \
                repr(sorted((gid, [(a.count, a.sum_i64, a.sum, a.min, a.max) for a in vals]) for gid, vals in got.groups.items()))
            rec["same_as_first_setting"] = first.setdefault(qname, key) == key
            if args.check and sname == args.settings.split(",")[0]:
                from oracle import oracle
                want = oracle.execute_sliced(sd, ospec)
                rec["bit_exact_vs_oracle"] = bool(oracle.matches_sliced(got, want, [fn for fn, _ in spec.aggregations]) and got.stats[0] == want["docs_scanned"])
            rec["docs_matched"] = got.stats[0]
            rec["entries_in_filter"] = got.stats[1]
            rec["entries_exact"] = got.filter_entries_exact
            print(json.dumps(rec), flush=True)
    for g in opened.values():
        g.close()


if __name__ == "__main__":
    main()
