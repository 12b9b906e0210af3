"""Round-6 A/B probe: the C5 (inverted-index AND) and C3 (group-by) shapes of bench.py on RESIDENT 1 B-row segments, every
setting of a list of environment switches timed in one process (Engine.reinit re-reads them), every answer held against the oracle once.

  python tools/ab_r6.py c5 [--rows N] [--steps K] [--out file.jsonl] [--settings "NAME=v,NAME=v;NAME=v;..."]
  python tools/ab_r6.py c3 ...
  python tools/ab_r6.py not ...      (AND-NOT-scan: the episode pass)

Two builds are compared by running the tool twice with PINOT_GPU_LIB set (the library is loaded once per process).
A setting is a comma-separated list of NAME=value (empty string: the defaults).  One JSON line per (setting, query)."""
import argparse
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def parse_settings(text):
    out = []
    for part in text.split(";"):
        env = {}
        for kv in part.split(","):
            kv = kv.strip()
            if kv:
                k, v = kv.split("=", 1)
                env[k] = v
        out.append(env)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shape", choices=["c5", "c3", "not", "typed"])
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--settings", default="")
    ap.add_argument("--out", default="")
    ap.add_argument("--match", default="")
    ap.add_argument("--no-check", action="store_true")
    args = ap.parse_args()

    from bench import Timer, v_dictionary
    from oracle import oracle
    from pinot_amd import _abi
    from pinot_amd import query as Q
    from pinot_amd import segment as S
    from pinot_amd.engine import Engine

    n = args.rows
    settings = parse_settings(args.settings)
    engine = Engine(device_id=0, time_kernels=True)
    timer = Timer(engine.lib, _abi)
    want = re.compile(args.match) if args.match else None
    sink = open(args.out, "a") if args.out else None
    names = sorted({k for env in settings for k in env})

    def emit(rec):
        line = json.dumps(rec)
        print(line, flush=True)
        if sink:
            sink.write(line + "\n")
            sink.flush()

    def sweep(label, gseg, seg, queries):
        checked = {}
        for env in settings:
            engine.reinit(**{k: env.get(k) for k in names})
            for qid, spec, ospec, nbytes in queries:
                if want and not want.search(qid):
                    continue
                t = timer.run(gseg, spec, args.steps, args.warmup)
                # the same query without PG_CFG_TIME_KERNELS: no event records on the stream, the waits the product path uses
                engine.reinit(time_kernels=False)
                untimed = timer.run(gseg, spec, args.steps, args.warmup)["step_ms_host_clock"]
                engine.reinit(time_kernels=True)
                got = gseg.execute(spec)
                if qid not in checked and not args.no_check:
                    w = oracle.execute_sliced(seg, ospec or spec)
                    checked[qid] = bool(oracle.matches_sliced(got, w, [f for f, _ in spec.aggregations]) and got.stats[0] == w["docs_scanned"])
                exact_now = None
                if qid in checked:
                    # every later setting must give the first one's answer (the first was held against the oracle)
                    answer = (got.stats[0], [(a.count, a.sum_i64, a.min, a.max) for a in got.aggregations],
                              [(gid, [(a.count, a.sum_i64, a.min, a.max) for a in got.groups[gid]]) for gid in sorted(got.groups)] if got.groups else None)
                    first = checked.setdefault(qid + "/answer", answer)
                    exact_now = checked[qid] and first == answer
                rec = {"shape": label, "query": qid, "setting": env, "rows": n, "kernel": t["kernel"], "kernel_ms": t["kernel_ms"], "all_kernels_ms": t["all_kernels_ms"],
                       "host_clock_ms": t["step_ms_host_clock"], "host_clock_untimed_ms": untimed, "bytes": nbytes, "frac_all_kernels": nbytes / t["all_kernels_ms"] / 1e6 / 8000.0 if t["all_kernels_ms"] > 0 else None,
                       "frac_host_clock": nbytes / t["step_ms_host_clock"] / 1e6 / 8000.0, "docs": got.stats[0], "entries": got.stats[1], "entries_exact": bool(got.filter_entries_exact),
                       "exact": exact_now}
                emit(rec)
        engine.reinit(**{k: None for k in names})

    B = lambda col: col.fwd.nbytes
    if args.shape == "c5":
        v = S.Column.synthetic_uniform("v", n, v_dictionary("affine"), seed=1)
        for vid, cards, seeds, picks in (("C5-sparse", (16, 64, 256), (11, 12, 13), (3, 5, 7)), ("C5-dense", (2, 4, 8), (21, 22, 23), (1, 2, 5))):
            if want and not want.search(vid):
                continue
            cols = []
            for name, card, seed in zip("pqr", cards, seeds):
                ids = S.synthetic_dict_ids(seed, 0, n, card)
                cols.append(S.Column.from_dict_ids(name, np.arange(card, dtype=np.int32), ids, with_inverted=True))
                del ids
            seg = S.SegmentData(vid, n, cols + [v])
            inv = lambda c, d: Q.leaf(Q.Pred.dict_range(c, d, d + 1, inverted=True))
            scan = lambda c, d: Q.leaf(Q.Pred.dict_range(c, d, d + 1))
            post = [int(c.inverted.nbytes / c.cardinality) for c in cols]
            survivors = int(n / (cards[0] * cards[1] * cards[2]))
            vb = min(B(v), survivors * 64)
            queries = [(vid, Q.QuerySpec([(Q.SUM, 3)], filter=Q.and_(inv(0, picks[0]), inv(1, picks[1]), inv(2, picks[2]))),
                        Q.QuerySpec([(Q.SUM, 3)], filter=Q.and_(scan(0, picks[0]), scan(1, picks[1]), scan(2, picks[2]))), sum(post) + vb),
                       (vid + "-count", Q.QuerySpec([(Q.COUNT, -1)], filter=Q.and_(inv(0, picks[0]), inv(1, picks[1]))),
                        Q.QuerySpec([(Q.COUNT, -1)], filter=Q.and_(scan(0, picks[0]), scan(1, picks[1]))), post[0] + post[1])]
            with engine.open(seg) as g:
                sweep(vid, g, seg, queries)
            del seg, cols
    elif args.shape == "c3":
        f = S.Column.synthetic_uniform("f", n, np.arange(1000, dtype=np.int32), seed=2)
        k = S.Column.synthetic_uniform("k", n, np.arange(1000, dtype=np.int32) * 3, seed=3)
        a = S.Column.synthetic_uniform("a", n, (np.arange(100000, dtype=np.int64) * 5 + 1).astype(np.int32), seed=4)
        b = S.Column.synthetic_uniform("b", n, np.arange(65536, dtype=np.int32) * 2, seed=5)
        lib = S.load_host_library()
        dv = np.ascontiguousarray(v_dictionary("irregular", seed=77), dtype=np.int32)
        dbytes = np.zeros(dv.shape[0] * 4, dtype=np.uint8)
        lib.ph_dict_write_int(S._i32p(dv), int(dv.shape[0]), S._u8p(dbytes))
        a_irr = S.Column("a_irr", a.encoding, a.bits, a.cardinality, a.fwd, dbytes, None, dv)
        seg = S.SegmentData("c3", n, [f, k, a, b, a_irr])
        flt = Q.leaf(Q.Pred.dict_range(0, 0, 100))
        queries = [("C3", Q.QuerySpec([(Q.SUM, 2), (Q.MAX, 3)], group_by=[1]), None, B(k) + B(a) + B(b)),
                   ("C3-filter", Q.QuerySpec([(Q.SUM, 2), (Q.MAX, 3)], filter=flt, group_by=[1]), None, B(k) + B(a) + B(b) + B(f)),
                   ("C3-irregular", Q.QuerySpec([(Q.SUM, 4), (Q.MAX, 3)], group_by=[1]), None, B(k) + B(a) + B(b))]
        with engine.open(seg) as g:
            sweep("C3", g, seg, queries)
    elif args.shape == "typed":
        # three and four aggregated raw / 8-byte columns: scan_private_typed_kernel<4> (and, in a batch of four segments, scan_typed_batch_kernel<4>)
        rng = np.random.default_rng(3)
        f = S.Column.synthetic_uniform("f", n, np.arange(1000, dtype=np.int32), seed=2)
        rl = S.Column.raw_typed("rl", rng.integers(-(2 ** 40), 2 ** 40, n, dtype=np.int64))
        rd = S.Column.raw_typed("rd", rng.normal(0, 1e6, n).astype(np.float64))
        ri = S.Column.raw("ri", rng.integers(-1000000, 1000000, n).astype(np.int32))
        seg = S.SegmentData("typed", n, [f, rl, rd, ri])
        flt = Q.leaf(Q.Pred.dict_range(0, 0, 500))
        queries = [("typed-4-slots", Q.QuerySpec([(Q.SUM, 1), (Q.MIN, 2), (Q.SUM, 3), (Q.AVG, 2)], filter=flt), None, B(f) + B(rl) + B(rd) + B(ri)),
                   ("typed-3-slots-no-filter", Q.QuerySpec([(Q.SUM, 1), (Q.MAX, 2), (Q.MIN, 3)]), None, B(rl) + B(rd) + B(ri))]
        with engine.open(seg) as g:
            sweep("typed", g, seg, queries)
    else:
        v = S.Column.synthetic_uniform("v", n, v_dictionary("affine"), seed=1)
        f = S.Column.synthetic_uniform("f", n, np.arange(1000, dtype=np.int32), seed=2)
        k = S.Column.synthetic_uniform("k", n, np.arange(1000, dtype=np.int32) * 3, seed=3)
        b = S.Column.synthetic_uniform("b", n, np.arange(65536, dtype=np.int32) * 2, seed=5)
        seg = S.SegmentData("not", n, [v, f, k, b])
        queries = [("AND-NOT-scan", Q.QuerySpec([(Q.SUM, 0)], filter=Q.and_(Q.leaf(Q.Pred.dict_range(1, 0, 300)), Q.not_(Q.leaf(Q.Pred.dict_range(2, 0, 500))))), None, B(v) + B(f) + B(k)),
                   ("AND3-scan", Q.QuerySpec([(Q.SUM, 0)], filter=Q.and_(Q.leaf(Q.Pred.dict_range(1, 0, 300)), Q.leaf(Q.Pred.dict_range(2, 0, 500)), Q.leaf(Q.Pred.dict_range(3, 0, 30000)))), None,
                    B(v) + B(f) + B(k) + B(b))]
        with engine.open(seg) as g:
            sweep("NOT", g, seg, queries)


if __name__ == "__main__":
    main()
