#!/usr/bin/env python3
"""One-off (round 6c): numEntriesScannedInFilter of a machine of 9 .. 16 states WITHOUT episodes at 1 B rows -- `(b OR k OR f) AND k AND
(b OR k OR f)`, the same predicates behind several leaves -- with the function-only tile pass + the range kernel as the counter (default)
and with the table walk (PINOT_GPU_FSM_PERM=0).  Prints one JSON line: all-kernels ms, entries, exactness per mode.
    python tools/count_pass_probe.py [rows]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
    from pinot_amd import query as Q
    from pinot_amd import segment as S
    from pinot_amd.engine import Engine
    f = S.Column.synthetic_uniform("f", n, np.arange(1000, dtype=np.int32), seed=2)
    k = S.Column.synthetic_uniform("k", n, np.arange(1000, dtype=np.int32) * 3, seed=3)
    b = S.Column.synthetic_uniform("b", n, np.arange(65536, dtype=np.int32) * 2, seed=5)
    seg = S.SegmentData("count_pass", n, [f, k, b])
    F, K, B = Q.leaf(Q.Pred.dict_range(0, 0, 300)), Q.leaf(Q.Pred.dict_range(1, 0, 100)), Q.leaf(Q.Pred.dict_range(2, 0, 3000))
    spec = Q.QuerySpec([(Q.COUNT, -1)], filter=Q.and_(Q.or_(B, K, F), K, Q.or_(B, K, F)))
    engine = Engine(device_id=0, time_kernels=True)
    out = {"rows": n}
    with engine.open(seg) as g:
        for mode, env in (("byte_functions", {"PINOT_GPU_FSM_PERM": None}), ("table_walk", {"PINOT_GPU_FSM_PERM": "0"})):
            engine.reinit(**env)
            ms = []
            for _ in range(4):
                r = g.execute(spec)
                ms.append(r.device_ms)
            out[mode] = {"all_kernels_ms": min(ms[1:]), "entries": int(r.stats[1]), "exact": bool(r.filter_entries_exact), "count": int(r.aggregations[0].count)}
        engine.reinit(PINOT_GPU_FSM_PERM=None)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
