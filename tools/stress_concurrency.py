#!/usr/bin/env python3
"""Stress of concurrent pg_execute calls on one segment under different completion settings (fold / finalize launch, polling / stream
synchronise): tools/stress_concurrency.py [rounds].  Prints failures per setting with the first mismatch."""
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    import numpy as np
    import torch  # noqa: F401
    from oracle import oracle
    from pinot_amd import query as Q
    from pinot_amd import segment as S
    from pinot_amd.engine import Engine
    import helpers as H

    rng = np.random.default_rng(8)
    n = 500000
    v, ids, dv = H.random_dict_column(rng, "v", n, 5000)
    seg = S.SegmentData("conc", n, [v])
    specs = [Q.QuerySpec([(Q.SUM, 0), (Q.COUNT, -1)], filter=Q.leaf(Q.Pred.dict_range(0, lo, lo + 1000))) for lo in range(0, 4000, 500)]
    want = [oracle.execute(seg, s) for s in specs]
    engine = Engine(device_id=0, time_kernels=True)
    settings = [("default", {}), ("poll0", {"PINOT_GPU_POLL_RESULT": "0"}), ("fold0", {"PINOT_GPU_FOLD_FINALIZE": "0"}),
                ("fold0_poll0", {"PINOT_GPU_FOLD_FINALIZE": "0", "PINOT_GPU_POLL_RESULT": "0"}), ("untimed", {"_untimed": "1"})]
    for name, env in settings:
        for k in ("PINOT_GPU_POLL_RESULT", "PINOT_GPU_FOLD_FINALIZE"):
            os.environ.pop(k, None)
        for k, val in env.items():
            if not k.startswith("_"):
                os.environ[k] = val
        eng = Engine(device_id=0, time_kernels="_untimed" not in env)
        failures, first = 0, None
        with eng.open(seg) as gseg:
            for r in range(rounds):
                errors = []

                def worker(i):
                    try:
                        for _ in range(5):
                            got = gseg.execute(specs[i])
                            if (got.aggregations[0].sum_i64, got.aggregations[1].count, got.stats[0]) != (want[i].aggregations[0].sum_i64, want[i].aggregations[1].count, want[i].stats[0]):
                                errors.append((i, got.aggregations[0].sum_i64, want[i].aggregations[0].sum_i64, got.aggregations[1].count, want[i].aggregations[1].count, got.stats))
                    except Exception as ex:  # noqa: BLE001
                        errors.append((i, repr(ex)))
                threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(specs))]
                [t.start() for t in threads]
                [t.join() for t in threads]
                if errors:
                    failures += 1
                    first = first or errors[:3]
        print(name, "rounds", rounds, "failed", failures, first, flush=True)


if __name__ == "__main__":
    main()
