"""tests/test_gpu_filter_stats.py runs this under `rocprofv3 --kernel-trace`, once per mode: four leap-frogging filters (two scan leaves, three,
an OR child, a NOT child) with (`bound`) or without (`exact`) PG_QUERY_STATS_UPPER_BOUND_OK.  Prints one JSON line: per query the answer,
the statistic and whether it is exact; the test compares the two runs' answers and reads the kernel names out of the traces."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def queries(Q, bound):
    a, b, c = (lambda lo, hi: Q.leaf(Q.Pred.dict_range(0, lo, hi))), (lambda lo, hi: Q.leaf(Q.Pred.dict_range(1, lo, hi))), (lambda lo, hi: Q.leaf(Q.Pred.dict_range(2, lo, hi)))
    filters = {"a AND b": Q.and_(a(0, 30), b(0, 5)), "a AND b AND c": Q.and_(a(0, 30), b(0, 5), c(0, 20)),
               "a AND (b OR c)": Q.and_(a(0, 30), Q.or_(b(0, 2), c(0, 4))), "a AND NOT b": Q.and_(a(0, 30), Q.not_(b(0, 5)))}
    return {name: Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, 3)], filter=f, stats_upper_bound_ok=bound) for name, f in filters.items()}


def main():
    bound = sys.argv[1] == "bound"
    from pinot_amd import query as Q
    from pinot_amd import segment as S
    from pinot_amd.engine import Engine
    n = 300_007
    cols = [S.Column.synthetic_uniform("a", n, np.arange(100, dtype=np.int32), seed=1), S.Column.synthetic_uniform("b", n, np.arange(10, dtype=np.int32), seed=2),
            S.Column.synthetic_uniform("c", n, np.arange(40, dtype=np.int32) * 3, seed=3), S.Column.synthetic_uniform("v", n, (np.arange(5000, dtype=np.int64) * 7 + 3).astype(np.int32), seed=4)]
    seg = S.SegmentData("stats_flag", n, cols)
    engine = Engine(device_id=0, time_kernels=False)
    out = {}
    with engine.open(seg) as g:
        for name, spec in queries(Q, bound).items():
            r = g.execute(spec)
            out[name] = {"count": r.aggregations[0].count, "sum": r.aggregations[1].sum_i64, "docs_scanned": r.stats[0], "entries": r.stats[1], "post": r.stats[2], "total": r.stats[3],
                         "exact": bool(r.filter_entries_exact), "scan_leaves_x_docs": sum(1 for _ in spec.predicates) * n}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
