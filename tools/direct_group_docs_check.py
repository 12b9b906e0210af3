"""Round-4 debug: docs scanned by the direct HBM-atomic group-by path against the exact filter count (small results: numGroupsLimit 10)."""
import sys

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: F401,E402
from pinot_amd import query as Q  # noqa: E402
from pinot_amd import segment as S  # noqa: E402
from pinot_amd.engine import Engine  # noqa: E402


def main():
    eng = Engine(device_id=0, time_kernels=True)
    for n in (5_000_000, 9_000_000, 30_000_000):
        v = S.Column.synthetic_uniform("v", n, (np.arange(100000, dtype=np.int64) * 7 + 3).astype(np.int32), seed=1)
        k = S.Column.synthetic_uniform("k", n, np.arange(1000, dtype=np.int32) * 3, seed=3)
        f = S.Column.synthetic_uniform("f", n, np.arange(1000, dtype=np.int32), seed=2)
        seg = S.SegmentData("dbg", n, [v, f, k])
        with eng.open(seg) as g:
            flt = Q.leaf(Q.Pred.dict_range(1, 0, 100))
            exact = g.execute(Q.QuerySpec([(Q.COUNT, -1)], filter=flt)).aggregations[0].count
            for aggs, name in (([(Q.COUNT, -1)], "count"), ([(Q.SUM, 1), (Q.MAX, 2)], "sum,max")):
                for f_, want in ((None, n), (flt, exact)):
                    got = g.execute(Q.QuerySpec(aggs, filter=f_, group_by=[0, 2], num_groups_limit=10))
                    print(n, name, "filter" if f_ else "nofilter", got.dominant_kernel, "docs", got.stats[0], "want", want, "OK" if got.stats[0] == want else "WRONG", "%.3f ms" % got.dominant_kernel_ms, flush=True)


main()
