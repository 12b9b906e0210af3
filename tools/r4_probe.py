"""Round-4 probes on the GPU box: (a) how long one map-based group-by over a 6 M-row segment takes, phase by phase; (b) nothing else yet."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import torch  # noqa: F401,E402
from pinot_amd import query as Q  # noqa: E402
from pinot_amd import segment as S  # noqa: E402
from pinot_amd.engine import Engine  # noqa: E402
import helpers as H  # noqa: E402


def main():
    eng = Engine(device_id=0, time_kernels=True)
    n = 6_000_011
    rng = np.random.default_rng(77)
    v = S.Column.synthetic_uniform("v", n, (np.arange(5000, dtype=np.int64) * 7 + 3).astype(np.int32), seed=5)
    k1, _, _ = H.random_dict_column(rng, "k1", n, 700)
    k2, _, _ = H.random_dict_column(rng, "k2", n, 300)
    f = S.Column.synthetic_uniform("f", n, np.arange(100, dtype=np.int32), seed=6)
    seg = S.SegmentData("heavy", n, [v, k1, k2, f])
    with eng.open(seg) as g:
        for name, spec in (("group-by k1,k2 filter", Q.QuerySpec([(Q.SUM, 0), (Q.COUNT, -1)], filter=Q.leaf(Q.Pred.dict_range(3, 0, 50)), group_by=[1, 2])),
                           ("group-by k1,k2", Q.QuerySpec([(Q.SUM, 0), (Q.COUNT, -1)], group_by=[1, 2])),
                           ("group-by k1", Q.QuerySpec([(Q.SUM, 0), (Q.COUNT, -1)], group_by=[1]))):
            for rep in range(4):
                t = time.perf_counter()
                r = g.execute(spec)
                dt = time.perf_counter() - t
                print("%-24s rep %d: %.2f ms host, device %.3f ms, kernel %.3f ms, groups %d" % (name, rep, dt * 1e3, r.device_ms, r.dominant_kernel_ms, len(r.groups)), flush=True)


main()
