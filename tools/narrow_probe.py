"""Filters over narrow dictionary columns (4 / 6 / 8 bits): kernel time, rate and the answers.  tools/narrow_ab.sh runs it with
PINOT_GPU_SCAN_NARROW=1 / 0 (scan_narrow_kernel vs scan_private_kernel) and compares the answers."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np

from pinot_amd import _abi, query as Q, segment as S
from pinot_amd.engine import Engine

n = int(os.environ.get("NARROW_ROWS", 250_000_000))
cols = []
for name, card, seed in zip("pqr", (16, 64, 256), (11, 12, 13)):
    cols.append(S.Column.from_dict_ids(name, np.arange(card, dtype=np.int32), S.synthetic_dict_ids(seed, 0, n, card)))
seg = S.SegmentData("narrow", n, cols)
eng = Engine(device_id=0, time_kernels=True)
g = eng.open(seg)
res = _abi.pg_result()
eq = lambda c, d: Q.leaf(Q.Pred.dict_range(c, d, d + 1))
bits = [c.bits for c in cols]
queries = [
    ("COUNT p=3 AND q=5 AND r=7", Q.and_(eq(0, 3), eq(1, 5), eq(2, 7)), sum(bits)),
    ("COUNT p<8", Q.leaf(Q.Pred.dict_range(0, 0, 8)), bits[0]),
    ("COUNT q<32", Q.leaf(Q.Pred.dict_range(1, 0, 32)), bits[1]),
    ("COUNT r in [7,100)", Q.leaf(Q.Pred.dict_range(2, 7, 100)), bits[2]),
    ("COUNT p<8 OR NOT q<32", Q.or_(Q.leaf(Q.Pred.dict_range(0, 0, 8)), Q.not_(Q.leaf(Q.Pred.dict_range(1, 0, 32)))), bits[0] + bits[1]),
]
for name, flt, bits_per_row in queries:
    spec = Q.QuerySpec([(Q.COUNT, -1)], filter=flt)
    ms = []
    for i in range(30):
        g.execute_raw(spec, res)
        ms.append(res.dominant_kernel_ms)
        count, kernel = int(res.aggregations[0].count), int(res.dominant_kernel)
        eng.lib.pg_result_free(C.byref(res))
    k = float(np.median(ms[10:]))
    print(json.dumps({"query": name, "rows": n, "kernel": _abi.KERNEL_NAMES.get(kernel), "kernel_ms": round(k, 4), "GBps": round(n * bits_per_row / 8 / k / 1e6, 1),
                      "frac_of_8TBps": round(n * bits_per_row / 8 / k / 1e6 / 8000, 3), "count": count}))
