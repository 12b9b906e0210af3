import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, ctypes as C
from pinot_amd import _abi, query as Q, segment as S
from pinot_amd.engine import Engine
n = 1_000_000_000
v = S.Column.synthetic_uniform("v", n, (np.arange(100000, dtype=np.int64) * 7 + 3).astype(np.int32), seed=1)
f = S.Column.synthetic_uniform("f", n, np.arange(1000, dtype=np.int32), seed=2)
seg = S.SegmentData("c2b", n, [v, f])
eng = Engine(device_id=0, time_kernels=True)
g = eng.open(seg)
res = _abi.pg_result()
for name, spec in (("10%", Q.QuerySpec([(Q.SUM, 0)], filter=Q.leaf(Q.Pred.dict_range(1, 0, 100)))), ("1%", Q.QuerySpec([(Q.SUM, 0)], filter=Q.leaf(Q.Pred.dict_range(1, 0, 10)))),
                   ("10% again", Q.QuerySpec([(Q.SUM, 0)], filter=Q.leaf(Q.Pred.dict_range(1, 0, 100))))):
    ms = []
    for i in range(40):
        g.execute_raw(spec, res); ms.append(res.dominant_kernel_ms); eng.lib.pg_result_free(C.byref(res))
    print(name, " ".join("%.3f" % x for x in ms))
