import sys, inspect, itertools
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_oracle_golden as A, test_oracle_nulls as B, test_oracle_group_map as Cm, test_oracle_typed as D, test_oracle_layouts as E
import test_oracle_iterator_scripts as F, test_oracle_rank_keys as G
ran=0
for mod in (A,B,Cm,D,E,F,G):
    for name, fn in inspect.getmembers(mod, inspect.isfunction):
        if not name.startswith("test_"): continue
        marks = getattr(fn, "pytestmark", [])
        params = [m for m in marks if m.name == "parametrize"]
        sig = list(inspect.signature(fn).parameters)
        if any(p in ("tmp_path",) for p in sig): continue
        if not params:
            if sig: continue
            fn(); ran+=1; continue
        names=[]; values=[]
        for m in params:
            n = [x.strip() for x in m.args[0].split(",")]
            names.append(n); values.append(m.args[1])
        for combo in itertools.product(*values):
            kw={}
            for n,v in zip(names, combo):
                if len(n)==1: kw[n[0]]=v
                else: kw.update(dict(zip(n,v)))
            fn(**kw); ran+=1
        print(mod.__name__, name, "ok", flush=True)
print("ran", ran, "test invocations under ASAN/UBSAN")
