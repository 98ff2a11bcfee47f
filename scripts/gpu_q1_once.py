"""One Q1-shaped GpuPreAgg fold over a resident chunk (for profilers): python gpu_q1_once.py [rows] [folds]"""
import sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpupreagg import GpuPreAgg
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
folds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
runtime.init()
rng = np.random.default_rng(5)
cols = [kds.Column("char1", rng.choice(np.array([65, 78, 82], dtype=np.int8), n)),
        kds.Column("char1", rng.choice(np.array([70, 79], dtype=np.int8), n)),
        kds.numeric_from_scaled(rng.integers(1, 51, n), 0),
        kds.numeric_from_scaled(rng.integers(90000, 10494951, n), 2),
        kds.numeric_from_scaled(rng.integers(0, 11, n), 2),
        kds.numeric_from_scaled(rng.integers(0, 9, n), 2),
        kds.Column("date", rng.integers(-2922, -2922 + 2526, n).astype(np.int32))]
ds = runtime.DeviceStore.upload(kds.build_kds("column", cols))
Q, P, D, T = "(var 3 numeric 0)", "(var 4 numeric 2)", "(var 5 numeric 2)", "(var 6 numeric 2)"
dp = "(numeric_mul %s (numeric_sub (const numeric 1) %s))" % (P, D)
ch = "(numeric_mul %s (numeric_add (const numeric 1) %s))" % (dp, T)
spec = ("(gpupreagg (qual (date_le (var 7 date) (const date '1998-09-02'))) (key (var 1 char1)) (key (var 2 char1))"
        " (psum %s 0) (psum %s 2) (psum %s 4) (psum %s 6) (nrows (isnotnull %s)) (nrows (isnotnull %s))"
        " (psum %s 2) (nrows (isnotnull %s)) (nrows))" % (Q, P, dp, ch, Q, P, D, D))
agg = GpuPreAgg(spec).begin([(65, 18), (70, 10)])
agg.census(ds)
agg.compact()
for _ in range(folds):
    st, pfm = agg.fold(ds)
    assert st == 0
print("fold %.1f us" % ((pfm["time_kern_exec_ns"] - pfm["time_kern_proj_ns"]) / 1e3))
agg.end()
