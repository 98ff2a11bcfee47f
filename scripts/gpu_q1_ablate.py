"""Where does the Q1-shaped kernel's time go?  Same chunk, growing target lists."""
import sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpupreagg import GpuPreAgg

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
runtime.init()
rng = np.random.default_rng(5)
rf = rng.choice(np.array([65, 78, 82], dtype=np.int8), n)
ls = rng.choice(np.array([70, 79], dtype=np.int8), n)
cols = [kds.Column("char1", rf), kds.Column("char1", ls),
        kds.numeric_from_scaled(rng.integers(1, 51, n), 0),
        kds.numeric_from_scaled(rng.integers(90000, 10494951, n), 2),
        kds.numeric_from_scaled(rng.integers(0, 11, n), 2),
        kds.numeric_from_scaled(rng.integers(0, 9, n), 2),
        kds.Column("date", rng.integers(-2922, -2922 + 2526, n).astype(np.int32))]
ds = runtime.DeviceStore.upload(kds.build_kds("column", cols))
Q, P, D, T = "(var 3 numeric 0)", "(var 4 numeric 2)", "(var 5 numeric 2)", "(var 6 numeric 2)"
dp = "(numeric_mul %s (numeric_sub (const numeric 1) %s))" % (P, D)
ch = "(numeric_mul %s (numeric_add (const numeric 1) %s))" % (dp, T)
head = "(gpupreagg (qual (date_le (var 7 date) (const date '1998-09-02'))) (key (var 1 char1)) (key (var 2 char1))"
cases = [
    ("nrows", " (nrows)", 6),
    ("+sum qty", " (nrows) (psum %s 0)" % Q, 14),
    ("+sum price", " (nrows) (psum %s 0) (psum %s 2)" % (Q, P), 22),
    ("+sum disc_price", " (nrows) (psum %s 0) (psum %s 2) (psum %s 4)" % (Q, P, dp), 30),
    ("+sum charge", " (nrows) (psum %s 0) (psum %s 2) (psum %s 4) (psum %s 6)" % (Q, P, dp, ch), 38),
    ("+sum disc", " (nrows) (psum %s 0) (psum %s 2) (psum %s 4) (psum %s 6) (psum %s 2)" % (Q, P, dp, ch, D), 38),
    ("+3 nrows(notnull)", " (nrows) (psum %s 0) (psum %s 2) (psum %s 4) (psum %s 6) (psum %s 2)"
     " (nrows (isnotnull %s)) (nrows (isnotnull %s)) (nrows (isnotnull %s))" % (Q, P, dp, ch, D, Q, P, D), 38),
    ("only 4 plain sums", " (psum %s 0) (psum %s 2) (psum %s 2) (psum %s 2)" % (Q, P, D, T), 38),
]
for compact in (True, False):
    for label, targets, bpr in cases:
        agg = GpuPreAgg(head + targets + ")").begin([(65, 18), (70, 10)])
        if compact:
            agg.census(ds)
            agg.compact()
        ts = []
        for _ in range(6):
            st, pfm = agg.fold(ds)
            assert st == 0
            ts.append(pfm["time_kern_exec_ns"] - pfm["time_kern_proj_ns"])
        t = float(np.median(ts[2:])) * 1e-9
        print("compact=%d %-20s main kernel %.1f us  %.0f GB/s of the %d B/row it reads" %
              (compact, label, t * 1e6, bpr * n / t / 1e9, bpr), flush=True)
        agg.end()
