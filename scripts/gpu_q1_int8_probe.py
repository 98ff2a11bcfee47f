"""What would Q1 cost if the numeric(*,2) columns were stored as int8 at their typmod scale
(no numeric -> fixed conversion per row)?  Same query shape over int8 columns."""
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
qty, prc = rng.integers(1, 51, n), rng.integers(90000, 10494951, n)
dsc, tax = rng.integers(0, 11, n), rng.integers(0, 9, n)
ship = rng.integers(-2922, -2922 + 2526, n).astype(np.int32)
head = "(gpupreagg (qual (date_le (var 7 date) (const date '1998-09-02'))) (key (var 1 char1)) (key (var 2 char1))"
for label, cols, spec in (
    ("numeric(*,2) as 64-bit numerics",
     [kds.Column("char1", rf), kds.Column("char1", ls), kds.numeric_from_scaled(qty, 0), kds.numeric_from_scaled(prc, 2),
      kds.numeric_from_scaled(dsc, 2), kds.numeric_from_scaled(tax, 2), kds.Column("date", ship)],
     head + " (psum (var 3 numeric 0) 0) (psum (var 4 numeric 2) 2)"
     " (psum (numeric_mul (var 4 numeric 2) (numeric_sub (const numeric 1) (var 5 numeric 2))) 4)"
     " (psum (numeric_mul (numeric_mul (var 4 numeric 2) (numeric_sub (const numeric 1) (var 5 numeric 2)))"
     " (numeric_add (const numeric 1) (var 6 numeric 2))) 6)"
     " (nrows (isnotnull (var 3 numeric 0))) (nrows (isnotnull (var 4 numeric 2)))"
     " (psum (var 5 numeric 2) 2) (nrows (isnotnull (var 5 numeric 2))) (nrows))"),
    ("the same values as int8 at their scale",
     [kds.Column("char1", rf), kds.Column("char1", ls), kds.Column("int8", qty), kds.Column("int8", prc),
      kds.Column("int8", dsc), kds.Column("int8", tax), kds.Column("date", ship)],
     head + " (psum (var 3 int8)) (psum (var 4 int8))"
     " (psum (int8mul (var 4 int8) (int8mi (const int8 100) (var 5 int8))))"
     " (psum (int8mul (int8mul (var 4 int8) (int8mi (const int8 100) (var 5 int8))) (int8pl (const int8 100) (var 6 int8))))"
     " (nrows (isnotnull (var 3 int8))) (nrows (isnotnull (var 4 int8)))"
     " (psum (var 5 int8)) (nrows (isnotnull (var 5 int8))) (nrows))")):
    ds = runtime.DeviceStore.upload(kds.build_kds("column", cols))
    agg = GpuPreAgg(spec).begin([(65, 18), (70, 10)])
    agg.census(ds)
    agg.compact()
    ts = []
    for _ in range(6):
        st, pfm = agg.fold(ds)
        assert st == 0, st
        ts.append(pfm["time_kern_exec_ns"] - pfm["time_kern_proj_ns"])
    t = float(np.median(ts[2:])) * 1e-9
    print("%-42s fold kernel %.1f us per %d rows = %.0f us per 1e8 rows (%.3f of 8 TB/s at 38 B/row)"
          % (label, t * 1e6, n, t * 1e6 * 1e8 / n, 38 * n / t / 8e12), flush=True)
    agg.end()
    ds.release()
