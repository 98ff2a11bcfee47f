"""how does GpuPreAgg kernel time scale with the number of LDS atomics per row?"""
import sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpupreagg import GpuPreAgg
n = 100_000_000
runtime.init()
rng = np.random.default_rng(3)
x = rng.integers(-10**6, 10**6, n, dtype=np.int64).astype(np.int32)
y = rng.random(n) * 100
for ngroups in (6, 5000):
    g = rng.integers(0, ngroups, n, dtype=np.int64).astype(np.int32)
    ds = runtime.DeviceStore.upload(kds.build_kds("column", [kds.Column("int4", g), kds.Column("int4", x), kds.Column("float8", y)]))
    for label, spec in (("key only (flags)", "(gpupreagg (key (var 1 int4)))"),
                        ("nrows", "(gpupreagg (key (var 1 int4)) (nrows))"),
                        ("nrows+psum_i8", "(gpupreagg (key (var 1 int4)) (nrows) (psum (int8 (var 2 int4))))"),
                        ("nrows+psum_i8+psum_f8", "(gpupreagg (key (var 1 int4)) (nrows) (psum (int8 (var 2 int4))) (psum (var 3 float8)))"),
                        ("psum_f8 x3", "(gpupreagg (key (var 1 int4)) (psum (var 3 float8)) (pmin (var 3 float8)) (pmax (var 3 float8)))")):
        agg = GpuPreAgg(spec).begin([(0, ngroups)])
        ts = []
        for it in range(6):
            st, pfm = agg.fold(ds)
            ts.append(pfm["time_kern_exec_ns"])
        print("ngroups=%d %-24s kern=%.1f us" % (ngroups, label, float(np.median(ts[2:])) / 1e3), flush=True)
        agg.end()
    ds.release()
