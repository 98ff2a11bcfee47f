"""C4 at 1e4 groups: one session with the id range split over two roles vs two sessions that split the AGGREGATES"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpupreagg import GpuPreAgg
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
ngroups = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10000
runtime.init()
rng = np.random.default_rng(3)
g = rng.integers(0, ngroups, n, dtype=np.int64).astype(np.int32)
x = rng.integers(-10**6, 10**6, n, dtype=np.int64).astype(np.int32)
y = rng.random(n) * 100
ds = runtime.DeviceStore.upload(kds.build_kds("column", [kds.Column("int4", g), kds.Column("int4", x), kds.Column("float8", y)]))
specs = {"all": "(gpupreagg (key (var 1 int4)) (nrows) (psum (int8 (var 2 int4))) (psum (var 3 float8)))",
         "A: count+sum(int)": "(gpupreagg (key (var 1 int4)) (nrows) (psum (int8 (var 2 int4))))",
         "B: sum(float8)": "(gpupreagg (key (var 1 int4)) (psum (var 3 float8)))",
         "A': count": "(gpupreagg (key (var 1 int4)) (nrows))",
         "B': sums": "(gpupreagg (key (var 1 int4)) (psum (int8 (var 2 int4))) (psum (var 3 float8)))"}
for name, spec in specs.items():
    agg = GpuPreAgg(spec).begin([(0, ngroups)])
    agg.program.wait()
    ts = []
    for it in range(6):
        st, pfm = agg.fold(ds)
        assert st == 0
        ts.append(pfm["time_kern_exec_ns"])
    print("%-20s kernel+merge %.1f us" % (name, np.median(ts[2:]) * 1e-3), flush=True)
    agg.end()
