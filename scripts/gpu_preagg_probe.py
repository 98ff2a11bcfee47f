"""GpuPreAgg C4-shape timing on a resident chunk: GROUP BY g (ngroups) COUNT(*), SUM(x), AVG(y)"""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpupreagg import GpuPreAgg
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
runtime.init()
rng = np.random.default_rng(3)
x = rng.integers(-10**6, 10**6, n, dtype=np.int64).astype(np.int32)
y = rng.random(n) * 100
spec = "(gpupreagg (key (var 1 int4)) (nrows) (psum (int8 (var 2 int4))) (psum (var 3 float8)))"
groups = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [1, 6, 30, 100, 1000, 5000, 10000, 20000]
for ngroups in groups:
    g = rng.integers(0, ngroups, n, dtype=np.int64).astype(np.int32)
    ds = runtime.DeviceStore.upload(kds.build_kds("column", [kds.Column("int4", g), kds.Column("int4", x), kds.Column("float8", y)]))
    agg = GpuPreAgg(spec).begin([(0, ngroups)])
    agg.program.wait()
    ts, tm = [], []
    for it in range(8):
        st, pfm = agg.fold(ds)
        assert st == 0
        ts.append(pfm["time_kern_exec_ns"])
        tm.append(pfm["time_kern_proj_ns"])
    pr = agg.fetch()
    cnt = np.bincount(g, minlength=ngroups)
    order = np.argsort(pr.column(0)[0])
    ok = np.array_equal(pr.column(1)[0][order], cnt * 8)
    t = float(np.median(ts[2:])) * 1e-9
    print("ngroups=%d kern=%.1f us (of which merge %.1f us)  %.0f Mrows/s  %.0f GB/s (%.1f%% of 8TB/s) counts_ok=%s" % (
        ngroups, t * 1e6, float(np.median(tm[2:])) * 1e-3, n / t / 1e6, 16.0 * n / t / 1e9,
        16.0 * n / t / 8e12 * 100, ok), flush=True)
    agg.end(); ds.release()
