"""hashed GpuPreAgg timing on a resident chunk: GROUP BY k (int8 keys spread over 2^60) COUNT(*), SUM(x), SUM(y)"""
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
spec = "(gpupreagg (key (var 1 int8)) (nrows) (psum (int8 (var 2 int4))) (psum (var 3 float8)))"
groups = [int(float(v)) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [1, 100, 10000, 1000000, 10000000]
for ngroups in groups:
    g = rng.integers(0, ngroups, n, dtype=np.int64)
    k = g * 1000003 * 65537 - 2**59
    ds = runtime.DeviceStore.upload(kds.build_kds("column", [kds.Column("int8", k), kds.Column("int4", x), kds.Column("float8", y)]))
    agg = GpuPreAgg(spec).begin_hashed(ngroups_hint=ngroups)
    agg.program.wait()
    ts, wall = [], []
    for it in range(5):
        t0 = time.perf_counter()
        st, pfm = agg.fold(ds)
        wall.append(time.perf_counter() - t0)
        assert st == 0
        ts.append(pfm["time_kern_exec_ns"])
    t0 = time.perf_counter()
    pr = agg.fetch()
    tf = time.perf_counter() - t0
    cnt = np.bincount(g, minlength=ngroups)
    order = np.argsort(pr.column(0)[0])
    ok = len(pr) == int((cnt > 0).sum()) and np.array_equal(pr.column(1)[0][order], cnt[cnt > 0] * 5)
    t = float(np.median(ts[2:])) * 1e-9
    print("ngroups=%d first fold %.1f ms (wall %.1f ms), steady check+fold=%.1f us  %.0f Mrows/s  %.0f GB/s (%.1f%% of 8TB/s) "
          "fetch %.1f ms  counts_ok=%s" % (
              ngroups, ts[0] * 1e-6, wall[0] * 1e3, t * 1e6, n / t / 1e6, 20.0 * n / t / 1e9,
              20.0 * n / t / 8e12 * 100, tf * 1e3, ok), flush=True)
    agg.end(); ds.release()
