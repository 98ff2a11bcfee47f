"""hashed GpuPreAgg over hash partitions against the global-table / hash-role paths:
GROUP BY k (int8 keys spread over 2^60) COUNT(*), SUM(x), SUM(y) on a resident chunk"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpupreagg import GpuPreAgg
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
groups = [int(float(v)) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [10000, 30000, 100000, 1000000, 10000000]
modes = sys.argv[3].split(',') if len(sys.argv) > 3 else ["parts", "table"]
runtime.init()
rng = np.random.default_rng(3)
x = rng.integers(-10**6, 10**6, n, dtype=np.int64).astype(np.int32)
y = rng.random(n) * 100
spec = "(gpupreagg (key (var 1 int8)) (nrows) (psum (int8 (var 2 int4))) (psum (var 3 float8)))"
for ngroups in groups:
    g = rng.integers(0, ngroups, n, dtype=np.int64)
    k = g * 1000003 * 65537 - 2**59
    ds = runtime.DeviceStore.upload(kds.build_kds("column", [kds.Column("int8", k), kds.Column("int4", x), kds.Column("float8", y)]))
    cnt = np.bincount(g, minlength=ngroups)
    sx = np.bincount(g, weights=x.astype(np.float64), minlength=ngroups).astype(np.int64)
    for mode in modes:
        os.environ.pop("STROM_GPUPREAGG_HASH_NO_PARTS", None)
        os.environ.pop("STROM_GPUPREAGG_HASH_PARTS_MIN", None)
        if mode == "table":
            os.environ["STROM_GPUPREAGG_HASH_NO_PARTS"] = "1"
        else:
            os.environ["STROM_GPUPREAGG_HASH_PARTS_MIN"] = "0"
        agg = GpuPreAgg(spec).begin_hashed(ngroups_hint=ngroups)
        agg.program.wait()
        ts, wall, nk = [], [], []
        for it in range(5):
            t0 = time.perf_counter()
            st, pfm = agg.fold(ds)
            wall.append(time.perf_counter() - t0)
            assert st == 0
            ts.append(pfm["time_kern_exec_ns"])
            nk.append(pfm["num_kern_exec"])
        pr = agg.fetch()
        order = np.argsort(pr.column(0)[0])
        ok = (len(pr) == int((cnt > 0).sum()) and np.array_equal(pr.column(1)[0][order], cnt[cnt > 0] * 5)
              and np.array_equal(pr.column(2)[0][order], sx[cnt > 0] * 5))
        t = float(np.median(ts[2:])) * 1e-9
        print("ngroups=%d %-5s first %.1f ms, steady %.1f us (wall %.1f us, %d launches)  %.0f Mrows/s  %.0f GB/s (%.1f%% of 8TB/s)  ok=%s"
              % (ngroups, mode, ts[0] * 1e-6, t * 1e6, np.median(wall[2:]) * 1e6, nk[-1], n / t / 1e6, 20.0 * n / t / 1e9,
                 20.0 * n / t / 8e12 * 100, ok), flush=True)
        agg.end()
    ds.release()
