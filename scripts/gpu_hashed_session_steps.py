"""where a stateless hashed GROUP BY request spends its time: create, fold, fetch, release"""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpupreagg import GpuPreAgg
from pg_strom_amd._lib import lib
runtime.init()
spec = "(gpupreagg (key (var 1 int8)) (nrows) (psum (int8 (var 2 int4))) (psum (var 3 float8)))"
rng = np.random.default_rng(1)
for n, ngroups in ((10_000_000, 100_000), (50_000_000, 1_000_000)):
    g = rng.integers(0, ngroups, n).astype(np.int64)
    k = g * 1000003 * 65537 - 2**59
    x = rng.integers(-10**6, 10**6, n).astype(np.int32)
    y = rng.random(n) * 100
    chunk = runtime.DeviceStore.upload(kds.build_kds("column", [kds.Column("int8", k), kds.Column("int4", x), kds.Column("float8", y)]))
    for rep in range(3):
        t0 = time.perf_counter()
        agg = GpuPreAgg(spec).begin_hashed(ngroups_hint=ngroups)
        t1 = time.perf_counter()
        st, pfm = agg.fold(chunk)
        t2 = time.perf_counter()
        need = lib.strom_gpupreagg_fetch(agg.session, None, 0)
        t3 = time.perf_counter()
        pr = agg.fetch()
        t4 = time.perf_counter()
        agg.end()
        t5 = time.perf_counter()
        print("%d rows %d groups: create %.2f ms, fold %.2f ms (kernels %.2f), fetch size query %.2f ms, fetch %.2f ms, end %.2f ms"
              % (n, len(pr), (t1 - t0) * 1e3, (t2 - t1) * 1e3, pfm["time_kern_exec_ns"] * 1e-6, (t3 - t2) * 1e3, (t4 - t3) * 1e3,
                 (t5 - t4) * 1e3), flush=True)
    chunk.release()
