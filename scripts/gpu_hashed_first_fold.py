"""where the first fold of a hashed session spends its wall clock ("one unexplained 22 ms host
stall" in profiles/r02_hashed_fetch.txt): wall and kernel time of five folds of one resident chunk, the
buffer pool cold and warm.  usage: gpu_hashed_first_fold.py [rows] [groups]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpupreagg import GpuPreAgg
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
ngroups = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
runtime.init()
rng = np.random.default_rng(3)
g = rng.integers(0, ngroups, n, dtype=np.int64) * 1000003 * 65537 - 2**59
ds = runtime.DeviceStore.upload(kds.build_kds("column", [kds.Column("int8", g), kds.Column("int4", rng.integers(-10**6, 10**6, n).astype(np.int32)),
                                                         kds.Column("float8", rng.random(n))]))
spec = "(gpupreagg (key (var 1 int8)) (nrows) (psum (int8 (var 2 int4))) (psum (var 3 float8)))"
for label in ("cold pool", "warm pool (second session)"):
    agg = GpuPreAgg(spec).begin_hashed(ngroups_hint=ngroups)
    agg.program.wait()
    out = []
    for it in range(5):
        t0 = time.perf_counter()
        st, pfm = agg.fold(ds)
        out.append("%.2f ms wall / %.2f ms kernels" % ((time.perf_counter() - t0) * 1e3, pfm["time_kern_exec_ns"] * 1e-6))
        assert st == 0
    print("%-28s %s" % (label, "; ".join(out)), flush=True)
    agg.end()
ds.release()
