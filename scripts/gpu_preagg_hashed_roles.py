"""hashed GpuPreAgg: sweep of the role count / LDS fill target (1e8 rows)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpupreagg import GpuPreAgg
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
runtime.init()
rng = np.random.default_rng(3)
x = rng.integers(-10**6, 10**6, n, dtype=np.int64).astype(np.int32)
y = rng.random(n) * 100
spec = "(gpupreagg (key (var 1 int8)) (nrows) (psum (int8 (var 2 int4))) (psum (var 3 float8)))"
plan = [(100, [1, 2, 8]), (1000, [2, 4]), (10000, [16, 32])]
if len(sys.argv) > 2:
    plan = eval(sys.argv[2])
for ngroups, roles in plan:
    g = rng.integers(0, ngroups, n, dtype=np.int64)
    k = g * 1000003 * 65537 - 2**59
    ds = runtime.DeviceStore.upload(kds.build_kds("column", [kds.Column("int8", k), kds.Column("int4", x), kds.Column("float8", y)]))
    for nroles in roles:
        os.environ["STROM_GPUPREAGG_HASH_ROLES"] = str(nroles)
        agg = GpuPreAgg(spec).begin_hashed(ngroups_hint=ngroups)
        agg.program.wait()
        ts = []
        for it in range(4):
            st, pfm = agg.fold(ds)
            assert st == 0
            ts.append(pfm["time_kern_exec_ns"])
        ng = agg.num_groups()
        t = float(np.median(ts[1:])) * 1e-9
        print("ngroups=%d roles=%d check+fold=%.1f us  %.0f Mrows/s groups=%d" % (ngroups, nroles, t * 1e6, n / t / 1e6, ng), flush=True)
        agg.end()
    ds.release()
