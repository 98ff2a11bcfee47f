"""scan+join+groupby as ONE kernel (strom_submit_gpupreagg_lookup) over a resident fact chunk:
fact(fk int4, a int4, b float8) x dim(key int4, grp int4), WHERE a < k AND b > c, GROUP BY dim.grp.
usage: gpu_lookup_probe.py [nrows] [ndim] [ngroups] [selectivity]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpuhashjoin import GpuHashJoin, build_multihash
from pg_strom_amd.gpupreagg import GpuPreAgg

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
nd = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
ngroups = int(float(sys.argv[3])) if len(sys.argv) > 3 else 10000
sel = float(sys.argv[4]) if len(sys.argv) > 4 else 0.5
runtime.init()
rng = np.random.default_rng(5)
fk = rng.integers(0, int(nd * 1.25), n, dtype=np.int64).astype(np.int32)
a = rng.integers(0, 2**31, n, dtype=np.int64).astype(np.int32)
b = rng.random(n)
fact = kds.build_kds("column", [kds.Column("int4", fk), kds.Column("int4", a), kds.Column("float8", b)])
dkey = rng.permutation(nd).astype(np.int32)
dgrp = (dkey % ngroups).astype(np.int32)
inner = kds.build_kds("row_flat", [kds.Column("int4", dkey), kds.Column("int4", dgrp)])
km = build_multihash([(inner, [1])])
ext = [np.int32(int(2**31 * sel) - 1), 0.0]
ds = runtime.DeviceStore.upload(fact)
qual = "(and (int4lt (var 2 int4) (param 0 int4)) (float8gt (var 3 float8) (param 1 float8)))"
join = GpuHashJoin("(gpuhashjoin (rel (hashkey (var 1 int4) 1 int4)))").begin(km)
agg = GpuPreAgg("(gpupreagg (qual " + qual + ") (key (var 1 int4)) (nrows) (psum (int8 (var 2 int4))) (psum (var 3 float8)))")
agg.begin([(0, ngroups)], ext_params=ext)
agg.program.wait()
for it in range(6):
    agg.reset()
    t0 = time.perf_counter()
    st, pfm = agg.collect(agg.submit_lookup(join, ds, [(1, 2, "int4"), (0, 2, "int4"), (0, 3, "float8")]))
    t1 = time.perf_counter()
    assert st == 0
    print("lookup pass %d: fold %.0f us + merge %.0f us, packed=%d | wall %.2f ms = %.0f Mrows/s" % (
        it, (pfm["time_kern_exec_ns"] - pfm["time_kern_proj_ns"]) * 1e-3, pfm["time_kern_proj_ns"] * 1e-3,
        pfm["num_kern_prep"], (t1 - t0) * 1e3, n / (t1 - t0) / 1e6), flush=True)
pr = agg.fetch()
m = (a < ext[0]) & (b > ext[1]) & (fk < nd)
pos = np.empty(nd, dtype=np.int64); pos[dkey] = np.arange(nd)
g = dgrp[pos[fk[m]]]
cnt = np.bincount(g, minlength=ngroups)
order = np.argsort(pr.column(0)[0])
print("counts_ok=%s" % np.array_equal(pr.column(1)[0][order], cnt[cnt > 0]))
agg.end(); join.end(); ds.release()
