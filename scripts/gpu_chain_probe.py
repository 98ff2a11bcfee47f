"""Scan -> HashJoin -> PreAgg over one resident chunk, nothing leaves HBM in between:
fact(fk int4, a int4, b float8) x dim(key int4, grp int4), WHERE a < k AND b > c,
GROUP BY dim.grp -> count(*), sum(a), sum(b).  Wall-clock per stage (each stage is waited for)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpuhashjoin import GpuHashJoin, build_multihash
from pg_strom_amd.gpupreagg import GpuPreAgg
from pg_strom_amd.gpuscan import GpuScan

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
nd = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
ngroups = int(float(sys.argv[3])) if len(sys.argv) > 3 else 1000
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
scan = GpuScan(qual).begin(ext_params=ext)
join = GpuHashJoin("(gpuhashjoin (rel (hashkey (var 1 int4) 1 int4)))").begin(km)
agg = GpuPreAgg("(gpupreagg (key (var 1 int4)) (nrows) (psum (int8 (var 2 int4))) (psum (var 3 float8)))")
agg.begin([(0, ngroups)])
agg.program.wait()
for it in range(4):
    agg.reset()
    t0 = time.perf_counter()
    rowmap, res = scan.scan_to_rowmap(ds)
    t1 = time.perf_counter()
    joined, nitems = join.join_to_column(ds, [(1, 2, "int4"), (0, 2, "int4"), (0, 3, "float8")], row_map=rowmap,
                                         nrooms=int(res.nitems * 0.82) + 1000)
    t2 = time.perf_counter()
    st, pfm = agg.fold(joined)
    pr = agg.fetch()
    t3 = time.perf_counter()
    rowmap.release(); joined.release()
    assert st == 0
    print("pass %d: scan %.2f ms (%d rows) | join+project %.2f ms (%d rows) | preagg+fetch %.2f ms (kernel %.0f us, %d groups)"
          " | total %.2f ms = %.0f Mrows/s of fact rows" % (
              it, (t1 - t0) * 1e3, res.nitems, (t2 - t1) * 1e3, nitems, (t3 - t2) * 1e3,
              pfm["time_kern_exec_ns"] * 1e-3, len(pr), (t3 - t0) * 1e3, n / (t3 - t0) / 1e6), flush=True)
# the same query with the WHERE pulled up into the join (the reference's plan shape,
# gpuhashjoin.c:2047-2050): no scan pass, no row map, the one-pass join kernel
join2 = GpuHashJoin("(gpuhashjoin (rel (hashkey (var 1 int4) 1 int4) (qual " + qual + ")))",
                    row_population_ratio=0.45).begin(km, ext_params=ext)
for it in range(4):
    agg.reset()
    t0 = time.perf_counter()
    joined, nitems = join2.join_to_column(ds, [(1, 2, "int4"), (0, 2, "int4"), (0, 3, "float8")], zone_maps=False)
    t1 = time.perf_counter()
    st, pfm = agg.fold(joined)
    pr = agg.fetch()
    t2 = time.perf_counter()
    joined.release()
    print("pulled-up pass %d: join+project %.2f ms (%d rows) | preagg+fetch %.2f ms | total %.2f ms = %.0f Mrows/s of fact rows" % (
        it, (t1 - t0) * 1e3, nitems, (t2 - t1) * 1e3, (t2 - t0) * 1e3, n / (t2 - t0) / 1e6), flush=True)
# and with the projection fused into the aggregate: GpuPreAgg reads the result pairs
from pg_strom_amd.gpuhashjoin import STROM_RESULTS_ON_DEVICE
for it in range(4):
    agg.reset()
    t0 = time.perf_counter()
    jp = join2.submit(ds, flags=STROM_RESULTS_ON_DEVICE)
    ap = agg.submit_joined(join2, jp, ds, [(1, 2, "int4"), (0, 2, "int4"), (0, 3, "float8")])
    st, pfm = agg.collect(ap)
    jr = join2.collect(jp)
    pr = agg.fetch()
    t1 = time.perf_counter()
    assert st == 0
    print("fused pass %d: join %.0f us + aggregate over %d pairs %.0f us (kernel times) | total wall %.2f ms = %.0f Mrows/s of fact rows" % (
        it, jr.perfmon["time_kern_exec_ns"] * 1e-3, jr.nitems, pfm["time_kern_exec_ns"] * 1e-3,
        (t1 - t0) * 1e3, n / (t1 - t0) / 1e6), flush=True)
# and with the join as a lookup inside the aggregate's own pass (no join request at all)
agg3 = GpuPreAgg("(gpupreagg (qual " + qual + ") (key (var 1 int4)) (nrows) (psum (int8 (var 2 int4))) (psum (var 3 float8)))")
agg3.begin([(0, ngroups)], ext_params=ext)
agg3.program.wait()
for it in range(4):
    agg3.reset()
    t0 = time.perf_counter()
    st, pfm = agg3.collect(agg3.submit_lookup(join2, ds, [(1, 2, "int4"), (0, 2, "int4"), (0, 3, "float8")]))
    pr3 = agg3.fetch()
    t1 = time.perf_counter()
    assert st == 0
    print("lookup pass %d: one kernel over the fact chunk %.0f us (+ merge) | total wall %.2f ms = %.0f Mrows/s of fact rows, %d groups" % (
        it, pfm["time_kern_exec_ns"] * 1e-3, (t1 - t0) * 1e3, n / (t1 - t0) / 1e6, len(pr3)), flush=True)
order3 = np.argsort(pr3.column(0)[0])
same = (np.array_equal(pr3.column(1)[0][order3], pr.column(1)[0][np.argsort(pr.column(0)[0])]) and
        np.array_equal(pr3.column(2)[0][order3], pr.column(2)[0][np.argsort(pr.column(0)[0])]))
print("lookup_equals_fused=%s" % same)
agg3.end()
join2.end()
m = (a < ext[0]) & (b > ext[1]) & (fk < nd)
pos = np.empty(nd, dtype=np.int64); pos[dkey] = np.arange(nd)
g = dgrp[pos[fk[m]]]
cnt = np.bincount(g, minlength=ngroups)
order = np.argsort(pr.column(0)[0])
print("counts_ok=%s sums_ok=%s" % (np.array_equal(pr.column(1)[0][order], cnt[cnt > 0]),
      np.array_equal(pr.column(2)[0][order], np.bincount(g, weights=a[m].astype(np.float64), minlength=ngroups)[cnt > 0].astype(np.int64))))
