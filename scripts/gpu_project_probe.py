"""column projection of joined rows: which part costs what (wall clock around join_to_column minus the join)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpuhashjoin import GpuHashJoin, build_multihash
n, nd = 100_000_000, 1_000_000
runtime.init()
rng = np.random.default_rng(5)
fk = rng.integers(0, int(nd * 1.25), n, dtype=np.int64).astype(np.int32)
a = rng.integers(0, 2**31, n, dtype=np.int64).astype(np.int32)
b = rng.random(n)
ds = runtime.DeviceStore.upload(kds.build_kds("column", [kds.Column("int4", fk), kds.Column("int4", a), kds.Column("float8", b)]))
dkey = rng.permutation(nd).astype(np.int32)
inner = kds.build_kds("row_flat", [kds.Column("int4", dkey), kds.Column("int4", (dkey % 1000).astype(np.int32))])
km = build_multihash([(inner, [1])])
join = GpuHashJoin("(gpuhashjoin (rel (hashkey (var 1 int4) 1 int4)))", row_population_ratio=0.82).begin(km)
for name, cols in [("outer int4", [(0, 2, "int4")]), ("outer float8", [(0, 3, "float8")]),
                   ("inner int4", [(1, 2, "int4")]), ("inner key + payload", [(1, 1, "int4"), (1, 2, "int4")]),
                   ("all three", [(1, 2, "int4"), (0, 2, "int4"), (0, 3, "float8")])]:
    ts = []
    for it in range(4):
        t0 = time.perf_counter()
        joined, nitems = join.join_to_column(ds, cols)
        ts.append(time.perf_counter() - t0)
        joined.release()
    print("%-22s join+project %.2f ms (%d rows)" % (name, np.median(ts[1:]) * 1e3, nitems), flush=True)
t = []
for it in range(4):
    t0 = time.perf_counter(); r = join.join_chunk(ds, flags=1); t.append(time.perf_counter() - t0)
print("join alone %.2f ms" % (np.median(t[1:]) * 1e3))
