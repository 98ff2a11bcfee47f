"""GpuHashJoin on a text key (HASH index, texteq on every candidate) and GpuScan with a text qual over
COLUMN chunks: kernel times per row for the record (no reference figure exists for either).
usage: gpu_textjoin_probe.py [outer rows] [dimension rows]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpuhashjoin import GpuHashJoin, build_multihash
from pg_strom_amd.gpuscan import GpuScan

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_000_000
nd = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000
runtime.init()
rng = np.random.default_rng(1)
words = [b"cust#%09d" % i for i in range(nd)]                      # 14-byte keys, short headers
inner = kds.build_kds("row_flat", [kds.Column("text", words), kds.Column("int4", np.arange(nd, dtype=np.int32))])
pick = rng.integers(0, int(nd * 1.25), n)
otxt = [b"cust#%09d" % i for i in pick]
t0 = time.perf_counter()
outer = kds.build_kds("column", [kds.Column("text", otxt), kds.Column("int8", np.arange(n, dtype=np.int64))])
print("outer COLUMN chunk: %d rows, %.1f MB (%.1f B/row), built in %.1f s" % (n, len(outer) / 1e6, len(outer) / n, time.perf_counter() - t0))
ds = runtime.DeviceStore.upload(outer)
want = int((pick < nd).sum())
join = GpuHashJoin("(gpuhashjoin (rel (hashkey (var 1 text) 1 text)))", row_population_ratio=0.85).begin(build_multihash([(inner, [1])]))
ts = []
for _ in range(5):
    r = join.join_chunk(ds, flags=1)
    assert r.nitems == want, (r.nitems, want)
    ts.append(r.perfmon["time_kern_exec_ns"] * 1e-3)
print("join on text key   %8.1f us  %.2f Grows/s  index %s" % (np.median(ts[1:]), n / np.median(ts[1:]) / 1e3, join.table_info()["mode"]))
join.end()
scan = GpuScan("(texteq (var 1 text) (const text 'cust#000000042'))").begin()
ts = []
for _ in range(5):
    r = scan.scan_chunk(ds)
    ts.append(r.perfmon["time_kern_exec_ns"] * 1e-3)
print("scan texteq        %8.1f us  %.2f Grows/s  %.0f GB/s of the chunk's bytes  (%d rows pass)"
      % (np.median(ts[1:]), n / np.median(ts[1:]) / 1e3, len(outer) / np.median(ts[1:]) / 1e3, r.nitems))
scan.end()
ds.release()
