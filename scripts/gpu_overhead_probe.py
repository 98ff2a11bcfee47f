"""where does per-request time go?  resident 1e8-row chunk, requests back to back"""
import sys, time, collections
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpuscan import GpuScan, STROM_RESULTS_ON_DEVICE
QUAL = "(and (int4lt (var 1 int4) (param 0 int4)) (float8gt (var 2 float8) (param 1 float8)))"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
rng = np.random.default_rng(5)
a = rng.integers(0, 2**31, n, dtype=np.int64).astype(np.int32); b = rng.random(n)
runtime.init()
buf = kds.build_kds("column", [kds.Column("int4", a), kds.Column("float8", b)])
ds = runtime.DeviceStore.upload(buf)
scan = GpuScan(QUAL).begin(ext_params=[np.int32(2**30), 0.8]); scan.program.wait()
for depth in (1, 2, 3, 4, 8):
    for _ in range(3): scan.scan_chunk(ds, flags=STROM_RESULTS_ON_DEVICE)
    w = collections.deque(); kern = []
    t0 = time.perf_counter(); N = 40
    for i in range(N):
        w.append(scan.submit(ds, flags=STROM_RESULTS_ON_DEVICE))
        if len(w) >= depth: kern.append(scan.collect(w.popleft()).perfmon["time_kern_exec_ns"])
    while w: kern.append(scan.collect(w.popleft()).perfmon["time_kern_exec_ns"])
    dt = (time.perf_counter() - t0) / N
    print("depth=%d  per-request wall %.1f us   kernel %.1f us  -> %.0f Mrows/s" % (depth, dt*1e6, np.mean(kern)/1e3, n/dt/1e6), flush=True)
t0=time.perf_counter()
for i in range(200): p = scan.submit(ds, flags=STROM_RESULTS_ON_DEVICE); scan.collect(p)
print("serial submit+collect %.1f us" % ((time.perf_counter()-t0)/200*1e6))
