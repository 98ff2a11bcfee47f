"""ablation of the GpuScan write path (diagnostic builds give WRONG results):
   0 = shipped kernel, 1 = no result stores, 2 = no reservation atomic"""
import os, sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpuscan import GpuScan, STROM_RESULTS_ON_DEVICE
QUAL = "(and (int4lt (var 1 int4) (param 0 int4)) (float8gt (var 2 float8) (param 1 float8)))"
n = 100_000_000
rng = np.random.default_rng(5)
a = rng.integers(0, 2**31, n, dtype=np.int64).astype(np.int32); b = rng.random(n)
runtime.init()
ds = runtime.DeviceStore.upload(kds.build_kds("column", [kds.Column("int4", a), kds.Column("float8", b)]))
for abl in ("0", "1", "2"):
    os.environ["STROM_GPUSCAN_ABLATE"] = abl
    os.environ["STROM_DIAGNOSTIC_BUILD"] = "1"      # the headers refuse *_ABLATE builds otherwise
    scan = GpuScan(QUAL).begin(ext_params=[np.int32(0), 0.0]); scan.program.wait()
    out = []
    for sa, sb in ((0.02, 0.5), (0.5, 0.8), (0.7, 0.3), (1.0, -1.0)):
        scan.parambuf = scan.codegen.parambuf([np.int32(min(int(2**31 * sa), 2**31 - 1)), sb])
        ts = [scan.scan_chunk(ds, flags=STROM_RESULTS_ON_DEVICE).perfmon["time_kern_exec_ns"] for _ in range(8)]
        out.append("%.1f" % (float(np.median(ts[2:])) / 1e3))
    print("ablate=%s  kernel us at sel 1%%/10%%/49%%/100%%: %s" % (abl, " / ".join(out)), flush=True)
    scan.end()
