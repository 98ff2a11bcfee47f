"""sweep the GpuScan kernel geometry knobs on one resident 1e8-row chunk"""
import os, sys, time, collections, itertools
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpuscan import GpuScan, STROM_RESULTS_ON_DEVICE
QUAL = "(and (int4lt (var 1 int4) (param 0 int4)) (float8gt (var 2 float8) (param 1 float8)))"
n = 100_000_000
rng = np.random.default_rng(5)
a = rng.integers(0, 2**31, n, dtype=np.int64).astype(np.int32); b = rng.random(n)
runtime.init()
buf = kds.build_kds("column", [kds.Column("int4", a), kds.Column("float8", b)])
ds = runtime.DeviceStore.upload(buf)
del buf
configs = []
for block, quads, stage, nt in [(256,1,8192,1),(256,2,8192,1),(256,2,16384,1),(256,4,16384,1),(512,1,8192,1),
                                (512,2,16384,1),(128,2,8192,1),(256,1,4096,1)]:
    for percu in ("", ):
        for snt in (0, 1):
            configs.append((block, quads, stage, nt, percu, snt))
sels = ((0.02, 0.5), (0.5, 0.8), (0.7, 0.3))
for block, quads, stage, nt, percu, snt in configs:
    os.environ["STROM_GPUSCAN_BLOCK"] = str(block)
    os.environ["STROM_GPUSCAN_QUADS"] = str(quads)
    os.environ["STROM_GPUSCAN_STAGE"] = str(stage)
    os.environ["STROM_COLUMN_LOAD_NT"] = str(nt)
    os.environ["STROM_GPUSCAN_STORE_NT"] = str(snt)
    if percu: os.environ["STROM_GPUSCAN_BLOCKS_PER_CU"] = percu
    else: os.environ.pop("STROM_GPUSCAN_BLOCKS_PER_CU", None)
    scan = GpuScan(QUAL).begin(ext_params=[np.int32(0), 0.0])
    try:
        scan.program.wait()
    except Exception as e:
        print("config", block, quads, stage, "build failed", str(e)[:200]); continue
    out = []
    for sa, sb in sels:
        scan.parambuf = scan.codegen.parambuf([np.int32(int(2**31*sa)), sb])
        ts = []
        for it in range(10):
            res = scan.scan_chunk(ds, flags=STROM_RESULTS_ON_DEVICE)
            ts.append(res.perfmon["time_kern_exec_ns"])
        t = float(np.median(ts[2:])) * 1e-9
        byts = 12.0 * n + 4.0 * res.nitems
        out.append("sel=%.2f %.1fus %.0fGB/s" % (res.nitems / n, t * 1e6, byts / t / 1e9))
    print("block=%d quads=%d stage=%d nt=%d store_nt=%d percu=%s | %s" % (block, quads, stage, nt, snt, percu or "auto", " | ".join(out)), flush=True)
    scan.end()
