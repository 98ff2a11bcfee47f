"""GpuScan over heap pages (KDS_FORMAT_ROW / ROW_FLAT, what the reference ships), resident:
kernel time of the row-at-a-time kernel.  python scripts/gpu_scan_row_probe.py [rows]"""
import sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpuscan import GpuScan, STROM_RESULTS_ON_DEVICE

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
runtime.init()
rng = np.random.default_rng(7)
a = rng.integers(0, 2**31, n, dtype=np.int64).astype(np.int32)
b = rng.random(n)
k, c = np.int32(2**30), 0.8
want = int(np.count_nonzero((a < k) & (b > c)))
scan = GpuScan("(and (int4lt (var 1 int4) (param 0 int4)) (float8gt (var 2 float8) (param 1 float8)))").begin(ext_params=[k, c])
for fmt in ("row", "row_flat", "tupslot"):
    buf = kds.build_kds(fmt, [kds.Column("int4", a), kds.Column("float8", b)])
    ds = runtime.DeviceStore.upload(buf)
    ts = []
    for _ in range(6):
        res = scan.scan_chunk(ds, flags=STROM_RESULTS_ON_DEVICE)
        assert res.nitems == want
        ts.append(res.perfmon["time_kern_exec_ns"])
    t = float(np.median(ts[2:])) * 1e-9
    print("%-8s %d rows %.1f MB: kernel %.1f us, %.0f Mrows/s, %.0f GB/s of the chunk's bytes"
          % (fmt, n, len(buf) / 1e6, t * 1e6, n / t / 1e6, len(buf) / t / 1e9), flush=True)
    ds.release()
scan.end()
