"""Q1-shaped GpuPreAgg (BASELINE configs[4] shape) over one resident 1e8-row chunk generated on the
device (bench.c5_chunk_device): fold kernel time for decimal columns and numeric images, with the
kernel family the environment selects (STROM_GPUPREAGG_PRIV_LDS, STROM_GPUPREAGG_NO_REG, ...).
usage: gpu_q1_probe.py [rows] [decimal|numeric|both]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pg_strom_amd import runtime
from pg_strom_amd.gpupreagg import GpuPreAgg
import bench

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
which = sys.argv[2] if len(sys.argv) > 2 else "both"
runtime.init()
for label, decimal, spec in (("decimal columns", True, bench.Q1_AGG), ("numeric images", False, bench.Q1_AGG_NUMERIC)):
    if which not in ("both", label.split()[0]):
        continue
    ds, ref = bench.c5_chunk_device(n, 0x5eed0005, decimal)
    agg = GpuPreAgg(spec).begin([(65, 18), (70, 10)])
    agg.census(ds)
    agg.compact()
    ts = []
    for _ in range(7):
        agg.reset()
        st, pfm = agg.fold(ds)
        assert st == 0
        ts.append((pfm["time_kern_exec_ns"] - pfm["time_kern_proj_ns"]) * 1e-3)
    pr = agg.fetch()
    bench.q1_check(pr, ref)
    t = float(np.median(ts[2:]))
    print("%-16s fold %7.1f us  %.0f GB/s (38 B/row)  frac %.3f  checked_folds=%d" % (label, t, 38.0 * n / t / 1e3, 38.0 * n / t / 1e3 / 8000, agg.checked_folds()), flush=True)
    agg.end()
    ds.release()
    del ds
    torch.cuda.empty_cache()
