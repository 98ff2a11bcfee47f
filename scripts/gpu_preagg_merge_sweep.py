"""C4 (1e8 rows, 1e4 groups, packed accumulators): slab-merge geometry sweep -- stripes of the
merge kernel (STROM_GPUPREAGG_MERGE_WS) against fold + merge time and the merge alone"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pg_strom_amd import runtime
from pg_strom_amd.gpupreagg import GpuPreAgg
runtime.init()
ds, cnt, sx = bench.c4_chunk_device(100_000_000, 5, 10_000)
for ws in ("0", "2", "4", "8", "16", "32", "64"):
    if ws == "0":
        os.environ.pop("STROM_GPUPREAGG_MERGE_WS", None)
    else:
        os.environ["STROM_GPUPREAGG_MERGE_WS"] = ws
    agg = GpuPreAgg(bench.C4_AGG).begin([(0, 10_000)])
    tot, mrg = [], []
    for it in range(8):
        agg.reset()
        st, pfm = agg.fold(ds)
        assert st == 0
        tot.append(pfm["time_kern_exec_ns"]); mrg.append(pfm["time_kern_proj_ns"])
    pr = agg.fetch()
    order = np.argsort(pr.column(0)[0])
    assert np.array_equal(pr.column(1)[0][order], cnt.cpu().numpy())
    print("merge_ws=%-3s fold+merge %.1f us   merge %.1f us   packed=%d" % (
        ws, np.median(tot[2:]) * 1e-3, np.median(mrg[2:]) * 1e-3, pfm["num_kern_prep"]), flush=True)
    agg.end()
