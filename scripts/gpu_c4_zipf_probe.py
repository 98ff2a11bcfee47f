"""C4 (GROUP BY int4, 1e4 groups, COUNT / SUM(int4) / SUM(float8)) with uniform and Zipf-1.0 keys
("also a Zipf-1.0 variant") on one resident 1e8-row chunk: fold kernel time of
gpupreagg_packed_column.  usage: gpu_c4_zipf_probe.py [rows] [groups]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pg_strom_amd import runtime
from pg_strom_amd.gpupreagg import GpuPreAgg
import bench

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
ng = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10000
runtime.init()
g = torch.Generator(device="cuda")
g.manual_seed(0x5eed0004)
for label in ("uniform", "zipf-1.0", "zipf-1.0, keys shuffled"):
    if label == "uniform":
        grp = torch.randint(0, ng, (n,), dtype=torch.int32, device="cuda", generator=g)
    else:
        # P(k) ~ 1 / (k + 1): inverse CDF by searching the cumulative weights
        w = 1.0 / torch.arange(1, ng + 1, dtype=torch.float64, device="cuda")
        cdf = torch.cumsum(w, 0)
        u = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * cdf[-1]
        grp = torch.searchsorted(cdf, u).clamp_(max=ng - 1).to(torch.int32)
        if "shuffled" in label:
            perm = torch.randperm(ng, device="cuda", generator=g).to(torch.int32)
            grp = perm[grp.long()]
    x = torch.randint(-10**6, 10**6, (n,), dtype=torch.int32, device="cuda", generator=g)
    y = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 100
    ds = runtime.DeviceStore.from_torch_columns(["int4", "int4", "float8"], [grp, x, y],
                                                [bench._minmax(grp), bench._minmax(x), bench._minmax(y)])
    cnt = torch.bincount(grp.long(), minlength=ng).cpu().numpy()
    sx = torch.zeros(ng, dtype=torch.int64, device="cuda").index_add_(0, grp.long(), x.long()).cpu().numpy()
    agg = GpuPreAgg(bench.C4_AGG).begin([(0, ng)])
    ts, packed = [], 0
    for _ in range(7):
        agg.reset()
        st, pfm = agg.fold(ds)
        assert st == 0
        ts.append((pfm["time_kern_exec_ns"] - pfm["time_kern_proj_ns"]) * 1e-3)
        packed = pfm["num_kern_prep"]
    pr = agg.fetch()
    order = np.argsort(pr.column(0)[0])
    assert np.array_equal(pr.column(1)[0][order], cnt[cnt > 0]) and np.array_equal(pr.column(2)[0][order], sx[cnt > 0])
    t = float(np.median(ts[2:]))
    print("%-26s fold %6.1f us  %.0f GB/s (16 B/row)  frac %.3f  packed=%d  hottest group %.1f %% of the rows"
          % (label, t, 16.0 * n / t / 1e3, 16.0 * n / t / 1e3 / 8000, packed, 100.0 * cnt.max() / n), flush=True)
    agg.end()
    ds.release()
    del ds, grp, x, y
    torch.cuda.empty_cache()
