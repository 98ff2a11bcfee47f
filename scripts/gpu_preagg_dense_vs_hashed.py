"""C4's targets over an int4 key with many groups: the dense-id session (id-range roles) against the
hashed session (partition plan) on the same resident chunk"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pg_strom_amd import runtime
from pg_strom_amd.gpupreagg import GpuPreAgg
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
groups = [int(float(v)) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [10000, 25000, 50000, 100000, 1000000]
runtime.init()
spec = "(gpupreagg (key (var 1 int4)) (nrows) (psum (int8 (var 2 int4))) (psum (var 3 float8)))"
gen = torch.Generator(device="cuda"); gen.manual_seed(5)
x = torch.randint(-10**6, 10**6, (n,), dtype=torch.int32, device="cuda", generator=gen)
y = torch.rand(n, dtype=torch.float64, device="cuda", generator=gen) * 100.0
mm = lambda t: (t.min().item(), t.max().item())
for ngroups in groups:
    g = torch.randint(0, ngroups, (n,), dtype=torch.int32, device="cuda", generator=gen)
    cnt = torch.bincount(g.long(), minlength=ngroups).cpu().numpy()
    ds = runtime.DeviceStore.from_torch_columns(["int4", "int4", "float8"], [g, x, y], [mm(g), mm(x), mm(y)])
    del g
    for mode in ("dense", "hashed"):
        agg = GpuPreAgg(spec)
        agg.begin([(0, ngroups)]) if mode == "dense" else agg.begin_hashed(ngroups_hint=ngroups)
        ts = []
        for it in range(4):
            agg.reset() if mode == "dense" else None
            st, pfm = agg.fold(ds)
            assert st == 0
            ts.append(pfm["time_kern_exec_ns"])
        pr = agg.fetch()
        order = np.argsort(pr.column(0)[0])
        nf = 1 if mode == "dense" else 4
        ok = len(pr) == int((cnt > 0).sum()) and np.array_equal(pr.column(1)[0][order], cnt[cnt > 0] * nf)
        t = float(np.median(ts[1:])) * 1e-9
        print("ngroups=%d %-6s %.1f us  %.0f Mrows/s  (%d launches, packed/plan=%d) ok=%s"
              % (ngroups, mode, t * 1e6, n / t / 1e6, pfm["num_kern_exec"], pfm["num_kern_prep"], ok), flush=True)
        agg.end()
    ds.release()
