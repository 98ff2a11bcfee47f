"""GpuHashJoin with a dimension small enough for LDS: the slot array staged in LDS
(gpuhashjoin_main_fast_lds) against the same probe through L2 (gpuhashjoin_main_fast),
1e8 fact rows generated on the device, every row probes (80 % find a partner)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpuhashjoin import GpuHashJoin, build_multihash, STROM_RESULTS_ON_DEVICE
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
runtime.init()
for nd in (2_000, 25_000):
    span = int(nd * 1.25)
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    fk = torch.randint(0, span, (nf,), dtype=torch.int32, device="cuda", generator=g)
    nmatch = int((fk < nd).sum().item())
    ds = runtime.DeviceStore.from_torch_columns(["int4"], [fk], [(0, span - 1)])
    pk = np.random.default_rng(3).permutation(nd).astype(np.int32)
    km = build_multihash([(kds.build_kds("row_flat", [kds.Column("int4", pk), kds.Column("int4", pk % 7)]), [1])])
    for label, env in (("LDS q8", {}), ("LDS q4", {"STROM_HASHJOIN_LDS_QUADS": "4"}),
                       ("LDS q16", {"STROM_HASHJOIN_LDS_QUADS": "16"}), ("L2", {"STROM_HASHJOIN_NO_LDS_SLOTS": "1"})):
        for k in ("STROM_HASHJOIN_LDS_QUADS", "STROM_HASHJOIN_NO_LDS_SLOTS"):
            os.environ.pop(k, None)
        os.environ.update(env)
        join = GpuHashJoin("(gpuhashjoin (rel (hashkey (var 1 int4) 1 int4)))", row_population_ratio=0.82).begin(km)
        ts = []
        for it in range(7):
            res = join.join_chunk(ds, flags=STROM_RESULTS_ON_DEVICE)
            assert res.errcode == 0 and res.nitems == nmatch, (res.errcode, res.nitems, nmatch)
            ts.append(res.perfmon["time_kern_exec_ns"])
        t = float(np.median(ts[2:])) * 1e-9
        byts = 4.0 * nf + 8.0 * nmatch
        print("dim %6d %-8s staged=%d kern=%.1f us  %.0f Mrows/s  %.0f GB/s algorithmic (%.1f%% of 8TB/s)" % (
            nd, label, res.perfmon["num_kern_prep"], t * 1e6, nf / t / 1e6, byts / t / 1e9, byts / t / 8e12 * 100), flush=True)
        join.end()
    ds.release()
