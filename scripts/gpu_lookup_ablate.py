"""scan+join+groupby as ONE kernel (strom_submit_gpupreagg_lookup): where the time goes.
The same resident 1e8-row fact chunk folded by diagnostic builds of the program that leave
work out (GPUPREAGG_ABLATE: 1 no accumulate, 2 no probe, 3 neither -- WRONG RESULTS, measurement
only, hence STROM_DIAGNOSTIC_BUILD), then the real build at several WHERE selectivities.
usage: gpu_lookup_ablate.py [nrows] [ndim] [ngroups]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpuhashjoin import GpuHashJoin, build_multihash
from pg_strom_amd.gpupreagg import GpuPreAgg
import bench

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
nd = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
ngroups = int(float(sys.argv[3])) if len(sys.argv) > 3 else 10000
runtime.init()
ds, (fk, a, b) = bench.c3_chunk_device(n, 0x5eed0003, nd)
dkey, dgrp = bench.c3_dimension(nd, ngroups)
km = build_multihash([(kds.build_kds("row_flat", [kds.Column("int4", dkey), kds.Column("int4", dgrp)]), [1])])
join = GpuHashJoin(bench.C3_JOIN, row_population_ratio=0.8).begin(km)


def run(label, sel=0.5):
    ext = [np.int32(int(2**31 * sel) - 1), 0.0]
    agg = GpuPreAgg(bench.CHAIN_AGG).begin([(0, ngroups)], ext_params=ext)
    agg.program.wait()
    ts = []
    for it in range(6):
        agg.reset()
        st, pfm = agg.collect(agg.submit_lookup(join, ds, [(1, 2, "int4"), (0, 2, "int4"), (0, 3, "float8")]))
        assert st == 0
        ts.append(((pfm["time_kern_exec_ns"] - pfm["time_kern_proj_ns"]) * 1e-3, pfm["time_kern_proj_ns"] * 1e-3,
                   pfm["num_kern_prep"]))
    agg.end()
    f = np.median([t[0] for t in ts[1:]])
    m = np.median([t[1] for t in ts[1:]])
    print("%-34s fold %6.0f us  merge %4.0f us  packed=%d" % (label, f, m, ts[-1][2]), flush=True)


os.environ["STROM_DIAGNOSTIC_BUILD"] = "1"
for abl, what in ((0, "real"), (1, "no accumulate"), (2, "no probe"), (3, "neither: stream + qual")):
    os.environ["STROM_GPUPREAGG_ABLATE"] = str(abl)
    run("ABLATE=%d (%s)" % (abl, what))
del os.environ["STROM_GPUPREAGG_ABLATE"]
del os.environ["STROM_DIAGNOSTIC_BUILD"]
for sel in (1.0, 0.5, 0.1, 0.01):
    run("WHERE keeps %.0f %%" % (sel * 100), sel)
join.end()
ds.release()
