"""C3 (1e8 fact x 1e6 dim, 80 % match) through the one-pass join kernel: 4-byte vs 3-byte slot array.
usage: gpu_c3_probe.py [nrows] [ndim]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpuhashjoin import GpuHashJoin, build_multihash
import bench

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
nd = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
runtime.init()
fact, (fk, a, b) = bench.c3_chunk_device(n, 0x5eed0003, nd)
nmatch = int((fk < nd).sum().item())
dkey, dgrp = bench.c3_dimension(nd, 10000)
km = build_multihash([(kds.build_kds("row_flat", [kds.Column("int4", dkey), kds.Column("int4", dgrp)]), [1])])
for label, env in (("3-byte slots", None), ("4-byte slots", "STROM_HASHJOIN_NO_NARROW_SLOTS")):
    if env:
        os.environ[env] = "1"
    join = GpuHashJoin(bench.C3_JOIN, row_population_ratio=0.8).begin(km)
    ts = []
    for _ in range(8):
        r = join.join_chunk(fact, flags=1)
        assert r.nitems == nmatch
        ts.append(r.perfmon["time_kern_exec_ns"] * 1e-3)
    join.end()
    t = float(np.median(ts[2:]))
    print("%-14s kernel %6.1f us  %.0f GB/s algorithmic (4 B/row + 8 B/match)" % (label, t, (4.0 * n + 8.0 * nmatch) / t / 1e3), flush=True)
    if env:
        del os.environ[env]
