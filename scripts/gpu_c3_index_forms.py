"""C3 (fact x dim on int4, 80 % match) through each form of the probe index: DIRECT (the one-pass
kernels), KEYED and HASH through the general kernel (what a sparse, a multi-column or a text key takes).
usage: gpu_c3_index_forms.py [nrows] [ndim]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
# (a second, always-true key column makes the program a two-key one: the general kernel over the HASH index)
for label, spec, env in (("DIRECT, one-pass kernel", bench.C3_JOIN, {}),
                         ("DIRECT, general kernel", bench.C3_JOIN, {"STROM_HASHJOIN_NO_FAST": "1"}),
                         ("HASH, general kernel", bench.C3_JOIN, {"STROM_HASHJOIN_FORCE_HASH": "1", "STROM_HASHJOIN_NO_KEYED": "1"})):
    os.environ.update(env)
    try:
        join = GpuHashJoin(spec, row_population_ratio=0.8).begin(km)
        ts = []
        for _ in range(6):
            r = join.join_chunk(fact, flags=1)
            assert r.nitems == nmatch
            ts.append(r.perfmon["time_kern_exec_ns"] * 1e-3)
        print("%-26s index %-6s kernel %8.1f us  %.2f Grows/s" % (label, join.table_info()["mode"], float(np.median(ts[2:])), n / float(np.median(ts[2:])) / 1e3), flush=True)
        join.end()
    finally:
        for k in env:
            del os.environ[k]
