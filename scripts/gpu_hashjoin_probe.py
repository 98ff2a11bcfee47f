"""GpuHashJoin C3-shape timing: 1e8 fact x 1e6 dim on int4 (80% hit), resident fact chunk"""
import os, sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpuhashjoin import GpuHashJoin, build_multihash, STROM_RESULTS_ON_DEVICE
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
nd = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
runtime.init()
rng = np.random.default_rng(3)
pk = rng.permutation(nd).astype(np.int32)
inner = kds.build_kds("row_flat", [kds.Column("int4", pk), kds.Column("int4", np.arange(nd, dtype=np.int32))])
t0 = time.time(); km = build_multihash([(inner, [1])]); print("host build %.2fs, %d MB" % (time.time() - t0, len(km) >> 20))
fk = rng.integers(0, int(nd * 1.25), nf, dtype=np.int64).astype(np.int32)
nmatch = int(np.count_nonzero(fk < nd))
ds = runtime.DeviceStore.upload(kds.build_kds("column", [kds.Column("int4", fk)]))
for label, env in (("direct+fast", {}), ("direct generic", {"STROM_HASHJOIN_NO_FAST": "1"}),
                   ("keyed one-pass", {"STROM_HASHJOIN_FORCE_HASH": "1"}),
                   ("keyed generic", {"STROM_HASHJOIN_FORCE_HASH": "1", "STROM_HASHJOIN_NO_FAST": "1"}),
                   ("hashed generic", {"STROM_HASHJOIN_FORCE_HASH": "1", "STROM_HASHJOIN_NO_KEYED": "1"})):
    for k in ("STROM_HASHJOIN_NO_FAST", "STROM_HASHJOIN_FORCE_HASH", "STROM_HASHJOIN_NO_KEYED"):
        os.environ.pop(k, None)
    os.environ.update(env)
    t0 = time.time()
    join = GpuHashJoin("(gpuhashjoin (rel (hashkey (var 1 int4) 1 int4)))").begin(km)
    tb = time.time() - t0
    ts = []
    for it in range(8):
        res = join.join_chunk(ds, flags=STROM_RESULTS_ON_DEVICE)
        assert os.environ.get('STROM_HASHJOIN_ABLATE') or (res.errcode == 0 and res.nitems == nmatch), (res.errcode, res.nitems, nmatch)
        ts.append(res.perfmon["time_kern_exec_ns"])
    t = float(np.median(ts[2:])) * 1e-9
    byts = 4.0 * nf + 8.0 * nmatch
    print("%-15s %s table+index %.2fs kern=%.1f us  %.0f Mrows/s  %.0f GB/s algorithmic (%.1f%% of 8TB/s)" % (
        label, join.table_info(1), tb, t * 1e6, nf / t / 1e6, byts / t / 1e9, byts / t / 8e12 * 100), flush=True)
    join.end()
