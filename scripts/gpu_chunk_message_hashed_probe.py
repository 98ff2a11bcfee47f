"""The per-chunk GpuPreAgg message over a key without dense ids (sparse int8): the library makes a
hashed session per message -- table, fold (partition plan for large chunks), export, release."""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpupreagg import GpuPreAgg

runtime.init()
spec = "(gpupreagg (key (var 1 int8)) (nrows) (psum (int8 (var 2 int4))) (psum (var 3 float8)))"
rng = np.random.default_rng(1)
for n, ngroups, hint in ((325_000, 1000, 0), (10_000_000, 1000, 0), (10_000_000, 100_000, 0), (10_000_000, 100_000, 100_000),
                         (50_000_000, 1_000_000, 1_000_000)):
    g = rng.integers(0, ngroups, n).astype(np.int64)
    k = g * 1000003 * 65537 - 2**59
    x = rng.integers(-10**6, 10**6, n).astype(np.int32)
    y = rng.random(n) * 100
    chunk = runtime.DeviceStore.upload(kds.build_kds("column", [kds.Column("int8", k), kds.Column("int4", x), kds.Column("float8", y)]))
    agg = GpuPreAgg(spec)
    ng = len(np.unique(g))
    agg.collect_chunk(agg.submit_chunk(chunk, dest_rooms=ng, num_groups=hint))            # program build, first use
    reps = 8
    t0 = time.perf_counter()
    for i in range(reps):
        st, pr = agg.collect_chunk(agg.submit_chunk(chunk, dest_rooms=ng, num_groups=hint))
        assert st == 0 and len(pr) == ng
    dt = (time.perf_counter() - t0) / reps
    print("%9d rows resident, %8d groups, planner's num_groups=%-8d: %9.1f us per message, %8.1f Mrows/s"
          % (n, ng, hint, dt * 1e6, n / dt / 1e6), flush=True)
    chunk.release()
