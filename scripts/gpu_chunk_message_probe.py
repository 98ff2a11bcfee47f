"""The reference's per-chunk GpuPreAgg message (strom_submit_gpupreagg_chunk): requests per second
and rows per second for the reference's own chunk size (325k rows, 15 MB as ROW) and larger ones,
host chunks (uploaded per request, as the reference does) and resident ones."""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpupreagg import GpuPreAgg

runtime.init()
spec = "(gpupreagg (key (var 1 int4)) (nrows) (psum (int8 (var 2 int4))) (psum (var 3 float8)))"
rng = np.random.default_rng(1)
for n, fmt in ((325_000, "row"), (325_000, "column"), (10_000_000, "column")):
    g = rng.integers(0, 10000, n).astype(np.int32)
    x = rng.integers(-10**6, 10**6, n).astype(np.int32)
    y = rng.random(n) * 100
    buf = kds.build_kds(fmt, [kds.Column("int4", g), kds.Column("int4", x), kds.Column("float8", y)])
    agg = GpuPreAgg(spec)
    agg.collect_chunk(agg.submit_chunk(buf))            # program build, first use
    for resident in (False, True):
        chunk = runtime.DeviceStore.upload(buf) if resident else buf
        for window in (1, 4):
            reps = 40 if n < 1_000_000 else 12
            t0 = time.perf_counter()
            pend = []
            for i in range(reps):
                pend.append(agg.submit_chunk(chunk, dest_rooms=10000))
                if len(pend) >= window:
                    st, pr = agg.collect_chunk(pend.pop(0))
                    assert st == 0 and len(pr) == 10000
            while pend:
                st, pr = agg.collect_chunk(pend.pop(0))
                assert st == 0 and len(pr) == 10000
            dt = (time.perf_counter() - t0) / reps
            print("%-7s %9d rows %-8s in flight %d: %7.1f us per message, %8.1f Mrows/s"
                  % (fmt, n, "resident" if resident else "host", window, dt * 1e6, n / dt / 1e6), flush=True)
        if resident:
            chunk.release()
