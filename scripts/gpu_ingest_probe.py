"""Device ingest throughput: ROW / ROW_FLAT / TUPSLOT -> COLUMN (C2 shape:
a int4, b float8).  Prints kernel time, source bytes/s and rows/s."""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pg_strom_amd import kds, runtime  # noqa: E402

runtime.init()
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
rng = np.random.default_rng(1)
cols = [kds.Column("int4", rng.integers(0, 2**31, n)), kds.Column("float8", rng.random(n))]
for fmt in ("row", "row_flat", "tupslot"):
    t0 = time.time()
    src = kds.build_kds(fmt, cols)
    t_build = time.time() - t0
    ds = runtime.DeviceStore.upload(src)
    best = None
    for _ in range(5):
        col, ns = ds.to_column([23, 701])
        col.release()
        best = ns if best is None else min(best, ns)
    ds.release()
    print("%-8s n=%d src=%.1f MB host_build=%.2fs kern=%.1f us  %.0f Mrows/s  src %.0f GB/s  out %.0f GB/s"
          % (fmt, n, len(src) / 1e6, t_build, best / 1e3, n / best * 1e3,
             len(src) / best, 12.0 * n / best), flush=True)
