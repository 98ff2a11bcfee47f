import sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pg_strom_amd import kds, runtime
runtime.init()
for fmt in ("row", "row_flat", "tupslot"):
    for n in (1, 2, 5):
        a = np.arange(n) + 1000
        b = np.arange(n) * 0.5 + 1
        cols = [kds.Column("int4", a), kds.Column("float8", b)]
        src = kds.build_kds(fmt, cols)
        ds = runtime.DeviceStore.upload(src)
        col, ns = ds.to_column([23, 701])
        img = col.download()
        dec = kds.decode_column_chunk(img)
        h = kds.KdsHead(img)
        print(fmt, n, "head", h.length, h.ncols, h.nitems, h.format, "ns", ns,
              "a", dec[0]["values"], dec[0]["notnull"], dec[0]["stat_flags"], dec[0]["minval"], dec[0]["maxval"],
              "b", dec[1]["values"].view(np.float64), dec[1]["notnull"], dec[1]["stat_flags"], flush=True)
        col.release(); ds.release()
