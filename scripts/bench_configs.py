"""Kernel-level numbers for every BASELINE config on one MI355X, one JSON
line per case (inputs resident in HBM, HIP events on the launch stream):

  C2  GpuScan 1e8 rows (a int4, b float8) WHERE a<k AND b>c: selectivity
      1/10/50 %, a 5 %-NULL variant, the ROW-format variant (what the
      reference feeds) and ROW -> COLUMN device ingest
  C3  GpuHashJoin 1e8 fact x 1e6 dim on int4, 80 % hit
  C4  GpuPreAgg GROUP BY int4 (1e4 groups) COUNT/SUM/SUM(float8), uniform + Zipf(1.0)
  C5  TPC-H Q1 shape: 2 char(1) keys, 4 numerics, date filter, 9 partials

Algorithmic bytes follow SURVEY.md section 8(d).  Not the driver's bench
(that is bench.py); the output is committed under profiles/.
"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pg_strom_amd import kds, runtime  # noqa: E402
from pg_strom_amd.gpuhashjoin import GpuHashJoin, build_multihash  # noqa: E402
from pg_strom_amd.gpupreagg import GpuPreAgg  # noqa: E402
from pg_strom_amd.gpuscan import GpuScan, STROM_RESULTS_ON_DEVICE  # noqa: E402

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
PEAK = 8.0e12
REPS = 8


def emit(config, kernel, rows, ns, alg_bytes, **extra):
    t = ns * 1e-9
    line = dict(config=config, kernel=kernel, rows=rows, kernel_us=round(ns / 1e3, 1),
                mrows_s=round(rows / t / 1e6), alg_bytes=alg_bytes,
                alg_gbs=round(alg_bytes / t / 1e9), frac_of_8tbs=round(alg_bytes / t / PEAK, 4))
    line.update(extra)
    print(json.dumps(line), flush=True)


def med(ts):
    return float(np.median(ts[2:]))


def c2():
    rng = np.random.default_rng(0x5eed0002)
    a = rng.integers(0, 2**31, N, dtype=np.int64).astype(np.int32)
    b = rng.random(N)
    qual = "(and (int4lt (var 1 int4) (param 0 int4)) (float8gt (var 2 float8) (param 1 float8)))"
    ds = runtime.DeviceStore.upload(kds.build_kds("column", [kds.Column("int4", a), kds.Column("float8", b)]))
    for s1, s2 in ((0.1, 0.1), (0.5, 0.2), (0.7, 0.7)):
        scan = GpuScan(qual).begin(ext_params=[np.int32(2**31 * s1), 1.0 - s2])
        ts = []
        for _ in range(REPS):
            res = scan.scan_chunk(ds, flags=STROM_RESULTS_ON_DEVICE)
            ts.append(res.perfmon["time_kern_exec_ns"])
        scan.end()
        emit("C2 sel=%.0f%%" % (100.0 * res.nitems / N), "gpuscan_qual_column", N, med(ts),
             12.0 * N + 4.0 * res.nitems, selected=res.nitems, format="COLUMN")
    ds.release()
    # 5 % NULLs in both columns: + N/8 bytes of bitmap per nullable column
    an = rng.random(N) < 0.05
    bn = rng.random(N) < 0.05
    ds = runtime.DeviceStore.upload(kds.build_kds("column", [kds.Column("int4", a, an), kds.Column("float8", b, bn)]))
    scan = GpuScan(qual).begin(ext_params=[np.int32(2**31 * 0.5), 0.8])
    ts = []
    for _ in range(REPS):
        res = scan.scan_chunk(ds, flags=STROM_RESULTS_ON_DEVICE)
        ts.append(res.perfmon["time_kern_exec_ns"])
    scan.end()
    ds.release()
    emit("C2 sel=%.0f%% 5%% NULLs" % (100.0 * res.nitems / N), "gpuscan_qual_column", N, med(ts),
         12.0 * N + 2 * N / 8.0 + 4.0 * res.nitems, selected=res.nitems, format="COLUMN")
    # the reference's own input: heap pages (ROW), generic kernel, then device ingest
    n = min(N, 10_000_000)
    cols = [kds.Column("int4", a[:n]), kds.Column("float8", b[:n])]
    src = kds.build_kds("row", cols)
    ds = runtime.DeviceStore.upload(src)
    scan = GpuScan(qual).begin(ext_params=[np.int32(2**31 * 0.5), 0.8])
    ts = []
    for _ in range(REPS):
        res = scan.scan_chunk(ds, flags=STROM_RESULTS_ON_DEVICE)
        ts.append(res.perfmon["time_kern_exec_ns"])
    emit("C2 sel=%.0f%% ROW format" % (100.0 * res.nitems / n), "gpuscan_qual_generic", n, med(ts),
         12.0 * n + 4.0 * res.nitems, selected=res.nitems, format="ROW", chunk_bytes=len(src),
         note="algorithmic bytes as for COLUMN; the chunk itself is %.1f B/row" % (len(src) / n))
    ts = []
    for _ in range(REPS):
        col, ns = ds.to_column([23, 701])
        ts.append(ns)
        if _ < REPS - 1:
            col.release()
    emit("ingest ROW->COLUMN", "ingest_to_column", n, med(ts), float(len(src)) + 12.0 * n,
         chunk_bytes=len(src), note="bytes = source chunk read + 12 B/row written")
    ts = []
    for _ in range(REPS):
        res = scan.scan_chunk(col, flags=STROM_RESULTS_ON_DEVICE)
        ts.append(res.perfmon["time_kern_exec_ns"])
    emit("C2 on ingested chunk", "gpuscan_qual_column", n, med(ts), 12.0 * n + 4.0 * res.nitems,
         selected=res.nitems, format="COLUMN (device ingest)")
    scan.end()
    col.release()
    ds.release()


def c2host():
    """the boundary hands over HOST chunks: upload + scan per request (PCIe-inclusive)"""
    rng = np.random.default_rng(0x5eed0002)
    n = 25_000_000
    a = rng.integers(0, 2**31, n, dtype=np.int64).astype(np.int32)
    b = rng.random(n)
    qual = "(and (int4lt (var 1 int4) (param 0 int4)) (float8gt (var 2 float8) (param 1 float8)))"
    for fmt in ("column", "row"):
        rows = n if fmt == "column" else 10_000_000
        buf = kds.build_kds(fmt, [kds.Column("int4", a[:rows]), kds.Column("float8", b[:rows])])
        runtime.lib.strom_pin_host_range(buf.ctypes.data, len(buf))
        scan = GpuScan(qual).begin(ext_params=[np.int32(2**31 * 0.5), 0.8])
        scan.scan_chunk(buf)
        t0 = time.perf_counter()
        reps = 6
        sel = 0
        for res in scan.scan_chunks([buf] * reps):
            sel = res.nitems
        dt = (time.perf_counter() - t0) / reps
        scan.end()
        runtime.lib.strom_unpin_host_range(buf.ctypes.data)
        print(json.dumps(dict(config="C2 host chunk, %s format (PCIe-inclusive)" % fmt.upper(), rows=rows,
                              chunk_bytes=len(buf), wall_us_per_chunk=round(dt * 1e6, 1),
                              mrows_s=round(rows / dt / 1e6), upload_gbs=round(len(buf) / dt / 1e9, 1),
                              selected=sel)), flush=True)


def c3():
    nd = 1_000_000
    rng = np.random.default_rng(0x5eed0003)
    pk = rng.permutation(nd).astype(np.int32)
    inner = kds.build_kds("row_flat", [kds.Column("int4", pk), kds.Column("int4", np.arange(nd, dtype=np.int32))])
    t0 = time.time()
    km = build_multihash([(inner, [1])])
    t_build = time.time() - t0
    fk = rng.integers(0, int(nd * 1.25), N, dtype=np.int64).astype(np.int32)
    nmatch = int(np.count_nonzero(fk < nd))
    ds = runtime.DeviceStore.upload(kds.build_kds("column", [kds.Column("int4", fk)]))
    join = GpuHashJoin("(gpuhashjoin (rel (hashkey (var 1 int4) 1 int4)))").begin(km)
    ts = []
    for _ in range(REPS):
        res = join.join_chunk(ds, flags=STROM_RESULTS_ON_DEVICE)
        assert res.errcode == 0 and res.nitems == nmatch
        ts.append(res.perfmon["time_kern_exec_ns"])
    emit("C3 1e6-row dim, 80% hit", "gpuhashjoin_main_fast", N, med(ts), 4.0 * N + 8.0 * nmatch,
         matches=nmatch, table=join.table_info(1), host_table_build_s=round(t_build, 2))
    join.end()
    ds.release()


def c4():
    rng = np.random.default_rng(0x5eed0004)
    x = rng.integers(-10**6, 10**6, N, dtype=np.int64).astype(np.int32)
    y = rng.random(N) * 100
    spec = "(gpupreagg (key (var 1 int4)) (nrows) (psum (int8 (var 2 int4))) (psum (var 3 float8)))"
    ng = 10000
    zipf_p = 1.0 / np.arange(1, ng + 1)
    zipf_p /= zipf_p.sum()
    for label, g in (("uniform", rng.integers(0, ng, N, dtype=np.int64).astype(np.int32)),
                     ("Zipf(1.0)", rng.choice(ng, N, p=zipf_p).astype(np.int32))):
        ds = runtime.DeviceStore.upload(kds.build_kds("column", [kds.Column("int4", g), kds.Column("int4", x),
                                                                 kds.Column("float8", y)]))
        agg = GpuPreAgg(spec).begin([(0, ng)])
        ts, tm = [], []
        for _ in range(REPS):
            st, pfm = agg.fold(ds)
            assert st == 0
            ts.append(pfm["time_kern_exec_ns"])
            tm.append(pfm["time_kern_proj_ns"])
        pr = agg.fetch()
        order = np.argsort(pr.column(0)[0])
        assert np.array_equal(pr.column(1)[0][order], np.bincount(g, minlength=ng) * REPS)
        emit("C4 1e4 groups %s" % label, "gpupreagg_dense_column+merge", N, med(ts), 16.0 * N,
             merge_us=round(med(tm) / 1e3, 1))
        agg.end()
        ds.release()


def c5(typmod=True):
    rng = np.random.default_rng(0x5eed0005)
    n = N
    rf = rng.choice(np.array([65, 78, 82], dtype=np.int8), n)
    ls = rng.choice(np.array([70, 79], dtype=np.int8), n)
    cols = [kds.Column("char1", rf), kds.Column("char1", ls),
            kds.numeric_from_scaled(rng.integers(1, 51, n), 0),
            kds.numeric_from_scaled(rng.integers(90000, 10494951, n), 2),
            kds.numeric_from_scaled(rng.integers(0, 11, n), 2),
            kds.numeric_from_scaled(rng.integers(0, 9, n), 2),
            kds.Column("date", rng.integers(-2922, -2922 + 2526, n).astype(np.int32))]
    qty, prc, dsc, txx = (("(var 3 numeric 0)", "(var 4 numeric 2)", "(var 5 numeric 2)", "(var 6 numeric 2)")
                          if typmod else
                          ("(var 3 numeric)", "(var 4 numeric)", "(var 5 numeric)", "(var 6 numeric)"))
    one_minus_d = "(numeric_sub (const numeric 1) %s)" % dsc
    one_plus_t = "(numeric_add (const numeric 1) %s)" % txx
    disc_price = "(numeric_mul %s %s)" % (prc, one_minus_d)
    spec = ("(gpupreagg (qual (date_le (var 7 date) (const date '1998-09-02')))"
            " (key (var 1 char1)) (key (var 2 char1))"
            " (psum %s 0) (psum %s 2) (psum %s 4) (psum (numeric_mul %s %s) 6)"
            " (nrows (isnotnull %s)) (nrows (isnotnull %s))"
            " (psum %s 2) (nrows (isnotnull %s)) (nrows))"
            % (qty, prc, disc_price, disc_price, one_plus_t, qty, prc, dsc, dsc))
    ds = runtime.DeviceStore.upload(kds.build_kds("column", cols))
    agg = GpuPreAgg(spec).begin([(65, 18), (70, 10)])
    t0 = time.time()
    agg.census(ds)
    nslots = agg.compact()
    t_census = time.time() - t0
    ts, tm = [], []
    for _ in range(REPS):
        st, pfm = agg.fold(ds)
        assert st == 0, st
        ts.append(pfm["time_kern_exec_ns"])
        tm.append(pfm["time_kern_proj_ns"])
    pr = agg.fetch()
    emit("C5 Q1 shape (6 groups, 9 partials)%s" % (", typmod scales" if typmod else ", scale-less numerics"),
         "gpupreagg_*_column+merge", n, med(ts), 38.0 * n,
         merge_us=round(med(tm) / 1e3, 1), groups=len(pr.column(0)[0]), table_slots=nslots,
         census_compact_ms=round(t_census * 1e3, 2))
    agg.end()
    ds.release()


if __name__ == "__main__":
    runtime.init()
    which = sys.argv[2].split(",") if len(sys.argv) > 2 else ["c2", "c3", "c4", "c5"]
    for name in which:
        {"c2": c2, "c2host": c2host, "c3": c3, "c4": c4, "c5": c5, "c5dyn": lambda: c5(False)}[name]()
