#!/usr/bin/env python3
"""
bench.py -- headline measurement of the per-chunk path (BASELINE.json metric:
"Mrows/sec scan+hashjoin+groupby on 1e9-row synth; achieved HBM GB/s vs peak").

Headline (`value`, `roofline`, the timed region): GpuScan over the rank's
resident table -- 1e9 rows per GPU of (a int4, b float8) in ten 1e8-row
KDS_FORMAT_COLUMN chunks (a chunk's `length` is 32-bit as in the reference, so
1e9 rows are 10 chunks), WHERE a < k AND b > c, every chunk submitted through
the C ABI (strom_submit_gpuscan) with results[] left in HBM.  A "step" is one
pass over all chunks; inputs are resident before the timed region starts.
With N GPUs the table is N x 1e9 rows sharded by row range, one process per
GPU, no data-path collective (weak scaling).

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks
itself (fresh child processes, before this process touches the GPU) and relays
rank 0's JSON line; under torch.distributed.run it is one of the ranks.

Next to the headline, in the same JSON line:
  roofline      algorithmic bytes per launch / mean kernel time per launch (HIP
                events on the launch stream, strom_perfmon.time_kern_exec_ns),
                against the nominal 8 TB/s and against the box's own
                streaming-read rate (strom_membw_probe)
  cpu_baseline  the CPU oracle (oracle/, a port: tuple-at-a-time over ROW
                format heap pages, expression tree interpreted per row -- the
                shape of PostgreSQL's SeqScan + ExecQual), bounded sample
  gpupreagg_c4  (every N) a second barrier-bracketed region, BASELINE
                configs[3]: every rank folds its 1e9-row range (GROUP BY int4,
                1e4 groups) into its resident table and the tables are merged by
                strom_gpupreagg_allreduce (RCCL over xGMI); merge time reported
                separately, merged result checked; roofline + CPU baselines
  operators     (N=1) GpuHashJoin C3 and scan+join+group-by in one pass: each
                with its own roofline block and CPU baselines (1 core, all cores)
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

C2_QUAL = "(and (int4lt (var 1 int4) (param 0 int4)) (float8gt (var 2 float8) (param 1 float8)))"
C3_JOIN = "(gpuhashjoin (rel (hashkey (var 1 int4) 1 int4)))"
HASHED_AGG = "(gpupreagg (key (var 1 int8)) (nrows) (psum (int8 (var 2 int4))) (psum (var 3 float8)))"
C4_AGG = "(gpupreagg (key (var 1 int4)) (nrows) (psum (int8 (var 2 int4))) (psum (var 3 float8)))"
CHAIN_QUAL = "(and (int4lt (var 2 int4) (param 0 int4)) (float8gt (var 3 float8) (param 1 float8)))"
CHAIN_AGG = "(gpupreagg (qual " + CHAIN_QUAL + ") " + C4_AGG[len("(gpupreagg "):]
CHAIN_EXT = [np.int32(2**30), 0.0]
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
CPU_CHUNK_ROWS = 325_000       # the reference's 15 MB ROW chunk


# --------------------------------------------------------------------------- #
# synthetic inputs
# --------------------------------------------------------------------------- #
def c2_columns(nrows, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 2**31, nrows, dtype=np.int32)
    b = rng.random(nrows)
    return a, b


def c3_columns(nrows, seed, nd=1_000_000):
    rng = np.random.default_rng(seed)
    fk = rng.integers(0, int(nd * 1.25), nrows, dtype=np.int32)      # 80 % of the keys have a partner
    a = rng.integers(0, 2**31, nrows, dtype=np.int32)
    b = rng.random(nrows)
    return fk, a, b


def c3_dimension(nd=1_000_000, ngroups=10_000):
    dkey = np.random.default_rng(0x5eed0013).permutation(nd).astype(np.int32)
    return dkey, (dkey % ngroups).astype(np.int32)


def c4_columns(nrows, seed, ngroups=10_000):
    rng = np.random.default_rng(seed)
    g = rng.integers(0, ngroups, nrows, dtype=np.int32)
    x = rng.integers(-10**6, 10**6, nrows, dtype=np.int32)
    y = rng.random(nrows) * 100.0
    return g, x, y


# ---- the same distributions generated ON the device (torch's CUDA generator): a 1e8-row
# chunk takes ~10 s of numpy on a host core and ~10 ms here; the chunk is assembled in HBM
# (strom_kds_column_head + device-to-device copies) and adopted with strom_dstore_wrap.
# Expected answers come from torch ops on the same tensors, never from this build's kernels.
def _gen(seed):
    import torch
    g = torch.Generator(device="cuda")
    g.manual_seed(int(seed))
    return g


def _minmax(t):
    return (t.min().item(), t.max().item())


def c2_chunk_device(nrows, seed, k, c):
    import torch
    from pg_strom_amd import runtime
    g = _gen(seed)
    a = torch.randint(0, 2**31, (nrows,), dtype=torch.int32, device="cuda", generator=g)
    b = torch.rand(nrows, dtype=torch.float64, device="cuda", generator=g)
    nsel = int(((a < int(k)) & (b > float(c))).sum().item())
    ds = runtime.DeviceStore.from_torch_columns(["int4", "float8"], [a, b], [_minmax(a), _minmax(b)])
    return ds, nsel


def c3_chunk_device(nrows, seed, nd=1_000_000):
    import torch
    from pg_strom_amd import runtime
    g = _gen(seed)
    fk = torch.randint(0, int(nd * 1.25), (nrows,), dtype=torch.int32, device="cuda", generator=g)
    a = torch.randint(0, 2**31, (nrows,), dtype=torch.int32, device="cuda", generator=g)
    b = torch.rand(nrows, dtype=torch.float64, device="cuda", generator=g)
    ds = runtime.DeviceStore.from_torch_columns(["int4", "int4", "float8"], [fk, a, b],
                                                [_minmax(fk), _minmax(a), _minmax(b)])
    return ds, (fk, a, b)


def c4_chunk_device(nrows, seed, ngroups=10_000):
    """returns (resident chunk, per-group counts, per-group integer sums) -- int64 tensors"""
    import torch
    from pg_strom_amd import runtime
    g = _gen(seed)
    grp = torch.randint(0, ngroups, (nrows,), dtype=torch.int32, device="cuda", generator=g)
    x = torch.randint(-10**6, 10**6, (nrows,), dtype=torch.int32, device="cuda", generator=g)
    y = torch.rand(nrows, dtype=torch.float64, device="cuda", generator=g) * 100.0
    gl = grp.long()
    cnt = torch.bincount(gl, minlength=ngroups)
    sx = torch.zeros(ngroups, dtype=torch.int64, device="cuda").index_add_(0, gl, x.long())
    ds = runtime.DeviceStore.from_torch_columns(["int4", "int4", "float8"], [grp, x, y],
                                                [_minmax(grp), _minmax(x), _minmax(y)])
    return ds, cnt, sx


def numeric_images_device(v, scale):
    """int64 tensor 'v' at 10^-scale -> the 64-bit device numeric images (normalised: trailing
    decimal zeros moved into the exponent), what kds.numeric_from_scaled does on the host"""
    import torch
    sign = (v < 0).to(torch.int64)
    mant = v.abs()
    exp = torch.full_like(v, -int(scale))
    for _ in range(19):
        strip = (mant != 0) & (mant % 10 == 0)
        if not bool(strip.any().item()):
            break
        mant = torch.where(strip, mant // 10, mant)
        exp = exp + strip.to(torch.int64)
    img = ((exp & 0x3f) << 58) | (sign << 57) | mant
    return torch.where(mant == 0, torch.zeros_like(img), img)


def c5_chunk_device(nrows, seed, decimal=True):
    """TPC-H Q1-shaped lineitem columns: returnflag, linestatus as
    char(1), quantity / extendedprice / discount / tax as 8-byte numerics, shipdate; returns
    (resident chunk, per-group row counts and quantity sums of the rows the date filter keeps)"""
    import torch
    from pg_strom_amd import runtime
    g = _gen(seed)

    def rnd(lo, hi, dtype=torch.int64):
        return torch.randint(lo, hi, (nrows,), dtype=dtype, device="cuda", generator=g)
    # l_returnflag / l_linestatus the way dbgen derives them from the dates (TPC-H 4.2.3): status 'O'
    # for lines shipped after 1995-06-17, else 'F'; flag 'N' for the open ones, 'A' or 'R' for the
    # rest -- three populated groups of a quarter, a quarter and a half of the table (dbgen's fourth,
    # N/F, holds 0.6 % of it), so the largest group's sum(charge) at scale 6 leaves int8 on one GPU's
    # share of C5 as it does on the real table
    ship = rnd(-2922, -2922 + 2526, torch.int32)
    is_open = ship > -1659                          # date '1995-06-17' in days since 2000-01-01
    ls = torch.where(is_open, 79, 70).to(torch.int8)
    rf = torch.where(is_open, 78, torch.tensor([65, 82], dtype=torch.int64, device="cuda")[rnd(0, 2)]).to(torch.int8)
    qty = rnd(1, 51)
    prc, dsc, tax = rnd(90000, 10494951), rnd(0, 11), rnd(0, 9)
    if decimal:
        # numeric(p,s) held as int8 at 10^-s ("8-byte fixed-scale numerics")
        cols, ntype = [rf, ls, qty, prc, dsc, tax], "decimal"
    else:
        # the reference's 64-bit float-decimal images (opencl_numeric.h:122-160)
        cols, ntype = [rf, ls, numeric_images_device(qty, 0), numeric_images_device(prc, 2),
                       numeric_images_device(dsc, 2), numeric_images_device(tax, 2)], "numeric"
    cutoff = -486                                   # date '1998-09-02' in days since 2000-01-01
    keep = ship <= cutoff
    gid = ((rf.long() - 65) * 16 + (ls.long() - 70))[keep]
    cnt = torch.bincount(gid, minlength=Q1_SLOTS)

    def exact_sums(v):
        # per-group sums as Python integers: two carry-free limbs, so that a total beyond int64
        # (sum(charge) over one GPU's share of the table is) is still exact
        v = v[keep]
        hi = torch.zeros(Q1_SLOTS, dtype=torch.int64, device="cuda").index_add_(0, gid, v >> 32)
        lo = torch.zeros(Q1_SLOTS, dtype=torch.int64, device="cuda").index_add_(0, gid, v & 0xffffffff)
        return [(int(h) << 32) + int(l) for h, l in zip(hi.cpu().tolist(), lo.cpu().tolist())]
    disc_price = prc * (100 - dsc)                  # scale 4
    ref = dict(count=cnt.cpu().tolist(), sum_qty=exact_sums(qty), sum_base_price=exact_sums(prc),
               sum_disc_price=exact_sums(disc_price), sum_charge=exact_sums(disc_price * (100 + tax)),
               sum_discount=exact_sums(dsc))
    # zone maps as the device ingest leaves them (ingest_minmax): integer-like columns -- the keys,
    # the date, decimal columns (int8 at their scale) -- have one, 64-bit numeric images do not
    # (... but bounds of their values' integer parts, outward: KDS_COLSTAT_INTPART)
    def intpart(t, scale):
        lo, hi = _minmax(t)
        return (lo // 10**scale, -((-hi) // 10**scale))
    zmaps = [_minmax(rf), _minmax(ls)] + \
        ([_minmax(c) for c in cols[2:]] if decimal else [intpart(qty, 0), intpart(prc, 2), intpart(dsc, 2), intpart(tax, 2)]) + \
        [_minmax(ship)]
    ds = runtime.DeviceStore.from_torch_columns(["char1", "char1", ntype, ntype, ntype, ntype, "date"],
                                                cols + [ship], zmaps)
    return ds, ref


Q1_SLOTS = 18 * 16
# target number -> (reference name, scale of the numeric partial) / row counts, in q1_agg()'s order
Q1_SUMS = {2: ("sum_qty", 0), 3: ("sum_base_price", 2), 4: ("sum_disc_price", 4), 5: ("sum_charge", 6),
           8: ("sum_discount", 2)}
Q1_COUNTS = (6, 7, 9, 10)


def q1_add_reference(total, ref):
    if total is None:
        return {k: list(v) for k, v in ref.items()}
    return {k: [a + b for a, b in zip(total[k], ref[k])] for k in ref}


def q1_check(pr, ref):
    """every aggregate of the Q1 program against torch's exact per-group answers: the partial
    rows of a group (one, plus extra rows for sums beyond the 57-bit numeric mantissa or beyond
    int8) add up to the reference -- big-integer arithmetic on both sides"""
    from pg_strom_amd import kds
    k1, k2 = pr.column(0)[0].astype(np.int64), pr.column(1)[0].astype(np.int64)
    gids = ((k1 - 65) * 16 + (k2 - 70)).tolist()
    want_groups = [g for g in range(Q1_SLOTS) if ref["count"][g]]
    assert sorted(set(gids)) == want_groups, "Q1: groups differ: %s vs %s" % (sorted(set(gids)), want_groups)
    for t in Q1_COUNTS:
        got = [0] * Q1_SLOTS
        for g, v in zip(gids, pr.column(t)[0].tolist()):
            got[g] += int(v)
        assert got == ref["count"], "Q1: row counts (target %d) differ" % t
    for t, (name, scale) in Q1_SUMS.items():
        img, isnull = pr.column(t)
        got = [0] * Q1_SLOTS
        for g, v, n in zip(gids, img.tolist(), isnull.tolist()):
            if not n:
                d = kds.numeric_decode(v).scaleb(scale)
                assert d == d.to_integral_value(), "Q1: %s is not at scale %d" % (name, scale)
                got[g] += int(d)
        assert got == ref[name], "Q1: %s differs: %s vs %s" % (name, got, ref[name])
    return len(want_groups)


def q1_agg(T):
    return ("(gpupreagg (qual (date_le (var 7 date) (const date '1998-09-02')))"
            " (key (var 1 char1)) (key (var 2 char1))"
            " (psum (var 3 {T} 0) 0) (psum (var 4 {T} 2) 2)"
            " (psum (numeric_mul (var 4 {T} 2) (numeric_sub (const numeric 1) (var 5 {T} 2))) 4)"
            " (psum (numeric_mul (numeric_mul (var 4 {T} 2) (numeric_sub (const numeric 1) (var 5 {T} 2)))"
            " (numeric_add (const numeric 1) (var 6 {T} 2))) 6)"
            " (nrows (isnotnull (var 3 {T} 0))) (nrows (isnotnull (var 4 {T} 2)))"
            " (psum (var 5 {T} 2) 2) (nrows (isnotnull (var 5 {T} 2))) (nrows))").replace("{T}", T)


Q1_AGG = q1_agg("decimal")              # the columns as int8 at their typmod scale
Q1_AGG_NUMERIC = q1_agg("numeric")      # the same query over 64-bit numeric images

T_START = time.perf_counter()


def log(msg):
    """progress on stderr (stdout carries the one JSON line)"""
    if os.environ.get("RANK", "0") == "0":
        print("[bench %6.1fs] %s" % (time.perf_counter() - T_START, msg), file=sys.stderr, flush=True)


def host_cores():
    try:
        return max(1, min(len(os.sched_getaffinity(0)), 64))
    except AttributeError:
        return max(1, min(os.cpu_count() or 1, 64))


# --------------------------------------------------------------------------- #
# CPU baselines: the oracle timed on this host (a reported baseline, never the
# thing shipped).  kind: scan | join | agg | chain.
# --------------------------------------------------------------------------- #
class CpuCase(object):
    """one PG-shaped unit of CPU work over a KDS_FORMAT_ROW chunk; run() returns
    the outer rows it processed"""

    def __init__(self, kind, k, c):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_binding as oracle
        from pg_strom_amd import kds
        self.oracle, self.kind, self.ext = oracle, kind, [k, c]
        oracle.load()
        n = CPU_CHUNK_ROWS
        if kind == "scan":
            a, b = c2_columns(n, 0x5eed0002)
            self.buf = kds.build_kds("row", [kds.Column("int4", a), kds.Column("float8", b)])
        elif kind == "agg":
            g, x, y = c4_columns(n, 0x5eed0004)
            self.buf = kds.build_kds("row", [kds.Column("int4", g), kds.Column("int4", x), kds.Column("float8", y)])
        else:
            # the hash table is built inside every oracle call (PostgreSQL builds it once
            # per query): a larger outer sample per call, and the build-only time of an
            # empty outer chunk is subtracted
            n = 8 * CPU_CHUNK_ROWS
            fk, a, b = c3_columns(n, 0x5eed0003)
            self.buf = kds.build_kds("row", [kds.Column("int4", fk), kds.Column("int4", a), kds.Column("float8", b)])
            dkey, dgrp = c3_dimension()
            self.dim = kds.build_kds("row_flat", [kds.Column("int4", dkey), kds.Column("int4", dgrp)])
            self.empty = kds.build_kds("row", [kds.Column("int4", fk[:0]), kds.Column("int4", a[:0]),
                                               kds.Column("float8", b[:0])])
            t0 = time.perf_counter()
            oracle.gpuhashjoin(C3_JOIN, self.empty, [self.dim], nrooms=16)
            self.build_s = time.perf_counter() - t0
            self.dgrp = dgrp
            self.cols = (fk, a, b)
        self.rows = n

    def run(self):
        """returns the seconds of oracle work for self.rows outer rows"""
        o = self.oracle
        t0 = time.perf_counter()
        if self.kind == "scan":
            o.gpuscan(C2_QUAL, self.buf, self.ext)
            return time.perf_counter() - t0
        if self.kind == "agg":
            o.gpupreagg(C4_AGG, self.buf, 4)
            return time.perf_counter() - t0
        if self.kind == "join":
            o.gpuhashjoin(C3_JOIN, self.buf, [self.dim], nrooms=self.rows)
            return max(1e-9, time.perf_counter() - t0 - self.build_s)
        # chain: WHERE (scan) -> join over the selected rows -> group by the dimension column;
        # the numpy gather that stands for PostgreSQL's projection of the joined rows is not timed
        from pg_strom_amd import kds
        rc, sel = o.gpuscan(CHAIN_QUAL, self.buf, CHAIN_EXT)      # the GPU leg's WHERE: keeps 50 %
        rc, nitems, recs = o.gpuhashjoin(C3_JOIN, self.buf, [self.dim], row_map=np.sort(sel - 1).astype(np.int32),
                                         nrooms=self.rows)
        t_ops = time.perf_counter() - t0 - self.build_s
        orow, irow = recs[:, 0] - 1, recs[:, 1]
        joined = kds.build_kds("row", [kds.Column("int4", self.dgrp[irow]), kds.Column("int4", self.cols[1][orow]),
                                       kds.Column("float8", self.cols[2][orow])])
        t0 = time.perf_counter()
        o.gpupreagg(C4_AGG, joined, 4)
        return max(1e-9, t_ops + time.perf_counter() - t0)


CPU_SAMPLE_TEXT = {
    "scan": "%d-row KDS_FORMAT_ROW chunks (heap pages), tuple-at-a-time deform + interpreted qual",
    "join": "%d-row KDS_FORMAT_ROW outer chunks probing a 1e6-row chained hash table of heap tuples "
            "(table build time excluded)",
    "agg": "%d-row KDS_FORMAT_ROW chunks, tuple-at-a-time deform + hash aggregate (1e4 groups)",
    "chain": "%d-row KDS_FORMAT_ROW outer chunks: seqscan qual -> hash join (1e6-row dim) -> hash aggregate "
             "(oracle calls only; the projection of joined rows is not timed)",
}


def cpu_baseline(kind, k, c, budget_s):
    case = CpuCase(kind, k, c)
    case.run()
    rows, secs, reps = 0, 0.0, 0
    t_end = time.perf_counter() + budget_s          # wall-clock budget (secs is oracle time only)
    while time.perf_counter() < t_end:
        secs += case.run()
        rows += case.rows
        reps += 1
    return {"value": rows / secs / 1e6, "unit": "Mrows/s", "cores": 1, "kind": "port",
            "sample": "%d x " % reps + CPU_SAMPLE_TEXT[kind] % case.rows + ", %.1f s" % secs}


def cpu_worker(kind, k, c, seconds):
    """one host core's share of the all-cores baseline (child process: no GPU)"""
    case = CpuCase(kind, k, c)
    case.run()
    rows, secs = 0, 0.0
    t_end = time.perf_counter() + seconds
    while time.perf_counter() < t_end:
        secs += case.run()
        rows += case.rows
    print(json.dumps({"rows": rows, "seconds": secs}), flush=True)


def cpu_baseline_all_cores(kind, k, c, seconds):
    """chunk-parallel flavour: one child process per host core"""
    ncores = host_cores()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker",
                               "%s,%d,%r,%f" % (kind, int(k), float(c), seconds)],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
             for _ in range(ncores)]
    rate = 0.0
    for p in procs:
        out, _ = p.communicate(timeout=seconds * 20 + 300)
        for line in out.splitlines():
            if line.startswith("{"):
                d = json.loads(line)
                rate += d["rows"] / d["seconds"]
    return {"value": rate / 1e6, "unit": "Mrows/s", "cores": ncores, "kind": "port",
            "sample": "%d processes x %.0f s of " % (ncores, seconds) +
                      CPU_SAMPLE_TEXT[kind] % (CPU_CHUNK_ROWS * (1 if kind in ("scan", "agg") else 8))}


def cpu_baseline_columnar(k, c, seconds=3.0):
    """best-effort columnar flavour: vectorised numpy over COLUMN arrays on
    one core -- shows the GPU/CPU ratio is not a row-format artefact"""
    n = 10_000_000
    a, b = c2_columns(n, 0x5eed0002)
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < seconds:
        np.flatnonzero((a < k) & (b > c)).astype(np.int32)
        reps += 1
    dt = time.perf_counter() - t0
    return {"value": reps * n / dt / 1e6, "unit": "Mrows/s", "cores": 1, "kind": "port",
            "sample": "%d x %d-row column arrays, numpy (a<k)&(b>c) -> row ids, %.1f s" % (reps, n, dt)}


def roofline_block(kernel, bytes_per_launch, kern_ns, measured_peak, traffic=None):
    mean_ns = float(np.mean(kern_ns))
    achieved = bytes_per_launch / (mean_ns * 1e-9) / 1e9
    blk = {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "bytes_per_launch": bytes_per_launch,
           "launch_us": mean_ns * 1e-3, "launches_timed": len(kern_ns)}
    if measured_peak:
        blk["measured_peak"] = measured_peak
        blk["frac_of_measured_peak"] = achieved / measured_peak
    return blk


# --------------------------------------------------------------------------- #
# the other operators of the metric, N=1, outside the headline's timed region
# --------------------------------------------------------------------------- #
def operator_figures(args, k, c, measured_peak, cpu_blocks):
    from pg_strom_amd import kds, runtime
    from pg_strom_amd.gpuhashjoin import GpuHashJoin, build_multihash
    from pg_strom_amd.gpupreagg import GpuPreAgg
    out = {}
    nd, ngroups = 1_000_000, 10_000
    chunk_rows = min(args.chunk_rows, 100_000_000)

    # ---- C3: 1e8 fact x 1e6 dim on an int4 key -------------------------------
    import torch
    fact, (fk, a, b) = c3_chunk_device(chunk_rows, 0x5eed0003, nd)
    dkey, dgrp = c3_dimension(nd, ngroups)
    km = build_multihash([(kds.build_kds("row_flat", [kds.Column("int4", dkey), kds.Column("int4", dgrp)]), [1])])
    join = GpuHashJoin(C3_JOIN, row_population_ratio=0.8).begin(km)
    nmatch = int((fk < nd).sum().item())
    ts, walls = [], []
    for _ in range(6):
        t0 = time.perf_counter()
        r = join.join_chunk(fact, flags=1)                      # result pairs stay on the device
        walls.append(time.perf_counter() - t0)
        assert r.nitems == nmatch, "GpuHashJoin: %d pairs, expected %d" % (r.nitems, nmatch)
        ts.append(r.perfmon["time_kern_exec_ns"])
    info = join.table_info()
    out["gpuhashjoin_c3"] = dict(
        workload="GpuHashJoin: %d fact x %d dim on int4 key, 80%% match (BASELINE configs[2]), index %s"
                 % (chunk_rows, nd, info["mode"]),
        value=chunk_rows / float(np.median(walls[1:])) / 1e6, unit="Mrows/s", results="device-resident",
        # (1.25e6 key values: the DIRECT slot array is probed in its 3-byte form, which fits an XCD's L2)
        roofline=roofline_block("gpuhashjoin_main_fast_narrow", 4.0 * chunk_rows + 8.0 * nmatch, ts[1:], measured_peak,
                                traffic=load_traffic(chunk_rows, "gpuhashjoin_main_fast_narrow")),
        **cpu_blocks("join"))

    # ---- scan + join + group-by, one pass over the WHOLE 1e9-row table (the metric's shape) ----
    ext = CHAIN_EXT
    grp = torch.from_numpy(dgrp[np.argsort(dkey)].astype(np.int64)).cuda()     # group of dimension key v
    cnt_ref = torch.zeros(ngroups, dtype=torch.int64, device="cuda")
    sum_ref = torch.zeros(ngroups, dtype=torch.int64, device="cuda")

    def add_reference(fk, a, b):
        sel = (a < int(ext[0])) & (b > float(ext[1])) & (fk < nd)
        g_sel = grp[fk[sel].long()]
        cnt_ref.add_(torch.bincount(g_sel, minlength=ngroups))
        sum_ref.index_add_(0, g_sel, a[sel].long())

    add_reference(fk, a, b)
    del fk, a, b
    facts = [fact]
    nfacts = max(1, args.rows // chunk_rows)
    for ci in range(1, nfacts):
        f2, (fk2, a2, b2) = c3_chunk_device(chunk_rows, 0x5eed0003 + ci, nd)
        add_reference(fk2, a2, b2)
        facts.append(f2)
        del fk2, a2, b2
    torch.cuda.empty_cache()
    cnt_ref, sum_ref = cnt_ref.cpu().numpy(), sum_ref.cpu().numpy()
    del grp
    agg = GpuPreAgg(CHAIN_AGG).begin([(0, ngroups)], ext_params=ext)
    cols = [(1, 2, "int4"), (0, 2, "int4"), (0, 3, "float8")]
    walls, kerns = [], []
    lookup_packed = True
    for step in range(6):
        agg.reset()
        t0 = time.perf_counter()
        pend = [agg.submit_lookup(join, f, cols) for f in facts]
        for p_ in pend:
            st, pfm = agg.collect(p_)
            assert st == 0
            if step > 0:
                kerns.append(pfm["time_kern_exec_ns"] - pfm["time_kern_proj_ns"])
            lookup_packed = lookup_packed and bool(pfm["num_kern_prep"])
        walls.append(time.perf_counter() - t0)
    pr = agg.fetch()
    order = np.argsort(pr.column(0)[0])
    assert np.array_equal(pr.column(1)[0][order], cnt_ref[cnt_ref > 0]), "chain: group counts differ"
    assert np.array_equal(pr.column(2)[0][order], sum_ref[cnt_ref > 0]), "chain: integer sums differ"
    total_rows = chunk_rows * len(facts)
    kname = "gpupreagg_packed_lookup" if lookup_packed else "gpupreagg_dense_lookup"
    out["scan_join_groupby"] = dict(
        workload="scan+hashjoin+groupby in ONE pass over a resident %d-row fact table (%d COLUMN chunks of %d rows): "
                 "WHERE a<k AND b>c (50%%), join 1e6-row dim (80%% match), GROUP BY dim column (%d groups) "
                 "COUNT/SUM/AVG; the join is a lookup inside the aggregate kernel (strom_submit_gpupreagg_lookup), "
                 "one resident table over all chunks" % (total_rows, len(facts), chunk_rows, ngroups),
        value=total_rows / float(np.median(walls[1:])) / 1e6, unit="Mrows/s", rows=total_rows,
        ms_per_pass=float(np.median(walls[1:])) * 1e3, groups=len(pr),
        checked="counts and integer sums over all %d rows equal torch's" % total_rows,
        roofline=roofline_block(kname, 16.0 * chunk_rows, kerns, measured_peak,
                                traffic=load_traffic(chunk_rows, kname)),
        **cpu_blocks("chain"))
    agg.end()
    join.end()
    for f in facts:
        f.release()
    del facts, fact
    torch.cuda.empty_cache()

    # ---- the ROW-format variant of C2 ("for honesty"): heap pages, the
    # format the reference ships -- the row-at-a-time kernel, and the device ingest + COLUMN scan
    from pg_strom_amd.gpuscan import GpuScan, STROM_RESULTS_ON_DEVICE
    nrow = 10_000_000
    ra, rb = c2_columns(nrow, 0x5eed0012)
    row_img = kds.build_kds("row", [kds.Column("int4", ra), kds.Column("float8", rb)])
    row_bytes = len(row_img)
    rds = runtime.DeviceStore.upload(row_img)
    del row_img
    nsel_row = int(np.count_nonzero((ra < k) & (rb > c)))
    scan = GpuScan(C2_QUAL).begin(ext_params=[k, c])
    gen_ns, col_ns, ing_ns = [], [], []
    for _ in range(4):
        res = scan.scan_chunk(rds, flags=STROM_RESULTS_ON_DEVICE)
        assert res.nitems == nsel_row, "ROW scan: %d rows, expected %d" % (res.nitems, nsel_row)
        gen_ns.append(res.perfmon["time_kern_exec_ns"])
    for _ in range(3):
        cds, ns = rds.to_column([23, 701])
        ing_ns.append(ns)
        res = scan.scan_chunk(cds, flags=STROM_RESULTS_ON_DEVICE)
        assert res.nitems == nsel_row
        col_ns.append(res.perfmon["time_kern_exec_ns"])
        cds.release()
    scan.end()
    rds.release()
    out["gpuscan_row_format"] = dict(
        workload="the ROW-format variant of C2: %d rows as PostgreSQL heap pages (KDS_FORMAT_ROW, %.1f B/row in "
                 "the chunk for 12 B referenced), resident; (1) the row-at-a-time kernel, (2) device ingest to "
                 "COLUMN once, then the streaming kernel" % (nrow, row_bytes / nrow),
        rows=nrow, chunk_bytes=row_bytes,
        generic_kernel_us=float(np.median(gen_ns[1:])) * 1e-3,
        ingest_us=float(np.median(ing_ns[1:])) * 1e-3,
        column_kernel_us=float(np.median(col_ns[1:])) * 1e-3,
        roofline=roofline_block("gpuscan_qual_generic", float(row_bytes) + 4.0 * nsel_row, gen_ns[1:], measured_peak),
        roofline_ingest=roofline_block("ingest_to_column(+ingest_minmax)", float(row_bytes) + 12.0 * nrow,
                                       ing_ns[1:], measured_peak))

    # ---- hashed GROUP BY: a key the dense ids cannot express (sparse int8), 1e6 groups ----
    # (C4's targets; the chunk is ordered by hash partition through LDS and folded unit by unit)
    hg = 1_000_000
    gen = _gen(0x5eed0014)
    grp = torch.randint(0, hg, (chunk_rows,), dtype=torch.int64, device="cuda", generator=gen)
    hx = torch.randint(-10**6, 10**6, (chunk_rows,), dtype=torch.int32, device="cuda", generator=gen)
    hy = torch.rand(chunk_rows, dtype=torch.float64, device="cuda", generator=gen) * 100.0
    hkey = grp * (1000003 * 65537) - 2**59
    hcnt = torch.bincount(grp, minlength=hg)
    hsx = torch.zeros(hg, dtype=torch.int64, device="cuda").index_add_(0, grp, hx.long())
    hds = runtime.DeviceStore.from_torch_columns(["int8", "int4", "float8"], [hkey, hx, hy],
                                                 [_minmax(hkey), _minmax(hx), _minmax(hy)])
    del grp, hx, hy
    agg = GpuPreAgg(HASHED_AGG).begin_hashed(ngroups_hint=hg)
    walls, kerns, plan = [], [], True
    nfold = 5
    for step in range(nfold):
        t0 = time.perf_counter()
        st, pfm = agg.fold(hds)
        walls.append(time.perf_counter() - t0)
        assert st == 0, "hashed fold status %d" % st
        kerns.append(pfm["time_kern_exec_ns"])
        plan = plan and bool(pfm["num_kern_prep"])
    pr = agg.fetch()
    agg.end()
    hds.release()
    order = np.argsort(pr.column(0)[0])
    hcnt, hsx = hcnt.cpu().numpy(), hsx.cpu().numpy()
    assert np.array_equal(pr.column(0)[0][order], np.sort(hkey.unique().cpu().numpy())), "hashed GROUP BY: keys differ"
    assert np.array_equal(pr.column(1)[0][order], hcnt[hcnt > 0] * nfold), "hashed GROUP BY: counts differ"
    assert np.array_equal(pr.column(2)[0][order], hsx[hcnt > 0] * nfold), "hashed GROUP BY: integer sums differ"
    del hkey
    torch.cuda.empty_cache()
    out["hashed_groupby_1e6"] = dict(
        workload="GpuPreAgg, hashed GROUP BY: %d rows, int8 key spread over 2^60 (%d groups), COUNT/SUM(int4)/SUM(float8); "
                 "%s" % (chunk_rows, len(pr), "hash partitions through LDS, folded unit by unit (check + plan + scatter + fold)"
                         if plan else "global table"),
        value=chunk_rows / float(np.median(walls[2:])) / 1e6, unit="Mrows/s", groups=len(pr),
        checked="keys, counts and integer sums equal torch's",
        roofline=roofline_block("gpupreagg_hash_check_parts+_scatter_lds+_fold_parts (whole chunk)",
                                20.0 * chunk_rows, kerns[2:], measured_peak))

    # ---- C5: TPC-H Q1-shaped scan + filter + group-by on numeric / date columns ----------
    # twice: the numeric(p,s) columns as int8 at their scale (decimal columns: what the device
    # ingest makes of typmod-scaled numerics), and as the reference's 64-bit numeric images
    q1rows = chunk_rows
    for label, decimal, spec in (("q1_shape_c5", True, Q1_AGG), ("q1_shape_c5_numeric_images", False, Q1_AGG_NUMERIC)):
        q1, q1ref = c5_chunk_device(q1rows, 0x5eed0005, decimal)
        agg = GpuPreAgg(spec).begin([(65, 18), (70, 10)])
        agg.census(q1)
        nslots = agg.compact()
        walls, kerns = [], []
        for step in range(6):
            agg.reset()
            t0 = time.perf_counter()
            st, pfm = agg.fold(q1)
            walls.append(time.perf_counter() - t0)
            assert st == 0, "Q1 fold status %d" % st
            kerns.append(pfm["time_kern_exec_ns"] - pfm["time_kern_proj_ns"])
        pr = agg.fetch()
        ngrp = q1_check(pr, q1ref)
        out[label] = dict(
            workload="TPC-H Q1-shaped GpuPreAgg (BASELINE configs[4]): %d lineitem-like rows, "
                     "WHERE shipdate <= date, GROUP BY returnflag, linestatus (%d groups), 9 partial aggregates "
                     "over 4 numeric(*,2) columns held as %s (38 B/row)"
                     % (q1rows, ngrp, "int8 at their scale (decimal columns)" if decimal
                        else "64-bit numeric images (the reference's device form)"),
            value=q1rows / float(np.median(walls[1:])) / 1e6, unit="Mrows/s", groups=ngrp,
            partial_rows=len(pr), table_slots=int(nslots), checked_folds=int(agg.checked_folds()),
            checked="groups, all four row counts and all five sums (quantity, base price, discounted price, "
                    "charge, discount) equal torch's exact integers",
            roofline=roofline_block("gpupreagg_dense_column", 38.0 * q1rows, kerns[1:], measured_peak,
                                    traffic=load_traffic(chunk_rows, "gpupreagg_dense_column")))
        agg.end()
        q1.release()
        del q1
        torch.cuda.empty_cache()

    # ---- C5 at one GPU's real share: 6e9 / 8 = 7.5e8 rows, 28.5 GB, in 8 resident chunks --------
    # sum(charge) at scale 6 passes 2^63 per group on the way (1.9e8 rows x ~5.5e10): the table's
    # 128-bit integer sums take it, no chunk comes back CpuReCheck, the fetch hands the totals out
    # as several partial rows each of which fits its numeric
    share = 750_000_000
    if args.rows >= 1_000_000_000 and not os.environ.get("STROM_BENCH_NO_Q1_SHARE"):
        nq = 8
        per = share // nq
        parts, ref_total = [], None
        for i in range(nq):
            ds, ref = c5_chunk_device(per, 0x5eed0050 + i, True)
            parts.append(ds)
            ref_total = q1_add_reference(ref_total, ref)
        agg = GpuPreAgg(Q1_AGG).begin([(65, 18), (70, 10)])
        agg.census(parts[0])
        nslots = agg.compact()
        walls, kerns = [], []
        for step in range(4):
            agg.reset()
            t0 = time.perf_counter()
            pend = [agg.submit(ds) for ds in parts]
            res = [agg.collect(p) for p in pend]
            walls.append(time.perf_counter() - t0)
            assert all(st == 0 for st, _ in res), "Q1 share: fold status %s" % [st for st, _ in res]
            kerns.extend(pfm["time_kern_exec_ns"] - pfm["time_kern_proj_ns"] for _, pfm in res)
        pr = agg.fetch()
        ngrp = q1_check(pr, ref_total)
        beyond = sum(1 for v in ref_total["sum_charge"] if v >= 2**63)
        assert beyond >= 1, "Q1 share: no group's sum(charge) left int8 -- the case this block is here for"
        out["q1_shape_c5_one_gpu_share"] = dict(
            workload="TPC-H Q1-shaped GpuPreAgg, ONE GPU's share of BASELINE configs[4]: 6e9 / 8 = %d lineitem-like "
                     "rows (%.1f GB) in %d resident chunks folded into one session, decimal columns (38 B/row)"
                     % (per * nq, 38.0 * per * nq / 1e9, nq),
            value=per * nq / float(np.median(walls[1:])) / 1e6, unit="Mrows/s", groups=ngrp,
            partial_rows=len(pr), table_slots=int(nslots), checked_folds=int(agg.checked_folds()),
            groups_whose_sum_charge_exceeds_int8=beyond,
            checked="groups, row counts and all five sums equal torch's exact integers (two-limb sums: sum(charge) "
                    "does not fit int8)",
            ms_per_pass=float(np.median(walls[1:])) * 1e3,
            roofline=roofline_block("gpupreagg_dense_column", 38.0 * per, kerns[nq:], measured_peak))
        agg.end()
        for ds in parts:
            ds.release()
        del parts
        torch.cuda.empty_cache()

    return out


# --------------------------------------------------------------------------- #
# every N: row-range sharded GpuPreAgg with the RCCL merge inside the C library
# --------------------------------------------------------------------------- #
def sharded_gpupreagg(args, rank, world, local_rank, barrier, dist, torch, measured_peak, cpu_blocks):
    from pg_strom_amd import kds, runtime, parallel
    from pg_strom_amd.gpupreagg import GpuPreAgg
    ngroups = 10_000
    nrows = args.agg_rows
    chunk_rows = min(args.chunk_rows, 100_000_000)
    chunks, cnt_ref, sx_ref = [], np.zeros(ngroups, dtype=np.int64), np.zeros(ngroups, dtype=np.int64)
    off = ci = 0
    while off < nrows:
        n = min(chunk_rows, nrows - off)
        ds, cnt, sx = c4_chunk_device(n, 0x5eed0400 + 1000 * rank + ci, ngroups)
        cnt_ref += cnt.cpu().numpy()
        sx_ref += sx.cpu().numpy()
        chunks.append(ds)
        off += n
        ci += 1
    torch.cuda.empty_cache()
    comm = parallel.RcclComm(rank, world, dindex=0)
    agg = GpuPreAgg(C4_AGG).begin([(0, ngroups)])
    agg.program.wait()
    merge_s, kern_ns, chunk_ns, packed_launches = [], [], [], [0]

    def one_step(record):
        agg.reset()
        pend = [agg.submit(ds) for ds in chunks]
        for p in pend:
            st, pfm = agg.collect(p)
            assert st == 0
            if record:
                # the dominant kernel alone (the fold); the slab merge behind it is reported apart
                kern_ns.append(pfm["time_kern_exec_ns"] - pfm["time_kern_proj_ns"])
                chunk_ns.append(pfm["time_kern_exec_ns"])
                packed_launches[0] += pfm["num_kern_prep"]
        t0 = time.perf_counter()
        agg.allreduce_rccl(comm)
        if record:
            merge_s.append(time.perf_counter() - t0)

    steps = max(2, min(args.steps, 10))
    one_step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        one_step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    tot_cnt, tot_sx = cnt_ref, sx_ref
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tc, ts = torch.from_numpy(cnt_ref).cuda(), torch.from_numpy(sx_ref).cuda()
        dist.all_reduce(tc)
        dist.all_reduce(ts)
        tot_cnt, tot_sx = tc.cpu().numpy(), ts.cpu().numpy()
    # every rank now holds the whole table: integers exact
    pr = agg.fetch()
    order = np.argsort(pr.column(0)[0])
    assert np.array_equal(pr.column(1)[0][order], tot_cnt), "merged group counts differ"
    assert np.array_equal(pr.column(2)[0][order], tot_sx), "merged integer sums differ"
    agg.end()
    comm.destroy()
    for ds in chunks:
        ds.release()
    per_step = elapsed / steps
    out = {"workload": "GpuPreAgg (BASELINE configs[3]): GROUP BY int4 (%d groups) COUNT/SUM(int4)/AVG(float8), "
                       "%d rows per GPU in %d resident COLUMN chunks x %d GPU(s) sharded by row range, per-GPU "
                       "tables merged by strom_gpupreagg_allreduce (RCCL, %d rank(s))"
                       % (ngroups, nrows, len(chunks), world, world),
           "value": world * nrows / per_step / 1e6, "unit": "Mrows/s", "n_gpus": world, "steps": steps,
           "ms_per_step": per_step * 1e3, "merge_ms": float(np.mean(merge_s)) * 1e3, "scaling": "weak",
           "checked": "merged counts and integer sums equal torch's over all ranks' rows"}
    if rank == 0:
        kname = ("gpupreagg_packed_column" if packed_launches[0] == len(kern_ns) else "gpupreagg_dense_column")
        out["roofline"] = roofline_block(kname, 16.0 * nrows / len(chunks), kern_ns,
                                         measured_peak, traffic=load_traffic(chunk_rows, kname))
        out["roofline"]["fold_plus_slab_merge_us"] = float(np.mean(chunk_ns)) * 1e-3
        out["roofline"]["note"] = ("launch_us is the fold kernel alone; gpupreagg_dense_merge adds the "
                                   "work-groups' slabs to the table behind it, on a stream of its own, while "
                                   "the next chunk is folded")
        out.update(cpu_blocks("agg"))
    return out


def claim_stdout():
    """stdout carries exactly ONE JSON line, but RCCL prints a version banner on the C stdout
    when a communicator is made (whenever that happens: at init, at the first collective).
    Keep a private handle on the real stdout for the line and point fd 1 at stderr for
    everything else in this process."""
    sys.stdout.flush()
    line_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return line_out


def load_traffic(chunk_rows, kernel=None):
    """per-launch HBM bytes from the committed rocprofv3 --pmc passes of this command
    (scripts/collect_traffic.sh), if any; kernel=None: the headline's gpuscan_qual_column"""
    path = os.path.join(ROOT, "profiles", "gpuscan_traffic.json")
    try:
        rec = json.load(open(path))
        if rec.get("chunk_rows") != chunk_rows:
            return None
        if kernel is None:
            return rec.get("hbm_bytes_per_launch")
        return rec.get("kernels", {}).get(kernel, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def spawn_ranks(args):
    """`--gpus N` outside torch.distributed.run: start the N ranks here.  This
    process has not touched the GPU (counting devices does not initialise it)
    and never will; every rank is a fresh child."""
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus:
        print("bench.py: --gpus %d asked for, %d GPU(s) visible" % (args.gpus, have), file=sys.stderr)
        sys.exit(2)
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out)
    sys.stdout.flush()
    sys.exit(0 if all(rc == 0 for rc in rcs) else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (a step is 2.1 ms: 200 of them keep the GPU busy for 0.4 s, long enough for an outside
    # utilisation sampler to see the timed region at all)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=1_000_000_000, help="rows per GPU (north star: 1e9)")
    ap.add_argument("--chunk-rows", type=int, default=100_000_000)
    ap.add_argument("--agg-rows", type=int, default=1_000_000_000,
                    help="rows per GPU of the sharded GpuPreAgg region (second timed region; C4: 1e9)")
    ap.add_argument("--window", type=int, default=2,
                    help="requests kept in flight (pg_strom.max_async_chunks)")
    ap.add_argument("--selectivity", type=float, default=0.10)
    ap.add_argument("--cpu-seconds", type=float, default=5.0, help="budget of each CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the GpuHashJoin / GpuPreAgg / chain figures reported next to the headline")
    ap.add_argument("--no-sharded", action="store_true", help="skip the sharded GpuPreAgg + RCCL merge region")
    ap.add_argument("--sharded-timeout", type=int, default=240,
                    help="seconds the sharded GpuPreAgg region may take before the headline is printed without it")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker:
        kind, kk, cc, secs = args.cpu_worker.split(",")
        cpu_worker(kind, np.int32(int(kk)), float(cc), float(secs))
        return
    if (args.gpus > 1 or os.environ.get("STROM_BENCH_FORCE_SPAWN")) and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)            # (the knob rehearses the launcher with one rank on a one-GPU box)
        return

    line_out = claim_stdout()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (the product has no CPU path)", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    ngpus = world

    from pg_strom_amd import kds, runtime
    from pg_strom_amd._lib import lib
    from pg_strom_amd.gpuscan import GpuScan, STROM_RESULTS_ON_DEVICE
    import ctypes

    runtime.init([local_rank])
    log("generating the table on the device")
    s1 = 0.5                                    # a<k passes 50 %, b>c the rest
    s2 = args.selectivity / s1
    k = np.int32(int(2**31 * s1))
    c = float(1.0 - s2)

    # resident table: row range [rank*rows, (rank+1)*rows) in COLUMN chunks
    nrows = args.rows
    chunks = []
    nsel_expect = off = ci = 0
    while off < nrows:
        n = min(args.chunk_rows, nrows - off)
        ds, nsel = c2_chunk_device(n, 0x5eed0002 + 1000 * rank + ci, k, c)
        nsel_expect += nsel
        chunks.append(ds)
        off += n
        ci += 1
    torch.cuda.empty_cache()

    log("table resident: %d chunks; timed region" % len(chunks))
    scan = GpuScan(C2_QUAL).begin(ext_params=[k, c])
    scan.program.wait()

    kern_ns = []
    nitems_seen = []

    import collections

    def run_steps(nsteps, record):
        """nsteps passes over the table; up to --window requests in flight,
        the way pgstrom_fetch_gpuscan keeps chunks in flight (gpuscan.c:
        1087-1108).  Returns rows selected per step (must be identical)."""
        window = collections.deque()
        totals = [0] * nsteps

        def retire():
            step, pend = window.popleft()
            res = scan.collect(pend)
            totals[step] += res.nitems
            if record:
                kern_ns.append(res.perfmon["time_kern_exec_ns"])
                nitems_seen.append(res.nitems)

        for step in range(nsteps):
            for ds in chunks:
                window.append((step, scan.submit(ds, flags=STROM_RESULTS_ON_DEVICE)))
                if len(window) > args.window:
                    retire()
        while window:
            retire()
        return totals

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for got in run_steps(args.warmup, False):
        assert got == nsel_expect, "row count mismatch: %d != %d" % (got, nsel_expect)
    barrier()
    t0 = time.perf_counter()
    totals = run_steps(args.steps, True)
    barrier()
    elapsed = time.perf_counter() - t0
    assert all(t == nsel_expect for t in totals)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the box's own streaming-read rate: the "measured peak" denominator
    measured_peak = None
    gbs = ctypes.c_double(0.0)
    if lib.strom_membw_probe(0, 1_200_000_000, 10, ctypes.byref(gbs)) == 0 and gbs.value > 0:
        measured_peak = gbs.value

    scan.end()
    for ds in chunks:
        ds.release()
    nchunks = len(chunks)
    del chunks

    with_cpu = (ngpus == 1 and not args.no_cpu_baseline)

    def cpu_blocks(kind):
        if not with_cpu:
            return {}
        log("cpu baselines: %s" % kind)
        return {"cpu_baseline": cpu_baseline(kind, k, c, args.cpu_seconds),
                "cpu_baseline_all_cores": cpu_baseline_all_cores(kind, k, c, args.cpu_seconds)}

    # the headline is complete at this point: build its line first, so that nothing
    # measured next to it can cost it
    out = None
    if rank == 0:
        per_step = elapsed / args.steps
        value = ngpus * nrows / per_step / 1e6
        # dominant kernel: gpuscan_qual_column, one launch per chunk
        rows_per_launch = nrows / nchunks
        alg_bytes = 12.0 * rows_per_launch + 4.0 * float(np.mean(nitems_seen))
        roof = roofline_block("gpuscan_qual_column", alg_bytes, kern_ns, measured_peak,
                              traffic=load_traffic(args.chunk_rows))
        roof["traffic_source"] = "profiles/gpuscan_traffic.json (committed rocprofv3 --pmc passes of this command)"
        out = {
            "metric": "GpuScan Mrows/s (WHERE a<k AND b>c over int4,float8 COLUMN chunks), "
                      "achieved HBM GB/s vs peak",
            "value": value,
            "unit": "Mrows/s",
            "n_gpus": ngpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": per_step * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32+f64",
            "data": "synthetic",
            "config": {
                "workload": "GpuScan over a %d-row int4+float8 table per GPU in %d KDS_FORMAT_COLUMN chunks, "
                            "WHERE a<k AND b>c (BASELINE metric's 1e9-row synth; configs[1] x %d chunks)"
                            % (nrows, nchunks, nchunks),
                "rows_per_gpu": nrows,
                "chunk_rows": args.chunk_rows,
                "chunks_per_gpu": nchunks,
                "requests_in_flight": args.window,
                "selectivity": nsel_expect / nrows,
                "format": "KDS_FORMAT_COLUMN",
                "results": "device-resident",
                "parallelism": "row-range x%d" % ngpus,
            },
            "roofline": roof,
            "whole_job_gbs": 12.0 * ngpus * nrows / per_step / 1e9,
        }

    printed = threading.Lock()

    def emit(extra=None):
        """rank 0 prints THE line, once"""
        if rank == 0 and printed.acquire(False):
            if extra:
                out.update(extra)
            print(json.dumps(out), file=line_out, flush=True)

    if not args.no_sharded:
        # second region: every rank takes part (RCCL).  A rank that does not come back from
        # it within the limit must not take the headline with it: the watchdog prints the
        # line (with the failure named) and ends the process.
        def give_up():
            log("sharded GpuPreAgg region did not finish in %d s" % args.sharded_timeout)
            emit({"gpupreagg_c4": {"error": "did not finish within %d s" % args.sharded_timeout}})
            os._exit(0 if rank == 0 else 3)

        watchdog = threading.Timer(args.sharded_timeout, give_up)
        watchdog.daemon = True
        watchdog.start()
        log("sharded GpuPreAgg + RCCL merge")
        try:
            sharded = sharded_gpupreagg(args, rank, world, local_rank, barrier, dist, torch, measured_peak,
                                        cpu_blocks)
        except Exception as e:
            sharded = {"error": "%s: %s" % (type(e).__name__, e)}
        watchdog.cancel()
        if rank == 0:
            out["gpupreagg_c4"] = sharded

    if rank == 0:
        if with_cpu:
            log("cpu baselines: scan")
            out["cpu_baseline"] = cpu_baseline("scan", k, c, 2 * args.cpu_seconds)
            out["cpu_baseline_all_cores"] = cpu_baseline_all_cores("scan", k, c, args.cpu_seconds)
            out["cpu_baseline_columnar"] = cpu_baseline_columnar(k, c)
        if ngpus == 1 and not args.no_extras:
            try:
                log("other operators")
                out["operators"] = operator_figures(args, k, c, measured_peak, cpu_blocks)
            except Exception as e:                  # never at the expense of the headline line
                out["operators"] = {"error": "%s: %s" % (type(e).__name__, e)}
        emit()

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
