#!/usr/bin/env python3
"""
bench.py -- headline measurement of the GpuScan path (BASELINE.json configs[1]).

A "step" is one pass of GpuScan over the rank's resident table: 1e8 rows of
(a int4, b float8) in KDS_FORMAT_COLUMN chunks, WHERE a < k AND b > c, every
chunk submitted through the C ABI (strom_submit_gpuscan) with the results left
in HBM.  Inputs are resident in HBM before the timed region starts.  With N
GPUs the table is N x 1e8 rows sharded by row range, one process per GPU,
no data-path collective (weak scaling).

Output: ONE JSON line on rank 0 (see the driver contract), with
  roofline     algorithmic bytes per launch / mean kernel time per launch,
               kernel time from HIP events recorded on the launch stream by
               the runtime (strom_perfmon.time_kern_exec_ns)
  cpu_baseline the CPU oracle (oracle/, a port: tuple-at-a-time over ROW
               format heap pages, expression tree interpreted per row, the
               shape of PostgreSQL's SeqScan+ExecQual) timed on a bounded
               sample of the same workload on this host, rank 0, N=1 only
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

C2_QUAL = "(and (int4lt (var 1 int4) (param 0 int4)) (float8gt (var 2 float8) (param 1 float8)))"
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def make_columns(nrows, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 2**31, nrows, dtype=np.int64).astype(np.int32)
    b = rng.random(nrows)
    return a, b


def cpu_baseline(k, c, budget_s=12.0):
    """time the CPU oracle on ROW-format chunks of the same distribution"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as oracle
    from pg_strom_amd import kds
    oracle.build_oracle()
    chunk_rows = 325_000            # the reference's 15MB ROW chunk
    a, b = make_columns(chunk_rows, 0x5eed0002)
    buf = kds.build_kds("row", [kds.Column("int4", a), kds.Column("float8", b)])
    t0 = time.perf_counter()
    oracle.gpuscan(C2_QUAL, buf, [k, c])
    t1 = time.perf_counter() - t0
    reps = int(max(1, min(400, budget_s / max(t1, 1e-6))))
    t0 = time.perf_counter()
    sel = 0
    for _ in range(reps):
        _, res = oracle.gpuscan(C2_QUAL, buf, [k, c])
        sel += len(res)
    dt = time.perf_counter() - t0
    rows = reps * chunk_rows
    return {
        "value": rows / dt / 1e6,
        "unit": "Mrows/s",
        "cores": 1,
        "kind": "port",
        "sample": "%d x %d-row KDS_FORMAT_ROW chunks (heap pages, 185 rows/page), "
                  "tuple-at-a-time deform + interpreted qual, %.1f s" % (reps, chunk_rows, dt),
    }


def cpu_worker(k, c, seconds):
    """one host core's share of the all-cores baseline (child process: no GPU)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as oracle
    from pg_strom_amd import kds
    chunk_rows = 325_000
    a, b = make_columns(chunk_rows, 0x5eed0002)
    buf = kds.build_kds("row", [kds.Column("int4", a), kds.Column("float8", b)])
    oracle.gpuscan(C2_QUAL, buf, [k, c])
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < seconds:
        oracle.gpuscan(C2_QUAL, buf, [k, c])
        reps += 1
    print(json.dumps({"rows": reps * chunk_rows, "seconds": time.perf_counter() - t0}), flush=True)


def cpu_baseline_all_cores(k, c, seconds=8.0):
    """chunk-parallel flavour of the same baseline: one child process per
    host core, each scanning its own ROW chunk"""
    import subprocess
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    ncores = max(1, min(ncores, 64))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker",
                               "%d,%r,%f" % (int(k), float(c), seconds)],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
             for _ in range(ncores)]
    rate = 0.0
    for p in procs:
        out, _ = p.communicate(timeout=seconds * 10 + 120)
        for line in out.splitlines():
            if line.startswith("{"):
                d = json.loads(line)
                rate += d["rows"] / d["seconds"]
    return {"value": rate / 1e6, "unit": "Mrows/s", "cores": ncores, "kind": "port",
            "sample": "%d processes x %.0f s of 325000-row KDS_FORMAT_ROW chunks" % (ncores, seconds)}


def cpu_baseline_columnar(k, c, seconds=3.0):
    """best-effort columnar flavour: vectorised numpy over COLUMN arrays on
    one core -- shows the GPU/CPU ratio is not a row-format artefact"""
    n = 10_000_000
    a, b = make_columns(n, 0x5eed0002)
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < seconds:
        sel = np.flatnonzero((a < k) & (b > c)).astype(np.int32)
        reps += 1
    dt = time.perf_counter() - t0
    return {"value": reps * n / dt / 1e6, "unit": "Mrows/s", "cores": 1, "kind": "port",
            "sample": "%d x %d-row column arrays, numpy (a<k)&(b>c) -> row ids, %.1f s" % (reps, n, dt)}


def other_operator_rates(nrows):
    """N=1 only, after the timed region: the other two operators of BASELINE.json's metric
    ("scan+hashjoin+groupby") on a resident chunk of the same size, kernel time from the
    runtime's events, plus the three chained on the device.  Reported next to the headline,
    never part of `value`."""
    from pg_strom_amd import kds, runtime
    from pg_strom_amd.gpuhashjoin import GpuHashJoin, build_multihash
    from pg_strom_amd.gpupreagg import GpuPreAgg
    from pg_strom_amd.gpuscan import GpuScan
    nd, ngroups = 1_000_000, 10_000
    rng = np.random.default_rng(0x5eed0003)
    fk = rng.integers(0, int(nd * 1.25), nrows, dtype=np.int64).astype(np.int32)
    a = rng.integers(0, 2**31, nrows, dtype=np.int64).astype(np.int32)
    b = rng.random(nrows)
    fact = runtime.DeviceStore.upload(kds.build_kds("column", [kds.Column("int4", fk), kds.Column("int4", a),
                                                               kds.Column("float8", b)]), 0)
    dkey = rng.permutation(nd).astype(np.int32)
    dgrp = (dkey % ngroups).astype(np.int32)
    km = build_multihash([(kds.build_kds("row_flat", [kds.Column("int4", dkey), kds.Column("int4", dgrp)]), [1])])
    out = {"rows": nrows}
    join = GpuHashJoin("(gpuhashjoin (rel (hashkey (var 1 int4) 1 int4)))", row_population_ratio=0.8).begin(km)
    ts = []
    for _ in range(5):
        r = join.join_chunk(fact, flags=1)                      # results stay on the device
        ts.append(r.perfmon["time_kern_exec_ns"])
    t = float(np.median(ts[1:])) * 1e-9
    out["gpuhashjoin_c3"] = {"workload": "%d fact x %d dim on int4 key, 80%% match" % (nrows, nd),
                             "kernel_us": t * 1e6, "mrows_s": nrows / t / 1e6,
                             "achieved_gbs": (4.0 * nrows + 8.0 * r.nitems) / t / 1e9}
    # C4: GROUP BY int4 (1e4 groups) count / sum / avg partials
    g = (fk % ngroups).astype(np.int32)
    c4 = runtime.DeviceStore.upload(kds.build_kds("column", [kds.Column("int4", g), kds.Column("int4", a),
                                                             kds.Column("float8", b)]), 0)
    agg = GpuPreAgg("(gpupreagg (key (var 1 int4)) (nrows) (psum (int8 (var 2 int4))) (psum (var 3 float8)))")
    agg.begin([(0, ngroups)])
    ts = []
    for _ in range(5):
        st, pfm = agg.fold(c4)
        ts.append(pfm["time_kern_exec_ns"])
    t = float(np.median(ts[1:])) * 1e-9
    out["gpupreagg_c4"] = {"workload": "GROUP BY int4 (%d groups) count, sum(int4), sum(float8) over %d rows" % (ngroups, nrows),
                           "kernel_us": t * 1e6, "mrows_s": nrows / t / 1e6, "achieved_gbs": 16.0 * nrows / t / 1e9}
    agg.end()
    c4.release()
    # scan -> join -> group by, nothing leaves HBM in between
    ext = [np.int32(2**30), 0.0]
    scan = GpuScan("(and (int4lt (var 2 int4) (param 0 int4)) (float8gt (var 3 float8) (param 1 float8)))").begin(ext_params=ext)
    agg = GpuPreAgg("(gpupreagg (key (var 1 int4)) (nrows) (psum (int8 (var 2 int4))) (psum (var 3 float8)))")
    agg.begin([(0, ngroups)])
    walls = []
    for _ in range(4):
        agg.reset()
        t0 = time.perf_counter()
        rowmap, res = scan.scan_to_rowmap(fact)
        joined, nitems = join.join_to_column(fact, [(1, 2, "int4"), (0, 2, "int4"), (0, 3, "float8")],
                                             row_map=rowmap, nrooms=int(res.nitems * 0.82) + 1000)
        agg.fold(joined)
        pr = agg.fetch()
        walls.append(time.perf_counter() - t0)
        rowmap.release()
        joined.release()
    t = float(np.median(walls[1:]))
    out["scan_join_groupby_chain"] = {"workload": "WHERE keeps 50%%, join 80%% match, GROUP BY dim column (%d groups), "
                                                  "device-resident hand-overs" % ngroups,
                                      "wall_ms": t * 1e3, "mrows_s": nrows / t / 1e6,
                                      "joined_rows": int(nitems), "groups": len(pr)}
    # the same query planned the reference's way: the WHERE pulled up into the join
    # (gpuhashjoin.c:2047-2050) -- no scan pass, the one-pass join kernel
    join2 = GpuHashJoin("(gpuhashjoin (rel (hashkey (var 1 int4) 1 int4) (qual (and (int4lt (var 2 int4) (param 0 int4))"
                        " (float8gt (var 3 float8) (param 1 float8))))))", row_population_ratio=0.45).begin(km, ext_params=ext)
    walls = []
    for _ in range(4):
        agg.reset()
        t0 = time.perf_counter()
        joined, nitems2 = join2.join_to_column(fact, [(1, 2, "int4"), (0, 2, "int4"), (0, 3, "float8")],
                                               zone_maps=False)     # the aggregate brings its key domain
        agg.fold(joined)
        pr = agg.fetch()
        walls.append(time.perf_counter() - t0)
        joined.release()
    t = float(np.median(walls[1:]))
    out["join_with_pulled_up_qual_groupby"] = {"workload": "same query, WHERE inside the join program",
                                               "wall_ms": t * 1e3, "mrows_s": nrows / t / 1e6,
                                               "joined_rows": int(nitems2), "groups": len(pr)}
    # and with the projection fused into the aggregate (strom_submit_gpupreagg_joined)
    walls = []
    for _ in range(4):
        agg.reset()
        t0 = time.perf_counter()
        jp = join2.submit(fact, flags=1)
        ap = agg.submit_joined(join2, jp, fact, [(1, 2, "int4"), (0, 2, "int4"), (0, 3, "float8")])
        agg.collect(ap)
        jr = join2.collect(jp)
        pr = agg.fetch()
        walls.append(time.perf_counter() - t0)
    t = float(np.median(walls[1:]))
    out["join_groupby_fused"] = {"workload": "same query, the aggregate reads the join's result pairs",
                                 "wall_ms": t * 1e3, "mrows_s": nrows / t / 1e6,
                                 "joined_rows": int(jr.nitems), "groups": len(pr)}
    # and with no join request at all: the join is a lookup inside the aggregate's own pass
    # over the fact chunk (strom_submit_gpupreagg_lookup), the WHERE is the aggregate's qual
    agg3 = GpuPreAgg("(gpupreagg (qual (and (int4lt (var 2 int4) (param 0 int4)) (float8gt (var 3 float8) (param 1 float8))))"
                     " (key (var 1 int4)) (nrows) (psum (int8 (var 2 int4))) (psum (var 3 float8)))")
    agg3.begin([(0, ngroups)], ext_params=ext)
    walls, kerns = [], []
    for _ in range(5):
        agg3.reset()
        t0 = time.perf_counter()
        st, pfm = agg3.collect(agg3.submit_lookup(join2, fact, [(1, 2, "int4"), (0, 2, "int4"), (0, 3, "float8")]))
        pr3 = agg3.fetch()
        walls.append(time.perf_counter() - t0)
        kerns.append(pfm["time_kern_exec_ns"])
    t = float(np.median(walls[1:]))
    out["join_as_lookup_groupby"] = {"workload": "same query, one pass over the fact chunk: the join is a lookup "
                                                 "inside the aggregate kernel",
                                     "wall_ms": t * 1e3, "kernel_us": float(np.median(kerns[1:])) * 1e-3,
                                     "mrows_s": nrows / t / 1e6, "groups": len(pr3)}
    agg3.end()
    join2.end()
    agg.end()
    scan.end()
    join.end()
    fact.release()
    return out


def load_traffic(chunk_rows):
    """per-launch HBM bytes from the committed rocprofv3 --pmc passes, if any"""
    path = os.path.join(ROOT, "profiles", "gpuscan_traffic.json")
    try:
        rec = json.load(open(path))
        if rec.get("chunk_rows") == chunk_rows:
            return rec.get("hbm_bytes_per_launch")
    except Exception:
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=100_000_000, help="rows per GPU")
    ap.add_argument("--chunk-rows", type=int, default=100_000_000)
    ap.add_argument("--window", type=int, default=2,
                    help="requests kept in flight (pg_strom.max_async_chunks)")
    ap.add_argument("--selectivity", type=float, default=0.10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the GpuHashJoin / GpuPreAgg / chain figures reported next to the headline")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker:
        kk, cc, secs = args.cpu_worker.split(",")
        cpu_worker(np.int32(int(kk)), float(cc), float(secs))
        return

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (the product has no CPU path)", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    ngpus = world

    from pg_strom_amd import kds, runtime
    from pg_strom_amd._lib import lib, strom_perfmon
    from pg_strom_amd.gpuscan import GpuScan, STROM_RESULTS_ON_DEVICE
    import ctypes

    runtime.init([local_rank])
    # selectivity s = s1 * s2 with s1 = sqrt-ish split: a<k passes 50%, b>c the rest
    s1 = 0.5
    s2 = args.selectivity / s1
    k = np.int32(int(2**31 * s1))
    c = float(1.0 - s2)

    # resident table: row range [rank*rows, (rank+1)*rows) in COLUMN chunks
    nrows = args.rows
    chunks = []
    nsel_expect = 0
    off = 0
    ci = 0
    while off < nrows:
        n = min(args.chunk_rows, nrows - off)
        a, b = make_columns(n, 0x5eed0002 + 1000 * rank + ci)
        nsel_expect += int(np.count_nonzero((a < k) & (b > c)))
        buf = kds.build_kds("column", [kds.Column("int4", a), kds.Column("float8", b)])
        chunks.append(runtime.DeviceStore.upload(buf, 0))
        del a, b, buf
        off += n
        ci += 1

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

    if rank == 0:
        per_step = elapsed / args.steps
        value = ngpus * nrows / per_step / 1e6
        # dominant kernel: gpuscan_qual_column, one launch per chunk
        launches = len(kern_ns)
        mean_ns = float(np.mean(kern_ns))
        rows_per_launch = nrows / len(chunks)
        alg_bytes = 12.0 * rows_per_launch + 4.0 * float(np.mean(nitems_seen))
        achieved = alg_bytes / (mean_ns * 1e-9) / 1e9
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
                "workload": "GpuScan: %d-row int4+float8 kern_data_store per GPU, "
                            "WHERE a<k AND b>c (BASELINE configs[1])" % nrows,
                "rows_per_gpu": nrows,
                "chunk_rows": args.chunk_rows,
                "chunks_per_gpu": len(chunks),
                "requests_in_flight": args.window,
                "selectivity": nsel_expect / nrows,
                "format": "KDS_FORMAT_COLUMN",
                "parallelism": "row-range x%d" % ngpus,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "gpuscan_qual_column",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": load_traffic(args.chunk_rows),
                "bytes_per_launch": alg_bytes,
                "launch_us": mean_ns * 1e-3,
                "launches_timed": launches,
            },
            "whole_job_gbs": 12.0 * ngpus * nrows / per_step / 1e9,
        }
        if ngpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(k, c)
            out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(k, c)
            out["cpu_baseline_columnar"] = cpu_baseline_columnar(k, c)
        if ngpus == 1 and not args.no_extras and not args.no_cpu_baseline:
            try:
                out["other_operators"] = other_operator_rates(min(nrows, 100_000_000))
            except Exception as e:                  # never at the expense of the headline line
                out["other_operators"] = {"error": "%s: %s" % (type(e).__name__, e)}
        print(json.dumps(out), flush=True)

    scan.end()
    for ds in chunks:
        ds.release()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
