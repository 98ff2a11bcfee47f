#!/usr/bin/env python3
"""
GpuPreAgg measurement (BASELINE configs[3] shape): GROUP BY g (1e4 groups)
COUNT(*), SUM(x int4), AVG(y float8) over rows sharded by row range, one
process per GPU, per-GPU tables merged with an RCCL all-reduce.
  1 GPU :  python scripts/bench_gpupreagg.py [--rows 100000000]
  N GPUs:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
               --master-addr 127.0.0.1 --master-port 29511 scripts/bench_gpupreagg.py
Prints one JSON line on rank 0 (same fields as bench.py; algorithmic bytes
16 B/row).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000_000, help="rows per GPU")
    ap.add_argument("--chunk-rows", type=int, default=100_000_000)
    ap.add_argument("--groups", type=int, default=10_000)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", init_method=None if world > 1 else "tcp://127.0.0.1:29533",
                            rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    from pg_strom_amd import kds, runtime
    from pg_strom_amd.gpupreagg import GpuPreAgg
    runtime.init([local_rank])
    rng = np.random.default_rng(0x5eed0004 + rank)
    chunks, cnt_ref = [], np.zeros(args.groups, dtype=np.int64)
    sx_ref = np.zeros(args.groups, dtype=np.int64)
    off = 0
    while off < args.rows:
        n = min(args.chunk_rows, args.rows - off)
        g = rng.integers(0, args.groups, n, dtype=np.int64).astype(np.int32)
        x = rng.integers(-10**6, 10**6, n, dtype=np.int64).astype(np.int32)
        y = rng.random(n) * 100
        cnt_ref += np.bincount(g, minlength=args.groups)
        sx_ref += np.bincount(g, weights=x.astype(np.float64), minlength=args.groups).astype(np.int64)
        chunks.append(runtime.DeviceStore.upload(
            kds.build_kds("column", [kds.Column("int4", g), kds.Column("int4", x), kds.Column("float8", y)])))
        off += n
    spec = "(gpupreagg (key (var 1 int4)) (nrows) (psum (int8 (var 2 int4))) (psum (var 3 float8)))"
    agg = GpuPreAgg(spec).begin([(0, args.groups)])
    agg.program.wait()
    agg.bind_torch_table()
    kern_ns = []

    def one_step(record):
        agg.reset()
        pend = [agg.submit(ds) for ds in chunks]
        for p in pend:
            st, pfm = agg.collect(p)
            assert st == 0
            if record:
                kern_ns.append(pfm["time_kern_exec_ns"])
        agg.allreduce()

    for _ in range(args.warmup):
        one_step(False)
    dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step(True)
    dist.barrier(); torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    # parity of the merged result (integers exact): every rank holds the global table
    tot_cnt = torch.from_numpy(cnt_ref).cuda(); tot_sx = torch.from_numpy(sx_ref).cuda()
    dist.all_reduce(tot_cnt); dist.all_reduce(tot_sx)
    pr = agg.fetch()
    order = np.argsort(pr.column(0)[0])
    assert np.array_equal(pr.column(1)[0][order], tot_cnt.cpu().numpy())
    assert np.array_equal(pr.column(2)[0][order], tot_sx.cpu().numpy())
    if rank == 0:
        per_step = elapsed / args.steps
        mean_ns = float(np.mean(kern_ns))
        rows_launch = args.rows / len(chunks)
        achieved = 16.0 * rows_launch / (mean_ns * 1e-9) / 1e9
        print(json.dumps({
            "metric": "GpuPreAgg Mrows/s (GROUP BY int4, COUNT/SUM/AVG), achieved HBM GB/s vs peak",
            "value": world * args.rows / per_step / 1e6, "unit": "Mrows/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": per_step * 1e3,
            "higher_is_better": True, "scaling": "weak", "dtype": "int64+f64", "data": "synthetic",
            "config": {"workload": "GpuPreAgg: GROUP BY int4 (%d groups) COUNT/SUM(int4)/AVG(float8), "
                                   "%d rows per GPU, RCCL table all-reduce" % (args.groups, args.rows),
                       "parallelism": "row-range x%d" % world},
            "roofline": {"bound": "hbm", "kernel": "gpupreagg_dense_column(+merge)", "achieved": achieved,
                         "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": None,
                         "bytes_per_launch": 16.0 * rows_launch, "launch_us": mean_ns * 1e-3},
        }), flush=True)
    agg.end()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
