"""
bench.py's host-side contract, without a GPU: `--gpus N` must start N ranks itself or fail
loudly (never silently run one rank and print n_gpus: 1), and the CPU-baseline legs
(the oracle timed as PostgreSQL's seqscan / hash join / hash aggregate stand-in) work.
"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_n_fails_loudly_when_the_gpus_are_not_there():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this box has the GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 2
    assert "--gpus 2 asked for" in p.stderr
    assert p.stdout.strip() == ""


def test_world_size_must_match_gpus():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 2 and "WORLD_SIZE=4" in p.stderr


def test_cpu_baseline_legs():
    sys.path.insert(0, ROOT)
    import bench
    k, c = np.int32(2**30), 0.8
    for kind in ("scan", "join", "agg", "chain"):
        r = bench.cpu_baseline(kind, k, c, 0.3)
        assert r["value"] > 0 and r["cores"] == 1 and r["kind"] == "port" and r["unit"] == "Mrows/s"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-worker", "agg,1073741824,0.8,0.3"],
                       capture_output=True, text=True, timeout=300)
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert d["rows"] > 0 and d["seconds"] > 0
