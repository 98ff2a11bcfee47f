"""
The callback half of the boundary, driven from a plain-C host (tests/c/
async_boundary.c, built by build() / conftest): real strom_done_cb, several
requests in flight from 3 threads, a COLD program cache so that the first
requests park behind the hiprtc build (opencl_devprog.c:291-527), exactly-once
on a non-caller thread with results final, build-failure and refused requests,
DataStoreNoSpace -> retry (gpuhashjoin.c:4330-4425), GpuPreAgg folds.
The CPU half only checks that the program exists and links the in-tree library.
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "c", "async_boundary")


def test_c_host_is_built_against_the_in_tree_library():
    assert os.path.exists(EXE)
    out = subprocess.run(["ldd", EXE], capture_output=True, text=True).stdout
    line = [l for l in out.splitlines() if "libstrom_hip.so" in l]
    assert line and os.path.join(ROOT, "pg_strom_amd", "libstrom_hip.so") in os.path.realpath(
        line[0].split("=>")[1].split("(")[0].strip())


@pytest.mark.gpu
def test_async_boundary_from_c():
    env = dict(os.environ)
    env.pop("STROM_HIP_CACHE_DIR", None)
    p = subprocess.run([EXE], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, "stdout:\n%s\nstderr:\n%s" % (p.stdout, p.stderr)
    assert "ALL OK" in p.stdout
    assert p.stdout.count("ok:") == 5
