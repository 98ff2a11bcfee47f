import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _build_everything():
    """build libstrom_hip.so and the oracle BEFORE pg_strom_amd is imported
    (the package loads the .so at import time)"""
    spec = importlib.util.spec_from_file_location(
        "_strom_build", os.path.join(ROOT, "pg_strom_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build_library(verbose=False)
    mod.build_c_tests(verbose=False)
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])


_build_everything()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
