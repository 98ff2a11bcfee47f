"""Compile an operator spec with hiprtc (no GPU needed) and print each
kernel's register / LDS / scratch use from the code object's metadata.

usage: python scripts/kernel_resources.py gpupreagg "(gpupreagg ...)"
"""
import re
import subprocess
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import os  # noqa: E402

import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location(
    "_strom_build", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                 "pg_strom_amd", "build.py"))
_build = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_build)
_build.build_library(verbose=False)          # the device library is embedded in the .so

from pg_strom_amd import gpuhashjoin, gpupreagg, runtime  # noqa: E402


def main():
    kind, spec = sys.argv[1], sys.argv[2]
    if kind == "ingest":
        class _S(object):
            source = ('#include "strom_kds.h"\n#include "strom_common.h"\n#include "strom_mathlib.h"\n'
                      '#include "strom_numeric.h"\n#include "strom_ingest.h"\n')
            extra_flags = 0
        cg = _S()
    elif kind == "gpuscan":
        cg = runtime.codegen_gpuscan(spec)
    elif kind == "gpupreagg":
        cg = gpupreagg.codegen_gpupreagg(spec)
        cg = cg[0] if isinstance(cg, tuple) else cg
    else:
        cg = gpuhashjoin.codegen_gpuhashjoin(spec)
        cg = cg[0] if isinstance(cg, tuple) else cg
    prog = runtime.DevProgram(cg.source, cg.extra_flags).wait()
    path = os.path.join(os.path.dirname(runtime.__file__), "_cache", "%016x.hsaco" % prog.key)
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", path],
                         capture_output=True, text=True).stdout
    for m in re.finditer(r"\.name:\s+(\S+)(.*?)\.wavefront_size", out, re.S):
        body = m.group(2)
        get = lambda k: re.search(r"\.%s:\s+(\d+)" % k, body)
        vals = {k: (get(k).group(1) if get(k) else "?")
                for k in ("vgpr_count", "agpr_count", "sgpr_count",
                          "group_segment_fixed_size", "private_segment_fixed_size")}
        print(m.group(1), vals)


if __name__ == "__main__":
    main()
