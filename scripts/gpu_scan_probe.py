"""GPU probe for the GpuScan path: parity vs numpy at a few sizes and kernel
timing on a resident chunk.  Run on the GPU box:  python scripts/gpu_scan_probe.py"""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import pg_strom_amd as ps
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpuscan import GpuScan, STROM_RESULTS_ON_DEVICE

QUAL = "(and (int4lt (var 1 int4) (param 0 int4)) (float8gt (var 2 float8) (param 1 float8)))"


def table(n, seed, null_frac=0.0):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 2**31, n, dtype=np.int64).astype(np.int32)
    b = rng.random(n)
    an = (rng.random(n) < null_frac) if null_frac > 0 else None
    return a, b, an


def expect(a, b, an, k, c):
    m = (a < k) & (b > c)
    if an is not None:
        m &= ~an
    return np.nonzero(m)[0]


def main():
    runtime.init()
    print(runtime.device_info(0), flush=True)
    scan = GpuScan(QUAL).begin(ext_params=[np.int32(0), 0.0])
    t0 = time.time()
    scan.program.wait()
    print("program ready in %.2fs" % (time.time() - t0), flush=True)
    for n, nf in ((1000, 0.0), (12345, 0.05), (1 << 20, 0.0), (3000001, 0.05)):
        a, b, an = table(n, n, nf)
        k, c = int(2**31 * 0.7), 0.3
        scan.parambuf = scan.codegen.parambuf([np.int32(k), c])
        want = expect(a, b, an, k, c)
        for fmt in ("column", "row", "row_flat", "tupslot"):
            if fmt != "column" and n > (1 << 20):
                continue
            buf = kds.build_kds(fmt, [kds.Column("int4", a, an), kds.Column("float8", b)])
            res = scan.scan_chunk(buf)
            got = res.passed_rows()
            ok = (len(got) == len(want) and np.array_equal(got, want) and res.nitems == len(want))
            print("n=%d nulls=%.2f fmt=%s nitems=%d expect=%d %s" % (
                n, nf, fmt, res.nitems, len(want), "OK" if ok else "MISMATCH"), flush=True)
            if not ok:
                sys.exit(1)
    # timing on a resident chunk
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
    a, b, an = table(n, 7)
    buf = kds.build_kds("column", [kds.Column("int4", a), kds.Column("float8", b)])
    ds = runtime.DeviceStore.upload(buf)
    for sel_a, sel_b in ((0.1, 0.9), (0.5, 0.8), (0.7, 0.3)):
        k, c = int(2**31 * sel_a), sel_b
        scan.parambuf = scan.codegen.parambuf([np.int32(k), c])
        want = len(expect(a, b, None, k, c))
        times = []
        for it in range(12):
            res = scan.scan_chunk(ds, flags=STROM_RESULTS_ON_DEVICE)
            assert res.nitems == want, (res.nitems, want)
            times.append(res.perfmon["time_kern_exec_ns"])
        t = np.median(times[2:]) * 1e-9
        byts = 12.0 * n + 4.0 * want
        print("n=%d sel=%.3f kern=%.1f us  %.1f Mrows/s  %.0f GB/s (%.1f%% of 8TB/s)" % (
            n, want / n, t * 1e6, n / t / 1e6, byts / t / 1e9, byts / t / 8e12 * 100), flush=True)
    ds.release()
    scan.end()


if __name__ == "__main__":
    main()
