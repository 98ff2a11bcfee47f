"""
GpuScan parity, HIP path vs CPU oracle (needs an MI355X: -m gpu).
Every request goes through the C ABI of libstrom_hip.so
(strom_submit_gpuscan).  Row selection is integer work: the bar is
bit-exact equality of the (row id, pass/recheck) sets -- order inside
results[] is unspecified by the reference (opencl_gpuscan.h:118-122), so
sets are compared after sorting.
"""
import numpy as np
import pytest

import oracle_binding as oracle
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpuscan import GpuScan, STROM_RESULTS_ON_DEVICE

pytestmark = pytest.mark.gpu

FORMATS = ("row", "row_flat", "tupslot", "column")
C2_QUAL = "(and (int4lt (var 1 int4) (param 0 int4)) (float8gt (var 2 float8) (param 1 float8)))"


def make_table(n, seed, null_frac=0.0):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 2**31, n, dtype=np.int64).astype(np.int32)
    b = rng.random(n)
    an = (rng.random(n) < null_frac) if null_frac else None
    bn = (rng.random(n) < null_frac) if null_frac else None
    return a, b, an, bn


def canon(results):
    """results[] as a sorted array: order is unspecified, the set is not"""
    r = np.asarray(results, dtype=np.int64)
    return r[np.argsort(np.abs(r), kind="stable")]


def check(qual, buf, ext=(), row_map=None):
    rc_o, res_o = oracle.gpuscan(qual, buf, ext, row_map=row_map)
    scan = GpuScan(qual).begin(ext_params=ext)
    try:
        res = scan.scan_chunk(buf, row_map=row_map)
    finally:
        scan.end()
    assert res.errcode == rc_o
    assert res.nitems == len(res_o)
    assert np.array_equal(canon(res.results), canon(res_o))
    return res


@pytest.mark.parametrize("fmt", FORMATS)
@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 2047, 2048, 2049, 100003])
def test_c2_parity_sizes_and_formats(fmt, n):
    a, b, an, bn = make_table(n, 100 + n, 0.05 if n % 2 else 0.0)
    buf = kds.build_kds(fmt, [kds.Column("int4", a, an), kds.Column("float8", b, bn)])
    check(C2_QUAL, buf, [np.int32(2**31 * 0.55), 0.3])


@pytest.mark.parametrize("sel", ["none", "all"])
def test_nothing_and_everything_selected(sel):
    a, b, _, _ = make_table(50000, 4)
    buf = kds.build_kds("column", [kds.Column("int4", a), kds.Column("float8", b)])
    ext = [np.int32(-1), 2.0] if sel == "none" else [np.int32(2**31 - 1), -1.0]
    res = check(C2_QUAL, buf, ext)
    assert res.nitems == (0 if sel == "none" else 50000)


@pytest.mark.parametrize("fmt", FORMATS)
def test_recheck_rows_are_negative(fmt):
    rng = np.random.default_rng(6)
    a = rng.integers(2**31 - 50, 2**31, 5000, dtype=np.int64).astype(np.int32)
    b = rng.random(5000)
    buf = kds.build_kds(fmt, [kds.Column("int4", a), kds.Column("float8", b)])
    res = check("(int4gt (int4pl (var 1 int4) (const int4 25)) (const int4 0))", buf)
    assert len(res.recheck_rows()) > 1000 and len(res.passed_rows()) > 1000
    check("(float8lt (float8div (var 2 float8) (float8 (int4mi (var 1 int4) (var 1 int4)))) (const float8 1))", buf)


def test_three_valued_logic_matches_oracle():
    x = np.array([1, 1, 1, 0, 0, 0, 9, 9, 9] * 40, dtype=np.int32)
    y = np.array([1, 0, 9, 1, 0, 9, 1, 0, 9] * 40, dtype=np.int32)
    X = "(int4eq (var 1 int4) (const int4 1))"
    Y = "(int4eq (var 2 int4) (const int4 1))"
    for fmt in ("tupslot", "column"):
        buf = kds.build_kds(fmt, [kds.Column("int4", x, x == 9), kds.Column("int4", y, y == 9)])
        for qual in ("(and %s %s)" % (X, Y), "(or %s %s)" % (X, Y),
                     "(not (and %s %s))" % (X, Y), "(not (or %s %s))" % (X, Y),
                     "(is_not_true (and %s %s))" % (X, Y), "(is_unknown (or %s %s))" % (X, Y),
                     "(isnull (var 1 int4))", "(isnotnull (var 2 int4))",
                     "(is_false (not %s))" % X):
            check(qual, buf)


def test_mixed_types_casts_case_and_nan():
    rng = np.random.default_rng(8)
    n = 30000
    s = rng.integers(-300, 300, n).astype(np.int16)
    l = rng.integers(-2**40, 2**40, n).astype(np.int64)
    f = (rng.random(n) * 100 - 50).astype(np.float32)
    d = rng.random(n) * 6e9 - 3e9
    d[::97] = np.nan
    d[::101] = np.inf
    cols = [kds.Column("int2", s), kds.Column("int8", l), kds.Column("float4", f),
            kds.Column("float8", d)]
    for fmt in ("row", "column"):
        buf = kds.build_kds(fmt, cols)
        check("(and (int28lt (var 1 int2) (var 2 int8)) (float48gt (var 3 float4) (float8 (var 1 int2))))", buf)
        check("(int4gt (int4 (var 4 float8)) (const int4 0))", buf)          # range-checked cast
        check("(float8gt (var 4 float8) (const float8 1e300))", buf)          # NaN / inf ordering
        check("(int8eq (case (when (int2lt (var 1 int2) (const int2 0)) (int82mul (var 2 int8) (var 1 int2)))"
              " (else (var 2 int8))) (var 2 int8))", buf)
        check("(int2gt (int2mod (var 1 int2) (const int2 7)) (const int2 2))", buf)


def test_date_predicates():
    rng = np.random.default_rng(12)
    days = rng.integers(-3000, 3000, 20000).astype(np.int32)
    dn = rng.random(20000) < 0.02
    buf = kds.build_kds("column", [kds.Column("date", days, dn)])
    res = check("(date_le (var 1 date) (const date '1998-09-02'))", buf)
    cutoff = -486          # 1998-09-02 is 486 days before 2000-01-01
    assert res.nitems == int(np.sum((days <= cutoff) & ~dn))
    check("(timestamp_lt_date (const timestamp '2001-03-04 05:06:07') (date_pli (var 1 date) (const int4 10)))", buf)


def test_row_map_and_async_window():
    a, b, an, bn = make_table(40000, 21, 0.03)
    cols = [kds.Column("int4", a, an), kds.Column("float8", b, bn)]
    ext = [np.int32(2**31 * 0.4), 0.5]
    for fmt in ("row", "column"):
        buf = kds.build_kds(fmt, cols)
        rmap = np.random.default_rng(3).permutation(40000)[:12345].astype(np.int32)
        check(C2_QUAL, buf, ext, row_map=rmap)
    # five chunks in flight through one operator instance
    scan = GpuScan(C2_QUAL, max_async_chunks=3).begin(ext_params=ext)
    chunks, want = [], []
    for i in range(5):
        a, b, an, bn = make_table(30000 + i, 50 + i, 0.02)
        buf = kds.build_kds("column" if i % 2 else "row",
                            [kds.Column("int4", a, an), kds.Column("float8", b, bn)])
        chunks.append(buf)
        want.append(oracle.gpuscan(C2_QUAL, buf, ext)[1])
    for res, w in zip(scan.scan_chunks(chunks), want):
        assert np.array_equal(canon(res.results), canon(w))
    scan.end()


def test_resident_chunk_and_results_on_device():
    a, b, _, _ = make_table(300000, 77)
    buf = kds.build_kds("column", [kds.Column("int4", a), kds.Column("float8", b)])
    ext = [np.int32(2**31 * 0.5), 0.5]
    ds = runtime.DeviceStore.upload(buf)
    scan = GpuScan(C2_QUAL).begin(ext_params=ext)
    res = scan.scan_chunk(ds)
    _, res_o = oracle.gpuscan(C2_QUAL, buf, ext)
    assert np.array_equal(canon(res.results), canon(res_o))
    res2 = scan.scan_chunk(ds, flags=STROM_RESULTS_ON_DEVICE)
    assert res2.nitems == len(res_o)
    assert res2.perfmon["num_kern_exec"] == 1 and res2.perfmon["time_kern_exec_ns"] > 0
    scan.end()
    ds.release()


def test_full_size_c2_properties():
    """BASELINE configs[1]: 1e8-row (int4, float8) chunk.  Too big for the
    tuple-at-a-time oracle, so size-independent properties are checked:
    ids unique and in range, every id satisfies the predicate, the count
    equals an independent vectorised count."""
    n = 100_000_000
    rng = np.random.default_rng(2024)
    a = rng.integers(0, 2**31, n, dtype=np.int64).astype(np.int32)
    b = rng.random(n)
    buf = kds.build_kds("column", [kds.Column("int4", a), kds.Column("float8", b)])
    k, c = np.int32(2**31 * 0.3), 0.6
    ds = runtime.DeviceStore.upload(buf)
    del buf
    scan = GpuScan(C2_QUAL).begin(ext_params=[k, c])
    res = scan.scan_chunk(ds)
    scan.end()
    ds.release()
    assert res.errcode == 0
    ids = res.passed_rows()
    assert len(res.recheck_rows()) == 0
    assert len(ids) == res.nitems
    assert ids[0] >= 0 and ids[-1] < n and np.all(np.diff(ids) > 0)     # unique
    assert np.all(a[ids] < k) and np.all(b[ids] > c)
    assert res.nitems == int(np.count_nonzero((a < k) & (b > c)))


MATH_FUNCS = [   # (expression over column c and c+1, error-free domain, domain with errors)
    ("(exp (var %d float8))", (-700.0, 700.0), (-760.0, 760.0)),
    ("(ln (var %d float8))", (1e-3, 1e6), (-1.0, 1e6)),
    ("(log (var %d float8))", (1e-3, 1e6), (-1.0, 1e6)),
    ("(cbrt (var %d float8))", (-1e9, 1e9), (-1e9, 1e9)),
    ("(power (var %d float8) (var %d float8))", (0.1, 20.0), (-20.0, 20.0)),
    ("(degrees (var %d float8))", (-1e3, 1e3), (-1e3, 1e3)),
    ("(radians (var %d float8))", (-1e3, 1e3), (-1e3, 1e3)),
    ("(acos (var %d float8))", (-1.0, 1.0), (-1.2, 1.2)),
    ("(asin (var %d float8))", (-1.0, 1.0), (-1.2, 1.2)),
    ("(atan (var %d float8))", (-1e3, 1e3), (-1e3, 1e3)),
    ("(atan2 (var %d float8) (var %d float8))", (-5.0, 5.0), (-5.0, 5.0)),
    ("(cos (var %d float8))", (-50.0, 50.0), (-50.0, 50.0)),
    ("(sin (var %d float8))", (-50.0, 50.0), (-50.0, 50.0)),
    ("(tan (var %d float8))", (-1.5, 1.5), (-1.5, 1.5)),
]


def _math_table(n, seed, which):
    """column 1: row number; then one (x, y) pair of columns per function"""
    rng = np.random.default_rng(seed)
    cols = [kds.Column("int4", np.arange(n, dtype=np.int32))]
    exprs = []
    for f, (expr, safe, wild) in enumerate(MATH_FUNCS):
        lo, hi = (safe if which == "safe" else wild)
        x = rng.uniform(lo, hi, n)
        if which == "wild":
            x[f * 8:f * 8 + 8] = [0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-310, 0.5]
        y = np.round(rng.uniform(-6, 6, n) * 2) / 2             # halves: non-integer powers of negatives
        cols += [kds.Column("float8", x), kds.Column("float8", y)]
        c = 2 + 2 * f
        exprs.append(expr % ((c, c + 1) if expr.count("%d") == 2 else (c,)))
    return cols, exprs


def test_transcendental_functions_on_device():
    """codegen.c:467-503 on the device (ocml) against the oracle (glibc).  Which rows must go
    back to the CPU (domain error, infinite result of finite arguments, underflow to zero) --
    the same set; the values of error-free rows -- 1e-13 relative (both libraries are within a
    few ulp of the true value; tan and power amplify the argument's last bit)"""
    from pg_strom_amd.gpupreagg import GpuPreAgg
    runtime.init()
    n = 5000
    cols, exprs = _math_table(n, 20240, "wild")
    buf = kds.build_kds("column", cols)
    qual = "(or " + " ".join("(isnotnull %s)" % e for e in exprs) + ")"
    scan = GpuScan(qual).begin()
    try:
        res = scan.scan_chunk(buf)
    finally:
        scan.end()
    rc, want = oracle.gpuscan(qual, buf, [])
    assert res.errcode == rc and np.array_equal(np.sort(res.results), np.sort(want))
    assert 100 < len(res.recheck_rows()) < n
    cols, exprs = _math_table(n, 20241, "safe")
    buf = kds.build_kds("column", cols)
    spec = "(gpupreagg (key (var 1 int4)) " + " ".join("(pmax %s)" % e for e in exprs) + ")"
    agg = GpuPreAgg(spec).begin([(0, n)])
    try:
        assert agg.fold(buf)[0] == 0
        pr = agg.fetch()
    finally:
        agg.end()
    order = np.argsort(pr.column(0)[0])
    for f, e in enumerate(exprs):
        oid, v, isn, err = oracle.eval_rows(e, buf)
        assert not err.any(), e
        got = pr.column(1 + f)[0][order]
        assert np.allclose(got, v.view(np.float64), rtol=1e-13, atol=0), (e, np.max(np.abs(got / v.view(np.float64) - 1)))
