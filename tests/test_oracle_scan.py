"""
The CPU oracle's GpuScan against independent numpy restatements (no GPU):
every chunk format, NULLs, three-valued logic, arithmetic that must come
back as CpuReCheck (negative row ids), row maps, empty chunks.
Semantics pinned: opencl_gpuscan.h:98-177 (row status -> results[]),
opencl_mathlib.h:34-53 (overflow => NULL + CpuReCheck),
opencl_common.h:132-144 (error priority).
"""
import numpy as np
import pytest

import oracle_binding as oracle
from pg_strom_amd import kds

FORMATS = ("row", "row_flat", "tupslot", "column")
C2_QUAL = "(and (int4lt (var 1 int4) (param 0 int4)) (float8gt (var 2 float8) (param 1 float8)))"


def make_table(n, seed, null_frac=0.0):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 2**31, n, dtype=np.int64).astype(np.int32)
    b = rng.random(n)
    an = (rng.random(n) < null_frac) if null_frac else None
    bn = (rng.random(n) < null_frac) if null_frac else None
    return a, b, an, bn


@pytest.mark.parametrize("fmt", FORMATS)
@pytest.mark.parametrize("null_frac", [0.0, 0.07])
def test_c2_predicate_all_formats(fmt, null_frac):
    a, b, an, bn = make_table(5000, 11, null_frac)
    buf = kds.build_kds(fmt, [kds.Column("int4", a, an), kds.Column("float8", b, bn)])
    k, c = np.int32(2**31 * 0.6), 0.25
    rc, res = oracle.gpuscan(C2_QUAL, buf, [k, c])
    m = (a < k) & (b > c)
    if an is not None:
        m &= ~an & ~bn
    assert rc == 0
    assert np.array_equal(res, np.nonzero(m)[0] + 1)


def test_empty_chunk_and_null_param():
    a, b, _, _ = make_table(0, 1)
    for fmt in FORMATS:
        buf = kds.build_kds(fmt, [kds.Column("int4", a), kds.Column("float8", b)])
        rc, res = oracle.gpuscan(C2_QUAL, buf, [np.int32(1), 0.5])
        assert rc == 0 and len(res) == 0
    a, b, _, _ = make_table(100, 2)
    buf = kds.build_kds("column", [kds.Column("int4", a), kds.Column("float8", b)])
    rc, res = oracle.gpuscan(C2_QUAL, buf, [None, 0.5])      # NULL parameter: nothing passes
    assert rc == 0 and len(res) == 0


def test_overflow_rows_come_back_negative():
    # int4pl overflow -> result NULL + CpuReCheck: the row is reported as -(i+1)
    a = np.array([1, 2**31 - 1, -5, 2**31 - 1, 10], dtype=np.int32)
    b = np.zeros(5)
    buf = kds.build_kds("row", [kds.Column("int4", a), kds.Column("float8", b)])
    rc, res = oracle.gpuscan("(int4gt (int4pl (var 1 int4) (const int4 1)) (const int4 0))", buf)
    assert rc == 0
    assert list(res) == [1, -2, -4, 5]
    # division by zero likewise
    rc, res = oracle.gpuscan("(int4eq (int4div (const int4 10) (var 1 int4)) (const int4 1))",
                             kds.build_kds("column", [kds.Column("int4", np.array([10, 0, 3], dtype=np.int32))]))
    assert list(res) == [1, -2]


def test_three_valued_logic():
    # columns x, y nullable booleans built from int4 comparisons
    x = np.array([1, 1, 1, 0, 0, 0, 9, 9, 9], dtype=np.int32)
    y = np.array([1, 0, 9, 1, 0, 9, 1, 0, 9], dtype=np.int32)
    xn, yn = x == 9, y == 9
    buf = kds.build_kds("tupslot", [kds.Column("int4", x, xn), kds.Column("int4", y, yn)])
    X = "(int4eq (var 1 int4) (const int4 1))"
    Y = "(int4eq (var 2 int4) (const int4 1))"
    T, F, N = True, False, None
    xs = [T, T, T, F, F, F, N, N, N]
    ys = [T, F, N, T, F, N, T, F, N]

    def k_and(p, q):
        if p is False or q is False: return False
        if p is None or q is None: return None
        return True

    def k_or(p, q):
        if p is True or q is True: return True
        if p is None or q is None: return None
        return False

    def k_not(p):
        return None if p is None else (not p)

    cases = {
        "(and %s %s)" % (X, Y): [k_and(p, q) for p, q in zip(xs, ys)],
        "(or %s %s)" % (X, Y): [k_or(p, q) for p, q in zip(xs, ys)],
        "(not (and %s %s))" % (X, Y): [k_not(k_and(p, q)) for p, q in zip(xs, ys)],
        "(not (or %s %s))" % (X, Y): [k_not(k_or(p, q)) for p, q in zip(xs, ys)],
        "(is_not_true (and %s %s))" % (X, Y): [k_and(p, q) is not True for p, q in zip(xs, ys)],
        "(is_unknown (or %s %s))" % (X, Y): [k_or(p, q) is None for p, q in zip(xs, ys)],
        "(isnull (var 1 int4))": [p is None for p in xs],
        "(isnotnull (var 2 int4))": [q is not None for q in ys],
    }
    for qual, truth in cases.items():
        rc, res = oracle.gpuscan(qual, buf)
        want = [i + 1 for i, t in enumerate(truth) if t is True]
        assert list(res) == want, qual


def test_row_map_restricts_and_orders_rows():
    a, b, _, _ = make_table(1000, 5)
    buf = kds.build_kds("row_flat", [kds.Column("int4", a), kds.Column("float8", b)])
    rmap = np.array([999, 3, 500, 4, 0, 77], dtype=np.int32)
    k, c = np.int32(2**31 - 1), -1.0           # everything passes
    rc, res = oracle.gpuscan(C2_QUAL, buf, [k, c], row_map=rmap)
    assert list(res) == list(rmap + 1)


def test_mixed_width_types_and_casts():
    rng = np.random.default_rng(8)
    n = 2000
    s = rng.integers(-300, 300, n).astype(np.int16)
    l = rng.integers(-2**40, 2**40, n).astype(np.int64)
    f = (rng.random(n) * 100 - 50).astype(np.float32)
    buf = kds.build_kds("row", [kds.Column("int2", s), kds.Column("int8", l), kds.Column("float4", f)])
    qual = "(and (int28lt (var 1 int2) (var 2 int8)) (float48gt (var 3 float4) (float8 (var 1 int2))))"
    rc, res = oracle.gpuscan(qual, buf)
    m = (s.astype(np.int64) < l) & (f.astype(np.float64) > s.astype(np.float64))
    assert np.array_equal(res, np.nonzero(m)[0] + 1)
    # float -> int cast rounds half to even like PostgreSQL's dtoi4
    v = np.array([0.5, 1.5, 2.5, -0.5, -1.5, 3e9], dtype=np.float64)
    buf = kds.build_kds("column", [kds.Column("float8", v)])
    rc, res = oracle.gpuscan("(int4eq (int4 (var 1 float8)) (const int4 2))", buf)
    assert list(res) == [2, 3, -6]           # 1.5->2, 2.5->2, 3e9 out of range -> recheck


def test_nan_orders_above_everything():
    v = np.array([1.0, np.nan, np.inf, -np.inf], dtype=np.float64)
    buf = kds.build_kds("column", [kds.Column("float8", v)])
    rc, res = oracle.gpuscan("(float8gt (var 1 float8) (const float8 1e308))", buf)
    assert list(res) == [2, 3]               # NaN and +inf
    rc, res = oracle.gpuscan("(float8eq (var 1 float8) (var 1 float8))", buf)
    assert list(res) == [1, 2, 3, 4]         # NaN = NaN in PostgreSQL's ordering


def test_transcendental_functions_follow_postgresql_float_c():
    """codegen.c:467-503 (cbrt exp ln log power degrees radians acos asin atan atan2 cos sin
    tan): values as libm gives them; PostgreSQL 9.4's float.c raises an ERROR for a domain
    error, an infinite result of finite arguments and an underflow to zero -- CpuReCheck here"""
    x = np.array([0.5, 2.0, -1.5, 0.0, 1.0, 1e308, np.nan, -0.3, 700.0, -800.0, np.inf, -np.inf, 1e-320])
    y = np.array([2.0, -1.0, 3.0, -1.0, 0.0, 2.0, 1.0, 0.5, 1.0, 1.0, 1.0, 2.0, 2.0])
    buf = kds.build_kds("column", [kds.Column("float8", x), kds.Column("float8", y)])

    def run(expr):
        oid, v, n, err = oracle.eval_rows(expr, buf)
        assert oid == 701
        return v.view(np.float64), n, err

    with np.errstate(all="ignore"):
        v, n, err = run("(exp (var 1 float8))")
        want = np.exp(x)
        bad = (np.isinf(want) & ~np.isinf(x)) | (want == 0.0)
        assert np.array_equal(err != 0, bad) and np.allclose(v[~bad], want[~bad], rtol=1e-15, equal_nan=True)
        v, n, err = run("(ln (var 1 float8))")
        bad = x <= 0.0
        assert np.array_equal(err != 0, bad) and np.allclose(v[~bad], np.log(x[~bad]), rtol=1e-15, equal_nan=True)
        v, n, err = run("(log (var 1 float8))")
        assert np.array_equal(err != 0, bad) and np.allclose(v[~bad], np.log10(x[~bad]), rtol=1e-15, equal_nan=True)
        v, n, err = run("(acos (var 1 float8))")
        bad = (x < -1) | (x > 1)
        assert np.array_equal(err != 0, bad) and np.allclose(v[~bad], np.arccos(x[~bad]), rtol=1e-15, equal_nan=True)
        v, n, err = run("(sin (var 1 float8))")
        bad = np.isinf(x)
        assert np.array_equal(err != 0, bad) and np.allclose(v[~bad], np.sin(x[~bad]), rtol=1e-15, equal_nan=True)
        v, n, err = run("(power (var 1 float8) (var 2 float8))")
        want = np.power(x, y)
        bad = ((x == 0) & (y < 0)) | ((x < 0) & (np.floor(y) != y)) | \
            (np.isinf(want) & ~(np.isinf(x) | np.isinf(y))) | ((want == 0) & (x != 0))
        assert np.array_equal(err != 0, bad) and np.allclose(v[~bad], want[~bad], rtol=1e-15, equal_nan=True)
        v, n, err = run("(atan2 (var 1 float8) (var 2 float8))")
        assert not err.any() and np.allclose(v, np.arctan2(x, y), rtol=1e-15, equal_nan=True)
        v, n, err = run("(cbrt (var 1 float8))")
        assert not err.any() and np.allclose(v, np.cbrt(x), rtol=1e-15, equal_nan=True)
