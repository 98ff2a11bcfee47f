"""
64-bit in-kernel NUMERIC, CPU side: the oracle's numeric arithmetic against
Python's Decimal (exact), the representable range the reference's
recheck_agg suite pins (expected/recheck_agg.out:23-51: sum(1E+48) and
sum(1E-32) stay on the device, 1E+49 / 1E-33 / 1E+-1000 go back to the CPU),
and fixed-point partial sums.
"""
from decimal import Decimal

import numpy as np
import pytest

import oracle_binding as oracle
from pg_strom_amd import kds, runtime


def test_representable_range_matches_recheck_suite():
    assert kds.numeric_encode("1E+48") is not None
    assert kds.numeric_encode("1E-32") is not None
    assert kds.numeric_encode("1E+49") is None
    assert kds.numeric_encode("1E-33") is None
    assert kds.numeric_encode("1E+1000") is None and kds.numeric_encode("1E-1000") is None
    for s in ("0", "1", "-1", "123.4500", "0.0001", "-987654321098.76543", "144115188075855871"):
        assert kds.numeric_decode(kds.numeric_encode(s)) == Decimal(s)
    assert kds.numeric_encode("144115188075855872") is None      # 2^57 does not fit
    # the expression IR's literal parser agrees
    assert runtime.expression_available("(numeric_gt (var 1 numeric) (const numeric 1E+48))")[0]
    assert not runtime.expression_available("(numeric_gt (var 1 numeric) (const numeric 1E+49))")[0]


def random_numerics(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        digits = int(rng.integers(1, 17))
        mant = int(rng.integers(0, 10 ** digits))
        scale = int(rng.integers(0, 12))
        sign = -1 if rng.random() < 0.4 else 1
        out.append(Decimal(sign * mant).scaleb(-scale))
    return out


@pytest.mark.parametrize("op,fn", [("numeric_add", lambda a, b: a + b),
                                   ("numeric_sub", lambda a, b: a - b),
                                   ("numeric_mul", lambda a, b: a * b)])
def test_arithmetic_is_exact_or_recheck(op, fn):
    a, b = random_numerics(3000, 1), random_numerics(3000, 2)
    buf = kds.build_kds("column", [kds.numeric_column(a), kds.numeric_column(b)])
    oid, v, isn, err = oracle.eval_rows("(%s (var 1 numeric) (var 2 numeric))" % op, buf)
    assert oid == 1700
    nre = 0
    for i in range(3000):
        exact = fn(a[i], b[i])
        if err[i] == 2:
            assert isn[i]
            nre += 1
        else:
            assert err[i] == 0 and not isn[i]
            assert kds.numeric_decode(v[i]) == exact, (a[i], b[i], exact)
            assert kds.numeric_encode(exact) == int(v[i])        # canonical image
    if op != "numeric_mul":
        assert nre < 1000         # rechecks are the exception, not the rule
    assert nre > 0 or op != "numeric_mul"


def test_compare_and_casts():
    a, b = random_numerics(2000, 3), random_numerics(2000, 4)
    b[:200] = a[:200]
    buf = kds.build_kds("row", [kds.numeric_column(a), kds.numeric_column(b)])
    for op, fn in (("lt", lambda x, y: x < y), ("le", lambda x, y: x <= y), ("eq", lambda x, y: x == y),
                   ("ne", lambda x, y: x != y), ("ge", lambda x, y: x >= y), ("gt", lambda x, y: x > y)):
        rc, res = oracle.gpuscan("(numeric_%s (var 1 numeric) (var 2 numeric))" % op, buf)
        want = [i + 1 for i in range(2000) if fn(a[i], b[i])]
        assert rc == 0 and list(res) == want
    # numeric -> int4 rounds half away from zero; out of range -> recheck
    vals = [Decimal("0.5"), Decimal("1.5"), Decimal("-0.5"), Decimal("-2.5"), Decimal("2.49"),
            Decimal("3000000000"), Decimal("12345.678")]
    buf = kds.build_kds("column", [kds.numeric_column(vals)])
    oid, v, isn, err = oracle.eval_rows("(int4 (var 1 numeric))", buf)
    got = [None if e else int(np.int64(x)) for x, e in zip(v.view(np.int64), err)]
    assert got == [1, 2, -1, -3, 2, None, 12346]
    oid, v, isn, err = oracle.eval_rows("(float8 (var 1 numeric))", buf)
    assert np.allclose(v.view(np.float64), [float(x) for x in vals], rtol=1e-15)
    oid, v, isn, err = oracle.eval_rows("(numeric (int4 (var 1 numeric)))", buf)
    assert kds.numeric_decode(v[6]) == Decimal(12346)


def test_fixed_point_partial_sums():
    rng = np.random.default_rng(9)
    price = [Decimal(int(rng.integers(90000, 10494950))).scaleb(-2) for _ in range(5000)]
    disc = [Decimal(int(rng.integers(0, 11))).scaleb(-2) for _ in range(5000)]
    flag = rng.integers(0, 3, 5000).astype(np.int8)
    buf = kds.build_kds("column", [kds.Column("char1", flag), kds.numeric_column(price),
                                   kds.numeric_column(disc)])
    spec = ("(gpupreagg (key (var 1 char1)) (nrows) (psum (var 2 numeric) 2)"
            " (psum (numeric_mul (var 2 numeric) (numeric_sub (const numeric 1) (var 3 numeric))) 4)"
            " (pmax (var 2 numeric) 2))")
    rc, v, isn = oracle.gpupreagg(spec, buf, 5)
    assert rc == 0 and len(v) == 3
    for i in range(3):
        k = int(v[i, 0].view(np.int64))
        rows = [j for j in range(5000) if flag[j] == k]
        assert int(v[i, 1]) == len(rows)
        assert Decimal(int(v[i, 2].view(np.int64))).scaleb(-2) == sum(price[j] for j in rows)
        assert Decimal(int(v[i, 3].view(np.int64))).scaleb(-4) == sum(price[j] * (1 - disc[j]) for j in rows)
        assert Decimal(int(v[i, 4].view(np.int64))).scaleb(-2) == max(price[j] for j in rows)
    # a value finer than the accumulator's scale sends the chunk back
    rc, v, isn = oracle.gpupreagg("(gpupreagg (psum (var 2 numeric) 1))", buf, 1)
    assert rc == 2


def test_typmod_scale_is_a_codegen_hint_only():
    """(var N numeric SCALE): the device emitter switches to fixed-point code
    (pg_fixed_t), the oracle ignores the hint -- same partial rows either way"""
    from pg_strom_amd.gpupreagg import codegen_gpupreagg
    rng = np.random.default_rng(10)
    price = [Decimal(int(rng.integers(90000, 10494950))).scaleb(-2) for _ in range(2000)]
    disc = [Decimal(int(rng.integers(0, 11))).scaleb(-2) for _ in range(2000)]
    buf = kds.build_kds("column", [kds.numeric_column(price), kds.numeric_column(disc)])
    plain = ("(gpupreagg (nrows) (psum (numeric_mul (var 1 numeric) "
             "(numeric_sub (const numeric 1) (var 2 numeric))) 4))")
    typed = plain.replace("(var 1 numeric)", "(var 1 numeric 2)").replace("(var 2 numeric)", "(var 2 numeric 2)")
    rc1, v1, n1 = oracle.gpupreagg(plain, buf, 2)
    rc2, v2, n2 = oracle.gpupreagg(typed, buf, 2)
    assert rc1 == rc2 == 0 and np.array_equal(v1, v2) and np.array_equal(n1, n2)
    assert Decimal(int(v1[0, 1].view(np.int64))).scaleb(-4) == sum(p * (1 - d) for p, d in zip(price, disc))

    def source_of(spec):
        cg = codegen_gpupreagg(spec)
        return (cg[0] if isinstance(cg, tuple) else cg).source
    src = source_of(typed)
    assert "pgfn_fixed_mul(" in src and "pg_fixed_cached(errcode, KV.KFIX_1_2)" in src and "pgfn_numeric_mul(" not in src
    # a scale-less operand anywhere pulls the expression back to the 64-bit numeric form
    mixed = source_of(plain.replace("(var 1 numeric)", "(var 1 numeric 2)"))
    assert "pgfn_numeric_mul(" in mixed and "pgfn_fixed_to_numeric(" in mixed
    # the literal meets a scale-less value as its kern_parambuf twin, not through a conversion
    assert "pg_fixed_lit(" not in source_of(plain)
    with pytest.raises(ValueError):
        codegen_gpupreagg("(gpupreagg (psum (var 1 int4 2)))")


def pg_numeric_varlena(d):
    """Decimal -> the bytes PostgreSQL 9.4 stores in a heap tuple for a
    numeric (utils/adt/numeric.c: base-10000 digits, short header when weight
    and dscale fit, 1-byte varlena header for these sizes).  Written from the
    format description, independently of the builder under test."""
    import struct
    sign, digs, exp = d.as_tuple()
    dscale = max(0, -exp)
    if not any(digs):
        groups, weight, sign = [], 0, 0
    else:
        s = "".join(map(str, digs))
        if exp >= 0:
            intpart, frac = s + "0" * exp, ""
        else:
            s = s.rjust(-exp + 1, "0")
            intpart, frac = s[:exp], s[exp:]
        intpart = intpart.lstrip("0")
        pad = (-len(intpart)) % 4
        intpart = "0" * pad + intpart
        frac = frac + "0" * ((-len(frac)) % 4)
        groups = [int(intpart[i:i + 4]) for i in range(0, len(intpart), 4)]
        weight = len(groups) - 1
        groups += [int(frac[i:i + 4]) for i in range(0, len(frac), 4)]
        while groups and groups[0] == 0:          # leading zero digits
            groups.pop(0)
            weight -= 1
        while groups and groups[-1] == 0:         # trailing zero digits
            groups.pop()
    if -64 <= weight <= 63 and dscale <= 63:
        body = struct.pack("<H", 0x8000 | (0x2000 if sign else 0) | (dscale << 7) |
                           (0x0040 if weight < 0 else 0) | (weight & 0x3F))
    else:
        body = struct.pack("<Hh", (0x4000 if sign else 0) | dscale, weight)
    body += b"".join(struct.pack("<H", g) for g in groups)
    total = 1 + len(body)
    assert total <= 126
    return bytes([(total << 1) | 1]) + body


def test_varlena_numeric_decode_and_heap_layout():
    """a21: PostgreSQL's on-disk numeric -> the 64-bit form (oracle side) and the
    host builder's heap tuples carry exactly those bytes"""
    cases = [Decimal(x) for x in ("0", "1", "-1", "0.5", "12.5", "-12.50", "0.0001", "10000", "9999.9999",
                                  "123456789.123456", "1E+20", "-1E-14", "100000000", "0.00000001",
                                  "99999999999999999", "-0.100", "5000", "1E+30", "1E-32")]
    for d in cases:
        raw = pg_numeric_varlena(d)
        assert oracle.numeric_from_varlena(raw) == kds.numeric_encode(d), d
    # long header (dscale beyond the short format's 6 bits is not reachable with 57-bit
    # mantissas, a weight beyond 63 is not either): decode a hand-made long datum
    import struct
    body = struct.pack("<Hh", 0x4000 | 2, 1) + struct.pack("<HHH", 12, 3456, 7800)   # -123456.78
    raw = bytes([((1 + len(body)) << 1) | 1]) + body
    assert oracle.numeric_from_varlena(raw) == kds.numeric_encode(Decimal("-123456.78"))
    # 4-byte varlena header, NaN, too many digits
    raw4 = struct.pack("<I", (4 + 4) << 2) + struct.pack("<HH", 0x8000, 7)
    assert oracle.numeric_from_varlena(raw4) == kds.numeric_encode(Decimal(7))
    assert oracle.numeric_from_varlena(bytes([(3 << 1) | 1]) + struct.pack("<H", 0xC000)) is None
    big = pg_numeric_varlena(Decimal("123456789012345678901234"))
    assert oracle.numeric_from_varlena(big) is None
    # the builder lays the same bytes into heap tuples: a scan through the oracle sees the values
    rng = np.random.default_rng(3)
    vals = [Decimal(int(rng.integers(-10**9, 10**9))).scaleb(-int(rng.integers(0, 7))) for _ in range(3000)]
    imgs = np.array([kds.numeric_encode(v) for v in vals], dtype=np.uint64)
    isnull = rng.random(3000) < 0.05
    a = np.arange(3000, dtype=np.int32)
    for fmt in ("row", "row_flat"):
        buf = kds.build_kds(fmt, [kds.Column("int4", a), kds.Column("numeric_varlena", imgs, isnull),
                                  kds.Column("int4", a * 2)])
        rc, res = oracle.gpuscan("(and (numeric_gt (var 2 numeric) (const numeric 12.5))"
                                 " (int4eq (var 3 int4) (int4mul (var 1 int4) (const int4 2))))", buf)
        assert rc == 0
        want = [i + 1 for i in range(3000) if not isnull[i] and vals[i] > Decimal("12.5")]
        assert sorted(res) == want
        rc, v, isn = oracle.gpupreagg("(gpupreagg (nrows) (psum (var 2 numeric) 6))", buf, 2)
        assert rc == 0
        assert Decimal(int(v[0, 1].view(np.int64))).scaleb(-6) == sum(x for x, n in zip(vals, isnull) if not n)
    with pytest.raises(ValueError):
        kds.build_kds("column", [kds.Column("numeric_varlena", imgs)])      # heap tuples only


def float_numeric_cases():
    rng = np.random.default_rng(77)
    f8 = np.concatenate([
        rng.normal(size=3000) * 10.0 ** rng.integers(-20, 30, 3000),
        rng.integers(-10**15, 10**15, 500).astype(np.float64),              # integers: exact
        np.array([0.0, -0.0, 1.0, -1.0, 0.1, 0.5, 1e15, 1e16, 999999999999999.9, 123456789012345.6,
                  2.5e-10, 1e-17, 9.999999999999999e22, 1e22, 1e23, 5e-324, 1e-33, 1e-40, 1.5e48, 1e49, 1e300,
                  np.inf, -np.inf, np.nan, 0.30000000000000004, 4.35, 2.675, 1e-5, 123456.5, 1234565.0])])
    f4 = np.concatenate([(rng.normal(size=2000) * 10.0 ** rng.integers(-12, 20, 2000)).astype(np.float32),
                         np.array([0.0, 1.0, 0.1, 16777216.0, 1234567.0, 999999.5, 3.4e38, 1e-30, 1e-38,
                                   np.inf, np.nan, 123456.5, 0.15625], dtype=np.float32)])
    return f8, f4


def postgres_float_numeric(x, dig):
    """float8_numeric / float4_numeric of PostgreSQL 9.4 (utils/adt/numeric.c): sprintf("%.*g", DBL_DIG /
    FLT_DIG, val), then numeric_in -- the value at 15 / 6 significant digits; NaN is a numeric NaN and
    infinity an error there: neither has a 64-bit device form"""
    if np.isnan(x) or np.isinf(x):
        return None
    return Decimal("%.*g" % (dig, float(x)))


def check_float_numeric(values, images, errs, dig):
    """an image per value (or CpuReCheck) against PostgreSQL's answer: exactly that value, as the
    canonical (normalised) image -- or CpuReCheck where the 64-bit form cannot hold it"""
    nre = 0
    for x, img, e in zip(values, images, errs):
        want = postgres_float_numeric(x, dig)
        if want is None or kds.numeric_encode(want) is None:
            assert e, (x, want)                     # not representable: CpuReCheck
            nre += 1
            continue
        assert not e, (x, want)
        assert kds.numeric_decode(img) == want and kds.numeric_encode(want) == int(img), (x, kds.numeric_decode(img), want)
    return nre


def test_float_to_numeric_is_postgresql_float_numeric():
    """numeric(float8) / numeric(float4) (codegen.c:519-520, float_to_numeric opencl_numeric.h:625-738):
    the oracle's statement against PostgreSQL's own definition, through Python's correctly rounded
    "%.15g" / "%.6g"."""
    f8, f4 = float_numeric_cases()
    buf = kds.build_kds("column", [kds.Column("float8", f8)])
    oid, v, isn, err = oracle.eval_rows("(numeric (var 1 float8))", buf)
    assert oid == 1700 and np.array_equal(isn, err != 0)    # a rechecked value is NULL + CpuReCheck
    assert check_float_numeric(f8, v, err, 15) >= 8
    buf = kds.build_kds("row", [kds.Column("float4", f4)])
    oid, v, isn, err = oracle.eval_rows("(numeric (var 1 float4))", buf)
    assert check_float_numeric(f4, v, err, 6) >= 2
    # and it composes: compared as numerics, summed as numerics
    buf = kds.build_kds("column", [kds.Column("float8", np.array([0.1, 0.2, 2.5]))])
    rc, res = oracle.gpuscan("(numeric_lt (numeric (var 1 float8)) (const numeric 0.15))", buf)
    assert rc == 0 and list(res) == [1]


def test_identity_casts_of_the_catalog():
    """date(date), time(time), timestamp(timestamp): alias casts (codegen.c:543-548)"""
    d = np.array([0, 7000, -1000], dtype=np.int32)
    t = np.array([0, 86399999999, 1], dtype=np.int64)
    buf = kds.build_kds("column", [kds.Column("date", d), kds.Column("time", t), kds.Column("timestamp", t * 1000)])
    for spec, col in (("(date (var 1 date))", d), ("(time (var 2 time))", t), ("(timestamp (var 3 timestamp))", t * 1000)):
        oid, v, isn, err = oracle.eval_rows(spec, buf)
        assert not err.any() and list(v.view(np.int64)[:3]) == [int(x) for x in col]


def test_numeric_image_columns_carry_integer_part_bounds():
    """KDS_COLSTAT_INTPART (include/strom_kds.h): a COLUMN chunk's numeric image column has no
    zone map of its bit patterns -- they do not order like the values -- but bounds of the values'
    integer parts, rounded outward; a value beyond int64 leaves the column without any"""
    import math
    vals = [Decimal("104949.50"), Decimal("-0.07"), Decimal("0"), Decimal("12345678.999"), Decimal("-99.01"), Decimal("3e10")]
    nul = np.array([0, 0, 0, 0, 0, 0], dtype=bool)
    buf = kds.build_kds("column", [kds.numeric_column(vals, nul), kds.Column("int4", np.arange(6, dtype=np.int32))])
    cd = kds.decode_column_chunk(buf)
    assert cd[0]["stat_flags"] == 4 and (cd[0]["minval"], cd[0]["maxval"]) == (-100, 30000000000)
    assert cd[1]["stat_flags"] == 1 and (cd[1]["minval"], cd[1]["maxval"]) == (0, 5)
    assert all(cd[0]["minval"] <= math.floor(v) and math.ceil(v) <= cd[0]["maxval"] for v in vals)
    nul[5] = True                                   # NULLs do not count
    cd = kds.decode_column_chunk(kds.build_kds("column", [kds.numeric_column(vals, nul)]))
    assert (cd[0]["stat_flags"], cd[0]["minval"], cd[0]["maxval"]) == (4, -100, 12345679)
    big = kds.decode_column_chunk(kds.build_kds("column", [kds.numeric_column([Decimal("1e25"), Decimal("1")])]))
    assert big[0]["stat_flags"] == 0
