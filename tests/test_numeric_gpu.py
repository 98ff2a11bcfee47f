"""
64-bit NUMERIC on the device (needs an MI355X: -m gpu): scalar functions,
numeric quals in GpuScan, fixed-point numeric partials in GpuPreAgg and the
TPC-H Q1-shaped aggregation (BASELINE configs[4] shape at test size).
Exact arithmetic: every result must equal the oracle's bit for bit; the
oracle is checked against Python's Decimal in tests/test_numeric_cpu.py.
"""
from decimal import Decimal

import numpy as np
import pytest

import oracle_binding as oracle
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpupreagg import GpuPreAgg
from pg_strom_amd.gpuscan import GpuScan
from test_numeric_cpu import random_numerics

pytestmark = pytest.mark.gpu


def scan_parity(qual, buf, ext=()):
    rc, want = oracle.gpuscan(qual, buf, ext)
    scan = GpuScan(qual).begin(ext_params=ext)
    res = scan.scan_chunk(buf)
    scan.end()
    assert res.errcode == rc
    assert np.array_equal(np.sort(res.results), np.sort(want))
    return res


@pytest.mark.parametrize("fmt", ["column", "row", "tupslot"])
def test_numeric_quals(fmt):
    a, b = random_numerics(20000, 11), random_numerics(20000, 12)
    an = np.random.default_rng(1).random(20000) < 0.03
    i4 = np.random.default_rng(2).integers(-1000, 1000, 20000).astype(np.int32)
    buf = kds.build_kds(fmt, [kds.numeric_column(a, an), kds.numeric_column(b), kds.Column("int4", i4)])
    for qual in ("(numeric_lt (var 1 numeric) (var 2 numeric))",
                 "(numeric_ge (numeric_add (var 1 numeric) (var 2 numeric)) (const numeric 0.5))",
                 "(numeric_gt (numeric_mul (var 1 numeric) (numeric (var 3 int4))) (const numeric -12.50))",
                 "(numeric_eq (numeric_sub (var 1 numeric) (var 1 numeric)) (const numeric 0))",
                 "(int4gt (int4 (numeric_abs (var 2 numeric))) (var 3 int4))",
                 "(float8lt (float8 (numeric_uminus (var 1 numeric))) (const float8 0.25))",
                 "(numeric_ne (var 1 numeric) (param 0 numeric))"):
        ext = [np.uint64(kds.numeric_encode("12.5"))] if "param" in qual else ()
        res = scan_parity(qual, buf, ext)
    assert res is not None


def test_device_arithmetic_equals_oracle_values():
    """device result == oracle result for every row that is not a recheck"""
    a, b = random_numerics(30000, 21), random_numerics(30000, 22)
    base = [kds.numeric_column(a), kds.numeric_column(b)]
    buf = kds.build_kds("column", base)
    for op in ("numeric_add", "numeric_sub", "numeric_mul"):
        expr = "(%s (var 1 numeric) (var 2 numeric))" % op
        oid, v, isn, err = oracle.eval_rows(expr, buf)
        exp_col = kds.Column("numeric", v, isn)
        buf3 = kds.build_kds("column", base + [exp_col])
        scan = GpuScan("(numeric_eq %s (var 3 numeric))" % expr).begin()
        res = scan.scan_chunk(buf3)
        scan.end()
        ok_rows = np.nonzero(err == 0)[0]
        re_rows = np.nonzero(err == 2)[0]
        assert np.array_equal(res.passed_rows(), ok_rows)
        assert np.array_equal(res.recheck_rows(), re_rows)


def fetch_rows(agg):
    pr = agg.fetch()
    out = []
    for i in range(len(pr)):
        row = []
        for t, (kind, oid) in enumerate(pr.targets):
            v, n = pr.column(t)
            row.append(None if n[i] else (kds.numeric_decode(v[i]) if oid == 1700 and kind != 2 else v[i].item()))
        out.append(row)
    return out


def test_fixed_point_numeric_partials_and_wide_sums():
    rng = np.random.default_rng(5)
    n = 40000
    price = [Decimal(int(rng.integers(90000, 10494950))).scaleb(-2) for _ in range(n)]
    disc = [Decimal(int(rng.integers(0, 11))).scaleb(-2) for _ in range(n)]
    flag = rng.integers(0, 3, n).astype(np.int8)
    pn = rng.random(n) < 0.02
    buf = kds.build_kds("column", [kds.Column("char1", flag), kds.numeric_column(price, pn),
                                   kds.numeric_column(disc)])
    spec = ("(gpupreagg (key (var 1 char1)) (nrows (isnotnull (var 2 numeric))) (psum (var 2 numeric) 2)"
            " (psum (numeric_mul (var 2 numeric) (numeric_sub (const numeric 1) (var 3 numeric))) 4)"
            " (pmin (var 2 numeric) 2) (pmax (var 2 numeric) 2))")
    agg = GpuPreAgg(spec).begin([(0, 3)])
    assert agg.fold(buf)[0] == 0
    rows = {r[0]: r for r in fetch_rows(agg)}
    agg.end()
    assert sorted(rows) == [0, 1, 2]
    for k in range(3):
        idx = [j for j in range(n) if flag[j] == k and not pn[j]]
        r = rows[k]
        assert r[1] == len(idx)
        assert r[2] == sum(price[j] for j in idx)
        assert r[3] == sum(price[j] * (1 - disc[j]) for j in idx)
        assert r[4] == min(price[j] for j in idx) and r[5] == max(price[j] for j in idx)
    # a sum wider than the 57-bit mantissa comes back as two partial rows that add up
    big = [Decimal(123456789012345678 + i) for i in range(10)]
    buf = kds.build_kds("column", [kds.numeric_column(big)])
    agg = GpuPreAgg("(gpupreagg (nrows) (psum (var 1 numeric) 0))").begin([])
    assert agg.fold(buf)[0] == 0
    rows = fetch_rows(agg)
    agg.end()
    assert len(rows) == 2
    assert sum(r[0] for r in rows) == 10
    assert sum(r[1] for r in rows if r[1] is not None) == sum(big)


@pytest.mark.parametrize("compact,typmod", [(False, False), (True, False), (True, True)])
def test_tpch_q1_shape(compact, typmod):
    """returnflag, linestatus, sum(qty), sum(price), sum(price*(1-disc)),
    sum(price*(1-disc)*(1+tax)), avg(qty), avg(price), avg(disc), count(*)
    WHERE shipdate <= date '1998-09-02' (SURVEY.md section 8d, C5 columns)"""
    rng = np.random.default_rng(1998)
    n = 200000
    rf = rng.choice(np.array([65, 78, 82], dtype=np.int8), n)          # A N R
    ls = rng.choice(np.array([70, 79], dtype=np.int8), n)              # F O
    qty = [Decimal(int(x)) for x in rng.integers(1, 51, n)]
    price = [Decimal(int(x)).scaleb(-2) for x in rng.integers(90000, 10494951, n)]
    disc = [Decimal(int(x)).scaleb(-2) for x in rng.integers(0, 11, n)]
    tax = [Decimal(int(x)).scaleb(-2) for x in rng.integers(0, 9, n)]
    ship = rng.integers(-2922, -2922 + 2526, n).astype(np.int32)       # 1992-01-02 .. 1998-12-01
    cols = [kds.Column("char1", rf), kds.Column("char1", ls), kds.numeric_column(qty),
            kds.numeric_column(price), kds.numeric_column(disc), kds.numeric_column(tax),
            kds.Column("date", ship)]
    # typmod: the plan passes the columns' numeric(p,s) scale -> fixed-point device code
    v_qty, v_prc, v_dsc, v_tax = (("(var 3 numeric 0)", "(var 4 numeric 2)", "(var 5 numeric 2)", "(var 6 numeric 2)")
                                  if typmod else
                                  ("(var 3 numeric)", "(var 4 numeric)", "(var 5 numeric)", "(var 6 numeric)"))
    one_minus_d = "(numeric_sub (const numeric 1) %s)" % v_dsc
    one_plus_t = "(numeric_add (const numeric 1) %s)" % v_tax
    disc_price = "(numeric_mul %s %s)" % (v_prc, one_minus_d)
    spec = ("(gpupreagg (qual (date_le (var 7 date) (const date '1998-09-02')))"
            " (key (var 1 char1)) (key (var 2 char1))"
            " (psum %s 0) (psum %s 2) (psum %s 4) (psum (numeric_mul %s %s) 6)"
            " (nrows (isnotnull %s)) (nrows (isnotnull %s))"
            " (psum %s 2) (nrows (isnotnull %s)) (nrows))"
            % (v_qty, v_prc, disc_price, disc_price, one_plus_t, v_qty, v_prc, v_dsc, v_dsc))
    half = n // 2
    chunks = [kds.build_kds("column", [kds.Column(c.sqltype, c.values[s], None) for c in cols])
              for s in (slice(0, half), slice(half, n))]
    agg = GpuPreAgg(spec).begin([(65, 18), (70, 10)])
    if compact:
        # 19 x 11 dense ids, 6 combinations occur: agree on 6 table slots first
        for b in chunks:
            agg.census(b)
        assert agg.compact() == 6
    for b in chunks:
        assert agg.fold(b)[0] == 0
    got = {(r[0], r[1]): r for r in fetch_rows(agg)}
    agg.end()
    cutoff = -486
    assert len(got) == 6
    for (f, s), r in got.items():
        idx = [j for j in range(n) if rf[j] == f and ls[j] == s and ship[j] <= cutoff]
        assert r[2] == sum(qty[j] for j in idx)
        assert r[3] == sum(price[j] for j in idx)
        assert r[4] == sum(price[j] * (1 - disc[j]) for j in idx)
        assert r[5] == sum(price[j] * (1 - disc[j]) * (1 + tax[j]) for j in idx)
        assert r[6] == r[7] == r[9] == r[10] == len(idx)
        assert r[8] == sum(disc[j] for j in idx)
    # and bit for bit against the oracle's partial rows
    merged = {}
    for b in chunks:
        rc, v, isn = oracle.gpupreagg(spec, b, 11)
        assert rc == 0
        for i in range(len(v)):
            key = (int(v[i, 0].view(np.int64)), int(v[i, 1].view(np.int64)))
            acc = merged.setdefault(key, [0] * 9)
            for t in range(2, 11):
                acc[t - 2] += int(v[i, t].view(np.int64))
    scales = [0, 2, 4, 6, None, None, 2, None, None]
    for key, acc in merged.items():
        r = got[key]
        for t, (a, sc) in enumerate(zip(acc, scales)):
            want = a if sc is None else Decimal(a).scaleb(-sc)
            assert r[t + 2] == want


@pytest.mark.parametrize("fmt", ["row", "row_flat"])
def test_varlena_numerics_in_heap_tuples(fmt):
    """a21: the kernels decode PostgreSQL's on-disk numeric (opencl_numeric.h:166-307)
    straight from heap tuples; scan, aggregate and device ingest against the oracle"""
    rng = np.random.default_rng(31)
    n = 50021
    vals = [Decimal(int(rng.integers(-10**9, 10**9))).scaleb(-int(rng.integers(0, 7))) for _ in range(n)]
    imgs = np.array([kds.numeric_encode(v) for v in vals], dtype=np.uint64)
    isnull = rng.random(n) < 0.05
    g = rng.integers(0, 7, n).astype(np.int32)
    cols = [kds.Column("int4", g), kds.Column("numeric_varlena", imgs, isnull),
            kds.Column("int4", np.arange(n, dtype=np.int32))]
    buf = kds.build_kds(fmt, cols)
    qual = "(and (numeric_gt (var 2 numeric) (const numeric 12.5)) (int4lt (var 3 int4) (const int4 40000)))"
    rc_o, res_o = oracle.gpuscan(qual, buf)
    scan = GpuScan(qual).begin()
    res = scan.scan_chunk(buf)
    scan.end()
    assert res.errcode == rc_o == 0
    assert np.array_equal(np.sort(res.results), np.sort(res_o))
    spec = ("(gpupreagg (key (var 1 int4)) (nrows) (psum (var 2 numeric) 6)"
            " (psum (numeric_mul (var 2 numeric) (const numeric 1.5)) 7))")
    agg = GpuPreAgg(spec).begin([(0, 7)])
    assert agg.fold(buf)[0] == 0
    got = {}
    for r in fetch_rows(agg):            # a wide sum comes back as several partial rows per key
        acc = got.setdefault(r[0], [0, Decimal(0), Decimal(0)])
        acc[0] += r[1]
        acc[1] += r[2] if r[2] is not None else 0
        acc[2] += r[3] if r[3] is not None else 0
    agg.end()
    for k in range(7):
        idx = [i for i in range(n) if g[i] == k and not isnull[i]]
        assert got[k][0] == int(np.count_nonzero(g == k))
        assert got[k][1] == sum(vals[i] for i in idx)
        assert got[k][2] == sum(vals[i] * Decimal("1.5") for i in idx)
    # device ingest turns the column into the 8-byte form with canonical images
    ds = runtime.DeviceStore.upload(buf)
    col, _ = ds.to_column([23, 1700, 23])
    dec = kds.decode_column_chunk(col.download())
    col.release()
    ds.release()
    assert np.array_equal(dec[1]["notnull"], ~isnull)
    assert np.array_equal(dec[1]["values"].view(np.uint64)[~isnull], imgs[~isnull])
    assert not dec[1]["values"][isnull].any()
    # ... and leaves bounds of the values' integer parts, outward (KDS_COLSTAT_INTPART): what lets
    # GpuPreAgg bound a sum over the column without measuring the rows
    import math
    live = [vals[i] for i in range(n) if not isnull[i]]
    assert dec[1]["stat_flags"] == 4
    assert dec[1]["minval"] == min(math.floor(v) for v in live) and dec[1]["maxval"] == max(math.ceil(v) for v in live)
    # the host builder states the same for the same column
    host = kds.decode_column_chunk(kds.build_kds("column", [kds.Column("int4", g), kds.Column("numeric", imgs, isnull),
                                                             kds.Column("int4", np.arange(n, dtype=np.int32))]))
    assert (host[1]["stat_flags"], host[1]["minval"], host[1]["maxval"]) == (4, dec[1]["minval"], dec[1]["maxval"])


@pytest.mark.parametrize("ftype,dig", [("float8", 15), ("float4", 6)])
def test_float_to_numeric_on_device_equals_oracle(ftype, dig):
    """numeric(float8) / numeric(float4) (codegen.c:519-520, float_to_numeric opencl_numeric.h:625-738):
    the device's 192-bit exact scaling == the oracle's (== PostgreSQL's "%.15g" / "%.6g",
    tests/test_numeric_cpu.py) for every row, rechecks included; and the catalog's alias casts
    date(date) / time(time) / timestamp(timestamp) in the same program"""
    from test_numeric_cpu import float_numeric_cases
    f8, f4 = float_numeric_cases()
    vals = f8 if ftype == "float8" else f4
    n = len(vals)
    d = (np.arange(n) % 9000 - 4000).astype(np.int32)
    expr = "(numeric (var 1 %s))" % ftype
    buf = kds.build_kds("column", [kds.Column(ftype, vals), kds.Column("date", d)])
    oid, v, isn, err = oracle.eval_rows(expr, buf)
    buf3 = kds.build_kds("column", [kds.Column(ftype, vals), kds.Column("date", d), kds.Column("numeric", v, isn)])
    scan = GpuScan("(and (numeric_eq %s (var 3 numeric)) (date_eq (date (var 2 date)) (var 2 date)))" % expr).begin()
    res = scan.scan_chunk(buf3)
    scan.end()
    assert np.array_equal(res.passed_rows(), np.nonzero(err == 0)[0])
    assert np.array_equal(res.recheck_rows(), np.nonzero(err == 2)[0])
    assert (err == 2).sum() >= 2 and (err == 0).sum() > 0.9 * n
