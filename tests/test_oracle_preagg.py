"""
CPU oracle for GpuPreAgg against the reference's own regression results
(no GPU).  This is the pin that keeps the oracle honest: 300+ queries whose
answers were printed by stock PostgreSQL (expected/*.out), over the
regenerated gpupreagg_test fixture -- integer aggregates exact, avg(int)
exact as numeric text, float aggregates at the precision PostgreSQL printed
(extra_float_digits = -3).
"""
import numpy as np
import pytest

import agg_golden
import oracle_binding as oracle
from pg_strom_amd import kds


def decode_oracle_column(values, isnull, kind):
    if kind == "float":
        return values.view(np.float64) if values.flags.c_contiguous else \
            np.ascontiguousarray(values).view(np.float64), isnull
    return np.ascontiguousarray(values).view(np.int64), isnull


def oracle_runner(chunks):
    def run(plan):
        vals, nulls = [], []
        for buf, rows in chunks[plan["table"]]:
            rc, v, n = oracle.gpupreagg(plan["spec"], buf, plan["ntargets"])
            assert rc in (0, 2)
            if rc == 2:
                # int8 partial sum overflowed: the chunk goes to the CPU path
                assert plan["type"] == "int8"
                v, n = agg_golden.cpu_fallback_rows(plan, rows)
            vals.append(v)
            nulls.append(n)
        return np.concatenate(vals), np.concatenate(nulls)
    return run


@pytest.mark.parametrize("fmt,nchunks", [("row", 1), ("column", 3), ("tupslot", 2)])
def test_reference_regression_suites(fmt, nchunks):
    chunks = {"gpupreagg_test": agg_golden.fixture_chunks(fmt, nchunks),
              "gpupreagg_zero_test": agg_golden.fixture_chunks(fmt, 1, empty=True)}
    run = oracle_runner(chunks)
    total = 0
    for suite in ("nogrp_agg", "group_agg", "where_agg", "zero_agg"):
        checked, skipped = agg_golden.check_suite(suite, run, decode_oracle_column)
        assert checked >= 50, (suite, checked, skipped)
        total += checked
    assert total >= 200        # 50 in-catalog aggregates x 4 suites


def test_partial_semantics_small():
    k = np.array([1, 1, 2, 2, 2, 0, 0], dtype=np.int32)
    kn = np.array([0, 0, 0, 0, 0, 1, 1], dtype=np.uint8)
    x = np.array([10, 20, 5, 0, 7, 1, 2], dtype=np.int32)
    xn = np.array([0, 1, 0, 1, 0, 1, 1], dtype=np.uint8)
    buf = kds.build_kds("row", [kds.Column("int4", k, kn), kds.Column("int4", x, xn)])
    spec = ("(gpupreagg (key (var 1 int4)) (nrows) (nrows (isnotnull (var 2 int4)))"
            " (psum (int8 (var 2 int4))) (pmin (var 2 int4)) (pmax (var 2 int4)))")
    rc, v, n = oracle.gpupreagg(spec, buf, 6)
    assert rc == 0 and len(v) == 3
    rows = {(None if n[i, 0] else int(v[i, 0])): i for i in range(3)}
    i = rows[1]
    assert list(v[i, 1:].view(np.int64)) == [2, 1, 10, 10, 10] and not n[i, 1:].any()
    i = rows[2]
    assert list(v[i, 1:].view(np.int64)) == [3, 2, 12, 5, 7]
    i = rows[None]                       # NULL keys form one group; all inputs NULL
    assert list(v[i, 1:3].view(np.int64)) == [2, 0] and list(n[i, 3:]) == [True, True, True]


def test_recheck_makes_the_chunk_recheck():
    x = np.array([1, 2**31 - 1, 3], dtype=np.int32)
    buf = kds.build_kds("column", [kds.Column("int4", x)])
    rc, v, n = oracle.gpupreagg("(gpupreagg (psum (int8 (int4pl (var 1 int4) (const int4 1)))))", buf, 1)
    assert rc == 2 and len(v) == 0


def test_integer_sum_overflow_is_the_reference_rule_and_nothing_else():
    """CHECK_OVERFLOW_INT(accum, newval) of GPUPREAGG_AGGCALC_PSUM_TEMPLATE
    (opencl_gpupreagg.h:142-143, 933-948): an accumulate that leaves int8 is CpuReCheck --
    and ONLY that: a single row of 2^63 - 1, or of -2^63, or four times 2^61 minus one, is a
    sum like any other (round 2's oracle also sent every |input| >= 2^62 back, mirroring a
    screen of the device; the reference has no such rule)"""
    spec = "(gpupreagg (key (var 1 int4)) (nrows) (psum (var 2 int8)))"
    big = (1 << 63) - 1
    g = np.array([0, 1, 2, 2, 2, 2, 3], dtype=np.int32)
    x = np.array([big, -big - 1, 1 << 61, 1 << 61, 1 << 61, (1 << 61) - 1, 5], dtype=np.int64)
    rc, v, n = oracle.gpupreagg(spec, kds.build_kds("column", [kds.Column("int4", g), kds.Column("int8", x)]), 3)
    assert rc == 0
    got = {int(v[i, 0]): (int(v[i, 1].view(np.int64)), int(v[i, 2].view(np.int64))) for i in range(len(v))}
    assert got == {0: (1, big), 1: (1, -big - 1), 2: (4, big), 3: (1, 5)}
    x[6], g[6] = 1, 2                    # ... and one more is one too many
    rc, v, n = oracle.gpupreagg(spec, kds.build_kds("row", [kds.Column("int4", g), kds.Column("int8", x)]), 3)
    assert rc == 2 and len(v) == 0
    g5 = np.zeros(5, dtype=np.int32)
    x5 = np.full(5, 1 << 61, dtype=np.int64)
    rc, v, n = oracle.gpupreagg(spec, kds.build_kds("column", [kds.Column("int4", g5), kds.Column("int8", x5)]), 3)
    assert rc == 2


def test_group_identity_is_the_keys_canonical_image():
    """float keys: -0 = +0, every NaN is one group; numeric keys: 1.50 = 1.5
    (what the types' comparators call equal, gpupreagg_keycomp
    opencl_gpupreagg.h:236) -- the hashed GROUP BY on the device uses the same images"""
    f = np.array([0.0, -0.0, np.nan, np.float64(np.nan) * -1, 1.5, 1.5, 0.0], dtype=np.float64)
    f4 = np.array([0.0, -0.0, 2.5, 2.5, np.nan, np.nan, -0.0], dtype=np.float32)
    norm = kds.numeric_from_scaled(np.array([15, 15, 20, 20, 15, 0, 0]), 1).values
    mant = norm & np.uint64((1 << 57) - 1)
    expo = norm.view(np.int64) >> 58
    loose = np.where(mant != 0, (((expo - 1) & 0x3f).astype(np.uint64) << np.uint64(58)) | (mant * np.uint64(10)), norm)
    img = np.where(np.arange(7) % 2 == 0, norm, loose)
    buf = kds.build_kds("column", [kds.Column("float8", f), kds.Column("float4", f4), kds.Column("numeric", img)])
    rc, v, n = oracle.gpupreagg("(gpupreagg (key (var 1 float8)) (nrows))", buf, 2)
    assert rc == 0
    got = {int(a): int(b) for a, b in zip(v[:, 0].view(np.int64), v[:, 1])}
    nan_img = np.array([0x7ff8000000000000], dtype=np.uint64).view(np.int64)[0]
    assert got == {0: 3, int(nan_img): 2, int(np.array([1.5]).view(np.int64)[0]): 2}
    rc, v, n = oracle.gpupreagg("(gpupreagg (key (var 2 float4)) (nrows))", buf, 2)
    assert rc == 0                     # float4 keys leave as float8 images, like float partials
    got = {int(a): int(b) for a, b in zip(v[:, 0].view(np.int64), v[:, 1])}
    assert got == {0: 3, int(np.array([2.5]).view(np.int64)[0]): 2, int(nan_img): 2}
    rc, v, n = oracle.gpupreagg("(gpupreagg (key (var 3 numeric)) (nrows))", buf, 2)
    assert rc == 0
    got = {int(a): int(b) for a, b in zip(v[:, 0], v[:, 1])}
    canon = kds.numeric_from_scaled(np.array([15, 20, 0]), 1).values
    assert got == {int(canon[0]): 3, int(canon[1]): 2, int(canon[2]): 2}
