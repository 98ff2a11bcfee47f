"""
The HIP path against the answers the reference holds for NUMERIC and for overflow handling
(numeric_golden.py; needs an MI355X: -m gpu): expected/recheck_agg.out, the nume_x queries of
the four aggregate suites, expected/overflow_agg.out.  NUMERIC partials are accumulated on the
device in the reference's own 64-bit form (compare-and-swap in LDS, checked slab merge),
exact or CpuReCheck; a rechecked chunk is answered on the CPU as the reference does.
"""
from decimal import Decimal

import numpy as np
import pytest

import numeric_golden as ng
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpupreagg import GpuPreAgg
from test_gpupreagg_gpu import partial_rows_as_raw8

pytestmark = pytest.mark.gpu


def hip_runner():
    cache = {}

    def run_chunk(plan, buf, chunk_no):
        spec, nt, picks = ng.superset(plan)
        key = (spec, plan["table"], chunk_no, buf.ctypes.data)
        if key not in cache:
            agg = GpuPreAgg(spec).begin([(1, 30)] if plan["grouped"] else [])
            try:
                status, _ = agg.fold(buf)
                v, n = partial_rows_as_raw8(agg.fetch()) if status == 0 else (None, None)
            finally:
                agg.end()
            cache[key] = (status, v, n)
        status, v, n = cache[key]
        if status != 0:
            return status, None, None
        return status, v[:, picks], n[:, picks]
    return run_chunk


def test_recheck_agg_literals_on_device():
    """recheck_agg.out through GpuPreAgg over a one-row heap chunk that carries the literal as
    PostgreSQL stores it: 0 / 1E+48 / 1E-32 are summed on the device, 1E-33 / 1E+49 / 1E+-1000
    come back CpuReCheck -- and a GpuScan over the seven rows marks exactly those four"""
    from pg_strom_amd.gpuscan import GpuScan
    runtime.init()
    qs = ng.load_expected()["recheck_agg"]
    lits, flags = [], []
    agg = GpuPreAgg("(gpupreagg (psum (var 1 numeric)))").begin([])
    try:
        for q in qs:
            lit = q["sql"][len("select sum("):-2]
            rechecked = any("re-checked by CPU" in n for n in q.get("notices", []))
            d = Decimal(lit)
            lits.append(d)
            flags.append(rechecked)
            for fmt in ("row", "row_flat"):
                agg.reset()
                status, _ = agg.fold(kds.build_kds(fmt, [kds.Column("numeric_raw", [ng.pg_numeric_varlena(d)])]))
                assert status == (2 if rechecked else 0), (lit, fmt, status)
                if status == 0:
                    pr = agg.fetch()
                    assert len(pr) == 1 and not pr.isnull[0, 0]
                    assert kds.numeric_decode(pr.values[0, 0]) == Decimal(q["rows"][0][0])
    finally:
        agg.end()
    buf = kds.build_kds("row", [kds.Column("numeric_raw", [ng.pg_numeric_varlena(d) for d in lits])])
    scan = GpuScan("(numeric_ge (var 1 numeric) (const numeric 0))").begin()
    try:
        res = scan.scan_chunk(buf)
    finally:
        scan.end()
    assert sorted(res.recheck_rows()) == [i for i, f in enumerate(flags) if f]
    assert sorted(res.passed_rows()) == [i for i, f in enumerate(flags) if not f]


@pytest.mark.parametrize("fmt,nchunks", [("column", 3), ("row", 2)])
def test_nume_x_queries_of_the_aggregate_suites_on_device(fmt, nchunks):
    runtime.init()
    chunks = {"gpupreagg_test": ng.table_chunks("gpupreagg_test", fmt, nchunks),
              "gpupreagg_zero_test": ng.table_chunks("gpupreagg_zero_test", fmt, 1)}
    exp = ng.load_expected()
    run, stats, held = hip_runner(), {}, 0
    for suite in ("nogrp_agg", "group_agg", "where_agg", "zero_agg"):
        for q in exp[suite]:
            if "nume_x" in q["sql"] and "gpupreagg_mix" not in q["sql"]:
                held += ng.run_query(q, chunks, run, stats)
    assert held == 4 * 11
    assert stats["rechecked_chunks"] > 0 and stats["device_chunks"] > 0


@pytest.mark.parametrize("fmt,nchunks", [("column", 2), ("row_flat", 3)])
def test_overflow_agg_suite_on_device(fmt, nchunks):
    runtime.init()
    chunks = {"gpupreagg_overflow_test": ng.table_chunks("gpupreagg_overflow_test", fmt, nchunks)}
    run, stats, held, errors = hip_runner(), {}, 0, 0
    for q in ng.load_expected()["overflow_agg"]:
        ok = ng.run_query(q, chunks, run, stats)
        held += ok
        errors += bool(ok and q.get("error"))
    assert held == 61 and errors == 16, (held, errors)
    assert stats["rechecked_chunks"] > 0 and stats["device_chunks"] > 0


def test_corr_and_covariance_known_answers_on_device():
    """corr / covar_pop / covar_samp through the pcov_* partials on the device (int4, float4,
    float8 and numeric columns: one program per argument type and query shape)"""
    runtime.init()
    exp = ng.load_expected()
    sessions = {}

    def run_chunk(plan, buf, chunk_no):
        agg = GpuPreAgg(plan["spec"]).begin([(1, 30)] if plan["grouped"] else [])
        try:
            status, _ = agg.fold(buf)
            v, n = partial_rows_as_raw8(agg.fetch()) if status == 0 else (None, None)
        finally:
            agg.end()
        return status, v, n

    held, cache, stats = 0, {}, {}
    for suite in ("nogrp_agg", "group_agg", "where_agg", "zero_agg"):
        for q in exp[suite]:
            if any(c in q["sql"] for c in ("integer_", "real_", "float_", "nume_")):
                held += ng.run_covar_query(q, run_chunk, "column", 2, stats=stats, chunk_cache=cache)
    assert held == 4 * (3 * 3 + 6), held
    assert stats.get("device_chunks", 0) > 0
