"""
The CPU oracle against the answers the reference holds for NUMERIC and for overflow handling
(see numeric_golden.py): expected/recheck_agg.out, the nume_x queries of the four aggregate
suites, expected/overflow_agg.out.  This is what pins SURVEY row a21 (and the CpuReCheck rules
of a18 / a20) to the reference's own files rather than to an independent restatement.
"""
from decimal import Decimal

import numpy as np
import pytest

import numeric_golden as ng
import oracle_binding as oracle
from pg_strom_amd import kds


def oracle_chunk(plan, buf, chunk_no):
    return oracle.gpupreagg(plan["spec"], buf, plan["ntargets"])


def test_superset_programs_give_the_same_partial_rows():
    """the HIP tests run one program per (column, shape, family) -- numeric_golden.superset();
    through the oracle that must equal the per-query programs"""
    chunks = ng.table_chunks("gpupreagg_overflow_test", "column", 1)
    for q in ng.load_expected()["overflow_agg"]:
        plan = ng.plan_query(q["sql"])
        if plan is None or plan["col"] not in ("nume_x", "real_x", "integer_x"):
            continue
        spec, nt, picks = ng.superset(plan)
        rc1, v1, n1 = oracle.gpupreagg(plan["spec"], chunks[0][0], plan["ntargets"])
        rc2, v2, n2 = oracle.gpupreagg(spec, chunks[0][0], nt)
        if rc2 == 0:
            assert rc1 == 0 and np.array_equal(n1, n2[:, picks]) and np.array_equal(v1[~n1], v2[:, picks][~n1])


def test_recheck_agg_literals_conversion_and_sum():
    """recheck_agg.out: 0, 1E+48, 1E-32 stay on the device; 1E-33, 1E+49, 1E+1000, 1E-1000 send
    the chunk back -- decided by the varlena -> 64-bit conversion (opencl_numeric.h:166-307)"""
    qs = ng.load_expected()["recheck_agg"]
    assert len(qs) == 7
    seen = {True: 0, False: 0}
    for q in qs:
        lit = q["sql"][len("select sum("):-2]
        rechecked = any("re-checked by CPU" in n for n in q.get("notices", []))
        d = Decimal(lit)
        raw = ng.pg_numeric_varlena(d)
        image = oracle.numeric_from_varlena(raw)
        assert (image is None) == rechecked, (lit, image)
        if image is not None:
            assert kds.numeric_decode(image) == d
        # the aggregate itself over a one-row heap chunk that carries the datum as PostgreSQL stores it
        for fmt in ("row", "row_flat"):
            buf = kds.build_kds(fmt, [kds.Column("numeric_raw", [raw])])
            rc, v, n = oracle.gpupreagg("(gpupreagg (psum (var 1 numeric)))", buf, 1)
            assert rc == (2 if rechecked else 0), (lit, fmt, rc)
            got = d if rc == 2 else kds.numeric_decode(v[0, 0])      # CPU fallback: the value itself
            assert format(got, "f") == q["rows"][0][0] or got == Decimal(q["rows"][0][0])
        seen[rechecked] += 1
    assert seen == {True: 4, False: 3}


@pytest.mark.parametrize("fmt,nchunks", [("column", 3), ("row", 2)])
def test_nume_x_queries_of_the_aggregate_suites(fmt, nchunks):
    chunks = {"gpupreagg_test": ng.table_chunks("gpupreagg_test", fmt, nchunks),
              "gpupreagg_zero_test": ng.table_chunks("gpupreagg_zero_test", fmt, 1)}
    exp = ng.load_expected()
    stats, held = {}, 0
    for suite in ("nogrp_agg", "group_agg", "where_agg", "zero_agg"):
        for q in exp[suite]:
            if "nume_x" in q["sql"] and "gpupreagg_mix" not in q["sql"]:
                held += ng.run_query(q, chunks, oracle_chunk, stats)
    assert held == 4 * 11                       # avg count max min sum + 6 stddev / variance flavours
    assert stats["rechecked_chunks"] > 0 and stats["device_chunks"] > 0


@pytest.mark.parametrize("fmt,nchunks", [("column", 2), ("row_flat", 3)])
def test_overflow_agg_suite(fmt, nchunks):
    """gpupreagg_overflow_test: sums of 32767 / 2147483647 / 9223372036854775807 / 1e38 / 1e308 and
    21-digit numerics per key -- results and PostgreSQL's errors as overflow_agg.out has them"""
    chunks = {"gpupreagg_overflow_test": ng.table_chunks("gpupreagg_overflow_test", fmt, nchunks)}
    stats, held, errors = {}, 0, 0
    for q in ng.load_expected()["overflow_agg"]:
        ok = ng.run_query(q, chunks, oracle_chunk, stats)
        held += ok
        errors += bool(ok and q.get("error"))
    # 9 columns: 5 catalog aggregates per int2/int4 column, 4 per int8, 11 per float / numeric column
    assert held == 61 and errors == 16, (held, errors)
    assert stats["rechecked_chunks"] > 0 and stats["device_chunks"] > 0


def test_corr_and_covariance_known_answers():
    """corr / covar_pop / covar_samp of the four suites -- every column with itself over
    gpupreagg_test, x / y against z over the gpupreagg_mix view -- through the pcov_* partials
    and the float8_corr / float8_covar_* finals"""
    exp = ng.load_expected()
    held, cache = 0, {}
    for suite in ("nogrp_agg", "group_agg", "where_agg", "zero_agg"):
        for q in exp[suite]:
            held += ng.run_covar_query(q, oracle_chunk, "column", 2, chunk_cache=cache)
    assert held == 135, held                 # 9 columns x 3 functions x 3 suites + 54 over the mix view
