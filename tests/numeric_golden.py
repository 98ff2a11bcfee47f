"""
Known answers the reference itself holds for NUMERIC and for overflow handling
(stock PostgreSQL output, parsed into tests/golden/expected_agg.json by
make_gpupreagg_fixture.py; the tables are regenerated there and verified against
the same files before they are written):

  expected/recheck_agg.out     select sum(LITERAL): which numerics the 64-bit device form
                               holds (1E+48, 1E-32) and which send the chunk back to the
                               CPU (1E-33, 1E+49, 1E+-1000) -- opencl_numeric.h:166-307
  expected/{nogrp,group,where,zero}_agg.out   the nume_x queries (56 of them)
  expected/overflow_agg.out    gpupreagg_overflow_test: columns pinned at the edges of their
                               types; partial sums overflow on the device, the chunk is
                               re-done on the CPU, the final cast succeeds or raises
                               PostgreSQL's error

A caller supplies run_chunk(plan, chunk image, chunk number) -> (status, values uint64
[n, plan["ntargets"]], isnull) -- the CPU oracle or the HIP path.  status 2 (CpuReCheck) is
answered the way gpupreagg_next_tuple_fallback does (gpupreagg.c:2507-2607): the chunk's rows
are aggregated on the CPU, exactly, and enter the final merge as partial rows.
"""
import json
import math
import os
import re
from decimal import Decimal, ROUND_HALF_UP

import numpy as np

from pg_strom_amd import aggregate, kds

HERE = os.path.dirname(os.path.abspath(__file__))

# column -> (attno, sql type) in the chunks built below
COLUMNS = {
    "id": (1, "int4"), "key": (2, "int4"),
    "smlint_x": (3, "int2"), "integer_x": (4, "int4"), "bigint_x": (5, "int8"),
    "real_x": (6, "float4"), "float_x": (7, "float8"), "nume_x": (8, "numeric"),
    "smlsrl_x": (9, "int2"), "serial_x": (10, "int4"), "bigsrl_x": (11, "int8"),
}
ORDER = ["id", "key", "smlint_x", "integer_x", "bigint_x", "real_x", "float_x", "nume_x",
         "smlsrl_x", "serial_x", "bigsrl_x"]
TABLES = {"gpupreagg_test": "gpupreagg_test.npz", "gpupreagg_zero_test": "gpupreagg_test.npz",
          "gpupreagg_overflow_test": "gpupreagg_overflow_test.npz"}
_cache = {}


def load_expected():
    return json.load(open(os.path.join(HERE, "golden", "expected_agg.json")))


def load_table(name):
    if name not in _cache:
        fx = dict(np.load(os.path.join(HERE, "golden", TABLES[name])))
        if name == "gpupreagg_zero_test":
            fx = {k: v[:0] for k, v in fx.items()}
        fx["nume_dec"] = np.array([None if n else Decimal(str(s)) for s, n in zip(fx["nume_x"], fx["nume_x_isnull"])],
                                  dtype=object)
        _cache[name] = fx
    return _cache[name]


def pg_numeric_varlena(d):
    """Decimal -> the bytes PostgreSQL 9.4 stores for a numeric (utils/adt/numeric.c: base-10000
    digits; short header when weight and dscale fit, long header otherwise; 1-byte varlena header
    up to 126 bytes, 4-byte header beyond)"""
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
        intpart = "0" * ((-len(intpart)) % 4) + intpart
        frac = frac + "0" * ((-len(frac)) % 4)
        groups = [int(intpart[i:i + 4]) for i in range(0, len(intpart), 4)]
        weight = len(groups) - 1
        groups += [int(frac[i:i + 4]) for i in range(0, len(frac), 4)]
        while groups and groups[0] == 0:
            groups.pop(0)
            weight -= 1
        while groups and groups[-1] == 0:
            groups.pop()
    if -64 <= weight <= 63 and dscale <= 63:
        body = struct.pack("<H", 0x8000 | (0x2000 if sign else 0) | (dscale << 7) |
                           (0x0040 if weight < 0 else 0) | (weight & 0x3F))
    else:
        body = struct.pack("<Hh", (0x4000 if sign else 0) | dscale, weight)
    body += b"".join(struct.pack("<H", g) for g in groups)
    if 1 + len(body) <= 126:
        return bytes([((1 + len(body)) << 1) | 1]) + body
    return struct.pack("<I", (4 + len(body)) << 2) + body


def table_chunks(name, fmt, nchunks):
    """[(kds image, row slice)] -- nume_x as 8-byte numerics in COLUMN / TUPSLOT chunks, as
    PostgreSQL's own varlena datums in heap-tuple chunks (decoded by the kernels per row)"""
    fx = load_table(name)
    n = len(fx["id"])
    bounds = np.linspace(0, n, nchunks + 1).astype(int)
    out = []
    for i in range(nchunks):
        rows = slice(bounds[i], bounds[i + 1])
        cols = []
        for cname in ORDER:
            typ = COLUMNS[cname][1]
            isn = fx[cname + "_isnull"][rows]
            if cname != "nume_x":
                cols.append(kds.Column(typ, fx[cname][rows], isn))
            elif fmt in ("row", "row_flat"):
                zero = pg_numeric_varlena(Decimal(0))
                datums = [pg_numeric_varlena(d) if d is not None else zero for d in fx["nume_dec"][rows]]
                cols.append(kds.Column("numeric_raw", datums, isn))
            else:
                imgs = np.array([0 if d is None else kds.numeric_encode(d) for d in fx["nume_dec"][rows]],
                                dtype=np.uint64)
                cols.append(kds.Column("numeric", imgs, isn))
        out.append((kds.build_kds(fmt, cols), rows))
    return out


QUERY_RE = re.compile(
    r"select\s+(key\s*,\s*)?(\w+)\((\w+)\)(::\w+)?\s+from\s+(\w+)\s*(where key=1\s*)?(group by key\s*(order by key)?)?\s*;?$",
    re.I)


def plan_query(sql):
    m = QUERY_RE.match(sql.strip())
    if not m:
        return None
    has_key, func, col, cast, table, where, groupby = m.group(1), m.group(2).lower(), m.group(3), m.group(4), \
        m.group(5), m.group(6), m.group(7)
    if table not in TABLES or col not in COLUMNS:
        return None
    attno, typ = COLUMNS[col]
    rw = aggregate.rewrite(func, typ, "(var %d %s)" % (attno, typ))
    if rw is None:
        return None
    targets, final = rw
    parts = []
    if where:
        parts.append("(qual (int4eq (var 2 int4) (const int4 1)))")
    if groupby:
        parts.append("(key (var 2 int4))")
    parts += targets
    return {"spec": "(gpupreagg " + " ".join(parts) + ")", "targets": targets, "final": final,
            "grouped": bool(groupby), "has_key": bool(has_key), "where": bool(where), "table": table,
            "func": func, "col": col, "type": typ, "cast": cast[2:].lower() if cast else None,
            "ntargets": len(targets) + (1 if groupby else 0)}


class PgError(Exception):
    pass


def _fsum(xs):
    """float8 sum; +-inf when it leaves the type (PostgreSQL then raises, see finalize_group)"""
    try:
        return math.fsum(xs)
    except OverflowError:
        with np.errstate(over="ignore"):
            return float(np.sum(np.array(xs, dtype=np.float64)))


def cpu_partial_rows(plan, rows):
    """one exact partial row per group of the chunk's rows (the CPU's answer for a chunk the
    device returned): {key: [python value or None per target]}"""
    fx = load_table(plan["table"])
    col = plan["col"]
    vals = fx["nume_dec"][rows] if col == "nume_x" else fx[col][rows]
    isn = fx[col + "_isnull"][rows].astype(bool)
    key = fx["key"][rows]
    keyn = fx["key_isnull"][rows].astype(bool)
    keep = np.ones(len(key), dtype=bool)
    if plan["where"]:
        keep = (key == 1) & ~keyn
    groups = {}
    if plan["grouped"]:
        for i in np.flatnonzero(keep):
            groups.setdefault(None if keyn[i] else int(key[i]), []).append(i)
    elif keep.any():
        groups[0] = list(np.flatnonzero(keep))
    out = {}
    typ = plan["type"]
    for k, idx in groups.items():
        idx = np.asarray(idx)
        nn = idx[~isn[idx]]
        if typ == "numeric":
            xs = [vals[i] for i in nn]
        elif typ in ("float4", "float8"):
            xs = [float(vals[i]) for i in nn]
        else:
            xs = [int(vals[i]) for i in nn]
        row = []
        for t in plan["targets"]:
            if t.startswith("(nrows"):
                row.append(len(nn) if "isnotnull" in t else len(idx))
            elif not xs:
                row.append(None)
            elif t.startswith("(psum_x2"):
                row.append(_fsum([x * x for x in xs]) if isinstance(xs[0], float) else sum(x * x for x in xs))
            elif t.startswith("(psum"):
                row.append(_fsum(xs) if isinstance(xs[0], float) else sum(xs))
            elif t.startswith("(pmin"):
                row.append(min(xs))
            else:
                row.append(max(xs))
        out[k] = row
    return out


def decode_rows(plan, values, isnull):
    """device / oracle partial rows -> {key: [[python value per target], ...]}"""
    out = {}
    first = 1 if plan["grouped"] else 0
    typ = plan["type"]
    for r in range(len(values)):
        k = 0
        if plan["grouped"]:
            k = None if isnull[r, 0] else int(np.int32(int(values[r, 0]) & 0xffffffff))
        row = []
        for t, tgt in enumerate(plan["targets"]):
            raw, n = values[r, first + t], isnull[r, first + t]
            if tgt.startswith("(nrows"):
                row.append(int(raw))
            elif n:
                row.append(None)
            elif typ == "numeric":
                row.append(kds.numeric_decode(raw))
            elif typ in ("float4", "float8"):
                # float partials arrive as float8 images (the oracle's convention; the HIP
                # runner widens float4 datums with partial_rows_as_raw8)
                row.append(float(np.array([raw], dtype=np.uint64).view(np.float64)[0]))
            else:
                width = {"int2": 16, "int4": 32, "int8": 64}[typ]
                if tgt.startswith("(psum"):
                    width = 64
                v = int(raw) & ((1 << width) - 1)
                row.append(v - (1 << width) if v >> (width - 1) else v)
        out.setdefault(k, []).append(row)
    return out


def finalize_group(plan, rows):
    """partial rows of one group -> python value, PostgreSQL's semantics (errors as PgError)"""
    typ, final = plan["type"], plan["final"]
    cols = []
    for t, tgt in enumerate(plan["targets"]):
        vals = [r[t] for r in rows]
        isn = np.array([v is None for v in vals], dtype=bool)
        if tgt.startswith("(nrows"):
            cols.append((np.array(vals, dtype=object), isn))
        elif typ in ("float4", "float8"):
            arr = np.array([0.0 if v is None else v for v in vals], dtype=np.float64)
            if tgt.startswith("(psum"):
                with np.errstate(over="ignore", invalid="ignore"):
                    total = float(np.sum(arr[~isn])) if (~isn).any() else 0.0
                if math.isinf(total) or math.isnan(total):
                    raise PgError("value out of range: overflow")   # float8pl / float8_accum
            cols.append((arr, isn))
        else:
            cols.append((np.array([0 if v is None else v for v in vals], dtype=object), isn))
    if typ == "numeric" and final != "count":
        return aggregate.finalize(final + "_exact", cols)
    with np.errstate(over="ignore", invalid="ignore"):
        res = aggregate.finalize(final, cols)
    if isinstance(res, float) and math.isinf(res):
        raise PgError("value out of range: overflow")         # float4pl / float8 accumulators
    return res


INT_RANGES = {"smallint": (-2**15, 2**15 - 1), "integer": (-2**31, 2**31 - 1), "bigint": (-2**63, 2**63 - 1)}


def apply_cast(value, cast):
    if value is None or cast is None or cast in ("numeric", "float"):
        return value
    if cast in INT_RANGES:
        lo, hi = INT_RANGES[cast]
        if isinstance(value, float):
            if math.isnan(value) or math.isinf(value):
                raise PgError("%s out of range" % cast)
            v = int(np.rint(value))                       # dtoi: rint()
        else:
            v = int(Decimal(value).quantize(Decimal(1), rounding=ROUND_HALF_UP))
        if v < lo or v > hi:
            raise PgError("%s out of range" % cast)
        return v
    if cast == "real":
        v = float(value)
        with np.errstate(over="ignore"):
            f = np.float32(v)
        if np.isinf(f) and not math.isinf(v):
            raise PgError("value out of range: overflow")
        return float(f)
    raise AssertionError(cast)


def float_spread_tolerance(plan, rows):
    """stddev / variance of floats are N*sum(x^2) - sum(x)^2 in float8: the cancellation error is
    relative to the MAGNITUDE of the data (1e38 for the overflow table), not to the result, and
    depends on the summation order, which on the device is not row order.  Absolute tolerance
    for the value PostgreSQL printed; 0 for everything else."""
    if plan["type"] not in ("float4", "float8") or not plan["final"].startswith(("stddev", "var")):
        return 0.0
    n = sum(r[0] for r in rows)
    sx = sum(r[1] for r in rows if r[1] is not None)
    if n == 0:
        return 0.0
    mag = abs(sx / n)
    return 1e-6 * mag if plan["final"].startswith("stddev") else 1e-12 * mag * mag


def matches(got, want_text, plan, abs_tol=0.0):
    if want_text == "":
        return got is None
    if got is None:
        return False
    if isinstance(got, float):
        want = float(want_text)
        if abs(got - want) <= abs_tol:
            return True
        tol = 6e-3 if (plan["type"] == "float4" or plan["cast"] == "real") else 2e-11
        if plan["final"] not in ("min", "max", "sum_float8", "sum_float4", "avg_float", "count"):
            tol = max(tol, 1e-9)
        return abs(got - want) <= tol * max(abs(want), 1e-300)
    ndec = len(want_text.split(".")[1]) if "." in want_text else 0
    g = Decimal(got).quantize(Decimal(1).scaleb(-ndec), rounding=ROUND_HALF_UP)
    w = Decimal(want_text)
    if g == w:
        return True
    # roots and quotients: PostgreSQL's last printed digit may differ by one unit
    root = plan["final"].startswith(("stddev", "var"))
    return root and abs(g - w) <= Decimal(1).scaleb(-ndec)


def run_query(q, chunks, run_chunk, stats):
    """returns True when the query was in scope and held"""
    plan = plan_query(q["sql"])
    if plan is None:
        return False
    groups = {}
    for ci, (buf, rows) in enumerate(chunks[plan["table"]]):
        status, values, isnull = run_chunk(plan, buf, ci)
        assert status in (0, 2), (q["sql"], status)
        if status == 2:
            stats["rechecked_chunks"] = stats.get("rechecked_chunks", 0) + 1
            for k, row in cpu_partial_rows(plan, rows).items():
                groups.setdefault(k, []).append(row)
        else:
            stats["device_chunks"] = stats.get("device_chunks", 0) + 1
            for k, rws in decode_rows(plan, values, isnull).items():
                groups.setdefault(k, []).extend(rws)
    tols = {k: float_spread_tolerance(plan, rws) for k, rws in groups.items()}
    try:
        results = {k: apply_cast(finalize_group(plan, rws), plan["cast"]) for k, rws in groups.items()}
        error = None
    except PgError as e:
        results, error = None, str(e)
    if q.get("error"):
        assert error is not None, (q["sql"], "expected ERROR: " + q["error"])
        assert error.split(":")[0] in q["error"], (q["sql"], error, q["error"])
        return True
    assert error is None, (q["sql"], error)
    if plan["grouped"]:
        want_rows = q["rows"] if plan["has_key"] else [["1", r[0]] for r in q["rows"]]
        assert len(want_rows) == len(results), (q["sql"], len(want_rows), len(results))
        for wr in want_rows:
            k = None if wr[0] == "" else int(wr[0])
            assert k in results, (q["sql"], k)
            assert matches(results[k], wr[1], plan, tols[k]), (q["sql"], k, results[k], wr[1])
    else:
        got = results.get(0) if results else _empty_result(plan)
        assert len(q["rows"]) == 1
        assert matches(got, q["rows"][0][0], plan, tols.get(0, 0.0)), (q["sql"], got, q["rows"][0][0])
    return True


def superset(plan):
    """One device program per (column, query shape, target family) instead of one per query:
    family A = what cannot overflow (counts, pmin, pmax), family B = the sums -- so that a sum
    that sends the chunk back to the CPU does not take min / max / count with it.  Returns
    (spec, ntargets, picks): plan target t is column picks[t] of the superset's partial rows."""
    attno, typ = COLUMNS[plan["col"]]
    var = "(var %d %s)" % (attno, typ)
    fam_a = all(t.startswith(("(nrows", "(pmin", "(pmax")) for t in plan["targets"])
    wanted = []
    funcs = ("count", "min", "max") if fam_a else \
        ("avg", "sum", "stddev", "stddev_pop", "stddev_samp", "variance", "var_pop", "var_samp")
    for func in funcs:
        rw = aggregate.rewrite(func, typ, var)
        for t in (rw[0] if rw else []):
            if t not in wanted:
                wanted.append(t)
    assert all(t in wanted for t in plan["targets"]), (plan["targets"], wanted)
    head = []
    if plan["where"]:
        head.append("(qual (int4eq (var 2 int4) (const int4 1)))")
    if plan["grouped"]:
        head.append("(key (var 2 int4))")
    nkeys = 1 if plan["grouped"] else 0
    picks = list(range(nkeys)) + [nkeys + wanted.index(t) for t in plan["targets"]]
    return "(gpupreagg " + " ".join(head + wanted) + ")", nkeys + len(wanted), picks


def _empty_result(plan):
    """no row at all reached the aggregate: count() is 0, everything else NULL"""
    return 0 if plan["final"] == "count" else None


# ---------------------------------------------------------------------------------------------
# corr / covar_pop / covar_samp (gpupreagg.c:303-332; finals float8_corr & friends through
# pgstrom.covariance_float8_accum, pg_strom--1.0.sql:368-401).  gpupreagg_mix is the
# materialised view of agg_init.sql:205-250: row i of the positive block (x) joined with row i
# of the negative (y) and of the mixed block (z) on id, key = x.key.
# ---------------------------------------------------------------------------------------------
COVAR_RE = re.compile(
    r"select\s+(key\s*,\s*)?(corr|covar_pop|covar_samp)\((\w+)\s*,\s*(\w+)\)\s+from\s+(\w+)\s*(where key=1\s*)?"
    r"(group by key\s*(order by key)?)?\s*;?$", re.I)


def covar_columns(table, a, b):
    """(key, key isnull, x values, x isnull, type, y values, y isnull, type) as numpy arrays"""
    fx = load_table("gpupreagg_test")
    if table == "gpupreagg_mix":
        block = {"x": slice(0, 10000), "y": slice(10000, 20000), "z": slice(20000, 30000)}

        def col(name):
            base, which = name[:-1] + "x", name[-1]
            rows = block[which]
            vals = fx["nume_dec"][rows] if base == "nume_x" else fx[base][rows]
            return vals, fx[base + "_isnull"][rows].astype(bool), COLUMNS[base][1]
        key, keyn = fx["key"][block["x"]], fx["key_isnull"][block["x"]].astype(bool)
    else:
        if table == "gpupreagg_zero_test":
            fx = load_table("gpupreagg_zero_test")

        def col(name):
            vals = fx["nume_dec"] if name == "nume_x" else fx[name]
            return vals, fx[name + "_isnull"].astype(bool), COLUMNS[name][1]
        key, keyn = fx["key"], fx["key_isnull"].astype(bool)
    xv, xn, xt = col(a)
    yv, yn, yt = col(b)
    return key, keyn, xv, xn, xt, yv, yn, yt


def _kds_column(vals, isn, typ):
    if typ == "numeric":
        imgs = np.array([0 if (d is None) else kds.numeric_encode(d) for d in vals], dtype=np.uint64)
        return kds.Column("numeric", imgs, isn)
    return kds.Column(typ, vals, isn)


def plan_covar(sql):
    m = COVAR_RE.match(sql.strip())
    if not m:
        return None
    has_key, func, a, b, table, where, groupby = (m.group(i) for i in range(1, 8))
    if table not in ("gpupreagg_test", "gpupreagg_zero_test", "gpupreagg_mix"):
        return None
    base_a = a if table != "gpupreagg_mix" else a[:-1] + "x"
    if base_a not in COLUMNS:
        return None
    key, keyn, xv, xn, xt, yv, yn, yt = covar_columns(table, a, b)
    rw = aggregate.rewrite2(func.lower(), xt, "(var 2 %s)" % xt, yt, "(var 3 %s)" % yt)
    if rw is None:
        return None
    targets, final = rw
    parts = []
    if where:
        parts.append("(qual (int4eq (var 1 int4) (const int4 1)))")
    if groupby:
        parts.append("(key (var 1 int4))")
    return {"spec": "(gpupreagg " + " ".join(parts + targets) + ")", "targets": targets, "final": final,
            "grouped": bool(groupby), "has_key": bool(has_key), "where": bool(where), "table": table,
            "cols": (key, keyn, xv, xn, xt, yv, yn, yt), "pair": (table, a, b),
            "ntargets": len(targets) + (1 if groupby else 0)}


def covar_chunks(plan, fmt, nchunks):
    key, keyn, xv, xn, xt, yv, yn, yt = plan["cols"]
    n = len(key)
    bounds = np.linspace(0, n, nchunks + 1).astype(int)
    out = []
    for i in range(nchunks):
        r = slice(bounds[i], bounds[i + 1])
        out.append((kds.build_kds(fmt, [kds.Column("int4", key[r], keyn[r]), _kds_column(xv[r], xn[r], xt),
                                        _kds_column(yv[r], yn[r], yt)]), r))
    return out


def covar_cpu_partials(plan, rows):
    key, keyn, xv, xn, xt, yv, yn, yt = plan["cols"]
    key, keyn = key[rows], keyn[rows]
    x = np.array([0.0 if (n or v is None) else float(v) for v, n in zip(xv[rows], xn[rows])])
    y = np.array([0.0 if (n or v is None) else float(v) for v, n in zip(yv[rows], yn[rows])])
    ok = ~xn[rows] & ~yn[rows]
    keep = np.ones(len(key), dtype=bool)
    if plan["where"]:
        keep = (key == 1) & ~keyn
    out = {}
    if plan["grouped"]:
        gids = [None if keyn[i] else int(key[i]) for i in np.flatnonzero(keep)]
        for g in set(gids):
            idx = np.array([i for i, gg in zip(np.flatnonzero(keep), gids) if gg == g])
            out[g] = idx
    elif keep.any():
        out[0] = np.flatnonzero(keep)
    res = {}
    for g, idx in out.items():
        sel = idx[ok[idx]]
        xs, ys = x[sel], y[sel]
        row = [len(sel)] + ([None] * 5 if len(sel) == 0 else
                            [_fsum(list(xs)), _fsum(list(xs * xs)), _fsum(list(ys)), _fsum(list(ys * ys)),
                             _fsum(list(xs * ys))])
        res[g] = row
    return res


def run_covar_query(q, run_chunk, fmt="column", nchunks=2, stats=None, chunk_cache=None):
    plan = plan_covar(q["sql"])
    if plan is None:
        return False
    ckey = (plan["pair"], fmt, nchunks)
    if chunk_cache is not None and ckey in chunk_cache:
        chunks = chunk_cache[ckey]
    else:
        chunks = covar_chunks(plan, fmt, nchunks)
        if chunk_cache is not None:
            chunk_cache[ckey] = chunks
    groups = {}
    first = 1 if plan["grouped"] else 0
    for ci, (buf, rows) in enumerate(chunks):
        status, values, isnull = run_chunk(plan, buf, ci)
        assert status in (0, 2), (q["sql"], status)
        if status == 2:
            if stats is not None:
                stats["rechecked_chunks"] = stats.get("rechecked_chunks", 0) + 1
            for k, row in covar_cpu_partials(plan, rows).items():
                groups.setdefault(k, []).append(row)
            continue
        if stats is not None:
            stats["device_chunks"] = stats.get("device_chunks", 0) + 1
        for r in range(len(values)):
            k = 0
            if plan["grouped"]:
                k = None if isnull[r, 0] else int(np.int32(int(values[r, 0]) & 0xffffffff))
            row = [int(values[r, first])]
            for t in range(1, 6):
                row.append(None if isnull[r, first + t]
                           else float(np.array([values[r, first + t]], dtype=np.uint64).view(np.float64)[0]))
            groups.setdefault(k, []).append(row)
    results = {}
    for k, rws in groups.items():
        cols = [(np.array([r[0] for r in rws], dtype=object), np.zeros(len(rws), dtype=bool))]
        for t in range(1, 6):
            vals = [r[t] for r in rws]
            cols.append((np.array([0.0 if v is None else v for v in vals]), np.array([v is None for v in vals])))
        with np.errstate(all="ignore"):
            results[k] = aggregate.finalize(plan["final"], cols)

    def close(got, text):
        if text == "":
            return got is None
        return got is not None and abs(got - float(text)) <= 5e-10 * max(abs(float(text)), 1e-300)
    if plan["grouped"]:
        want_rows = q["rows"] if plan["has_key"] else [["1", r[0]] for r in q["rows"]]
        assert len(want_rows) == len(results), (q["sql"], len(want_rows), len(results))
        for wr in want_rows:
            k = None if wr[0] == "" else int(wr[0])
            assert close(results.get(k), wr[1]), (q["sql"], k, results.get(k), wr[1])
    else:
        assert close(results.get(0), q["rows"][0][0]), (q["sql"], results.get(0), q["rows"][0][0])
    return True
