"""
Aggregate rewriting and partial -> final merge, host side.

rewrite()  mirrors the catalog + gpupreagg_rewrite_expr of the reference
           (gpupreagg.c:134-333, 729-1166): an SQL aggregate over a column
           becomes one or more partial-function targets for the device plus
           the name of the final function that consumes them.
finalize() mirrors the pgstrom.* final aggregates the reference installs for
           PostgreSQL's Agg node (pg_strom--1.0.sql:247-401; accumulators
           gpupreagg.c:4430-4773) and PostgreSQL's own final functions they
           delegate to (int8_avg, float8_avg, float8_var_samp, ...).
Partial rows may come from any number of chunks / work-groups / GPUs: every
merge below is associative (nrows, psum -> +; pmin/pmax -> min/max).
"""
import math
from decimal import Decimal, getcontext

import numpy as np

getcontext().prec = 60

INT_TYPES = ("int2", "int4", "int8")
FLOAT_TYPES = ("float4", "float8")


def rewrite(func, coltype, var, scale=None):
    """(targets, final) for aggregate `func` over expression text `var` of
    SQL type `coltype`; None when the reference's catalog has no entry
    (the aggregate then stays on the CPU).  numeric needs `scale`, the
    number of fractional digits partial sums are kept with."""
    notnull = "(isnotnull %s)" % var
    if coltype == "numeric" and func != "count" and scale is None:
        # the reference's own form (gpupreagg.c:169-313, NUMERICOID entries): partials stay
        # 64-bit numerics, added / compared on the device exact-or-CpuReCheck
        if func == "avg":
            return ["(nrows %s)" % notnull, "(psum %s)" % var], "avg_numeric"
        if func == "sum":
            return ["(psum %s)" % var], "sum_numeric"
        if func in ("min", "max"):
            return ["(p%s %s)" % (func, var)], func + "_numeric"
        if func in ("stddev", "stddev_samp", "stddev_pop", "variance", "var_samp", "var_pop"):
            return ["(nrows %s)" % notnull, "(psum %s)" % var, "(psum_x2 %s)" % var], func + "_numeric"
        return None
    if coltype == "numeric" and func != "count":
        if func == "avg":
            return ["(nrows %s)" % notnull, "(psum %s %d)" % (var, scale)], "avg_numeric"
        if func == "sum":
            return ["(psum %s %d)" % (var, scale)], "sum_numeric"
        if func in ("min", "max"):
            return ["(p%s %s %d)" % (func, var, scale)], func + "_numeric"
        if func in ("stddev", "stddev_samp", "stddev_pop", "variance", "var_samp", "var_pop"):
            return ["(nrows %s)" % notnull, "(psum %s %d)" % (var, scale),
                    "(psum (numeric_mul %s %s) %d)" % (var, var, 2 * scale)], func + "_numeric"
        return None
    if func == "count":
        return (["(nrows %s)" % notnull] if var else ["(nrows)"]), "count"
    if coltype in INT_TYPES:
        as8 = var if coltype == "int8" else "(int8 %s)" % var
        if func == "avg":
            return ["(nrows %s)" % notnull, "(psum %s)" % as8], "avg_int"
        if func == "sum" and coltype != "int8":
            return ["(psum %s)" % as8], "sum_int8"
        if func in ("min", "max"):
            return ["(p%s %s)" % (func, var)], func
        return None
    if coltype in FLOAT_TYPES:
        as8 = var if coltype == "float8" else "(float8 %s)" % var
        if func == "avg":
            return ["(nrows %s)" % notnull, "(psum %s)" % as8], "avg_float"
        if func == "sum":
            return ["(psum %s)" % var], "sum_float4" if coltype == "float4" else "sum_float8"
        if func in ("min", "max"):
            return ["(p%s %s)" % (func, var)], func
        if func in ("stddev", "stddev_samp", "stddev_pop", "variance", "var_samp", "var_pop"):
            return ["(nrows %s)" % notnull, "(psum %s)" % as8, "(psum_x2 %s)" % as8], func
        return None
    return None


def rewrite2(func, xtype, xvar, ytype, yvar):
    """two-argument aggregates of the reference's catalog (gpupreagg.c:303-332): corr,
    covar_pop, covar_samp over float8 -- other argument types are cast first, as PostgreSQL's
    parser does.  Returns (targets, final) or None."""
    if func not in ("corr", "covar_pop", "covar_samp"):
        return None

    def as8(t, v):
        if t == "float8":
            return v
        if t in INT_TYPES + ("float4", "numeric"):
            return "(float8 %s)" % v
        return None
    x, y = as8(xtype, xvar), as8(ytype, yvar)
    if x is None or y is None:
        return None
    filt = "(and (isnotnull %s) (isnotnull %s))" % (xvar, yvar)
    targets = ["(nrows (isnotnull %s) (isnotnull %s))" % (xvar, yvar)] + \
        ["(pcov_%s %s %s %s)" % (k, filt, x, y) for k in ("x", "x2", "y", "y2", "xy")]
    return targets, func


def _finalize_covariance(final, cols):
    """float8_corr / float8_covar_pop / float8_covar_samp (PostgreSQL float.c) over the merged
    (N, Sx, Sxx, Sy, Syy, Sxy) -- pgstrom.covariance_float8_accum adds the partial rows up"""
    def total(c):
        v, n = c
        v = v[~n]
        return float(np.sum(v.astype(np.float64))) if len(v) else 0.0
    n = int(np.sum(cols[0][0].astype(object))) if len(cols[0][0]) else 0
    if n < 1:
        return None
    sx, sxx, sy, syy, sxy = (total(c) for c in cols[1:6])
    num_xy = n * sxy - sx * sy
    if final == "covar_pop":
        return num_xy / (n * float(n))
    if final == "covar_samp":
        return None if n < 2 else num_xy / (n * (n - 1.0))
    num_x = n * sxx - sx * sx
    num_y = n * syy - sy * sy
    if num_x <= 0.0 or num_y <= 0.0:
        return None
    return num_xy / math.sqrt(num_x * num_y)


def _select_div_scale(num, den):
    """numeric.c select_div_scale(): result scale of num/den so that at
    least NUMERIC_MIN_SIG_DIGITS (16) significant digits survive; base
    10000 digit weights as PostgreSQL stores them"""
    def weight_and_first(d):
        d = abs(d)
        if d == 0:
            return 0, 0
        digits = d.adjusted()            # decimal exponent of the leading digit
        w = math.floor(digits / 4)
        first = int(d.scaleb(-4 * w))    # leading base-10000 digit
        return w, first
    w1, f1 = weight_and_first(num)
    w2, f2 = weight_and_first(den)
    qweight = w1 - w2
    if f1 <= f2:
        qweight -= 1
    rscale = 16 - qweight * 4
    dscale1 = max(0, -num.as_tuple().exponent)
    dscale2 = max(0, -den.as_tuple().exponent)
    rscale = max(rscale, dscale1, dscale2, 0)
    return min(rscale, 1000)


def numeric_div(num, den):
    num, den = Decimal(num), Decimal(den)
    rscale = _select_div_scale(num, den)
    q = (num / den).quantize(Decimal(1).scaleb(-rscale))   # ROUND_HALF_EVEN ~ PG round
    # PostgreSQL rounds half away from zero
    from decimal import ROUND_HALF_UP
    q = (num / den).quantize(Decimal(1).scaleb(-rscale), rounding=ROUND_HALF_UP)
    return q


def finalize(final, cols):
    """cols: list of (values ndarray, isnull ndarray), one per partial target
    of this aggregate, each holding the partial rows of ONE group.  Returns
    a python value (int / float / Decimal) or None for SQL NULL."""
    def sum_nonnull(c, as_float=False):
        v, n = c
        v = v[~n]
        if len(v) == 0:
            return None
        if as_float:
            return float(np.sum(v.astype(np.float64)))
        return int(np.sum(v.astype(object)))

    if final == "count":
        return int(np.sum(cols[0][0].astype(object))) if len(cols[0][0]) else 0
    if final in ("corr", "covar_pop", "covar_samp"):
        return _finalize_covariance(final, cols)
    if final.endswith("_numeric_exact"):
        return _finalize_numeric(final[:-14], cols, exact=True)
    if final.endswith("_numeric"):
        return _finalize_numeric(final[:-8], cols)
    if final == "sum_int8":
        return sum_nonnull(cols[0])
    if final in ("sum_float8", "sum_float4"):
        s = sum_nonnull(cols[0], True)
        if s is not None and final == "sum_float4":
            s = float(np.float32(s))
        return s
    if final in ("min", "max"):
        v, n = cols[0]
        v = v[~n]
        if len(v) == 0:
            return None
        if v.dtype.kind == "f":
            # PostgreSQL ordering: NaN is the largest value
            if final == "max":
                return float("nan") if np.any(np.isnan(v)) else float(np.max(v))
            nn = v[~np.isnan(v)]
            return float(np.min(nn)) if len(nn) else float("nan")
        return int(np.min(v) if final == "min" else np.max(v))
    n = int(np.sum(cols[0][0].astype(object))) if len(cols[0][0]) else 0
    if final == "avg_int":
        s = sum_nonnull(cols[1])
        return None if n == 0 or s is None else numeric_div(Decimal(s), Decimal(n))
    if final == "avg_float":
        s = sum_nonnull(cols[1], True)
        return None if n == 0 or s is None else s / n
    # float8_var_samp & friends over (N, sum x, sum x^2)
    sx = sum_nonnull(cols[1], True)
    sxx = sum_nonnull(cols[2], True)
    if n == 0 or sx is None:
        return None
    samp = final in ("stddev", "stddev_samp", "variance", "var_samp")
    if samp and n <= 1:
        return None
    numer = n * sxx - sx * sx
    if numer <= 0.0:
        return 0.0
    var = numer / (n * (n - 1.0)) if samp else numer / (n * float(n))
    return math.sqrt(var) if final.startswith("stddev") else var


def _finalize_numeric(final, cols, exact=False):
    """numeric aggregates: cols hold python Decimals (object arrays).  exact=True: quotients
    and roots at the context's full precision (60 digits) instead of PostgreSQL's result scale,
    for callers that round to a printed value themselves"""
    def nonnull(c):
        v, n = c
        return [x for x, isn in zip(v, n) if not isn]
    if final in ("sum", "min", "max"):
        v = nonnull(cols[0])
        if not v:
            return None
        return sum(v, Decimal(0)) if final == "sum" else (min(v) if final == "min" else max(v))
    n = int(np.sum(cols[0][0].astype(object))) if len(cols[0][0]) else 0
    sx = nonnull(cols[1])
    if n == 0 or not sx:
        return None
    sx = sum(sx, Decimal(0))
    if final == "avg":
        return (sx / Decimal(n)) if exact else numeric_div(sx, Decimal(n))
    sxx = sum(nonnull(cols[2]), Decimal(0))
    samp = final in ("stddev", "stddev_samp", "variance", "var_samp")
    if samp and n <= 1:
        return None
    # numeric_stddev_internal (numeric.c): N*sumX2 - sumX^2 over N*(N-1) or N*N
    numer = n * sxx - sx * sx
    if numer <= 0:
        return Decimal(0)
    denom = Decimal(n) * (n - 1 if samp else n)
    if exact:
        var = numer / denom
        return var.sqrt() if final.startswith("stddev") else var
    var = numeric_div(numer, denom)
    if final.startswith("stddev"):
        rscale = max(-var.as_tuple().exponent, 0)
        return var.sqrt().quantize(Decimal(1).scaleb(-max(rscale, 16)))
    return var
