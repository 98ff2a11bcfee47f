"""numeric(p,s) columns stored as int8 at 10^-s ("decimal", STROM_DECIMALOID): the same table
twice -- once as 64-bit numeric images, once as scaled integers -- and the same queries written
with (var N numeric S) and (var N decimal S).  Results must be identical."""
import numpy as np

from pg_strom_amd import kds


def tables(n, seed, nulls=0.03):
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 7, n).astype(np.int32)
    qty = rng.integers(-50, 51, n)                      # scale 0, negative values too
    prc = rng.integers(-5000, 10494951, n)              # scale 2
    dsc = rng.integers(0, 11, n)                        # scale 2
    wide = rng.integers(-10**15, 10**15, n)             # scale 4, 15 digits
    qn, pn = rng.random(n) < nulls, rng.random(n) < nulls
    num = [kds.Column("int4", g), kds.numeric_from_scaled(qty, 0, qn), kds.numeric_from_scaled(prc, 2, pn),
           kds.numeric_from_scaled(dsc, 2), kds.numeric_from_scaled(wide, 4)]
    dec = [kds.Column("int4", g), kds.Column("decimal", qty, qn), kds.Column("decimal", prc, pn),
           kds.Column("decimal", dsc), kds.Column("decimal", wide)]
    return num, dec


def specs():
    """[(spec with numeric vars, the same with decimal vars)]"""
    out = []
    agg = ("(gpupreagg (qual (numeric_ge (var 3 {T} 2) (const numeric -10.5))) (key (var 1 int4))"
           " (nrows) (nrows (isnotnull (var 2 {T} 0))) (psum (var 2 {T} 0) 0) (psum (var 3 {T} 2) 2)"
           " (psum (numeric_mul (var 3 {T} 2) (numeric_sub (const numeric 1) (var 4 {T} 2))) 4)"
           " (pmin (var 3 {T} 2) 2) (pmax (var 5 {T} 4) 4) (psum (var 5 {T} 4) 4))")
    out.append((agg.replace("{T}", "numeric"), agg.replace("{T}", "decimal")))
    return out


SCAN_QUALS = [
    "(numeric_lt (var 3 {T} 2) (const numeric 5.25))",
    "(and (numeric_ge (var 2 {T} 0) (const numeric 0)) (numeric_ne (var 4 {T} 2) (const numeric 0.05)))",
    "(numeric_gt (numeric_mul (var 3 {T} 2) (var 4 {T} 2)) (numeric_add (var 5 {T} 4) (const numeric 100)))",
    "(isnull (var 2 {T} 0))",
]
