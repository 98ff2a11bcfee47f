"""
(var N decimal S): a numeric(p,s) column held as int8 at 10^-s in a COLUMN chunk.  CPU side: the
emitter types it as fixed-point straight away (no per-row decode), and the oracle -- which
computes with the SQL value -- gives the same answers as over the numeric-image form of the same
table.  The aggregates' known answers (the reference's suites) pin the numeric-image form;
this pins the second encoding to the first.
"""
import numpy as np
import pytest

import decimal_cases
import oracle_binding as oracle
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpupreagg import codegen_gpupreagg


def test_codegen_types_a_decimal_column_as_fixed_point():
    num, dec = decimal_cases.specs()[0]
    cgn, cgd = codegen_gpupreagg(num), codegen_gpupreagg(dec)
    # numeric images: decoded to fixed point once per row (the KFIX_<attno>_<scale> cache of strom_kvars);
    # a decimal column IS fixed point already
    assert "STROM_KFIXED_LIST(X) X(" in cgn.source and "pg_fixed_cached(errcode, KV.KFIX_3_2)" in cgn.source
    assert "pg_fixed_cached" not in cgd.source and "pgfn_numeric_as_fixed" not in cgd.source
    assert "pg_fixed_from_decimal(KV.KVAR_3)" in cgd.source
    assert [t for _, t in cgn.targets] == [t for _, t in cgd.targets]       # same partial-row types
    # sums of expressions over decimal columns carry a bound formula over the columns' zone maps
    # (the host evaluates it per chunk; the fold then does not measure the rows' magnitudes)
    q1 = ("(gpupreagg (key (var 1 char1)) (psum (numeric_mul (var 4 decimal 2) (numeric_sub (const numeric 1) (var 5 decimal 2))) 4)"
          " (psum (numeric_add (var 4 decimal 2) (const numeric 0.5)) 3) (psum (numeric_mul (var 4 decimal 2) (param 0 numeric)) 4))")
    src = codegen_gpupreagg(q1).source
    assert '#define GPUPREAGG_SUMBITS_0 66' in src and '#define GPUPREAGG_SUMBOUND_0 " c4 k1 e2 c5 + * "' in src
    assert '#define GPUPREAGG_SUMBOUND_1 " c4 k05 e1 + e1 "' in src          # 0.5 is 05 at scale 1; the sum at scale 2 -> 3
    assert '#define GPUPREAGG_SUMBITS_2 64' in src and 'GPUPREAGG_SUMBOUND_2' not in src    # a parameter: measured
    with pytest.raises(ValueError):
        runtime.codegen_gpuscan("(numeric_lt (var 1 decimal) (const numeric 1))")       # the scale is not optional


@pytest.mark.parametrize("fmt", ["column", "tupslot"])
def test_oracle_agrees_over_both_encodings(fmt):
    num, dec = decimal_cases.tables(20000, 3)
    bn, bd = kds.build_kds(fmt, num), kds.build_kds(fmt, dec)
    for q in decimal_cases.SCAN_QUALS:
        rn = oracle.gpuscan(q.replace("{T}", "numeric"), bn)
        rd = oracle.gpuscan(q.replace("{T}", "decimal"), bd)
        assert rn[0] == rd[0] == 0 and np.array_equal(np.sort(rn[1]), np.sort(rd[1])), q
    for sn, sd in decimal_cases.specs():
        nt = len(codegen_gpupreagg(sn).targets)
        rc1, v1, n1 = oracle.gpupreagg(sn, bn, nt)
        rc2, v2, n2 = oracle.gpupreagg(sd, bd, nt)
        assert rc1 == rc2 == 0
        o1, o2 = np.argsort(v1[:, 0]), np.argsort(v2[:, 0])
        assert np.array_equal(v1[o1], v2[o2]) and np.array_equal(n1[o1], n2[o2])


def test_sum_bound_formula_over_zone_maps():
    """GPUPREAGG_SUMBOUND_<a>: the code generator's bound of a summed expression, evaluated over a
    COLUMN chunk's zone maps (what the launch path does per chunk instead of measuring every row's
    magnitude): never below the largest |value| the rows really produce, and within a few bits of it"""
    import re
    from pg_strom_amd._lib import lib
    rng = np.random.default_rng(12)
    n = 5000
    prc = rng.integers(90000, 10494951, n)
    dsc = rng.integers(0, 11, n)
    tax = rng.integers(0, 9, n)
    dec = kds.build_kds("column", [kds.Column("char1", np.full(n, 65, dtype=np.int8)), kds.Column("int4", np.zeros(n, dtype=np.int32)),
                                   kds.Column("int4", np.zeros(n, dtype=np.int32)), kds.Column("decimal", prc),
                                   kds.Column("decimal", dsc), kds.Column("decimal", tax)])
    spec = ("(gpupreagg (key (var 1 char1))"
            " (psum (numeric_mul (numeric_mul (var 4 decimal 2) (numeric_sub (const numeric 1) (var 5 decimal 2)))"
            " (numeric_add (const numeric 1) (var 6 decimal 2))) 6))")
    src = codegen_gpupreagg(spec).source
    formula = re.search(r'#define GPUPREAGG_SUMBOUND_0 "([^"]*)"', src).group(1)
    bits = lib.strom_gpupreagg_sum_bound_bits(formula.encode(), dec.ctypes.data)
    true_max = int(max(abs(int(p) * (100 - int(d)) * (100 + int(t))) for p, d, t in zip(prc, dsc, tax)))
    assert true_max.bit_length() <= bits <= true_max.bit_length() + 2
    # a numeric image column answers through its integer-part bounds (nN), a column without a zone
    # map -- here: the formula names a float column -- not at all
    num = kds.build_kds("column", [kds.numeric_from_scaled(prc, 2), kds.Column("float8", rng.random(n))])
    b2 = lib.strom_gpupreagg_sum_bound_bits(b" n1 e2 ", num.ctypes.data)
    assert int(prc.max()).bit_length() <= b2 <= int(prc.max()).bit_length() + 1
    assert lib.strom_gpupreagg_sum_bound_bits(b" c2 ", num.ctypes.data) == -1
    assert lib.strom_gpupreagg_sum_bound_bits(b" c1 + ", dec.ctypes.data) == -1            # malformed
    assert lib.strom_gpupreagg_sum_bound_bits(b" c4 ", kds.build_kds("tupslot", [kds.Column("int8", prc)]).ctypes.data) == -1
