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
