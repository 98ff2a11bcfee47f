"""Expression emitter (no GPU): text shape, type checking, parameter
de-duplication, and that the generated programs compile for gfx950 through
the runtime's own program cache (hiprtc cross-compiles without a device)."""
import numpy as np
import pytest

from pg_strom_amd import runtime

C2_QUAL = "(and (int4lt (var 1 int4) (param 0 int4)) (float8gt (var 2 float8) (param 1 float8)))"


def test_gpuscan_codegen_shape():
    cg = runtime.codegen_gpuscan(C2_QUAL)
    assert "#define STROM_KPARAM_LIST(X) X(0,int4) X(1,float8)" in cg.source
    assert "#define STROM_KVAR_LIST(X) X(1,0,int4) X(2,1,float8)" in cg.source
    assert "pgfn_boolop_and2(pgfn_int4lt(errcode, KV.KVAR_1, KP.KPARAM_0), " \
           "pgfn_float8gt(errcode, KV.KVAR_2, KP.KPARAM_1))" in cg.source
    assert cg.vars == [(1, 23), (2, 701)]
    assert [p[:3] for p in cg.params] == [(23, False, 0), (701, False, 1)]


def test_constants_are_deduplicated_and_params_packed():
    cg = runtime.codegen_gpuscan(
        "(or (int4eq (var 1 int4) (const int4 7)) (int4eq (var 3 int4) (const int4 7)))")
    assert len(cg.params) == 1
    pb = cg.parambuf()
    length, nparams, poff0 = np.frombuffer(pb[:12], dtype=np.uint32)
    assert nparams == 1 and poff0 == 16 and length == 32 and len(pb) == 32
    assert np.frombuffer(pb[16:20], dtype=np.int32)[0] == 7


def test_external_param_null_and_value():
    cg = runtime.codegen_gpuscan(C2_QUAL)
    pb = cg.parambuf([np.int32(5), None])
    length, nparams, p0, p1 = np.frombuffer(pb[:16], dtype=np.uint32)
    assert (nparams, p0, p1) == (2, 16, 0)
    assert np.frombuffer(pb[16:20], dtype=np.int32)[0] == 5


@pytest.mark.parametrize("expr,msg", [
    ("(int4lt (var 1 int4) (const float8 1.0))", "not supported"),
    ("(and (var 1 int4))", "not bool"),
    ("(frobnicate (var 1 int4))", "not supported"),
    ("(int4lt (var 1 int4)", "missing"),
    ("(const int2 70000)", "out of range"),
])
def test_codegen_rejects(expr, msg):
    ok, err = runtime.expression_available(expr)
    assert not ok and msg in err


def test_available_expression():
    assert runtime.expression_available("(float8mul (float8 (var 1 int4)) (const float8 2.5))")[0]
    assert runtime.expression_available(
        "(case (when (int4gt (var 1 int4) (const int4 0)) (var 2 float8)) (else (const float8 0)))")[0]


@pytest.mark.parametrize("qual", [
    C2_QUAL,
    "(not (isnull (var 2 float8)))",
    "(int8gt (int48pl (var 1 int4) (var 3 int8)) (const int8 100))",
    "(is_not_true (float4lt (var 4 float4) (float4 (var 2 float8))))",
    "(int4eq (case (when (int2lt (var 5 int2) (const int2 3)) (const int4 1)) (else (int4um (var 1 int4)))) (const int4 1))",
    "(date_le (var 6 date) (const date '1998-09-02'))",
])
def test_generated_programs_build_for_gfx950(qual):
    cg = runtime.codegen_gpuscan(qual)
    prog = runtime.DevProgram(cg.source, cg.extra_flags)
    prog.wait()          # raises with the compiler log on failure
    assert prog.state() == 1


def test_bad_program_reports_build_log():
    prog = runtime.DevProgram("#include \"strom_kds.h\"\nthis is not HIP;\n", 0)
    with pytest.raises(runtime.StromError) as e:
        prog.wait()
    assert e.value.errcode == -11 and "error" in str(e.value)
