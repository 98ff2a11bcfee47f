"""
Device ingest (ROW / ROW_FLAT / TUPSLOT -> KDS_FORMAT_COLUMN) parity
(needs an MI355X: -m gpu).  The transposed chunk must carry the same
values, NULLs and zone maps as the host builder (strom_kds_build /
strom_kds_to_column) lays out for the same rows, and the operators must
return the same results from it as from the source chunk.
"""
import numpy as np
import pytest

import oracle_binding as oracle
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpuscan import GpuScan

pytestmark = pytest.mark.gpu

TYPES = ("int4", "float8", "int2", "int8", "float4", "bool", "date")


def make_columns(n, seed, null_frac):
    rng = np.random.default_rng(seed)
    cols = []
    for i, t in enumerate(TYPES):
        if t in ("float8", "float4"):
            v = rng.normal(size=n) * 1000
        elif t == "bool":
            v = rng.integers(0, 2, n)
        elif t == "int2":
            v = rng.integers(-30000, 30000, n)
        elif t == "int8":
            v = rng.integers(-2**62, 2**62, n)
        else:
            v = rng.integers(-2**31, 2**31, n)
        isnull = None
        if null_frac and i % 2 == 0:
            isnull = rng.random(n) < null_frac
        cols.append(kds.Column(t, v, isnull))
    return cols


def check_same(dev_cols, host_cols, columns):
    for c, (d, h) in enumerate(zip(dev_cols, host_cols)):
        hn = h["notnull"]
        dn = d["notnull"]
        if hn is None:
            assert dn is None, "column %d: device kept a bitmap the host dropped" % c
        else:
            assert dn is not None and np.array_equal(dn, hn)
        assert np.array_equal(d["values"], h["values"]), "column %d values" % c
        assert d["stat_flags"] == h["stat_flags"], "column %d stat flags" % c
        if h["stat_flags"] & 1:
            if h["stat_flags"] & 2:
                dm = np.array([d["minval"], d["maxval"]], dtype=np.int64).view(np.float64)
                hm = np.array([h["minval"], h["maxval"]], dtype=np.int64).view(np.float64)
                assert np.array_equal(dm, hm)
            else:
                assert (d["minval"], d["maxval"]) == (h["minval"], h["maxval"])


@pytest.mark.parametrize("fmt", ["row", "row_flat", "tupslot"])
@pytest.mark.parametrize("n", [0, 1, 31, 32, 33, 63, 64, 65, 255, 256, 257, 4097, 100003])
def test_transposed_chunk_equals_host_layout(fmt, n):
    runtime.init()
    columns = make_columns(n, 7000 + n, 0.1 if n % 2 else 0.0)
    src = kds.build_kds(fmt, columns)
    host = kds.decode_column_chunk(kds.build_kds("column", columns))
    ds = runtime.DeviceStore.upload(src)
    try:
        col, _ = ds.to_column([c.type_oid for c in columns])
        try:
            dev = kds.decode_column_chunk(col.download())
        finally:
            col.release()
    finally:
        ds.release()
    check_same(dev, host, columns)


def test_all_null_column_and_no_type_oids():
    runtime.init()
    n = 1000
    a = kds.Column("int4", np.arange(n), np.ones(n, dtype=bool))
    b = kds.Column("float8", np.arange(n) * 0.5)
    src = kds.build_kds("row", [a, b])
    ds = runtime.DeviceStore.upload(src)
    col, _ = ds.to_column([a.type_oid, b.type_oid])
    dev = kds.decode_column_chunk(col.download())
    col.release()
    assert not dev[0]["notnull"].any() and dev[0]["stat_flags"] == 0
    assert not dev[0]["values"].any()
    assert dev[1]["notnull"] is None and dev[1]["stat_flags"] == 3
    # without type oids: values and NULLs only, no zone maps
    col, _ = ds.to_column(None)
    dev = kds.decode_column_chunk(col.download())
    col.release()
    ds.release()
    assert dev[0]["stat_flags"] == 0 and dev[1]["stat_flags"] == 0
    assert np.array_equal(dev[1]["values"].view(np.float64), np.arange(n) * 0.5)


@pytest.mark.parametrize("fmt", ["row", "row_flat", "tupslot"])
def test_gpuscan_on_ingested_chunk_matches_oracle_on_source(fmt):
    runtime.init()
    n = 200003
    rng = np.random.default_rng(5)
    a = rng.integers(0, 2**31, n)
    b = rng.random(n)
    an = rng.random(n) < 0.03
    columns = [kds.Column("int4", a, an), kds.Column("float8", b)]
    qual = "(and (int4lt (var 1 int4) (param 0 int4)) (float8gt (var 2 float8) (param 1 float8)))"
    ext = [np.int32(2**30), 0.25]
    src = kds.build_kds(fmt, columns)
    rc_o, res_o = oracle.gpuscan(qual, src, ext)
    ds = runtime.DeviceStore.upload(src)
    col, _ = ds.to_column([c.type_oid for c in columns])
    scan = GpuScan(qual).begin(ext_params=ext)
    try:
        res = scan.scan_chunk(col)
    finally:
        scan.end()
        col.release()
        ds.release()
    assert res.errcode == rc_o
    assert np.array_equal(np.sort(np.asarray(res.results)), np.sort(np.asarray(res_o)))


def test_rejects_column_source_and_bad_type_count():
    runtime.init()
    columns = make_columns(100, 1, 0.0)
    ds = runtime.DeviceStore.upload(kds.build_kds("column", columns))
    with pytest.raises(runtime.StromError):
        ds.to_column(None)
    ds.release()
    ds = runtime.DeviceStore.upload(kds.build_kds("row", columns))
    with pytest.raises(runtime.StromError):
        ds.to_column([23])
    ds.release()
