"""
(var N decimal S) on the device (needs an MI355X: -m gpu): numeric(p,s) columns held as int8 at
10^-s.  The same table in both encodings through GpuScan and GpuPreAgg: identical row sets,
identical partial rows (integers: bit-exact), and both equal the oracle.
"""
import numpy as np
import pytest

import decimal_cases
import oracle_binding as oracle
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpupreagg import GpuPreAgg
from test_gpuscan_gpu import check
from test_gpupreagg_gpu import partial_rows_as_raw8

pytestmark = pytest.mark.gpu


def test_scan_quals_over_both_encodings():
    num, dec = decimal_cases.tables(50003, 11)
    bn, bd = kds.build_kds("column", num), kds.build_kds("column", dec)
    for q in decimal_cases.SCAN_QUALS:
        rn = check(q.replace("{T}", "numeric"), bn)
        rd = check(q.replace("{T}", "decimal"), bd)            # (check: HIP == oracle)
        assert np.array_equal(np.sort(rn.results[:rn.nitems]), np.sort(rd.results[:rd.nitems])), q


@pytest.mark.parametrize("resident", [False, True])
def test_preagg_partial_rows_are_identical_over_both_encodings(resident):
    num, dec = decimal_cases.tables(120007, 12)
    bn, bd = kds.build_kds("column", num), kds.build_kds("column", dec)
    for sn, sd in decimal_cases.specs():
        rows = []
        for spec, buf in ((sn, bn), (sd, bd)):
            agg = GpuPreAgg(spec).begin([(0, 7)])
            ds = runtime.DeviceStore.upload(buf) if resident else None
            try:
                assert agg.fold(ds if resident else buf)[0] == 0
                v, n = partial_rows_as_raw8(agg.fetch())
            finally:
                agg.end()
                if ds is not None:
                    ds.release()
            o = np.argsort(v[:, 0])
            rows.append((v[o], n[o]))
        assert np.array_equal(rows[0][0], rows[1][0]) and np.array_equal(rows[0][1], rows[1][1])
        # and the oracle over the numeric-image form: its scaled partials are int8 at 10^-scale,
        # the device's fetched rows carry them as numerics (tests/test_numeric_gpu.py)
        from decimal import Decimal
        rc, ov, on = oracle.gpupreagg(sn, bn, rows[0][0].shape[1])
        oo = np.argsort(ov[:, 0])
        assert rc == 0 and np.array_equal(on[oo], rows[0][1])
        scales = [None, None, None, 0, 2, 4, 2, 4, 4]
        for t, sc in enumerate(scales):
            a, b, isn = ov[oo][:, t], rows[0][0][:, t], rows[0][1][:, t]
            if sc is None:
                assert np.array_equal(a, b), t
                continue
            for i in np.flatnonzero(~isn):
                assert kds.numeric_decode(b[i]) == Decimal(int(a[i].view(np.int64))).scaleb(-sc), (t, i)


@pytest.mark.parametrize("fmt,coltype", [("row", "numeric_varlena"), ("row_flat", "numeric_varlena"),
                                         ("tupslot", "numeric")])
def test_ingest_turns_numeric_columns_into_decimal_columns(fmt, coltype):
    """strom_dstore_to_column with STROM_DECIMAL_TYPE(scale): heap numerics (varlena) and 8-byte
    numeric images become int8 at 10^-scale, exactly; the decimal program over the result gives
    the numeric program's partial rows"""
    n = 30011
    num, dec = decimal_cases.tables(n, 21)
    src_cols = [num[0]] + [kds.Column(coltype, c.values, c.isnull) for c in num[1:]]
    src = runtime.DeviceStore.upload(kds.build_kds(fmt, src_cols))
    col, _ = src.to_column([23, kds.decimal_type(0), kds.decimal_type(2), kds.decimal_type(2), kds.decimal_type(4)])
    try:
        img = col.download()
        head = kds.KdsHead(img)
        assert head.format == 4 and head.nitems == n
        sn, sd = decimal_cases.specs()[0]
        rows = []
        for spec, chunk in ((sd, col), (sn, kds.build_kds("column", num))):
            agg = GpuPreAgg(spec).begin([(0, 7)])
            try:
                assert agg.fold(chunk)[0] == 0
                v, nn = partial_rows_as_raw8(agg.fetch())
            finally:
                agg.end()
            o = np.argsort(v[:, 0])
            rows.append((v[o], nn[o]))
        assert np.array_equal(rows[0][0], rows[1][0]) and np.array_equal(rows[0][1], rows[1][1])
    finally:
        col.release()
        src.release()


def test_ingest_refuses_a_scale_the_values_do_not_fit():
    """1.234 is not a numeric(*,2): the conversion fails as a whole (CpuReCheck), the chunk stays
    in its row format"""
    vals = kds.numeric_from_scaled(np.array([100, 1234, 5], dtype=np.int64), 3)
    src = runtime.DeviceStore.upload(kds.build_kds("row", [kds.Column("numeric_varlena", vals.values)]))
    try:
        with pytest.raises(runtime.StromError) as ei:
            src.to_column([kds.decimal_type(2)])
        assert ei.value.errcode == 2
        col, _ = src.to_column([kds.decimal_type(3)])
        col.release()
    finally:
        src.release()


@pytest.mark.parametrize("ngroups", [10000, 25000])
def test_sum_of_a_decimal_column_at_its_own_scale_is_a_packed_accumulator_input(ngroups, monkeypatch):
    """sum(numeric(p,2)) over a decimal column is the sum of an int8 column bounded by its zone
    map: it shares the packed 64-bit word with count(*) (num_kern_prep marks the packed path),
    and gives the standard image's partial rows bit for bit -- and numpy's"""
    from decimal import Decimal
    n = 200003
    rng = np.random.default_rng(5)
    g = rng.integers(0, ngroups, n).astype(np.int32)
    g[:ngroups] = np.arange(ngroups, dtype=np.int32)
    x = rng.integers(-10**6, 10**6, n)
    y = rng.random(n) * 100
    buf = kds.build_kds("column", [kds.Column("int4", g), kds.Column("decimal", x), kds.Column("float8", y)])
    spec = "(gpupreagg (key (var 1 int4)) (nrows) (psum (var 2 decimal 2) 2) (psum (var 3 float8)))"
    rows = []
    for packed in (True, False):
        if not packed:
            monkeypatch.setenv("STROM_GPUPREAGG_NO_PACKED", "1")
        agg = GpuPreAgg(spec).begin([(0, ngroups)])
        ds = runtime.DeviceStore.upload(buf)
        try:
            status, pfm = agg.fold(ds)
            assert status == 0 and pfm["num_kern_prep"] == (1 if packed else 0)
            v, isn = partial_rows_as_raw8(agg.fetch())
        finally:
            agg.end()
            ds.release()
        o = np.argsort(v[:, 0])
        rows.append((v[o], isn[o]))
    assert np.array_equal(rows[0][0][:, :3], rows[1][0][:, :3]) and np.array_equal(rows[0][1], rows[1][1])
    assert np.allclose(rows[0][0][:, 3].view(np.float64), rows[1][0][:, 3].view(np.float64), rtol=1e-12)
    want_n = np.bincount(g, minlength=ngroups)
    want_x = np.zeros(ngroups, dtype=np.int64)
    np.add.at(want_x, g, x)
    v = rows[0][0]
    assert np.array_equal(v[:, 0].view(np.int64), np.arange(ngroups)) and np.array_equal(v[:, 1].view(np.int64), want_n)
    for i in range(0, ngroups, 37):
        assert kds.numeric_decode(v[i, 2]) == Decimal(int(want_x[i])).scaleb(-2), i
    # a sum at another scale than the column's is an expression: the standard image
    other = "(gpupreagg (key (var 1 int4)) (nrows) (psum (var 2 decimal 2) 4) (psum (var 3 float8)))"
    monkeypatch.delenv("STROM_GPUPREAGG_NO_PACKED")
    agg = GpuPreAgg(other).begin([(0, ngroups)])
    try:
        status, pfm = agg.fold(buf)
        assert status == 0 and pfm["num_kern_prep"] == 0
    finally:
        agg.end()
