"""
Host chunk builders (csrc/datastore.cpp; reference datastore.c:312-828):
every format must hand back the cell values it was given, and the host
ROW -> COLUMN conversion must equal a COLUMN chunk built directly.
CPU only: no device call.
"""
import numpy as np
import pytest

from pg_strom_amd import kds


def columns(n, seed):
    rng = np.random.default_rng(seed)
    return [kds.Column("int4", rng.integers(-2**31, 2**31, n), rng.random(n) < 0.2),
            kds.Column("float8", rng.normal(size=n)),
            kds.Column("int2", rng.integers(-100, 100, n), rng.random(n) < 0.5),
            kds.Column("int8", rng.integers(-2**62, 2**62, n)),
            kds.Column("float4", rng.normal(size=n))]     # 5 columns: head length needs STROMALIGN


@pytest.mark.parametrize("fmt", ["row", "row_flat", "tupslot", "column"])
def test_fetch_returns_what_was_stored(fmt):
    n = 777
    cols = columns(n, 3)
    buf = kds.build_kds(fmt, cols)
    head = kds.KdsHead(buf)
    assert head.nitems == n and head.ncols == len(cols)
    for r in (0, 1, 184, 185, 186, 400, n - 1):
        for c, col in enumerate(cols):
            isnull, image = kds.kds_fetch(buf, r, c)
            want_null = bool(col.isnull[r]) if col.isnull is not None else False
            assert bool(isnull) == want_null
            if not want_null:
                raw = int(np.frombuffer(col.values[r:r + 1].tobytes().ljust(8, b"\0"), dtype="<u8")[0])
                mask = (1 << (8 * col.attlen)) - 1
                assert (image & mask) == (raw & mask)


@pytest.mark.parametrize("fmt", ["row", "row_flat", "tupslot"])
def test_host_conversion_equals_direct_column_build(fmt):
    n = 1000
    cols = columns(n, 11)
    direct = kds.decode_column_chunk(kds.build_kds("column", cols))
    conv = kds.decode_column_chunk(kds.kds_to_column(kds.build_kds(fmt, cols)))
    for d, c in zip(direct, conv):
        assert np.array_equal(d["values"], c["values"])
        assert (d["notnull"] is None) == (c["notnull"] is None)
        if d["notnull"] is not None:
            assert np.array_equal(d["notnull"], c["notnull"])


def test_zone_map_of_a_column_chunk():
    a = np.array([5, -3, 9, 100], dtype=np.int32)
    an = np.array([0, 0, 0, 1], dtype=bool)       # the 100 is NULL
    b = np.array([0.5, -2.0, np.nan, 7.25])
    dec = kds.decode_column_chunk(kds.build_kds("column", [kds.Column("int4", a, an),
                                                           kds.Column("float8", b)]))
    assert dec[0]["stat_flags"] == 1 and (dec[0]["minval"], dec[0]["maxval"]) == (-3, 9)
    assert dec[1]["stat_flags"] == 3
    mm = np.array([dec[1]["minval"], dec[1]["maxval"]], dtype=np.int64).view(np.float64)
    assert list(mm) == [-2.0, 7.25]
    assert dec[1]["notnull"] is None and list(dec[0]["notnull"]) == [True, True, True, False]


def test_column_head_equals_the_head_of_a_built_chunk():
    """strom_kds_column_head (head only, payload filled elsewhere) lays a NULL-free COLUMN
    chunk out exactly as strom_kds_build does: same head bytes, offsets and zone maps"""
    import numpy as np
    from pg_strom_amd import kds
    rng = np.random.default_rng(8)
    n = 100003
    a = rng.integers(-5, 1000, n).astype(np.int32)
    b = rng.normal(size=n)
    c = rng.integers(0, 2**40, n).astype(np.int64)
    d = rng.integers(0, 3, n).astype(np.int8)
    full = kds.build_kds("column", [kds.Column("int4", a), kds.Column("float8", b), kds.Column("int8", c),
                                    kds.Column("char1", d)])
    head, total, voff = kds.column_head(["int4", "float8", "int8", "char1"], n,
                                        [(a.min(), a.max()), (b.min(), b.max()), (c.min(), c.max()),
                                         (d.min(), d.max())])
    assert total == len(full)
    # hostptr (first 8 bytes) is the buffer's own address: not comparable
    assert np.array_equal(head[8:], full[8:len(head)])
    assert np.array_equal(full[voff[1]:voff[1] + 8 * n].view(np.float64), b)
    assert np.array_equal(full[voff[3]:voff[3] + n].view(np.int8), d)


def test_fixup_kernel_numeric_makes_postgresql_numerics():
    """strom_fixup_kernel_numeric / strom_kernel_numeric_cstring <- pgstrom_fixup_kernel_numeric
    (datastore.c:150-167): a 64-bit device numeric as PostgreSQL's own datum.  The datum must be
    byte for byte what PostgreSQL 9.4 stores for the value (numeric_golden.pg_numeric_varlena, the
    encoder the reference's recheck_agg literals go through; 4-byte varlena header here), the
    oracle's varlena reader (opencl_numeric.h:166-307) must give the image back, and the text must
    be the reference's own "%c%lue%d"."""
    import ctypes
    from decimal import Decimal
    import numeric_golden as ng
    import oracle_binding as oracle
    from pg_strom_amd._lib import lib
    from test_numeric_cpu import random_numerics
    vals = list(random_numerics(3000, 31)) + [Decimal(0), Decimal("1E+31"), Decimal("-1E-32"), Decimal("144115188075855871"),
                                              Decimal("-14411518807585587E+30"), Decimal("0.0001"), Decimal("10000"),
                                              Decimal("99999999.99990000"), Decimal("1E+48"), Decimal("5E-30")]
    buf = ctypes.create_string_buffer(256)
    datums = []
    for d in vals:
        img = kds.numeric_encode(d)
        assert img is not None
        n = lib.strom_fixup_kernel_numeric(img, buf, 256)
        assert n > 0
        got = buf.raw[:n]
        want = ng.pg_numeric_varlena(kds.numeric_decode(img))          # (the image's own exponent: its display scale)
        body = want[1:] if want[0] & 1 else want[4:]
        assert got[4:] == body and int.from_bytes(got[:4], "little") == n << 2, (d, got.hex(), want.hex())
        datums.append(got)
        m = lib.strom_kernel_numeric_cstring(img, buf, 256)
        exp = int(img) >> 58
        exp = exp - 64 if exp >= 32 else exp
        assert buf.raw[:m].decode() == "%s%de%d" % ("-" if (int(img) >> 57) & 1 else "+", int(img) & ((1 << 57) - 1), exp)
        assert Decimal(buf.raw[:m].decode()) == d
    assert lib.strom_fixup_kernel_numeric(kds.numeric_encode(Decimal("123.456")), buf, 7) == -301
    # round trip through a heap chunk: the oracle reads the datums back as the same images
    chunk = kds.build_kds("row", [kds.Column("numeric_raw", datums)])
    oid, v, isn, err = oracle.eval_rows("(numeric_uplus (var 1 numeric))", chunk)
    assert not err.any() and [int(x) for x in v] == [int(kds.numeric_encode(d)) for d in vals]
