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
