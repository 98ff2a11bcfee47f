"""
Wire-format layout pins (no GPU).  The numbers are SURVEY.md Appendix A --
sizes and offsets measured by compiling the reference's own headers
(opencl_common.h:335-486, opencl_hashjoin.h:102-165, opencl_gpupreagg.h:67-106)
-- and PostgreSQL 9.4 heap-page arithmetic for an (int4, float8) table.
"""
import os
import re

import numpy as np

import oracle_binding
from pg_strom_amd import kds
from pg_strom_amd._lib import lib, PROTOTYPES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_struct_layout_matches_reference():
    lay = oracle_binding.layout()
    assert lay["sizeof_kern_data_store_head"] == 48
    assert lay["sizeof_kern_colmeta"] == 8
    assert lay["sizeof_kern_rowitem"] == 4
    assert lay["sizeof_kern_blkitem"] == 16
    assert lay["offsetof_resultbuf_results"] == 20
    assert lay["sizeof_kern_parambuf_head"] == 8
    assert lay["sizeof_kern_hashentry"] == 40
    assert lay["offsetof_hashentry_htup"] == 16
    assert lay["offsetof_htup_t_bits"] == 23
    assert lay["sizeof_kern_multihash_head"] == 1040 - 4   # + htable_offset[0]
    assert lay["offsetof_gpupreagg_kparams"] == 16
    assert lay["sizeof_kern_coldir"] == 32


def _table(n, nulls=False, seed=3):
    rng = np.random.default_rng(seed)
    a = rng.integers(-2**31, 2**31, n, dtype=np.int64).astype(np.int32)
    b = rng.random(n)
    an = (rng.random(n) < 0.1) if nulls else None
    bn = (rng.random(n) < 0.1) if nulls else None
    return a, b, an, bn


def test_row_format_page_arithmetic():
    # header 23 -> t_hoff 24; a at 24, b at 32 -> t_len 40; 185 rows per 8KB page
    a, b, _, _ = _table(185 * 3 + 1)
    buf = kds.build_kds("row", [kds.Column("int4", a), kds.Column("float8", b)])
    head = kds.KdsHead(buf)
    assert head.format == kds.KDS_FORMAT_ROW
    assert head.nblocks == 4 and head.nitems == 185 * 3 + 1
    assert list(head.colmeta["attcacheoff"]) == [24, 32]
    assert list(head.colmeta["attlen"]) == [4, 8]
    # pages start BLCKSZ aligned after head + blkitems + rowitems
    off = 48 + 16            # KDS_HEAD_LENGTH(2) = STROMALIGN(48+16) = 64
    off = ((off + 15) & ~15) + ((16 * head.maxblocks + 15) & ~15) + ((4 * head.nitems + 15) & ~15)
    off = (off + 8191) & ~8191
    page0 = buf[off:off + 8192]
    pd_lower, pd_upper = np.frombuffer(page0[12:16].tobytes(), dtype=np.uint16)
    assert pd_lower == 24 + 4 * 185
    assert pd_upper == 8192 - 40 * 185
    itemid = int(np.frombuffer(page0[24:28].tobytes(), dtype=np.uint32)[0])
    assert itemid & 0x7fff == 8192 - 40          # lp_off
    assert (itemid >> 15) & 3 == 1               # LP_NORMAL
    assert (itemid >> 17) & 0x7fff == 40         # lp_len
    t_hoff = page0[8192 - 40 + 22]
    assert t_hoff == 24


def test_tupslot_and_column_layout():
    a, b, an, bn = _table(1000, nulls=True)
    cols = [kds.Column("int4", a, an), kds.Column("float8", b, bn)]
    ts = kds.build_kds("tupslot", cols)
    assert len(ts) == ((64 + 24 * 1000 + 15) & ~15)   # LONGALIGN(9*2)=24 per row
    col = kds.build_kds("column", cols)
    head = kds.KdsHead(col)
    assert head.format == kds.KDS_FORMAT_COLUMN and head.length == len(col)
    coldir = np.frombuffer(col[64:64 + 64].tobytes(), dtype=np.uint32).reshape(2, 8)
    for c in range(2):
        assert coldir[c, 0] % 256 == 0 and coldir[c, 1] % 256 == 0 and coldir[c, 1] != 0
    # zone map of the int4 column
    mn, mx = np.frombuffer(col[64 + 16:64 + 32].tobytes(), dtype=np.int64)
    assert mn == a[~an].min() and mx == a[~an].max()


def test_all_formats_round_trip_every_cell():
    a, b, an, bn = _table(777, nulls=True, seed=9)
    cols = [kds.Column("int4", a, an), kds.Column("float8", b, bn)]
    for fmt in ("row", "row_flat", "tupslot", "column"):
        buf = kds.build_kds(fmt, cols)
        for r in (0, 1, 184, 185, 186, 500, 776):
            isnull, v = kds.kds_fetch(buf, r, 0)
            assert isnull == bool(an[r])
            if not isnull:
                assert np.int32(np.uint32(v & 0xffffffff)) == a[r]
            isnull, v = kds.kds_fetch(buf, r, 1)
            assert isnull == bool(bn[r])
            if not isnull:
                assert np.array([v], dtype=np.uint64).view(np.float64)[0] == b[r]
    conv = kds.kds_to_column(kds.build_kds("row", cols))
    direct = kds.build_kds("column", cols)
    # identical past the head (hostptr differs) and the zone-map fields
    assert bytes(conv[256:]) == bytes(direct[256:])


def test_c_abi_exports_every_declared_symbol():
    """every function include/*.h declares resolves in libstrom_hip.so"""
    declared = set()
    for name in os.listdir(os.path.join(ROOT, "include")):
        text = open(os.path.join(ROOT, "include", name)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared |= set(re.findall(r"\b(strom_[a-z0-9_]+)\s*\(", text))
    declared -= {"strom_done_cb"}
    assert len(declared) >= 30
    for sym in sorted(declared):
        assert hasattr(lib, sym), "libstrom_hip.so does not export " + sym
        assert sym in PROTOTYPES, "binding lacks a prototype for " + sym


def test_merge_entry_points_refuse_null_sessions_without_a_gpu():
    """the multi-GPU / hashed merge calls answer BadRequestMessage for a missing session or
    communicator before anything touches a device (a backend must get an error code, never a crash)"""
    from pg_strom_amd._lib import lib
    assert lib.strom_gpupreagg_merge(None, None) == 101
    assert lib.strom_gpupreagg_allreduce(None, None, None) == 101
    assert lib.strom_gpupreagg_census_allreduce(None, None, None) != 0
