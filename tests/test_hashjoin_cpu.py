"""
GpuHashJoin host-side pieces (no GPU): the kern_multihash the product's
builder lays out is walked by the CPU oracle the way the reference's device
code walks it (KERN_HASH_FIRST_ENTRY / NEXT_ENTRY, opencl_hashjoin.h:167-192)
and must hold every inner row once, under slot = pg_crc32(keys) % nslots;
the oracle's join itself is checked against numpy; codegen shapes.
"""
import numpy as np
import pytest

import oracle_binding as oracle
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpuhashjoin import build_multihash, codegen_gpuhashjoin, entry_rowids


def dim_table(n, seed, dup=False):
    rng = np.random.default_rng(seed)
    pk = rng.permutation(n * 2)[:n].astype(np.int32)
    if dup:
        pk[n // 2:] = pk[:n - n // 2]
    payload = rng.integers(0, 1000, n).astype(np.int32)
    pkn = rng.random(n) < 0.02
    return pk, payload, pkn


def test_layout_of_kern_multihash():
    pk, payload, pkn = dim_table(5000, 1)
    for fmt in ("row", "row_flat", "column"):
        inner = kds.build_kds(fmt, [kds.Column("int4", pk, pkn), kds.Column("int4", payload)])
        km = build_multihash([(inner, [1])])
        assert oracle.check_hashtable(km, 1, inner, [1], [4]) == 5000
        ntables = np.frombuffer(km[1032:1036].tobytes(), dtype=np.uint32)[0]
        assert ntables == 1
        # a key-only (int4,int4) entry: 16 + t_len(32) -> 48 bytes, LONGALIGNed
        toff = int(np.frombuffer(km[1036:1040].tobytes(), dtype=np.uint32)[0])
        length, ncols, nslots = np.frombuffer(km[toff:toff + 12].tobytes(), dtype=np.uint32)
        assert ncols == 2 and nslots == int(5000 * 1.15) + 1


def test_two_tables_and_multi_key():
    rng = np.random.default_rng(3)
    a = rng.integers(0, 50, 3000).astype(np.int32)
    b = rng.integers(0, 7, 3000).astype(np.int64)
    c = rng.random(3000)
    t1 = kds.build_kds("row", [kds.Column("int4", a), kds.Column("int8", b), kds.Column("float8", c)])
    pk, payload, pkn = dim_table(400, 4)
    t2 = kds.build_kds("row_flat", [kds.Column("int4", pk, pkn), kds.Column("int4", payload)])
    km = build_multihash([(t1, [1, 2]), (t2, [1])])
    assert oracle.check_hashtable(km, 1, t1, [1, 2], [4, 8]) == 3000
    assert oracle.check_hashtable(km, 2, t2, [1], [4]) == 400


def test_pg_crc32_is_the_legacy_variant():
    # PostgreSQL 9.4's COMP_CRC32 (reflected table, MSB-first update) is NOT zlib's crc32
    import zlib
    assert oracle.pg_crc32(b"123456789") != zlib.crc32(b"123456789")
    assert oracle.pg_crc32(b"") == 0
    # one-byte message: table[0xFF ^ b] ^ 0xFFFFFF00, finalised
    t = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (0xEDB88320 ^ (c >> 1)) if c & 1 else (c >> 1)
        t.append(c)
    for byte in (0, 1, 0x41, 0xFF):
        want = (t[(0xFF ^ byte) & 0xFF] ^ 0xFFFFFF00) ^ 0xFFFFFFFF
        assert oracle.pg_crc32(bytes([byte])) == want & 0xFFFFFFFF


def test_oracle_join_matches_numpy():
    rng = np.random.default_rng(7)
    pk, payload, pkn = dim_table(1000, 8, dup=True)
    inner = kds.build_kds("row", [kds.Column("int4", pk, pkn), kds.Column("int4", payload)])
    fk = rng.integers(0, 2200, 20000).astype(np.int32)
    fkn = rng.random(20000) < 0.03
    outer = kds.build_kds("column", [kds.Column("int4", fk, fkn)])
    rc, n, recs = oracle.gpuhashjoin("(gpuhashjoin (rel (hashkey (var 1 int4) 1 int4)))", outer, [inner])
    assert rc == 0
    want = set()
    index = {}
    for r in range(1000):
        if not pkn[r]:
            index.setdefault(int(pk[r]), []).append(r)
    for o in range(20000):
        if not fkn[o]:
            for r in index.get(int(fk[o]), ()):
                want.add((o + 1, r))
    assert n == len(want) and set(map(tuple, recs.tolist())) == want
    # with a join qual on the inner payload and NoSpace reporting
    rc, n2, recs2 = oracle.gpuhashjoin(
        "(gpuhashjoin (rel (hashkey (var 1 int4) 1 int4) (qual (int4lt (ivar 1 2 int4) (const int4 500)))))",
        outer, [inner], nrooms=10)
    want2 = {(o, r) for (o, r) in want if payload[r] < 500}
    assert rc == 301 and n2 == len(want2)


def test_codegen_shapes_and_errors():
    cg = codegen_gpuhashjoin("(gpuhashjoin (rel (hashkey (var 1 int4) 1 int4)))")
    assert cg.nrels == 1 and "#define HASHJOIN_FAST_ELIGIBLE 1" in cg.source
    cg = codegen_gpuhashjoin("(gpuhashjoin (rel (hashkey (var 1 int4) 1 int4) (qual (int4gt (ivar 1 2 int4) (const int4 3)))))")
    assert "#define HASHJOIN_FAST_ELIGIBLE 0" in cg.source and "IVAR_1_2" in cg.source
    with pytest.raises(ValueError):
        codegen_gpuhashjoin("(gpuhashjoin (rel (hashkey (var 1 int4) 1 int8)))")      # type mismatch
    with pytest.raises(ValueError):
        codegen_gpuhashjoin("(gpuhashjoin (rel (hashkey (ivar 1 1 int4) 1 int4)))")   # own depth in a key
    prog = runtime.DevProgram(cg.source, cg.extra_flags)
    prog.wait()
