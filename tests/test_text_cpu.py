"""
text / character(n) on the CPU side: the oracle's comparison against Python's own bytes
comparison (PostgreSQL "C" collation = memcmp; character(n) without trailing blanks), the
expression emitter's catalogue (codegen.c:616-629) and the kern_parambuf of varlena
constants / parameters (datastore.c:100-127).  The reference holds no fixture for these
operators (its suites are aggregates over numeric columns): parity is "oracle = restatement
checked against Python", stated as such in DESIGN.md.
"""
import numpy as np
import pytest

import oracle_binding as oracle
import text_cases
from pg_strom_amd import kds, runtime


@pytest.mark.parametrize("fmt", ["row", "row_flat", "column"])
def test_oracle_text_compare_matches_python(fmt):
    buf, txt, chr10, num, tnull = text_cases.text_table(3000, 5, fmt)
    for qual, fn, ext in text_cases.CASES:
        rc, rows = oracle.gpuscan(qual, buf, ext)
        assert rc == 0, qual
        want = text_cases.expected_rows(fn, txt, chr10, num, tnull)
        assert np.array_equal(np.sort(rows) - 1, want), qual


def test_oracle_unreadable_varlena_is_rechecked():
    """a compressed datum (4-byte header, bit 1 set) or an external TOAST pointer cannot be
    read in place: the row goes back to the CPU (opencl_common.h:1142-1151)"""
    plain = kds.varlena_datum(b"abc")
    compressed = np.array([(20 << 2) | 2], dtype="<u4").tobytes() + b"\0" * 16
    external = bytes([0x01, 18]) + b"\0" * 16
    buf = kds.build_kds("row", [kds.Column("text_raw", [plain, compressed, external, plain])])
    rc, rows = oracle.gpuscan("(texteq (var 1 text) (const text 'abc'))", buf)
    assert rc == 0 and sorted(rows.tolist()) == [-3, -2, 1, 4]


def test_column_chunk_layout_of_a_text_column():
    """include/strom_kds.h: a varlena column of a COLUMN chunk is an array of 8-byte offsets
    (from the chunk head; 0 = NULL) to complete datums in the chunk's heap area -- 4-byte headers on
    4-byte boundaries, short ones packed; strom_kds_to_column moves the datums of heap tuples there
    verbatim; the generated program lists its text variables for strom_kvars_from_column"""
    words = [b"", b"a", b"x" * 126, b"x" * 127, b"hello world", b"y" * 1000, b"ab"]
    nul = np.array([0, 0, 0, 0, 1, 0, 0], dtype=bool)
    cols = [kds.Column("int4", np.arange(7, dtype=np.int32)), kds.Column("text", words, nul)]
    buf = kds.build_kds("column", cols)
    head = kds.KdsHead(buf)
    assert head.format == 4 and head.length == len(buf) and head.usage <= head.length
    cd = kds.decode_column_chunk(buf)[1]
    offs = cd["values"].view(np.uint64)
    assert len(offs) == 7 and offs[4] == 0 and not cd["notnull"][4]
    seen = []
    for r, w in enumerate(words):
        if nul[r]:
            continue
        o = int(offs[r])
        assert cd["extra_off"] <= o < head.usage
        if len(w) + 1 <= 127:
            assert int(buf[o]) == ((len(w) + 1) << 1) | 1 and buf[o + 1:o + 1 + len(w)].tobytes() == w
            seen.append((o, o + 1 + len(w)))
        else:
            assert o % 4 == 0 and int(buf[o:o + 4].view(np.uint32)[0]) == (len(w) + 4) << 2
            assert buf[o + 4:o + 4 + len(w)].tobytes() == w
            seen.append((o, o + 4 + len(w)))
    seen.sort()
    assert all(a[1] <= b[0] for a, b in zip(seen, seen[1:]))             # no datum overlaps another
    # heap tuples -> COLUMN on the host: the same rows come out
    row = kds.build_kds("row", cols)
    col = kds.kds_to_column(row)
    for q in ("(texteq (var 2 text) (const text 'hello world'))", "(text_gt (var 2 text) (const text 'a'))",
              "(isnull (var 2 text))"):
        assert oracle.gpuscan(q, buf)[1].tolist() == oracle.gpuscan(q, row)[1].tolist() == oracle.gpuscan(q, col)[1].tolist()
    cg = runtime.codegen_gpuscan("(and (texteq (var 2 text) (const text 'a')) (int4gt (var 1 int4) (const int4 0)))")
    assert "#define STROM_KVARLENA_LIST(X) X(2,text)\n" in cg.source
    cg = runtime.codegen_gpuscan("(int4gt (var 1 int4) (const int4 0))")
    assert "STROM_KVARLENA_LIST" not in cg.source
    # a TUPSLOT chunk still cannot hold the datums
    with pytest.raises(ValueError):
        kds.build_kds("tupslot", cols)


def test_codegen_catalogue_and_parambuf():
    cg = runtime.codegen_gpuscan("(and (texteq (var 2 text) (const text 'hello world'))"
                                 " (texteq (var 2 text) (const text 'hello world'))"
                                 " (bpcharlt (var 3 character) (param 1 character)))")
    assert '#include "strom_textlib.h"' in cg.source
    assert "pgfn_texteq" in cg.source and "pgfn_bpcharlt" in cg.source
    assert cg.extra_flags & 0x0010                      # DEVFUNC_NEEDS_TEXTLIB
    assert len(cg.params) == 2                          # equal literals share one KPARAM
    pb = np.frombuffer(cg.parambuf([None, b"x" * 200]), dtype=np.uint8)
    length, nparams = pb[:8].view(np.uint32)
    assert nparams == 2 and length == len(pb)
    off = pb[8:16].view(np.uint32)
    # constant: 4-byte header + 11 bytes
    assert int(pb[off[0]:off[0] + 4].view(np.uint32)[0]) >> 2 == 15
    assert pb[off[0] + 4:off[0] + 15].tobytes() == b"hello world"
    # parameter: the caller's datum verbatim (4-byte header, 200 bytes)
    assert int(pb[off[1]:off[1] + 4].view(np.uint32)[0]) >> 2 == 204
    assert pb[off[1] + 4:off[1] + 204].tobytes() == b"x" * 200
    # NULL parameter
    pb = np.frombuffer(cg.parambuf([None, None]), dtype=np.uint8)
    assert pb[8:16].view(np.uint32)[1] == 0


def test_codegen_refuses_text_where_the_device_cannot_hold_it():
    from pg_strom_amd.gpupreagg import codegen_gpupreagg
    from pg_strom_amd.gpuhashjoin import codegen_gpuhashjoin
    with pytest.raises(ValueError):
        codegen_gpupreagg("(gpupreagg (key (var 1 text)) (nrows))")
    with pytest.raises(ValueError):
        codegen_gpupreagg("(gpupreagg (key (var 1 int4)) (pmax (var 2 text)))")
    with pytest.raises(ValueError):
        runtime.codegen_gpuscan("(texteq (var 1 text) (var 2 character))")
    # a qual over text inside an aggregate is fine
    cg = codegen_gpupreagg("(gpupreagg (qual (texteq (var 2 text) (const text 'MAIL'))) (key (var 1 int4)) (nrows))")
    assert "pgfn_texteq" in cg.source


def test_multihash_entries_carry_varlena_inner_columns():
    """the inner side of a join may have text / character(n) columns: the builder copies the
    datum into the entry's heap tuple verbatim (the reference copies whole inner tuples,
    gpuhashjoin.c:3717-3805)"""
    from pg_strom_amd.gpuhashjoin import build_multihash
    words = [text_cases.WORDS[i % len(text_cases.WORDS)] for i in range(300)]
    pk = np.arange(300, dtype=np.int32)
    inner = kds.build_kds("row", [kds.Column("int4", pk), kds.Column("text", words, np.arange(300) % 11 == 5),
                                  kds.Column("int2", (pk % 5).astype(np.int16))])
    km = build_multihash([(inner, [1])])
    assert oracle.check_hashtable(km, 1, inner, [1], [4]) == 300
    outer = kds.build_kds("column", [kds.Column("int4", np.arange(-20, 400, dtype=np.int32))])
    spec = ("(gpuhashjoin (rel (hashkey (var 1 int4) 1 int4)"
            " (qual (and (text_ge (ivar 1 2 text) (const text 'a')) (int2ne (ivar 1 3 int2) (const int2 0))))))")
    rc, n, recs = oracle.gpuhashjoin(spec, outer, [inner])
    want = [i for i in range(300) if i % 11 != 5 and words[i] >= b"a" and i % 5 != 0]
    assert rc == 0 and n == len(want) and sorted(recs[:, 1].tolist()) == want


def test_text_and_character_hash_keys_builder_and_oracle():
    """join ON outer.text = inner.text (and character(n), and text + int together): the builder
    hashes the datum's PAYLOAD (COMP_CRC32 over VARDATA_ANY / VARSIZE_ANY_EXHDR, gpuhashjoin.c:
    3775-3779; device side opencl_hashjoin.h:935-953), the generated probe compares with texteq /
    bpchareq.  Python's own bytes equality is the answer; NULL never joins; character(n) ignores
    trailing blanks on both sides; 'hello' <> 'hello ' as text."""
    from pg_strom_amd.gpuhashjoin import build_multihash
    from pg_strom_amd.gpuhashjoin import codegen_gpuhashjoin
    W = text_cases.WORDS
    nd = len(W) * 2
    words = [W[i % len(W)] for i in range(nd)]                       # every word twice: duplicates
    wnull = np.arange(nd) % 9 == 4
    chr8 = [(w[:8] + b" " * 8)[:8] for w in words]
    inner = kds.build_kds("row", [kds.Column("text", words, wnull), kds.Column("character", chr8),
                                  kds.Column("int4", (np.arange(nd) % 3).astype(np.int32))])
    km = build_multihash([(inner, [1])])
    assert oracle.check_hashtable(km, 1, inner, [1], [-1]) == nd
    # the hash of an entry is the reference's: pg_crc32 over the payload bytes alone
    assert oracle.pg_crc32(b"hello") != oracle.pg_crc32(b"hello ")
    km2 = build_multihash([(inner, [2, 3])])
    assert oracle.check_hashtable(km2, 1, inner, [2, 3], [-1, 4]) == nd

    n = 500
    rng = np.random.default_rng(4)
    otxt = [W[i] for i in rng.integers(0, len(W), n)]
    otxt[7], otxt[8] = b"no such word", b"hello   "
    onull = np.arange(n) % 13 == 2
    ochr = [(w[:12] + b" " * 12)[:12] for w in otxt]                 # character(12) against character(8)
    onum = rng.integers(0, 4, n).astype(np.int32)
    outer = kds.build_kds("row", [kds.Column("text", otxt, onull), kds.Column("character", ochr),
                                  kds.Column("int4", onum)])
    # text = text
    rc, cnt, recs = oracle.gpuhashjoin("(gpuhashjoin (rel (hashkey (var 1 text) 1 text)))", outer, [inner])
    want = sorted((o + 1, d) for o in range(n) if not onull[o]
                  for d in range(nd) if not wnull[d] and words[d] == otxt[o])
    assert rc == 0 and cnt == len(want) and sorted(map(tuple, recs.tolist())) == want
    # character(n) = character(m), and an int4 key beside it
    spec2 = "(gpuhashjoin (rel (hashkey (var 2 character) 2 character) (hashkey (var 3 int4) 3 int4)))"
    rc, cnt, recs = oracle.gpuhashjoin(spec2, outer, [inner])
    want = sorted((o + 1, d) for o in range(n) for d in range(nd)
                  if chr8[d].rstrip(b" ") == ochr[o].rstrip(b" ") and int(onum[o]) == d % 3)
    assert rc == 0 and cnt == len(want) and cnt > 0 and sorted(map(tuple, recs.tolist())) == want
    # the generated program: images are hashes of the bytes, so no KEYED index, and the probe
    # compares by the type's equality function
    cg = codegen_gpuhashjoin("(gpuhashjoin (rel (hashkey (var 1 text) 1 text)))")
    assert "#define HASHJOIN_KEYED_OK_1 0" in cg.source and "pgfn_texteq" in cg.source
    assert "hashjoin_varlena_image(okey_1_0.value, false)" in cg.source
    cg = codegen_gpuhashjoin(spec2)
    assert "pgfn_bpchareq" in cg.source and "hashjoin_varlena_image(okey_1_0.value, true)" in cg.source
    cg = codegen_gpuhashjoin("(gpuhashjoin (rel (hashkey (var 3 int4) 3 int4)))")
    assert "#define HASHJOIN_KEYED_OK_1 1" in cg.source
    # types must agree (text against character(n) needs a cast the catalog does not have)
    with pytest.raises(ValueError):
        codegen_gpuhashjoin("(gpuhashjoin (rel (hashkey (var 1 text) 2 character)))")


def test_multihash_builder_sends_a_compressed_key_datum_to_the_cpu():
    """a key datum the device cannot read in place (compressed 4-byte header, external pointer)
    cannot be compared there: the builder answers CpuReCheck for the relation"""
    from pg_strom_amd.gpuhashjoin import build_multihash
    comp = np.frombuffer(np.uint32((12 << 2) | 2).tobytes() + b"\0" * 8, dtype=np.uint8).tobytes()
    inner = kds.build_kds("row", [kds.Column("text_raw", [comp, comp]), kds.Column("int4", np.arange(2, dtype=np.int32))])
    assert len(build_multihash([(inner, [2])])) > 0                  # as a payload column: fine
    with pytest.raises(runtime.StromError) as ei:
        build_multihash([(inner, [1])])
    assert ei.value.errcode == 2                                      # StromError_CpuReCheck
