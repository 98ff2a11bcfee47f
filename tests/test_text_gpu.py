"""
text / character(n) on the device (needs an MI355X: -m gpu), through the C ABI.  Row
selection is byte work: bit-exact against the CPU oracle AND against Python's own bytes
comparison (tests/text_cases.py).  Text columns live in heap tuples (ROW / ROW_FLAT, the
formats the reference ships) and in the heap area of a COLUMN chunk (8-byte offsets in the column
array, include/strom_kds.h), which the streaming kernels read; TUPSLOT chunks are refused.
"""
import numpy as np
import pytest

import oracle_binding as oracle
import text_cases
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpuscan import GpuScan
from pg_strom_amd.gpupreagg import GpuPreAgg
from test_gpuscan_gpu import check, canon
from test_gpuhashjoin_gpu import run_and_compare

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fmt", ["row", "row_flat", "column"])
def test_text_quals_match_oracle_and_python(fmt):
    buf, txt, chr10, num, tnull = text_cases.text_table(20011, 12, fmt)
    for qual, fn, ext in text_cases.CASES:
        res = check(qual, buf, ext)
        assert res.errcode == 0, qual
        want = text_cases.expected_rows(fn, txt, chr10, num, tnull)
        assert np.array_equal(np.sort(np.asarray(res.results[:res.nitems])) - 1, want), qual


@pytest.mark.parametrize("fmt", ["row", "column"])
def test_unreadable_varlena_rows_go_back_to_the_cpu(fmt):
    plain = kds.varlena_datum(b"abc")
    compressed = np.array([(20 << 2) | 2], dtype="<u4").tobytes() + b"\0" * 16
    external = bytes([0x01, 18]) + b"\0" * 16
    datums = [plain, compressed, external, plain] * 500
    buf = kds.build_kds(fmt, [kds.Column("text_raw", datums)])
    res = check("(texteq (var 1 text) (const text 'abc'))", buf)
    assert len(res.passed_rows()) == 1000 and len(res.recheck_rows()) == 1000


def test_chunks_without_the_datums_are_refused_for_text_programs():
    """a TUPSLOT chunk holds by-value datums: refused on the host.  A COLUMN chunk whose column is
    by-value where the program reads text: the chunk lies about itself (DataStoreCorruption, as the
    row formats answer for such a column) -- a cl_ulong is never followed as an address"""
    a = np.arange(1000, dtype=np.int64) * 8
    scan = GpuScan("(texteq (var 1 text) (const text 'abc'))").begin()
    try:
        buf = kds.build_kds("tupslot", [kds.Column("int8", a)])
        with pytest.raises(runtime.StromError) as ei:
            scan.scan_chunk(buf)
        assert ei.value.errcode == 101
        # a datum whose header claims more bytes than the chunk has: never read by that length
        liar = kds.build_kds("column", [kds.Column("text", [b"abc"] * 999 + [b"x" * 200])])
        at = int(kds.decode_column_chunk(liar)[0]["values"].view(np.uint64)[999])
        assert int(liar[at:at + 4].view(np.uint32)[0]) == 204 << 2
        liar[at:at + 4] = np.array([(1 << 29) << 2], dtype="<u4").view(np.uint8)
        with pytest.raises(runtime.StromError) as ei:
            scan.scan_chunk(liar)
        assert ei.value.errcode == 300
        for row_map in (None, np.arange(0, 1000, 3, dtype=np.int32)):
            buf = kds.build_kds("column", [kds.Column("int8", a)])
            with pytest.raises(runtime.StromError) as ei:
                scan.scan_chunk(buf, row_map=row_map)
            assert ei.value.errcode == 300                         # StromError_DataStoreCorruption
    finally:
        scan.end()


def test_text_column_through_row_maps_ingest_and_resident_chunks():
    """the same text table as heap pages, as a COLUMN chunk built on the host, as the device's own
    ROW -> COLUMN ingest of the heap pages (datums moved to the heap area by the wave allocator),
    and behind a row map: the same rows pass"""
    runtime.init()
    n = 50021
    buf, txt, chr10, num, tnull = text_cases.text_table(n, 77, "row")
    col = kds.kds_to_column(buf)
    quals = [text_cases.CASES[i] for i in (0, 3, 5, 7, 8)]
    rm = np.sort(np.random.default_rng(2).choice(n, n // 3, replace=False)).astype(np.int32)
    src = runtime.DeviceStore.upload(buf)
    dev, _ = src.to_column([kds.SQL_TYPES[t][0] for t in ("int4", "text", "character", "int8")])
    try:
        back = dev.download()
        for qual, fn, ext in quals:
            want = text_cases.expected_rows(fn, txt, chr10, num, tnull)
            for chunk in (col, back):
                res = check(qual, chunk, ext)
                assert np.array_equal(np.sort(np.asarray(res.results[:res.nitems])) - 1, want), qual
            res = check(qual, col, ext, row_map=rm)
            assert np.array_equal(np.sort(np.asarray(res.results[:res.nitems])) - 1, np.intersect1d(want, rm)), qual
            scan = GpuScan(qual).begin(ext_params=ext)
            try:
                res = scan.scan_chunk(dev)
            finally:
                scan.end()
            assert res.errcode == 0
            assert np.array_equal(np.sort(np.asarray(res.results[:res.nitems])) - 1, want), qual
    finally:
        dev.release()
        src.release()


@pytest.mark.parametrize("fmt", ["row", "column"])
def test_text_qual_inside_gpupreagg_through_the_chunk_message(fmt):
    buf, txt, chr10, num, tnull = text_cases.text_table(30000, 21, fmt)
    spec = ("(gpupreagg (qual (and (text_ge (var 2 text) (const text 'a')) (bpcharne (var 3 character) (param 0 character))))"
            " (key (var 1 int4)) (nrows) (psum (var 4 int8)))")
    agg = GpuPreAgg(spec)
    status, pr = agg.collect_chunk(agg.submit_chunk(buf, ext_params=[b"MAIL"]))
    assert status == 0
    from test_gpupreagg_gpu import assert_matches_oracle
    assert_matches_oracle(spec, agg, [buf], pr, ext=[b"MAIL"])
    # and against Python
    keep = [i for i in range(30000) if not tnull[i] and txt[i] >= b"a" and text_cases.bpchar_key(chr10[i]) != b"MAIL"]
    k, _ = pr.column(0)
    c, _ = pr.column(1)
    s, _ = pr.column(2)
    got = {int(a): (int(b), int(d)) for a, b, d in zip(k, c, s)}
    want = {}
    for i in keep:
        cnt, sm = want.get(int(num[i]), (0, 0))
        want[int(num[i])] = (cnt + 1, sm + i)
    assert got == want
    # the same through sessions -- dense ids and the hashed table -- over the resident chunk
    ds = runtime.DeviceStore.upload(buf)
    try:
        for hashed in (False, True):
            sess = GpuPreAgg(spec).begin_hashed(ext_params=[b"MAIL"]) if hashed else \
                GpuPreAgg(spec).begin([(-50, 100)], ext_params=[b"MAIL"])
            try:
                assert sess.fold(ds)[0] == 0
                assert_matches_oracle(spec, sess, [buf], sess.fetch(), ext=[b"MAIL"])
            finally:
                sess.end()
    finally:
        ds.release()


@pytest.mark.parametrize("ngroups", [0, 4, 2000])
def test_text_qual_in_every_streaming_aggregate_kernel(ngroups):
    """a COLUMN chunk with a text column through the kernel families the dense path picks by group
    count: no GROUP BY (register accumulators), a handful of groups (lane-private accumulators),
    many (LDS atomics, packed accumulators) -- each turns the text column's offsets into addresses
    per row (strom_kvars_from_column) before the qual reads them; an unreadable datum in a row the
    qual has to look at sends the chunk back (CpuReCheck), as it does from heap tuples"""
    n = 200000
    rng = np.random.default_rng(5)
    W = text_cases.WORDS
    txt = [W[i] for i in rng.integers(0, len(W), n)]
    tnull = rng.random(n) < 0.05
    g = rng.integers(0, max(ngroups, 1), n).astype(np.int32)
    x = rng.integers(-1000, 1000, n).astype(np.int32)
    cols = [kds.Column("int4", g), kds.Column("text", txt, tnull), kds.Column("int4", x)]
    buf = kds.build_kds("column", cols)
    key = "(key (var 1 int4)) " if ngroups else ""
    spec = ("(gpupreagg (qual (or (text_lt (var 2 text) (const text 'b')) (texteq (var 2 text) (const text 'zebra')))) "
            + key + "(nrows) (psum (int8 (var 3 int4))) (pmax (var 3 int4)))")
    from test_gpupreagg_gpu import assert_matches_oracle
    agg = GpuPreAgg(spec).begin([(0, ngroups)] if ngroups else [])
    ds = runtime.DeviceStore.upload(buf)
    try:
        assert agg.fold(ds)[0] == 0
        pr = agg.fetch()
        assert_matches_oracle(spec, agg, [buf], pr)
        keep = [i for i in range(n) if not tnull[i] and (txt[i] < b"b" or txt[i] == b"zebra")]
        assert int(pr.column(1 if ngroups else 0)[0].sum()) == len(keep)
        # a compressed datum: the chunk is the CPU's
        raw = [kds.varlena_datum(t) for t in txt[:1000]]
        raw[500] = np.array([(20 << 2) | 2], dtype="<u4").tobytes() + b"\0" * 16
        bad = kds.build_kds("column", [kds.Column("int4", g[:1000]), kds.Column("text_raw", raw), kds.Column("int4", x[:1000])])
        agg.reset()
        assert agg.fold(bad)[0] == 2
    finally:
        ds.release()
        agg.end()


@pytest.mark.parametrize("fmt", ["row_flat", "column"])
def test_text_residual_qual_in_a_join(fmt):
    buf, txt, chr10, num, tnull = text_cases.text_table(20000, 33, fmt)
    pk = np.arange(-50, 50, dtype=np.int32)
    inner = kds.build_kds("row", [kds.Column("int4", pk), kds.Column("int4", pk * 3)])
    spec = ("(gpuhashjoin (rel (hashkey (var 1 int4) 1 int4)"
            " (qual (or (bpchareq (var 3 character) (const character 'SHIP')) (text_lt (var 2 text) (const text 'B'))))))")
    res, _ = run_and_compare(spec, buf, [inner], [[1]])
    want = sum(1 for i in range(20000)
               if text_cases.bpchar_key(chr10[i]) == b"SHIP" or (not tnull[i] and txt[i] < b"B"))
    assert res.nitems == want


def _walk_row_flat(dest, nitems, ncols):
    """every datum of a ROW_FLAT chunk through the ORACLE's tuple walker (independent of the
    device code that built the tuples): [(bytes or None) per column] per row"""
    import ctypes
    lib = oracle.load()
    lib.oracle_get_datum.restype = ctypes.c_void_p
    lib.oracle_get_datum.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32]
    attlen = [int(dest[48 + 8 * c + 2:48 + 8 * c + 4].view(np.int16)[0]) for c in range(ncols)]
    base = dest.ctypes.data
    rows = []
    for r in range(nitems):
        row = []
        for c in range(ncols):
            addr = lib.oracle_get_datum(base, c, r)
            if not addr:
                row.append(None)
                continue
            off = addr - base
            assert 0 <= off < len(dest)
            if attlen[c] > 0:
                row.append(dest[off:off + attlen[c]].tobytes())
            else:
                b0 = int(dest[off])
                if b0 & 1:
                    n = (b0 >> 1) & 0x7f
                    row.append(dest[off + 1:off + n].tobytes())
                else:
                    n = int(dest[off:off + 4].view(np.uint32)[0]) >> 2
                    row.append(dest[off + 4:off + n].tobytes())
        rows.append(row)
    return rows


@pytest.mark.parametrize("ofmt", ["row", "row_flat"])
def test_join_projection_into_heap_tuples(ofmt):
    """kern_gpuhashjoin_projection_row: joined rows as ROW_FLAT heap tuples, NULLs, short and
    long text datums, columns of both sides -- every datum checked against the source rows"""
    from pg_strom_amd.gpuhashjoin import GpuHashJoin, build_multihash, entry_rowids
    n = 6000
    buf, txt, chr10, num, tnull = text_cases.text_table(n, 44, ofmt)
    rng = np.random.default_rng(4)
    pk = np.arange(-50, 50, dtype=np.int32)
    pay = rng.integers(0, 1000, 100).astype(np.int32)
    payn = rng.random(100) < 0.2
    small = rng.integers(-9, 9, 100).astype(np.int16)
    inner = kds.build_kds("row", [kds.Column("int4", pk), kds.Column("int4", pay, payn), kds.Column("int2", small)])
    spec = "(gpuhashjoin (rel (hashkey (var 1 int4) 1 int4)))"
    km = build_multihash([(inner, [1])])
    join = GpuHashJoin(spec).begin(km)
    try:
        dest_columns = [(0, 2, "text"), (1, 2, "int4"), (0, 1, "int4"), (1, 3, "int2"), (0, 4, "int8"),
                        (0, 3, "character")]
        # a deliberately small store first: the tuples do not fit -> DataStoreNoSpace -> retry
        nitems, dest, recs = join.join_chunk_project_rows(buf, dest_columns, data_bytes=4096)
        dkm = join.device_kmhash()
    finally:
        join.end()
    assert nitems == n                          # every fact key has its dimension row
    inner_row = entry_rowids(dkm, 1, recs[:, 1])
    rows = _walk_row_flat(dest, nitems, len(dest_columns))
    for i in range(nitems):
        o = int(recs[i, 0]) - 1
        d = int(inner_row[i])
        want = [None if tnull[o] else txt[o],
                None if payn[d] else pay[d].tobytes(),
                num[o].tobytes(), small[d].tobytes(), np.int64(o).tobytes(), chr10[o]]
        assert rows[i] == want, (i, o, d)
        assert int(pk[d]) == int(num[o])
    # head bookkeeping: nitems, usage inside the buffer
    u32 = dest[:48].view(np.uint32)
    assert u32[5] == nitems and 0 < u32[3] < len(dest)


def test_inner_text_columns_in_join_quals_and_in_projected_tuples():
    """a dimension with a text column: the multihash entries carry the datum, a residual qual
    compares it on the device, and the ROW_FLAT projection copies it into the joined tuples"""
    from pg_strom_amd.gpuhashjoin import GpuHashJoin, build_multihash, entry_rowids
    nd, n = 400, 20000
    rng = np.random.default_rng(9)
    words = [text_cases.WORDS[i % len(text_cases.WORDS)] for i in range(nd)]
    wnull = np.arange(nd) % 13 == 4
    pk = rng.permutation(nd).astype(np.int32)
    inner = kds.build_kds("row_flat", [kds.Column("int4", pk), kds.Column("text", words, wnull),
                                       kds.Column("int2", (pk % 5).astype(np.int16))])
    fk = rng.integers(-10, nd + 50, n).astype(np.int32)
    outer = kds.build_kds("column", [kds.Column("int4", fk), kds.Column("int8", np.arange(n, dtype=np.int64))])
    spec = ("(gpuhashjoin (rel (hashkey (var 1 int4) 1 int4)"
            " (qual (or (text_lt (ivar 1 2 text) (const text 'b')) (isnull (ivar 1 2 text))))))")
    res, _ = run_and_compare(spec, outer, [inner], [[1]])          # HIP == oracle
    pos = np.full(nd + 60, -1, dtype=np.int64)
    pos[pk] = np.arange(nd)
    di = np.where((fk >= 0) & (fk < nd), pos[np.clip(fk, 0, nd + 59)], -1)
    want = sum(1 for d in di if d >= 0 and (wnull[d] or words[d] < b"b"))
    assert res.nitems == want
    # and the joined rows as heap tuples: outer int8, inner text, inner int2
    join = GpuHashJoin(spec).begin(build_multihash([(inner, [1])]))
    try:
        nitems, dest, recs = join.join_chunk_project_rows(outer, [(0, 2, "int8"), (1, 2, "text"), (1, 3, "int2")])
        dkm = join.device_kmhash()
    finally:
        join.end()
    assert nitems == want
    rows = _walk_row_flat(dest, nitems, 3)
    irow = entry_rowids(dkm, 1, recs[:, 1])
    for i in range(nitems):
        o, d = int(recs[i, 0]) - 1, int(irow[i])
        assert rows[i] == [np.int64(o).tobytes(), None if wnull[d] else words[d], np.int16(pk[d] % 5).tobytes()], i


@pytest.mark.parametrize("ofmt", ["row", "row_flat", "column"])
def test_text_and_character_hash_keys(ofmt):
    """join ON outer.text = inner.text, and ON (character(n), int4): the probe index hashes the
    payload bytes (opencl_hashjoin.h:935-953) and, a hash not being the value, the HASH form with
    texteq / bpchareq on every candidate is chosen -- never the KEYED one, which would take two
    strings with one image for equal"""
    W = text_cases.WORDS
    rng = np.random.default_rng(21)
    nd, n = 3000, 40000
    # 1000 generated strings (short and 4-byte headers), each three times; the WORDS at the front
    gen = [(b"k%05d" % i) * (1 + i % 40) for i in range(1000)]
    words = (W + gen * 3)[:nd]
    wnull = np.arange(nd) % 17 == 4
    chr8 = [(w[:8] + b" " * 8)[:8] for w in words]
    inner = kds.build_kds("column" if ofmt == "column" else "row",
                          [kds.Column("text", words, wnull), kds.Column("character", chr8),
                           kds.Column("int4", (np.arange(nd) % 3).astype(np.int32))])
    pool = W + gen + [b"no such word", b"hello   ", b"k00001k00001 "]
    otxt = [pool[i] for i in rng.integers(0, len(pool), n)]
    onull = rng.random(n) < 0.03
    ochr = [(w[:12] + b" " * 12)[:12] for w in otxt]
    outer = kds.build_kds(ofmt, [kds.Column("text", otxt, onull), kds.Column("character", ochr),
                                 kds.Column("int4", rng.integers(0, 4, n).astype(np.int32))])
    res, info = run_and_compare("(gpuhashjoin (rel (hashkey (var 1 text) 1 text)))", outer, [inner], [[1]],
                                expect_mode="hash")
    by_word = {}
    for d in range(nd):
        if not wnull[d]:
            by_word[words[d]] = by_word.get(words[d], 0) + 1
    assert res.nitems == sum(by_word.get(otxt[o], 0) for o in range(n) if not onull[o]) > n
    res, info = run_and_compare("(gpuhashjoin (rel (hashkey (var 2 character) 2 character) (hashkey (var 3 int4) 3 int4)"
                                " (qual (text_ge (var 1 text) (ivar 1 1 text)))))", outer, [inner], [[2, 3]],
                                expect_mode="hash")
    assert res.nitems > 0


def test_text_key_taken_from_an_inner_relation_in_a_two_level_join():
    """fact -> dim1 on int4, dim1.name -> dim2 on text: the second relation's key is a text column
    of the FIRST inner relation's tuple (an address inside its hash entry), its image hashed on the
    device, the candidates compared with texteq"""
    rng = np.random.default_rng(17)
    n, n1 = 30000, 500
    names = [b"region-%03d" % (i % 37) for i in range(n1)]
    nnull = np.arange(n1) % 29 == 3
    dim1 = kds.build_kds("row", [kds.Column("int4", np.arange(n1, dtype=np.int32)), kds.Column("text", names, nnull)])
    words = [b"region-%03d" % i for i in range(0, 40, 2)] + [b"region-%03d" % 4]      # even regions, one twice
    dim2 = kds.build_kds("row_flat", [kds.Column("text", words), kds.Column("int4", np.arange(len(words), dtype=np.int32))])
    fk = rng.integers(-5, n1 + 20, n).astype(np.int32)
    outer = kds.build_kds("column", [kds.Column("int4", fk), kds.Column("int8", np.arange(n, dtype=np.int64))])
    spec = "(gpuhashjoin (rel (hashkey (var 1 int4) 1 int4)) (rel (hashkey (ivar 1 2 text) 1 text)))"
    res, info = run_and_compare(spec, outer, [dim1, dim2], [[1], [1]])
    assert info[1]["mode"] == "hash"
    cnt = {}
    for w in words:
        cnt[w] = cnt.get(w, 0) + 1
    want = sum(cnt.get(names[k], 0) for k in fk if 0 <= k < n1 and not nnull[k])
    assert res.nitems == want > 0


def test_text_qual_behind_a_row_map_in_the_hashed_group_by():
    """hashed GROUP BY (float key) over a COLUMN chunk with a text qual, behind a row map: the
    one-role walk turns the text column's offsets into addresses for the mapped rows only"""
    n = 60000
    rng = np.random.default_rng(23)
    W = text_cases.WORDS
    txt = [W[i] for i in rng.integers(0, len(W), n)]
    f = rng.integers(0, 300, n).astype(np.float64) / 4
    x = rng.integers(-1000, 1000, n).astype(np.int32)
    buf = kds.build_kds("column", [kds.Column("float8", f), kds.Column("text", txt, rng.random(n) < 0.04), kds.Column("int4", x)])
    rm = np.sort(rng.choice(n, n // 2, replace=False)).astype(np.int32)
    spec = ("(gpupreagg (qual (text_ge (var 2 text) (const text 'b'))) (key (var 1 float8)) (nrows)"
            " (psum (int8 (var 3 int4))) (pmin (var 3 int4)))")
    from test_gpupreagg_gpu import assert_matches_oracle
    agg = GpuPreAgg(spec).begin_hashed()
    try:
        assert agg.fold(buf, row_map=rm)[0] == 0
        assert_matches_oracle(spec, agg, [buf], agg.fetch(), row_maps=[rm])
    finally:
        agg.end()
