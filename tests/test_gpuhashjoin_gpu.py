"""
GpuHashJoin parity, HIP path vs CPU oracle (needs an MI355X: -m gpu), through
strom_hashjoin_table_create / strom_submit_gpuhashjoin.  Integer work: the
set of (outer row, inner row per relation) records must be identical; the
order of records is unspecified (as in the reference).
"""
import numpy as np
import pytest

import oracle_binding as oracle
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpuhashjoin import GpuHashJoin, build_multihash, entry_rowids, STROM_RESULTS_ON_DEVICE

pytestmark = pytest.mark.gpu

C3_SPEC = "(gpuhashjoin (rel (hashkey (var 1 int4) 1 int4)))"


def run_and_compare(spec, outer, inners, keys, ext=(), row_map=None, expect_mode=None, ratio=1.0):
    rc, n, want = oracle.gpuhashjoin(spec, outer, inners, ext, row_map=row_map)
    assert rc == 0
    km = build_multihash(list(zip(inners, keys)))
    join = GpuHashJoin(spec, row_population_ratio=ratio).begin(km, ext_params=ext)
    try:
        if expect_mode:
            assert join.table_info(1)["mode"] == expect_mode
        res = join.join_chunk(outer, row_map=row_map)
        dkm = join.device_kmhash()
        info = [join.table_info(d + 1) for d in range(len(inners))]
    finally:
        join.end()
    assert res.errcode == 0 and res.nitems == n
    got = np.empty_like(res.records)
    got[:, 0] = res.records[:, 0]
    for d in range(len(inners)):
        got[:, d + 1] = entry_rowids(dkm, d + 1, res.records[:, d + 1])
    a = got[np.lexsort(got.T[::-1])]
    b = want[np.lexsort(want.T[::-1])]
    assert np.array_equal(a, b)
    return res, info


def fact_dim(nfact, ndim, seed, dup=False, nulls=0.02, key_span=None):
    rng = np.random.default_rng(seed)
    span = key_span or int(ndim * 1.25)
    pk = rng.permutation(span)[:ndim].astype(np.int32)
    if dup:
        pk[ndim // 3:2 * (ndim // 3)] = pk[:ndim // 3]
    payload = rng.integers(0, 1000, ndim).astype(np.int32)
    pkn = (rng.random(ndim) < nulls) if nulls else None
    fk = rng.integers(0, span, nfact).astype(np.int32)
    fkn = (rng.random(nfact) < nulls) if nulls else None
    return pk, payload, pkn, fk, fkn


@pytest.mark.parametrize("ofmt", ["column", "row", "row_flat", "tupslot"])
@pytest.mark.parametrize("dup", [False, True])
def test_single_key_all_outer_formats(ofmt, dup):
    pk, payload, pkn, fk, fkn = fact_dim(30011, 2000, 5, dup=dup)
    inner = kds.build_kds("row", [kds.Column("int4", pk, pkn), kds.Column("int4", payload)])
    outer = kds.build_kds(ofmt, [kds.Column("int4", fk, fkn), kds.Column("float8", np.zeros(len(fk)))])
    res, info = run_and_compare(C3_SPEC, outer, [inner], [[1]], expect_mode="direct",
                                ratio=2.0 if dup else 1.0)
    assert info[0]["unique"] == (not dup)


def test_sparse_keys_use_the_keyed_index(monkeypatch):
    """one key, sparse: 16-byte slots that carry the key image (KEYED); with the switch off the
    chained HASH index over the same entries must give the same pairs"""
    rng = np.random.default_rng(11)
    pk = rng.integers(-2**31, 2**31, 3000, dtype=np.int64).astype(np.int32)
    fk = np.concatenate([pk[rng.integers(0, 3000, 20000)],
                         rng.integers(-2**31, 2**31, 20000, dtype=np.int64).astype(np.int32)])
    inner = kds.build_kds("row_flat", [kds.Column("int4", pk), kds.Column("int4", np.arange(3000, dtype=np.int32))])
    for ofmt in ("column", "row"):
        outer = kds.build_kds(ofmt, [kds.Column("int4", fk)])
        run_and_compare(C3_SPEC, outer, [inner], [[1]], expect_mode="keyed", ratio=1.2)
    monkeypatch.setenv("STROM_HASHJOIN_NO_KEYED", "1")
    run_and_compare(C3_SPEC, outer, [inner], [[1]], expect_mode="hash", ratio=1.2)


def test_result_overflow_is_retried_with_exact_room():
    pk, payload, pkn, fk, fkn = fact_dim(20000, 50, 21, dup=True, nulls=0)
    inner = kds.build_kds("row", [kds.Column("int4", pk), kds.Column("int4", payload)])
    outer = kds.build_kds("column", [kds.Column("int4", fk)])
    rc, n, want = oracle.gpuhashjoin(C3_SPEC, outer, [inner])
    km = build_multihash([(inner, [1])])
    join = GpuHashJoin(C3_SPEC).begin(km)
    first = join.collect(join.submit(outer, nrooms=100))
    assert first.errcode == 301 and first.nitems == n          # DataStoreNoSpace + room needed
    res = join.join_chunk(outer, nrooms=100)
    join.end()
    assert res.errcode == 0 and res.nitems == n and getattr(res, "retried", False)


def test_two_relations_multi_key_quals_and_row_map():
    rng = np.random.default_rng(31)
    n1, n2, nf = 3000, 500, 40000
    a = rng.integers(0, 60, n1).astype(np.int32)
    b = rng.integers(0, 5, n1).astype(np.int64)
    c = rng.random(n1)
    link = rng.integers(0, 700, n1).astype(np.int32)
    t1 = kds.build_kds("row", [kds.Column("int4", a), kds.Column("int8", b), kds.Column("float8", c),
                               kds.Column("int4", link, rng.random(n1) < 0.05)])
    pk2 = rng.permutation(700)[:n2].astype(np.int32)
    t2 = kds.build_kds("row_flat", [kds.Column("int4", pk2), kds.Column("int2", rng.integers(0, 9, n2).astype(np.int16))])
    fa = rng.integers(0, 70, nf).astype(np.int32)
    fb = rng.integers(0, 6, nf).astype(np.int16)
    fc = rng.random(nf)
    spec = ("(gpuhashjoin (rel (hashkey (var 1 int4) 1 int4) (hashkey (int8 (var 2 int2)) 2 int8)"
            " (qual (float8gt (ivar 1 3 float8) (var 3 float8))))"
            " (rel (hashkey (ivar 1 4 int4) 1 int4) (qual (int2lt (ivar 2 2 int2) (param 0 int2)))))")
    for ofmt in ("column", "row"):
        outer = kds.build_kds(ofmt, [kds.Column("int4", fa, rng.random(nf) < 0.02), kds.Column("int2", fb),
                                     kds.Column("float8", fc)])
        run_and_compare(spec, outer, [t1, t2], [[1, 2], [1]], ext=[np.int16(6)], ratio=30.0)
        rmap = rng.permutation(nf)[:7000].astype(np.int32)
        run_and_compare(spec, outer, [t1, t2], [[1, 2], [1]], ext=[np.int16(6)], row_map=rmap, ratio=30.0)


def test_float_and_int8_keys():
    rng = np.random.default_rng(41)
    k = np.round(rng.random(800) * 50, 1)
    k[5] = np.nan
    k[6] = -0.0
    inner = kds.build_kds("row", [kds.Column("float8", k), kds.Column("int4", np.arange(800, dtype=np.int32))])
    f = np.round(rng.random(20000) * 50, 1)
    f[::97] = np.nan
    f[::89] = 0.0
    outer = kds.build_kds("column", [kds.Column("float8", f)])
    run_and_compare("(gpuhashjoin (rel (hashkey (var 1 float8) 1 float8)))", outer, [inner], [[1]],
                    expect_mode="keyed", ratio=30.0)
    big = rng.integers(-2**62, 2**62, 1000)
    inner = kds.build_kds("row", [kds.Column("int8", big)])
    outer = kds.build_kds("row_flat", [kds.Column("int8", np.concatenate([big[::3], big[::7] + 1]))])
    run_and_compare("(gpuhashjoin (rel (hashkey (var 1 int8) 1 int8)))", outer, [inner], [[1]], expect_mode="keyed")


def test_projection_into_tupslot():
    """kern_gpuhashjoin_projection_slot: joined rows materialised as TUPSLOT"""
    pk, payload, pkn, fk, fkn = fact_dim(50000, 3000, 77, dup=True)
    amount = np.random.default_rng(7).random(len(fk))
    inner = kds.build_kds("row", [kds.Column("int4", pk, pkn), kds.Column("int4", payload, payload % 17 == 0)])
    km = build_multihash([(inner, [1])])
    for ofmt in ("row", "column"):
        outer = kds.build_kds(ofmt, [kds.Column("int4", fk, fkn), kds.Column("float8", amount)])
        rc, n, want = oracle.gpuhashjoin(C3_SPEC, outer, [inner])
        join = GpuHashJoin(C3_SPEC, row_population_ratio=0.2).begin(km)     # forces one retry
        nitems, cols = join.join_chunk_project(outer, [(0, 2, "float8"), (1, 2, "int4"), (0, 1, "int4"), (1, 1, "int4")])
        join.end()
        assert nitems == n
        (amt, amt_n), (pay, pay_n), (ofk, ofk_n), (ipk, ipk_n) = cols
        assert not amt_n.any() and not ofk_n.any() and not ipk_n.any()
        assert np.array_equal(ofk, ipk)                                   # join keys equal
        got = sorted(zip(ofk.tolist(), amt.tolist(), [None if x else int(v) for v, x in zip(pay, pay_n)]),
                     key=lambda t: (t[0], t[1], -1 if t[2] is None else t[2]))
        exp = sorted(((int(fk[o - 1]), float(amount[o - 1]),
                       None if payload[r] % 17 == 0 else int(payload[r])) for o, r in want.tolist()),
                     key=lambda t: (t[0], t[1], -1 if t[2] is None else t[2]))
        assert got == exp


def test_c3_shape_properties_at_1e8():
    """BASELINE configs[2] shape: 1e8 fact x 1e6 dim on int4, 80% hit.
    Properties instead of the tuple-at-a-time oracle: every record's keys are
    equal, outer rows are unique (dim keys are unique) and the count equals an
    independent numpy membership count."""
    nf, nd = 100_000_000, 1_000_000
    rng = np.random.default_rng(2025)
    pk = rng.permutation(nd).astype(np.int32)
    payload = rng.integers(0, 2**31, nd, dtype=np.int64).astype(np.int32)
    inner = kds.build_kds("row_flat", [kds.Column("int4", pk), kds.Column("int4", payload)])
    km = build_multihash([(inner, [1])])
    fk = rng.integers(0, int(nd * 1.25), nf, dtype=np.int64).astype(np.int32)
    outer = runtime.DeviceStore.upload(kds.build_kds("column", [kds.Column("int4", fk)]))
    join = GpuHashJoin(C3_SPEC).begin(km)
    info = join.table_info(1)
    assert info["mode"] == "direct" and info["unique"] and info["nentries"] == nd
    res = join.join_chunk(outer)
    dkm = join.device_kmhash()
    join.end()
    outer.release()
    assert res.errcode == 0
    assert res.nitems == int(np.count_nonzero(fk < nd))
    orow = res.records[:, 0].astype(np.int64) - 1
    irow = entry_rowids(dkm, 1, res.records[:, 1])
    assert np.array_equal(fk[orow], pk[irow])
    srt = np.sort(orow)
    assert np.all(np.diff(srt) > 0)


@pytest.mark.parametrize("nfact", [1, 255, 4096, 16383, 16384, 16385, 40001])
@pytest.mark.parametrize("fanout", [1, 40])
def test_general_kernel_tile_and_slice_boundaries(nfact, fanout):
    """the general kernel counts per 16k-row tile and emits per 4096-row slice
    through a 32 KB LDS stage; a slice whose matches do not fit the stage
    (fan-out 40) takes the direct path.  A row map forces the general kernel."""
    rng = np.random.default_rng(1000 + nfact + fanout)
    ndim_keys = 300
    pk = np.repeat(np.arange(ndim_keys, dtype=np.int32), fanout)
    rng.shuffle(pk)
    inner = kds.build_kds("row_flat", [kds.Column("int4", pk),
                                       kds.Column("int4", np.arange(len(pk), dtype=np.int32))])
    fk = rng.integers(0, int(ndim_keys * 1.3), nfact).astype(np.int32)
    fkn = rng.random(nfact) < 0.01
    outer = kds.build_kds("column", [kds.Column("int4", fk, fkn), kds.Column("float8", np.zeros(nfact))])
    row_map = np.arange(nfact, dtype=np.int32)[::-1].copy()       # every row, reversed
    res, info = run_and_compare(C3_SPEC, outer, [inner], [[1]], row_map=row_map, ratio=float(fanout))
    assert info[0]["unique"] == (fanout == 1)
    want = int(np.count_nonzero((fk < ndim_keys) & ~fkn)) * fanout
    assert res.nitems == want


@pytest.mark.parametrize("ofmt", ["row", "row_flat"])
def test_numeric_key_from_varlena_outer(ofmt):
    """outer key: PostgreSQL's varlena numeric inside heap tuples (decoded and
    normalised per row); inner key: the 8-byte form -- the index hashes images"""
    from decimal import Decimal
    rng = np.random.default_rng(77)
    nd, nf = 500, 20011
    dim_vals = [Decimal(int(x)).scaleb(-2) for x in rng.permutation(5000)[:nd]]
    dim_img = np.array([kds.numeric_encode(v) for v in dim_vals], dtype=np.uint64)
    inner = kds.build_kds("row", [kds.Column("numeric", dim_img), kds.Column("int4", np.arange(nd, dtype=np.int32))])
    # the same values written with trailing zeros / other scales on the outer side
    pick = rng.integers(0, 5000, nf)
    fact_img = np.array([kds.numeric_encode(Decimal(int(x)).scaleb(-2)) for x in pick], dtype=np.uint64)
    fn = rng.random(nf) < 0.02
    outer = kds.build_kds(ofmt, [kds.Column("numeric_varlena", fact_img, fn),
                                 kds.Column("int4", np.arange(nf, dtype=np.int32))])
    spec = "(gpuhashjoin (rel (hashkey (var 1 numeric) 1 numeric)))"
    res, info = run_and_compare(spec, outer, [inner], [[1]])
    present = set(int(round(float(v) * 100)) for v in dim_vals)
    assert res.nitems == sum(1 for x, isn in zip(pick, fn) if not isn and int(x) in present)


@pytest.mark.parametrize("ofmt", ["column", "row"])
@pytest.mark.parametrize("dup", [False, True])
def test_outer_only_qual_pulled_up_into_the_join(ofmt, dup):
    """a scan's WHERE pulled up into the join (gpuhashjoin.c:2047-2050): with a
    COLUMN outer chunk and unique keys this runs in the one-pass kernel, which
    evaluates the qual for rows that found their entry, like the general kernel
    (other formats / duplicate keys); NULL quals reject"""
    pk, payload, pkn, fk, fkn = fact_dim(50021, 3000, 19, dup=dup)
    rng = np.random.default_rng(23)
    a = rng.integers(0, 2**31, len(fk), dtype=np.int64).astype(np.int32)
    an = rng.random(len(fk)) < 0.03
    b = rng.random(len(fk))
    inner = kds.build_kds("row", [kds.Column("int4", pk, pkn), kds.Column("int4", payload)])
    outer = kds.build_kds(ofmt, [kds.Column("int4", fk, fkn), kds.Column("int4", a, an), kds.Column("float8", b)])
    spec = ("(gpuhashjoin (rel (hashkey (var 1 int4) 1 int4)"
            " (qual (and (int4lt (var 2 int4) (param 0 int4)) (float8gt (var 3 float8) (param 1 float8))))))")
    res, info = run_and_compare(spec, outer, [inner], [[1]], ext=[np.int32(2**30), 0.25],
                                expect_mode="direct", ratio=2.0 if dup else 1.0)
    assert 0 < res.nitems < len(fk) // 2


def test_pulled_up_qual_errors_only_count_for_rows_that_match():
    """int4 overflow in the qual: CpuReCheck if a MATCHING row hits it, nothing if
    only rows without a partner do (the qual is not evaluated for those)"""
    pk = np.arange(100, dtype=np.int32)
    inner = kds.build_kds("row", [kds.Column("int4", pk), kds.Column("int4", pk)])
    spec = ("(gpuhashjoin (rel (hashkey (var 1 int4) 1 int4)"
            " (qual (int4gt (int4pl (var 2 int4) (const int4 1)) (const int4 0)))))")
    km = build_multihash([(inner, [1])])
    for bad_matches, want_rc in ((False, 0), (True, 2)):
        fk = np.array([5, 500, 7, 9] * 50, dtype=np.int32)
        a = np.ones(len(fk), dtype=np.int32)
        a[1 if not bad_matches else 0] = 2**31 - 1         # row 1 has no partner (fk = 500), row 0 has
        outer = kds.build_kds("column", [kds.Column("int4", fk), kds.Column("int4", a)])
        rc_o, n_o, _ = oracle.gpuhashjoin(spec, outer, [inner], [])
        assert rc_o == want_rc
        join = GpuHashJoin(spec).begin(km)
        try:
            if want_rc == 0:
                res = join.join_chunk(outer)
                assert res.errcode == 0 and res.nitems == n_o == 150
            else:
                with pytest.raises(runtime.StromError) as ei:
                    join.join_chunk(outer)
                assert ei.value.errcode == 2
        finally:
            join.end()


@pytest.mark.parametrize("nd,span,limit,expect_lds", [(5000, 6000, None, True), (20000, 25000, None, False),
                                                       (30000, 31000, "131072", True), (40000, 50000, "1000000", False)])
def test_small_dimension_is_probed_from_lds(nd, span, limit, expect_lds, monkeypatch):
    """'inner hash staged in LDS' (BASELINE configs[2]): a DIRECT slot array of up to 32 KB is
    copied into the work-group's LDS once and probed with ds_reads (gpuhashjoin_main_fast_lds;
    num_kern_prep marks it); larger ones are probed through the caches, where they measured
    faster -- STROM_HASHJOIN_LDS_SLOT_LIMIT moves the line up to what fits next to the result
    stage.  NULL keys, keys outside the table, 32-row-per-thread tiles with a ragged tail,
    tiny result room"""
    if limit:
        monkeypatch.setenv("STROM_HASHJOIN_LDS_SLOT_LIMIT", limit)
    rng = np.random.default_rng(nd)
    pk = rng.permutation(span)[:nd].astype(np.int32)
    inner = kds.build_kds("row_flat", [kds.Column("int4", pk), kds.Column("int4", np.arange(nd, dtype=np.int32))])
    n = 300011
    fk = rng.integers(-100, span + 100, n).astype(np.int32)
    outer = kds.build_kds("column", [kds.Column("int4", fk, rng.random(n) < 0.03), kds.Column("float8", rng.random(n))])
    rc, nwant, want = oracle.gpuhashjoin(C3_SPEC, outer, [inner])
    assert rc == 0
    km = build_multihash([(inner, [1])])
    join = GpuHashJoin(C3_SPEC, row_population_ratio=1.0).begin(km)
    try:
        info = join.table_info(1)
        assert info["mode"] == "direct" and info["unique"]
        first = join.collect(join.submit(outer, nrooms=1000))         # too small: the room needed comes back
        assert first.errcode == 301 and first.nitems == nwant
        res = join.join_chunk(outer)
        dkm = join.device_kmhash()
    finally:
        join.end()
    assert res.perfmon["num_kern_prep"] == (1 if expect_lds else 0)
    assert res.errcode == 0 and res.nitems == nwant
    got = np.stack([res.records[:, 0], entry_rowids(dkm, 1, res.records[:, 1])], axis=1)
    a = got[np.lexsort(got.T[::-1])]
    b = want[np.lexsort(want.T[::-1])]
    assert np.array_equal(a, b)


@pytest.mark.parametrize("narrow", [True, False])
def test_large_dimension_is_probed_through_three_byte_slots(narrow, monkeypatch):
    """a DIRECT index over 5e5 key values is 2 MB of 4-byte slots; with unique keys the fast kernel
    probes the 3-byte form instead (hashjoin_index_rel.slots3_off: entry offset >> 3, entries are
    LONGALIGNed) -- same pairs as the oracle and as the 4-byte form (STROM_HASHJOIN_NO_NARROW_SLOTS),
    including the last slot of the array, NULL keys on both sides and keys outside the range"""
    if not narrow:
        monkeypatch.setenv("STROM_HASHJOIN_NO_NARROW_SLOTS", "1")
    pk, payload, pkn, fk, fkn = fact_dim(200003, 400000, 21, nulls=0.01, key_span=500000)
    pk[0], pk[1] = 0, 499999                                     # both ends of the slot array
    pk[2:] = np.random.default_rng(3).permutation(np.arange(1, 499999))[:len(pk) - 2].astype(np.int32)
    fk[:4] = [499999, 0, 500000, -1]
    inner = kds.build_kds("row_flat", [kds.Column("int4", pk, pkn), kds.Column("int4", payload)])
    outer = kds.build_kds("column", [kds.Column("int4", fk, fkn)])
    res, info = run_and_compare(C3_SPEC, outer, [inner], [[1]], expect_mode="direct")
    assert info[0]["unique"]
