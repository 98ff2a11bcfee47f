"""
Chained operators over a device-resident row map (SURVEY.md section 8 f2;
needs an MI355X: -m gpu).  GpuScan leaves the ids of the rows it selected in
HBM (strom_rowmap_from_task), GpuPreAgg / GpuHashJoin / a second GpuScan then
run over exactly those rows of the same resident chunk.  The bar: identical
to feeding the oracle the same row ids through a host kern_row_map, and to
the single operator with the scan's qual pulled in.
"""
import numpy as np
import pytest

import oracle_binding as oracle
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpuhashjoin import GpuHashJoin, build_multihash
from pg_strom_amd.gpupreagg import GpuPreAgg
from pg_strom_amd.gpuscan import GpuScan

pytestmark = pytest.mark.gpu

QUAL = "(and (int4lt (var 2 int4) (param 0 int4)) (float8gt (var 3 float8) (param 1 float8)))"


def table(n, seed):
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 40, n).astype(np.int32)
    a = rng.integers(0, 2**31, n, dtype=np.int64).astype(np.int32)
    b = rng.random(n)
    return [kds.Column("int4", g, rng.random(n) < 0.01), kds.Column("int4", a), kds.Column("float8", b)]


def test_scan_then_preagg_over_the_device_row_map():
    runtime.init()
    cols = table(300007, 21)
    buf = kds.build_kds("column", cols)
    ext = [np.int32(2**30), 0.5]
    ds = runtime.DeviceStore.upload(buf)
    scan = GpuScan(QUAL).begin(ext_params=ext)
    spec = "(gpupreagg (key (var 1 int4)) (nrows) (psum (int8 (var 2 int4))) (psum (var 3 float8)) (pmax (var 2 int4)))"
    agg = GpuPreAgg(spec).begin([(0, 40)])
    try:
        rowmap, res = scan.scan_to_rowmap(ds)
        assert rowmap.nvalids == res.nitems
        st, _ = agg.fold(ds, row_map=rowmap)
        assert st == 0
        pr = agg.fetch()
        rowmap.release()
    finally:
        agg.end()
        scan.end()
        ds.release()
    # oracle: the scan's rows as a host row map into the oracle's preagg
    rc, ids = oracle.gpuscan(QUAL, buf, ext)
    assert rc == 0 and len(ids) == res.nitems
    rows = np.sort(np.asarray(ids, dtype=np.int64) - 1).astype(np.int32)
    rc, v, isn = oracle.gpupreagg(spec, buf, 5, row_map=rows)
    assert rc == 0
    want = {(None if isn[i, 0] else int(v[i, 0].view(np.int64))): v[i] for i in range(len(v))}
    keys, knull = pr.column(0)
    assert len(keys) == len(want)
    for i in range(len(keys)):
        k = None if knull[i] else int(keys[i])
        w = want[k]
        assert int(pr.column(1)[0][i]) == int(w[1].view(np.int64))
        assert int(pr.column(2)[0][i]) == int(w[2].view(np.int64))
        assert abs(float(pr.column(3)[0][i]) - float(w[3].view(np.float64))) <= 1e-12 * abs(float(w[3].view(np.float64)))
        assert int(pr.column(4)[0][i]) == int(w[4].view(np.int64))


def test_scan_then_join_over_the_device_row_map():
    runtime.init()
    n, nd = 200003, 5000
    rng = np.random.default_rng(22)
    fk = rng.integers(0, int(nd * 1.3), n).astype(np.int32)
    a = rng.integers(0, 2**31, n, dtype=np.int64).astype(np.int32)
    b = rng.random(n)
    buf = kds.build_kds("column", [kds.Column("int4", fk), kds.Column("int4", a), kds.Column("float8", b)])
    inner = kds.build_kds("row_flat", [kds.Column("int4", rng.permutation(nd).astype(np.int32)),
                                       kds.Column("int4", np.arange(nd, dtype=np.int32))])
    km = build_multihash([(inner, [1])])
    ext = [np.int32(2**30), 0.25]
    ds = runtime.DeviceStore.upload(buf)
    scan = GpuScan(QUAL).begin(ext_params=ext)
    join = GpuHashJoin("(gpuhashjoin (rel (hashkey (var 1 int4) 1 int4)))").begin(km)
    try:
        rowmap, res = scan.scan_to_rowmap(ds)
        out = join.join_chunk(ds, row_map=rowmap)
        rowmap.release()
    finally:
        join.end()
        scan.end()
        ds.release()
    sel = (a < ext[0]) & (b > ext[1])
    assert res.nitems == int(sel.sum())
    want_outer = np.sort(np.flatnonzero(sel & (fk < nd)))
    assert out.errcode == 0 and out.nitems == len(want_outer)
    assert np.array_equal(np.sort(out.records[:, 0].astype(np.int64) - 1), want_outer)


def test_scan_over_a_device_row_map_and_recheck_fallback():
    runtime.init()
    cols = table(100003, 23)
    buf = kds.build_kds("column", cols)
    ds = runtime.DeviceStore.upload(buf)
    first = GpuScan("(int4lt (var 2 int4) (param 0 int4))").begin(ext_params=[np.int32(2**30)])
    second = GpuScan("(float8gt (var 3 float8) (param 0 float8))").begin(ext_params=[0.5])
    try:
        rowmap, r1 = first.scan_to_rowmap(ds)
        r2 = second.scan_chunk(ds, row_map=rowmap)
        rowmap.release()
    finally:
        first.end()
        second.end()
    a, b = cols[1].values, cols[2].values
    first_rows = np.flatnonzero(a < 2**30)
    assert r1.nitems == len(first_rows)
    # with a row map the reported id is the position in the map (+1): count is what can be compared
    assert r2.nitems == int(np.count_nonzero(b[first_rows] > 0.5))
    # a chunk with rows to re-check cannot be chained
    over = GpuScan("(int4gt (int4pl (var 2 int4) (const int4 2147483000)) (const int4 0))").begin()
    try:
        with pytest.raises(runtime.StromError) as ei:
            over.scan_to_rowmap(ds)
        assert ei.value.errcode == 2            # StromError_CpuReCheck
    finally:
        over.end()
        ds.release()


def test_scan_join_preagg_chain_stays_in_hbm():
    """GpuScan -> GpuHashJoin -> GpuPreAgg without a host round trip: the scan
    leaves a row map, the join leaves its joined rows as a COLUMN chunk
    (strom_hashjoin_project_column), GpuPreAgg runs its streaming kernels on it"""
    runtime.init()
    n, nd = 300007, 4000
    rng = np.random.default_rng(61)
    fk = rng.integers(0, int(nd * 1.25), n).astype(np.int32)
    a = rng.integers(0, 2**31, n, dtype=np.int64).astype(np.int32)
    an = rng.random(n) < 0.03
    b = rng.random(n)
    fact = kds.build_kds("column", [kds.Column("int4", fk), kds.Column("int4", a, an), kds.Column("float8", b)])
    dkey = rng.permutation(nd).astype(np.int32)
    dgrp = (dkey % 37).astype(np.int32)
    dval = rng.random(nd) * 10
    dvaln = rng.random(nd) < 0.05
    inner = kds.build_kds("row_flat", [kds.Column("int4", dkey), kds.Column("int4", dgrp),
                                       kds.Column("float8", dval, dvaln)])
    km = build_multihash([(inner, [1])])
    ext = [np.int32(2**30), 0.25]
    ds = runtime.DeviceStore.upload(fact)
    scan = GpuScan(QUAL).begin(ext_params=ext)
    join = GpuHashJoin("(gpuhashjoin (rel (hashkey (var 1 int4) 1 int4)))").begin(km)
    spec = ("(gpupreagg (key (var 1 int4)) (nrows) (nrows (isnotnull (var 2 int4))) (psum (int8 (var 2 int4)))"
            " (psum (var 3 float8)) (pmax (var 4 float8)))")
    agg = GpuPreAgg(spec)
    try:
        rowmap, _ = scan.scan_to_rowmap(ds)
        # a deliberately small first result buffer: the resize-and-retry path
        joined, nitems = join.join_to_column(ds, [(1, 2, "int4"), (0, 2, "int4"), (0, 3, "float8"), (1, 3, "float8")],
                                             row_map=rowmap, nrooms=1000)
        rowmap.release()
        image = joined.download()
        agg.begin([(0, 37)])
        assert agg.fold(joined)[0] == 0
        pr = agg.fetch()
    finally:
        agg.end()
        join.end()
        scan.end()
        ds.release()
    # the joined rows, as a set
    an_eff = an.copy()
    sel = np.flatnonzero((~an) & (a < ext[0]) & (b > ext[1]) & (fk < nd))
    pos = np.empty(nd, dtype=np.int64)
    pos[dkey] = np.arange(nd)
    di = pos[fk[sel]]
    assert nitems == len(sel)
    cols = kds.decode_column_chunk(image)
    assert len(cols) == 4 and all(len(c["values"]) == nitems for c in cols)
    got = np.stack([cols[0]["values"].astype(np.int64), cols[1]["values"].astype(np.int64),
                    cols[2]["values"].view(np.int64), cols[3]["values"].view(np.int64)], axis=1)
    dval_img = np.where(dvaln, 0.0, dval)
    want = np.stack([dgrp[di].astype(np.int64), a[sel].astype(np.int64), b[sel].view(np.int64),
                     dval_img[di].view(np.int64)], axis=1)
    order_g = np.lexsort(got.T[::-1])
    order_w = np.lexsort(want.T[::-1])
    assert np.array_equal(got[order_g], want[order_w])
    assert cols[0]["notnull"] is None and cols[1]["notnull"] is None      # no NULL survives the qual / comes from dgrp
    nn = cols[3]["notnull"]
    assert nn is not None and int((~nn).sum()) == int(dvaln[di].sum())
    assert cols[0]["stat_flags"] & 1 and cols[0]["minval"] == int(dgrp[di].min()) and cols[0]["maxval"] == int(dgrp[di].max())
    # and the aggregate over them
    g = dgrp[di]
    keys, _ = pr.column(0)
    order = np.argsort(keys)
    ug, inv = np.unique(g, return_inverse=True)
    assert np.array_equal(keys[order], ug)
    assert np.array_equal(pr.column(1)[0][order], np.bincount(inv))
    assert np.array_equal(pr.column(2)[0][order], np.bincount(inv))
    assert np.array_equal(pr.column(3)[0][order], np.bincount(inv, weights=a[sel].astype(np.float64)).astype(np.int64))
    wb = np.bincount(inv, weights=b[sel])
    assert np.allclose(pr.column(4)[0][order], wb, rtol=1e-12)
    wmax = np.full(len(ug), -np.inf)
    ok = ~dvaln[di]
    np.maximum.at(wmax, inv[ok], dval[di][ok])
    gmax, gnull = pr.column(5)
    assert np.array_equal(gnull[order], np.isinf(wmax))
    assert np.array_equal(gmax[order][~np.isinf(wmax)], wmax[~np.isinf(wmax)])


def test_join_to_column_without_zone_maps_and_with_the_general_kernel():
    """same joined rows whether the inner column comes from the slot-indexed array
    (DIRECT index, unique keys) or from the entries (duplicate keys: general
    kernel, no such array), with or without the min/max pass"""
    runtime.init()
    n, nd = 100003, 3000
    rng = np.random.default_rng(71)
    fk = rng.integers(0, int(nd * 1.2), n).astype(np.int32)
    a = rng.integers(-1000, 1000, n).astype(np.int32)
    fact = kds.build_kds("column", [kds.Column("int4", fk, rng.random(n) < 0.02), kds.Column("int4", a)])
    fkn = kds.decode_column_chunk(fact)[0]["notnull"]
    for dup in (False, True):
        dkey = rng.permutation(nd).astype(np.int32)
        if dup:
            dkey[:500] = dkey[500:1000]
        dpay = rng.integers(0, 10**6, nd).astype(np.int32)
        dpn = rng.random(nd) < 0.1
        inner = kds.build_kds("row", [kds.Column("int4", dkey), kds.Column("int4", dpay, dpn)])
        km = build_multihash([(inner, [1])])
        ds = runtime.DeviceStore.upload(fact)
        join = GpuHashJoin("(gpuhashjoin (rel (hashkey (var 1 int4) 1 int4)))", row_population_ratio=1.3).begin(km)
        try:
            assert join.table_info(1)["unique"] == (not dup)
            for zm in (True, False):
                joined, nitems = join.join_to_column(ds, [(0, 2, "int4"), (1, 2, "int4"), (1, 1, "int4")], zone_maps=zm)
                cols = kds.decode_column_chunk(joined.download())
                joined.release()
                # expected pairs: every (outer row, inner row) with equal keys
                order = np.argsort(dkey, kind="stable")
                sk = dkey[order]
                lo = np.searchsorted(sk, fk, "left")
                hi = np.searchsorted(sk, fk, "right")
                cnt = np.where(fkn, hi - lo, 0)
                assert nitems == int(cnt.sum())
                orow = np.repeat(np.arange(n), cnt)
                irow = (order[np.concatenate([np.arange(l, h) for l, h, c in zip(lo, hi, cnt) if c > 0])]
                        if nitems else np.zeros(0, dtype=np.int64))
                want = np.stack([a[orow].astype(np.int64), np.where(dpn[irow], 0, dpay[irow]).astype(np.int64),
                                 dkey[irow].astype(np.int64), dpn[irow].astype(np.int64)], axis=1)
                nn = cols[1]["notnull"]
                got = np.stack([cols[0]["values"].astype(np.int64), cols[1]["values"].astype(np.int64),
                                cols[2]["values"].astype(np.int64),
                                (~nn).astype(np.int64) if nn is not None else np.zeros(nitems, dtype=np.int64)], axis=1)
                assert np.array_equal(got[np.lexsort(got.T[::-1])], want[np.lexsort(want.T[::-1])])
                assert bool(cols[2]["stat_flags"] & 1) == zm
        finally:
            join.end()
            ds.release()


@pytest.mark.parametrize("ngroups", [53, 9000])
def test_preagg_straight_over_the_join_result_pairs(ngroups):
    """strom_submit_gpupreagg_joined: the projection fused into its consumer -- the
    aggregate reads (outer row, slot) pairs, outer columns from the COLUMN chunk and
    inner columns from the table's slot-indexed arrays; same partial rows as numpy
    over the joined rows, with a pulled-up qual and NULL keys / inputs; 9000 groups
    do not fit one LDS image, so id-range roles split them"""
    from pg_strom_amd.gpuhashjoin import STROM_RESULTS_ON_DEVICE
    runtime.init()
    n, nd = 250007, 20000
    rng = np.random.default_rng(83)
    fk = rng.integers(0, int(nd * 1.25), n).astype(np.int32)
    fkn = rng.random(n) < 0.02
    a = rng.integers(0, 2**31, n, dtype=np.int64).astype(np.int32)
    an = rng.random(n) < 0.03
    b = rng.random(n)
    fact = kds.build_kds("column", [kds.Column("int4", fk, fkn), kds.Column("int4", a, an), kds.Column("float8", b)])
    dkey = rng.permutation(nd).astype(np.int32)
    dgrp = (dkey % ngroups).astype(np.int32)
    dgn = rng.random(nd) < 0.04                          # NULL group keys: their own group
    dval = rng.random(nd) * 10
    dvn = rng.random(nd) < 0.05
    inner = kds.build_kds("row_flat", [kds.Column("int4", dkey), kds.Column("int4", dgrp, dgn),
                                       kds.Column("float8", dval, dvn)])
    km = build_multihash([(inner, [1])])
    ext = [np.int32(2**30), 0.25]
    ds = runtime.DeviceStore.upload(fact)
    join = GpuHashJoin("(gpuhashjoin (rel (hashkey (var 1 int4) 1 int4) (qual " + QUAL + ")))").begin(km, ext_params=ext)
    spec = ("(gpupreagg (key (var 1 int4)) (nrows) (psum (int8 (var 2 int4))) (psum (var 3 float8)) (pmax (var 4 float8)))")
    columns = [(1, 2, "int4"), (0, 2, "int4"), (0, 3, "float8"), (1, 3, "float8")]
    agg = GpuPreAgg(spec)
    try:
        agg.begin([(0, ngroups)])
        jp = join.submit(ds, flags=STROM_RESULTS_ON_DEVICE)
        ap = agg.submit_joined(join, jp, ds, columns)
        if ngroups == 53:
            assert agg.collect(ap)[0] == 0
            jr = join.collect(jp)
        else:
            # the join's device results belong to the aggregate's request now: any order
            jr = join.collect(jp)
            assert agg.collect(ap)[0] == 0
        pr = agg.fetch()
    finally:
        agg.end()
        join.end()
        ds.release()
    sel = np.flatnonzero((~fkn) & (~an) & (a < ext[0]) & (b > ext[1]) & (fk < nd))
    pos = np.empty(nd, dtype=np.int64)
    pos[dkey] = np.arange(nd)
    di = pos[fk[sel]]
    assert jr.nitems == len(sel)
    g = np.where(dgn[di], 10**6, dgrp[di])               # 10**6 stands for the NULL key
    ug, inv = np.unique(g, return_inverse=True)
    keys, knull = pr.column(0)
    gk = np.where(knull, 10**6, keys)
    order = np.argsort(gk)
    assert np.array_equal(gk[order], ug)
    assert np.array_equal(pr.column(1)[0][order], np.bincount(inv))
    assert np.array_equal(pr.column(2)[0][order], np.bincount(inv, weights=a[sel].astype(np.float64)).astype(np.int64))
    assert np.allclose(pr.column(3)[0][order], np.bincount(inv, weights=b[sel]), rtol=1e-12)
    wmax = np.full(len(ug), -np.inf)
    ok = ~dvn[di]
    np.maximum.at(wmax, inv[ok], dval[di][ok])
    gmax, gnull = pr.column(4)
    assert np.array_equal(gnull[order], np.isinf(wmax))
    assert np.array_equal(gmax[order][~np.isinf(wmax)], wmax[~np.isinf(wmax)])


@pytest.mark.parametrize("ngroups,keytype", [(53, "int4"), (9000, "int4"), (53, "int8")])
def test_join_as_a_lookup_inside_the_aggregate(ngroups, keytype):
    """strom_submit_gpupreagg_lookup: fact JOIN dim WHERE ... GROUP BY in ONE pass over
    the fact chunk -- no join request, no result pairs; rows without a partner (NULL key,
    key outside the table, empty slot) are dropped, the WHERE is the aggregate's qual"""
    runtime.init()
    n, nd = 250007, 20000
    rng = np.random.default_rng(89)
    span = int(nd * 1.3)
    fk = rng.integers(-50, span + 50, n).astype(np.int32)      # also keys below / above the table
    fkn = rng.random(n) < 0.02
    a = rng.integers(0, 2**31, n, dtype=np.int64).astype(np.int32)
    an = rng.random(n) < 0.03
    b = rng.random(n)
    kdt = np.int32 if keytype == "int4" else np.int64
    fact = kds.build_kds("column", [kds.Column(keytype, fk.astype(kdt), fkn), kds.Column("int4", a, an),
                                    kds.Column("float8", b)])
    dkey = rng.permutation(span)[:nd].astype(np.int32)         # holes in the key range
    dgrp = (dkey % ngroups).astype(np.int32)
    dgn = rng.random(nd) < 0.04
    dval = rng.random(nd) * 10
    dvn = rng.random(nd) < 0.05
    inner = kds.build_kds("row_flat", [kds.Column(keytype, dkey.astype(kdt)), kds.Column("int4", dgrp, dgn),
                                       kds.Column("float8", dval, dvn)])
    km = build_multihash([(inner, [1])])
    ext = [np.int32(2**30), 0.25]
    ds = runtime.DeviceStore.upload(fact)
    join = GpuHashJoin("(gpuhashjoin (rel (hashkey (var 1 %s) 1 %s)))" % (keytype, keytype)).begin(km)
    spec = ("(gpupreagg (qual " + QUAL + ") (key (var 1 int4)) (nrows) (psum (int8 (var 2 int4)))"
            " (psum (var 3 float8)) (pmax (var 4 float8)))")
    agg = GpuPreAgg(spec)
    try:
        assert join.table_info(1)["mode"] == "direct" and join.table_info(1)["unique"]
        agg.begin([(0, ngroups)], ext_params=ext)
        for _ in range(2):                                      # two chunks into the same table
            assert agg.collect(agg.submit_lookup(join, ds, [(1, 2, "int4"), (0, 2, "int4"), (0, 3, "float8"),
                                                            (1, 3, "float8")]))[0] == 0
        pr = agg.fetch()
    finally:
        agg.end()
        join.end()
        ds.release()
    pos = np.full(span + 100, -1, dtype=np.int64)
    pos[dkey] = np.arange(nd)
    inrange = (~fkn) & (fk >= 0) & (fk < span)
    di_all = np.where(inrange, pos[np.clip(fk, 0, span - 1)], -1)
    sel = np.flatnonzero((di_all >= 0) & (~an) & (a < ext[0]) & (b > ext[1]))
    di = di_all[sel]
    g = np.where(dgn[di], 10**6, dgrp[di])
    ug, inv = np.unique(g, return_inverse=True)
    keys, knull = pr.column(1 - 1)
    gk = np.where(knull, 10**6, keys)
    order = np.argsort(gk)
    assert np.array_equal(gk[order], ug)
    assert np.array_equal(pr.column(1)[0][order], 2 * np.bincount(inv))
    sums = np.zeros(len(ug), dtype=np.int64)
    np.add.at(sums, inv, a[sel].astype(np.int64))
    assert np.array_equal(pr.column(2)[0][order], 2 * sums)
    assert np.allclose(pr.column(3)[0][order], 2 * np.bincount(inv, weights=b[sel]), rtol=1e-12)
    wmax = np.full(len(ug), -np.inf)
    ok = ~dvn[di]
    np.maximum.at(wmax, inv[ok], dval[di][ok])
    gmax, gnull = pr.column(4)
    assert np.array_equal(gnull[order], np.isinf(wmax))
    assert np.array_equal(gmax[order][~np.isinf(wmax)], wmax[~np.isinf(wmax)])


@pytest.mark.parametrize("epochs,narrow,generic", [(False, True, False), (True, True, False), (False, False, False),
                                                   (False, True, True), (False, False, True)])
def test_join_as_a_lookup_with_packed_accumulators(epochs, narrow, generic, monkeypatch):
    """generic: the program as built for the session, which reads the column mapping and the
    record form from the map at run time (STROM_GPUPREAGG_LOOKUP_GENERIC); otherwise the program
    built FOR the mapping (gpupreagg.cpp: lookup_mapping_program) -- same answers.
    narrow: the slot records are 2 bytes (presence, NULL bit, 14 bits of the group column:
    hashjoin_dimrec_narrow_kernel) instead of 8 -- same answers either way.
    1e4 groups, summed OUTER columns without NULLs: the lookup aggregate takes the packed
    LDS image (gpupreagg_packed_lookup, one id-range role instead of two); the grouping key
    is an inner column with NULLs, rows without a partner are dropped.  epochs (the name of
    round 2's mechanism): a 31-bit value range next to the count does not fit the packed word
    for all rows of a work-group -- narrow fields (capped at 15 bits here), returning LDS adds,
    and most of the fact rows point at ONE dimension row, so that group's word is moved to the
    slab again and again while the fold runs (gpupreagg_packed_spill)."""
    runtime.init()
    n, nd, ngroups = (16_000_003 if epochs else 300007), 40000, 10000
    if epochs:
        monkeypatch.setenv("STROM_GPUPREAGG_PACK_COUNT_BITS", "15")
    if not narrow:
        monkeypatch.setenv("STROM_HASHJOIN_NO_NARROW_RECS", "1")
    if generic:
        monkeypatch.setenv("STROM_GPUPREAGG_LOOKUP_GENERIC", "1")
    rng = np.random.default_rng(97)
    span = int(nd * 1.25)
    fk = rng.integers(0, span, n).astype(np.int32)
    a = (rng.integers(0, 2**31 - 1, n) if epochs else rng.integers(-5 * 10**5, 2**20, n)).astype(np.int32)
    b = rng.random(n)
    fact = kds.build_kds("column", [kds.Column("int4", fk), kds.Column("int4", a), kds.Column("float8", b)])
    dkey = rng.permutation(span)[:nd].astype(np.int32)
    dgrp = (dkey % ngroups).astype(np.int32)
    dgn = rng.random(nd) < 0.02
    inner = kds.build_kds("row_flat", [kds.Column("int4", dkey), kds.Column("int4", dgrp, dgn)])
    if epochs:
        fk[rng.random(n) < 0.6] = dkey[np.flatnonzero(~dgn)[0]]        # the hot group: ~14000 surviving rows per work-group
    fact = kds.build_kds("column", [kds.Column("int4", fk), kds.Column("int4", a), kds.Column("float8", b)])
    km = build_multihash([(inner, [1])])
    ext = [np.int32(2**30 if epochs else 2**19), 0.25]
    ds = runtime.DeviceStore.upload(fact)
    join = GpuHashJoin("(gpuhashjoin (rel (hashkey (var 1 int4) 1 int4)))").begin(km)
    spec = ("(gpupreagg (qual " + QUAL + ") (key (var 1 int4)) (nrows) (psum (int8 (var 2 int4)))"
            " (psum (var 3 float8)))")
    agg = GpuPreAgg(spec)
    try:
        agg.begin([(0, ngroups)], ext_params=ext)
        pfms = []
        for _ in range(2):
            st, pfm = agg.collect(agg.submit_lookup(join, ds, [(1, 2, "int4"), (0, 2, "int4"), (0, 3, "float8")]))
            assert st == 0
            pfms.append(pfm)
        pr = agg.fetch()
    finally:
        agg.end()
        join.end()
        ds.release()
    assert all(p["num_kern_prep"] == 1 for p in pfms)             # the packed path was taken
    pos = np.full(span, -1, dtype=np.int64)
    pos[dkey] = np.arange(nd)
    di_all = pos[fk]
    sel = np.flatnonzero((di_all >= 0) & (a < ext[0]) & (b > ext[1]))
    di = di_all[sel]
    g = np.where(dgn[di], 10**6, dgrp[di])
    ug, inv = np.unique(g, return_inverse=True)
    keys, knull = pr.column(0)
    gk = np.where(knull, 10**6, keys)
    order = np.argsort(gk)
    assert np.array_equal(gk[order], ug)
    assert np.array_equal(pr.column(1)[0][order], 2 * np.bincount(inv))
    sums = np.zeros(len(ug), dtype=np.int64)
    np.add.at(sums, inv, a[sel].astype(np.int64))
    assert np.array_equal(pr.column(2)[0][order], 2 * sums)
    assert np.allclose(pr.column(3)[0][order], 2 * np.bincount(inv, weights=b[sel]), rtol=1e-12)


def test_narrow_slot_records_negative_values_two_inner_columns():
    """two integer inner columns in one 4-byte record: a group column with negative values
    and NULLs, an int2 column the WHERE reads (so the qual runs after the probe); the same
    query with the standard records must give the same partial rows"""
    runtime.init()
    n, nd = 400003, 30000
    rng = np.random.default_rng(5)
    span = int(nd * 1.2)
    fk = rng.integers(0, span, n).astype(np.int32)
    a = rng.integers(-10**6, 10**6, n).astype(np.int32)
    fact = kds.build_kds("column", [kds.Column("int4", fk), kds.Column("int4", a)])
    dkey = rng.permutation(span)[:nd].astype(np.int32)
    dgrp = (dkey % 300 - 150).astype(np.int32)
    dgn = rng.random(nd) < 0.03
    dflag = rng.integers(-3, 4, nd).astype(np.int16)
    dfn = rng.random(nd) < 0.03
    inner = kds.build_kds("row", [kds.Column("int4", dkey), kds.Column("int4", dgrp, dgn),
                                  kds.Column("int2", dflag, dfn)])
    km = build_multihash([(inner, [1])])
    spec = ("(gpupreagg (qual (int2gt (var 3 int2) (const int2 -2))) (key (var 1 int4)) (nrows)"
            " (psum (int8 (var 2 int4))) (pmin (var 3 int2)))")
    cols = [(1, 2, "int4"), (0, 2, "int4"), (1, 3, "int2")]
    got = {}
    import os
    for narrow in (True, False):
        if narrow:
            os.environ.pop("STROM_HASHJOIN_NO_NARROW_RECS", None)
        else:
            os.environ["STROM_HASHJOIN_NO_NARROW_RECS"] = "1"
        ds = runtime.DeviceStore.upload(fact)
        join = GpuHashJoin("(gpuhashjoin (rel (hashkey (var 1 int4) 1 int4)))").begin(km)
        agg = GpuPreAgg(spec)
        try:
            agg.begin([(-150, 300)])
            assert agg.collect(agg.submit_lookup(join, ds, cols))[0] == 0
            pr = agg.fetch()
        finally:
            agg.end()
            join.end()
            ds.release()
            os.environ.pop("STROM_HASHJOIN_NO_NARROW_RECS", None)
        rows = [(None if pr.isnull[i, 0] else int(pr.column(0)[0][i]), int(pr.column(1)[0][i]),
                 int(pr.column(2)[0][i]), None if pr.isnull[i, 3] else int(pr.column(3)[0][i]))
                for i in range(len(pr))]
        got[narrow] = sorted(rows, key=lambda r: (r[0] is not None, r[0] if r[0] is not None else 0))
    assert got[True] == got[False]
    # and numpy
    pos = np.full(span, -1, dtype=np.int64)
    pos[dkey] = np.arange(nd)
    di = pos[fk]
    sel = np.flatnonzero((di >= 0) & ~dfn[np.clip(di, 0, nd - 1)] & (dflag[np.clip(di, 0, nd - 1)] > -2))
    want = {}
    for i in sel:
        d = di[i]
        k = None if dgn[d] else int(dgrp[d])
        c, sm, mn = want.get(k, (0, 0, 99))
        want[k] = (c + 1, sm + int(a[i]), min(mn, int(dflag[d])))
    wrows = sorted(((k, c, sm, mn) for k, (c, sm, mn) in want.items()),
                   key=lambda r: (r[0] is not None, r[0] if r[0] is not None else 0))
    grows = sorted(got[True], key=lambda r: (r[0] is not None, r[0] if r[0] is not None else 0))
    assert grows == wrows
