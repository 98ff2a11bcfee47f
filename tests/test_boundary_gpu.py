"""
Boundary behaviour of the C ABI on the device (needs an MI355X: -m gpu):
  - requests the library must REFUSE on the host instead of launching a kernel
    with indexes it would dereference (a faulting kernel aborts the host
    process -- a PostgreSQL backend here): bad projection mappings, a broken
    kern_multihash image, a lookup aggregate over a table whose WHERE lives in
    the join program;
  - the callback protocol through ctypes: results left in HBM are chained
    AFTER the callback fired (the handle outlives the callback);
  - strom_gpupreagg_allreduce (RCCL inside the C library) at world size 1:
    the table comes back unchanged, equal to the torch statement of the merge;
  - the streaming-read probe.
"""
import ctypes
import threading

import numpy as np
import pytest

import oracle_binding as oracle
from pg_strom_amd import kds, runtime, parallel
from pg_strom_amd._lib import lib, DONE_CB, strom_perfmon
from pg_strom_amd.gpuhashjoin import GpuHashJoin, build_multihash
from pg_strom_amd.gpupreagg import GpuPreAgg
from pg_strom_amd.gpuscan import GpuScan, STROM_RESULTS_ON_DEVICE
from pg_strom_amd.kds import make_kern_gpuscan

pytestmark = pytest.mark.gpu

QUAL = "(and (int4lt (var 2 int4) (param 0 int4)) (float8gt (var 3 float8) (param 1 float8)))"
ERR_BAD_REQUEST = 101
ERR_CORRUPTION = 300


def _fact_and_dim(n=60011, nd=5000, seed=5):
    rng = np.random.default_rng(seed)
    fk = rng.integers(0, int(nd * 1.2), n).astype(np.int32)
    a = rng.integers(0, 2**31, n, dtype=np.int64).astype(np.int32)
    b = rng.random(n)
    fact = kds.build_kds("column", [kds.Column("int4", fk), kds.Column("int4", a), kds.Column("float8", b)])
    dkey = rng.permutation(nd).astype(np.int32)
    dgrp = (dkey % 37).astype(np.int32)
    inner = kds.build_kds("row_flat", [kds.Column("int4", dkey), kds.Column("int4", dgrp)])
    return (fk, a, b), fact, (dkey, dgrp), inner


def test_lookup_refuses_a_table_whose_where_lives_in_the_join_program():
    """strom_submit_gpupreagg_lookup runs the aggregate program only; a join
    program with a pulled-up qual must be refused there, while _joined applies the qual"""
    runtime.init()
    (fk, a, b), fact, (dkey, dgrp), inner = _fact_and_dim()
    km = build_multihash([(inner, [1])])
    ext = [np.int32(2**30), 0.5]
    ds = runtime.DeviceStore.upload(fact)
    jspec = "(gpuhashjoin (rel (hashkey (var 1 int4) 1 int4) (qual " + QUAL + ")))"
    join = GpuHashJoin(jspec, row_population_ratio=0.5).begin(km, ext_params=ext)
    spec = "(gpupreagg (key (var 1 int4)) (nrows) (psum (int8 (var 2 int4))) (psum (var 3 float8)))"
    agg = GpuPreAgg(spec).begin([(0, 37)])
    cols = [(1, 2, "int4"), (0, 2, "int4"), (0, 3, "float8")]
    try:
        with pytest.raises(runtime.StromError) as ei:
            agg.submit_lookup(join, ds, cols)
        assert ei.value.errcode == ERR_BAD_REQUEST
        # the fused plan over the join's result pairs does apply the join's WHERE
        jp = join.submit(ds, flags=STROM_RESULTS_ON_DEVICE)
        ap = agg.submit_joined(join, jp, ds, cols)
        assert agg.collect(ap)[0] == 0
        jr = join.collect(jp)
        pr = agg.fetch()
    finally:
        agg.end()
        join.end()
        ds.release()
    # oracle: join with the qual, then group the joined rows
    rc, nitems, recs = oracle.gpuhashjoin(jspec, fact, [inner], ext)
    assert rc == 0 and nitems == jr.nitems
    orow, irow = recs[:, 0] - 1, recs[:, 1]
    g = dgrp[irow]
    cnt = np.bincount(g, minlength=37)
    sx = np.bincount(g, weights=a[orow].astype(np.float64), minlength=37).astype(np.int64)
    order = np.argsort(pr.column(0)[0])
    assert np.array_equal(pr.column(0)[0][order], np.flatnonzero(cnt))
    assert np.array_equal(pr.column(1)[0][order], cnt[cnt > 0])
    assert np.array_equal(pr.column(2)[0][order], sx[cnt > 0])


@pytest.mark.parametrize("bad", [(2, 0), (-1, 0), (0, 3), (0, -1), (1, 2), (1, 99)])
def test_projection_mappings_are_validated_on_the_host(bad):
    """(relation, column) pairs outside the outer chunk / the inner relations never reach
    gpuhashjoin_projection_column / _slot: StromError_BadRequestMessage"""
    runtime.init()
    _, fact, _, inner = _fact_and_dim(n=20011)
    km = build_multihash([(inner, [1])])
    ds = runtime.DeviceStore.upload(fact)
    join = GpuHashJoin("(gpuhashjoin (rel (hashkey (var 1 int4) 1 int4)))").begin(km)
    try:
        pending = join.submit(ds, flags=STROM_RESULTS_ON_DEVICE)
        depth = np.array([0, bad[0]], dtype=np.int32)
        colidx = np.array([1, bad[1]], dtype=np.int32)
        oids = np.array([23, 23], dtype=np.int32)
        err = ctypes.c_int(0)
        handle = lib.strom_hashjoin_project_column(pending[0], join.table, ds.handle, 2, depth.ctypes.data,
                                                   colidx.ctypes.data, oids.ctypes.data, ctypes.byref(err))
        assert not handle and err.value == ERR_BAD_REQUEST
        join.collect(pending)
        # the TUPSLOT projection request takes the same mapping
        with pytest.raises(runtime.StromError) as ei:
            join.join_chunk_project(ds, [(0, 2, "int4"), (bad[0], bad[1] + 1, "int4")])
        assert ei.value.errcode == ERR_BAD_REQUEST
        # a sane mapping on the same objects still works afterwards
        out, n = join.join_to_column(ds, [(0, 2, "int4"), (1, 2, "int4")])
        assert n == out.nitems
        out.release()
    finally:
        join.end()
        ds.release()


def test_a_broken_multihash_image_is_refused():
    runtime.init()
    _, _, _, inner = _fact_and_dim(n=1000)
    km = build_multihash([(inner, [1])])
    join = GpuHashJoin("(gpuhashjoin (rel (hashkey (var 1 int4) 1 int4)))")
    join.program = runtime.DevProgram(join.codegen.source, join.codegen.extra_flags)
    join.program.wait()
    for what in ("offset", "length", "ncols"):
        bad = km.copy()
        toff = int(bad[1036:1040].view(np.uint32)[0])
        if what == "offset":
            bad[1036:1040] = np.array([len(bad) + 4096], dtype=np.uint32).view(np.uint8)
        elif what == "length":
            bad[toff:toff + 4] = np.array([len(bad) * 2], dtype=np.uint32).view(np.uint8)
        else:
            bad[toff + 4:toff + 8] = np.array([60000], dtype=np.uint32).view(np.uint8)
        err = ctypes.c_int(0)
        tbl = lib.strom_hashjoin_table_create(join.program.key, bad.ctypes.data, len(bad), 0, ctypes.byref(err))
        assert not tbl and err.value == ERR_CORRUPTION, what
    join.program.release()


def test_callback_then_chain_the_device_results():
    """done() fires on a runtime thread with the result head final; the handle stays
    valid, so the ids the scan left in HBM become a row map afterwards"""
    runtime.init()
    rng = np.random.default_rng(11)
    n = 200003
    g = rng.integers(0, 40, n).astype(np.int32)
    a = rng.integers(0, 2**31, n, dtype=np.int64).astype(np.int32)
    b = rng.random(n)
    buf = kds.build_kds("column", [kds.Column("int4", g), kds.Column("int4", a), kds.Column("float8", b)])
    ext = [np.int32(2**30), 0.5]
    want = int(np.count_nonzero((a < ext[0]) & (b > ext[1])))
    ds = runtime.DeviceStore.upload(buf)
    scan = GpuScan(QUAL).begin(ext_params=ext)
    scan.program.wait()
    kgs, res_off = make_kern_gpuscan(scan.parambuf, n, host_results=False)
    seen = {}
    fired = threading.Event()

    def on_done(arg, errcode, pfm):
        head = np.frombuffer(kgs[res_off:res_off + 20].tobytes(), dtype=np.int32)
        seen["thread"] = threading.get_ident()
        seen["errcode"] = errcode
        seen["nitems"] = int(head[2])
        seen["calls"] = seen.get("calls", 0) + 1
        fired.set()

    cb = DONE_CB(on_done)
    err = ctypes.c_int(0)
    task = lib.strom_submit_gpuscan(scan.program.key, kgs.ctypes.data, None, ds.handle, None,
                                    STROM_RESULTS_ON_DEVICE, ctypes.cast(cb, ctypes.c_void_p), None,
                                    ctypes.byref(err))
    assert task and err.value == 0
    assert fired.wait(60)
    assert seen["errcode"] == 0 and seen["nitems"] == want
    assert seen["thread"] != threading.get_ident()
    # after the callback: the handle is still good for the device-resident results
    handle = lib.strom_rowmap_from_task(task, ctypes.byref(err))
    assert handle and err.value == 0
    rowmap = runtime.DeviceRowMap(handle)
    assert rowmap.nvalids == want
    assert lib.strom_task_wait(task, None) == 0
    lib.strom_synchronize()
    assert seen["calls"] == 1
    spec = "(gpupreagg (key (var 1 int4)) (nrows))"
    agg = GpuPreAgg(spec).begin([(0, 40)])
    try:
        assert agg.fold(ds, row_map=rowmap)[0] == 0
        pr = agg.fetch()
    finally:
        agg.end()
        rowmap.release()
        scan.end()
        ds.release()
    sel = (a < ext[0]) & (b > ext[1])
    order = np.argsort(pr.column(0)[0])
    assert np.array_equal(pr.column(1)[0][order], np.bincount(g[sel], minlength=40))


def test_rccl_merge_of_a_hashed_session_world_size_1():
    """hashed GROUP BY sessions merge by all-gathering their packed groups: over a 1-rank
    communicator the export, the count gather and the early exit run, and the table is as before"""
    runtime.init()
    rng = np.random.default_rng(4)
    n = 80000
    key = rng.integers(0, 3000, n).astype(np.float64) * 0.5
    x = rng.integers(-1000, 1000, n).astype(np.int32)
    buf = kds.build_kds("column", [kds.Column("float8", key), kds.Column("int4", x, rng.random(n) < 0.05)])
    agg = GpuPreAgg("(gpupreagg (key (var 1 float8)) (nrows) (psum (int8 (var 2 int4))) (pmax (var 2 int4)))").begin_hashed()
    try:
        assert agg.fold(buf)[0] == 0
        before = agg.fetch()
        comm = parallel.RcclComm(0, 1)
        agg.allreduce_rccl(comm)
        comm.destroy()
        after = agg.fetch()
    finally:
        agg.end()
    ob, oa = np.argsort(before.column(0)[0]), np.argsort(after.column(0)[0])
    assert len(before) == len(after) == len(np.unique(key))
    assert np.array_equal(before.values[ob], after.values[oa]) and np.array_equal(before.isnull[ob], after.isnull[oa])


def test_rccl_merge_behind_the_abi_world_size_1():
    """strom_gpupreagg_allreduce over a 1-rank RCCL communicator made by the library:
    fetch() after the merge == fetch() without it (every section goes through its
    prepare -> collective -> finish path: int sum, float sum, int/float min and max,
    never-touched groups, NULL-only aggregates), and == the torch statement of the merge"""
    import torch
    import torch.distributed as dist
    runtime.init()
    rng = np.random.default_rng(3)
    n = 150001
    g = rng.integers(0, 90, n).astype(np.int32)                # groups 90..99 never occur
    x = rng.integers(-10**6, 10**6, n).astype(np.int32)
    xn = (rng.random(n) < 0.1) | (g == 7)                      # group 7: only NULL inputs
    y = rng.normal(size=n) * 50
    yn = rng.random(n) < 0.2
    buf = kds.build_kds("column", [kds.Column("int4", g, rng.random(n) < 0.01), kds.Column("int4", x, xn),
                                   kds.Column("float8", y, yn)])
    spec = ("(gpupreagg (key (var 1 int4)) (nrows) (nrows (isnotnull (var 2 int4))) (psum (int8 (var 2 int4)))"
            " (psum (var 3 float8)) (pmin (var 3 float8)) (pmax (var 3 float8)) (pmin (var 2 int4)) (pmax (var 2 int4)))")
    ds = runtime.DeviceStore.upload(buf)

    def partials(merge):
        agg = GpuPreAgg(spec).begin([(0, 100)])
        try:
            assert agg.fold(ds)[0] == 0
            if merge == "rccl":
                comm = parallel.RcclComm(0, 1)
                agg.allreduce_rccl(comm)
                agg.allreduce_rccl(comm)                        # idempotent at world size 1
                comm.destroy()
            elif merge == "torch":
                agg.allreduce()
            pr = agg.fetch()
            # (partial rows come in no particular order -- the device packs them: by key, NULL key last)
            order = np.lexsort((pr.values[:, 0].view(np.int64), pr.isnull[:, 0]))
            return pr.values[order].copy(), pr.isnull[order].copy()
        finally:
            agg.end()

    def same(va, na, vb, nb):
        # two folds of the same chunk: everything bit-equal except the float8 sum (column 4),
        # whose LDS-atomic accumulation order differs from launch to launch (tolerance 1e-12)
        assert np.array_equal(na, nb)
        for col in range(va.shape[1]):
            ok = ~na[:, col]
            if col == 4:
                assert np.allclose(va[ok, col].view(np.float64), vb[ok, col].view(np.float64), rtol=1e-12, atol=0)
            else:
                assert np.array_equal(va[ok, col], vb[ok, col]), col

    v0, n0 = partials(None)
    v1, n1 = partials("rccl")
    same(v0, n0, v1, n1)
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29541", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
    try:
        v2, n2 = partials("torch")
    finally:
        dist.destroy_process_group()
    same(v0, n0, v2, n2)
    ds.release()


def test_census_union_behind_the_abi_world_size_1():
    runtime.init()
    rng = np.random.default_rng(4)
    n = 50000
    k1 = rng.choice(np.array([3, 17, 900], dtype=np.int32), n)
    k2 = rng.choice(np.array([-5, 40], dtype=np.int32), n)
    x = rng.integers(0, 1000, n).astype(np.int32)
    buf = kds.build_kds("column", [kds.Column("int4", k1), kds.Column("int4", k2), kds.Column("int4", x)])
    spec = "(gpupreagg (key (var 1 int4)) (key (var 2 int4)) (nrows) (psum (int8 (var 3 int4))))"
    ds = runtime.DeviceStore.upload(buf)
    agg = GpuPreAgg(spec).begin([(3, 898), (-5, 46)])
    try:
        agg.census(ds)
        comm = parallel.RcclComm(0, 1)
        agg.census_allreduce(comm)
        assert agg.compact() == 6
        assert agg.fold(ds)[0] == 0
        agg.allreduce_rccl(comm)
        comm.destroy()
        pr = agg.fetch()
    finally:
        agg.end()
        ds.release()
    assert len(pr) == 6
    got = {(int(a), int(b)): (int(c), int(s)) for a, b, c, s in zip(pr.column(0)[0], pr.column(1)[0],
                                                                       pr.column(2)[0], pr.column(3)[0])}
    for a in (3, 17, 900):
        for b in (-5, 40):
            m = (k1 == a) & (k2 == b)
            assert got[(a, b)] == (int(m.sum()), int(x[m].sum()))


def test_streaming_read_probe():
    runtime.init()
    gbs = ctypes.c_double(0.0)
    assert lib.strom_membw_probe(0, 1 << 30, 5, ctypes.byref(gbs)) == 0
    assert 1000.0 < gbs.value < 8000.0, gbs.value          # above PCIe-class rates, below the HBM3E spec
