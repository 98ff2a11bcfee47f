"""
Multi-GPU GpuPreAgg merge rehearsed on CPU: world_size 2, gloo.  Each rank
takes a row range of the reference's fixture, reduces it with the CPU
oracle, packs its partial rows into a table with the product's dense
layout, runs the same all-reduce the GPU path runs over RCCL
(pg_strom_amd.parallel.allreduce_table), and rank 0 checks the merged table
against the oracle over the whole table -- integers exact, float sums 1e-12.
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SPEC = ("(gpupreagg (key (var 2 int4)) (nrows) (nrows (isnotnull (var 4 int4))) (psum (int8 (var 4 int4)))"
        " (psum (var 7 float8)) (pmin (var 7 float8)) (pmax (var 4 int4)) (pmin (var 5 int8)))")
NT = 8


def _worker(rank, world, port, outq):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    import torch
    import torch.distributed as dist
    import agg_golden
    import oracle_binding as oracle
    from pg_strom_amd import kds, parallel
    from pg_strom_amd.gpupreagg import codegen_gpupreagg
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        fx = agg_golden.load_fixture()
        n = len(fx["id"])
        lo, hi = rank * n // world, (rank + 1) * n // world
        buf = kds.build_kds("column", agg_golden.fixture_columns(fx, slice(lo, hi)))
        rc, v, isn = oracle.gpupreagg(SPEC, buf, NT)
        assert rc == 0
        targets = codegen_gpupreagg(SPEC).targets
        domain = [(1, 30)]
        layout = parallel.TableLayout(targets, 31)
        tbl = torch.from_numpy(parallel.pack_rows(layout, domain, v, isn))
        parallel.allreduce_table(tbl, layout)
        if rank == 0:
            whole = kds.build_kds("column", agg_golden.fixture_columns(fx))
            rc, wv, wn = oracle.gpupreagg(SPEC, whole, NT)
            gv, gn = parallel.unpack_rows(layout, domain, tbl.numpy())
            order_w = np.lexsort((wv[:, 0].view(np.int64), wn[:, 0]))
            order_g = np.lexsort((gv[:, 0].view(np.int64), gn[:, 0]))
            wv, wn, gv, gn = wv[order_w], wn[order_w], gv[order_g], gn[order_g]
            ok = (gv.shape == wv.shape) and np.array_equal(gn, wn)
            for t, (kind, oid) in enumerate(targets):
                if not ok:
                    break
                if oid in (700, 701) and kind == 3:
                    ok &= np.allclose(gv[:, t].view(np.float64)[~gn[:, t]],
                                      wv[:, t].view(np.float64)[~wn[:, t]], rtol=1e-12, atol=0)
                else:
                    ok &= np.array_equal(gv[:, t][~gn[:, t]], wv[:, t][~wn[:, t]])
            outq.put(bool(ok))
    finally:
        dist.destroy_process_group()


def test_two_rank_merge_matches_single_reduction():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _wide_worker(rank, world, port, outq):
    sys.path.insert(0, os.path.dirname(HERE))
    import torch
    import torch.distributed as dist
    from pg_strom_amd import parallel
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        # key int4, nrows, psum(int8): group 0 near +2^63 on both ranks, group 1 near -2^63,
        # group 2 only on rank 1, group 3 opposite signs that cancel, group 4 already beyond int8
        # on rank 0 (a table that has folded several chunks)
        targets = [(1, 23), (2, 23), (3, 20)]
        layout = parallel.TableLayout(targets, 6)
        big = 2**63 - 5
        mine = {0: {0: big, 1: -big - 3, 3: big, 4: 3 * big}, 1: {0: big - 1, 1: -big, 2: 17, 3: -big}}[rank]
        tbl = np.zeros(layout.nbytes, dtype=np.uint8)
        flags = tbl[:4 * 6].view(np.uint32)
        n = layout.ngroups
        for gid, total in mine.items():
            flags[gid] = 1 | (2 << 1)
            tbl[layout.vals_offset(0):layout.vals_offset(0) + 8 * n].view(np.int64)[gid] = 1 + rank
            tbl[layout.vals_offset(1):layout.vals_offset(1) + 8 * n].view(np.uint64)[gid] = total & (2**64 - 1)
            tbl[layout.hi_offset(1):layout.hi_offset(1) + 8 * n].view(np.int64)[gid] = total >> 64
        t = torch.from_numpy(tbl)
        parallel.allreduce_table(t, layout)
        got = {g: layout.int_sum(t.numpy(), 1, g) for g in range(5)}
        outq.put((rank, got, [int(x) for x in t.numpy()[:24].view(np.uint32)]))
    finally:
        dist.destroy_process_group()


def test_two_rank_merge_of_integer_sums_never_wraps():
    """psum(int8) is 128 bits wide in the table and travels as three carry-free limbs
    (strom_merge.h / parallel.allreduce_table): two ranks whose sums are each at the edge of int8
    merge to the exact big-integer total -- ncclSum on the raw int64 would have wrapped"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_wide_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = [q.get(timeout=5) for _ in range(2)]
    big = 2**63 - 5
    want = {0: 2 * big - 1, 1: -2 * big - 3, 2: 17, 3: 0, 4: 3 * big}
    for rank, got, flags in res:
        assert got == want, (rank, got)
        assert flags == [5, 5, 5, 5, 5, 0]


def test_pack_unpack_round_trip():
    sys.path.insert(0, HERE)
    import agg_golden
    import oracle_binding as oracle
    from pg_strom_amd import kds, parallel
    from pg_strom_amd.gpupreagg import codegen_gpupreagg
    fx = agg_golden.load_fixture()
    buf = kds.build_kds("row", agg_golden.fixture_columns(fx))
    rc, v, isn = oracle.gpupreagg(SPEC, buf, NT)
    targets = codegen_gpupreagg(SPEC).targets
    layout = parallel.TableLayout(targets, 31)
    tbl = parallel.pack_rows(layout, [(1, 30)], v, isn)
    gv, gn = parallel.unpack_rows(layout, [(1, 30)], tbl)
    a = np.lexsort((v[:, 0].view(np.int64), isn[:, 0]))
    b = np.lexsort((gv[:, 0].view(np.int64), gn[:, 0]))
    assert np.array_equal(isn[a], gn[b])
    assert np.array_equal(np.where(isn[a], 0, v[a]), np.where(gn[b], 0, gv[b]))


def _census_worker(rank, world, port, outq):
    sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist
    from pg_strom_amd import parallel
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        nbits = 209                                  # the Q1 domain: 19 x 11 dense ids
        mine = np.zeros((nbits + 31) // 32, dtype=np.uint32)
        for d in ((0, 5, 40, 208) if rank == 0 else (5, 77, 100)):
            mine[d >> 5] |= np.uint32(1 << (d & 31))
        merged = parallel.allreduce_census(mine)
        got = np.flatnonzero(np.unpackbits(merged.view(np.uint8), bitorder="little"))
        outq.put((rank, [int(x) for x in got]))
    finally:
        dist.destroy_process_group()


def test_two_ranks_agree_on_group_slots():
    """census bitmaps are OR-ed over the ranks: both compact to the same slots"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_census_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(2))
    assert res[0] == res[1] == [0, 5, 40, 77, 100, 208]


HASHED_SPEC = ("(gpupreagg (key (var 1 float8)) (key (var 2 int8)) (nrows) (psum (int8 (var 3 int4)))"
               " (psum (var 4 float8)) (pmin (var 4 float8)) (pmax (var 3 int4)))")


def _hashed_table(n, seed):
    from pg_strom_amd import kds
    rng = np.random.default_rng(seed)
    f = rng.integers(-3, 4, n).astype(np.float64) / 2
    f[rng.random(n) < 0.04] = np.nan
    big = (rng.integers(0, 5, n).astype(np.int64) - 2) * (2**41 + 7)
    x = rng.integers(-10**6, 10**6, n).astype(np.int32)
    y = rng.random(n) * 100
    return [kds.Column("float8", f, rng.random(n) < 0.03), kds.Column("int8", big, rng.random(n) < 0.03),
            kds.Column("int4", x, rng.random(n) < 0.05), kds.Column("float8", y, rng.random(n) < 0.05)]


def _hashed_worker(rank, world, port, outq):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    import torch.distributed as dist
    import oracle_binding as oracle
    from pg_strom_amd import kds, parallel
    from pg_strom_amd.gpupreagg import codegen_gpupreagg
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        cols = _hashed_table(20000, 97)
        n = len(cols[0].values)
        lo, hi = rank * n // world, (rank + 1) * n // world
        part = [kds.Column(c.sqltype, c.values[lo:hi], c.isnull[lo:hi]) for c in cols]
        rc, v, isn = oracle.gpupreagg(HASHED_SPEC, kds.build_kds("column", part), 7)
        assert rc == 0
        targets = codegen_gpupreagg(HASHED_SPEC).targets
        gv, gn = parallel.gather_partial_rows(targets, v, isn)
        if rank == 0:
            rc, wv, wn = oracle.gpupreagg(HASHED_SPEC, kds.build_kds("column", cols), 7)
            def order(v_, n_):
                return np.lexsort((v_[:, 1], n_[:, 1], v_[:, 0], n_[:, 0]))
            og, ow = order(gv, gn), order(wv, wn)
            gv, gn, wv, wn = gv[og], gn[og], wv[ow], wn[ow]
            ok = gv.shape == wv.shape and np.array_equal(gn, wn)
            for t, (kind, oid) in enumerate(targets):
                if not ok:
                    break
                live = ~wn[:, t]
                if oid in (700, 701) and kind == 3:
                    ok &= np.allclose(gv[:, t].view(np.float64)[live], wv[:, t].view(np.float64)[live], rtol=1e-12, atol=0)
                else:
                    ok &= np.array_equal(gv[:, t][live], wv[:, t][live])
            outq.put(bool(ok))
    finally:
        dist.destroy_process_group()


def test_two_rank_merge_of_hashed_group_by_partial_rows():
    """hashed GROUP BY sessions have no common table layout: ranks gather their partial
    rows and combine equal keys (float keys incl. NaN / NULL, sparse int8 keys)"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    outq = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 7
    procs = [ctx.Process(target=_hashed_worker, args=(r, 2, port, outq)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    assert all(p.exitcode == 0 for p in procs)
    assert outq.get(timeout=5) is True


def _scatter_worker(rank, world, port, outq):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    import torch.distributed as dist
    import oracle_binding as oracle
    from pg_strom_amd import kds, parallel, runtime
    from pg_strom_amd.gpupreagg import codegen_gpupreagg
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        cols = _hashed_table(20000, 98)
        n = len(cols[0].values)
        lo, hi = rank * n // world, (rank + 1) * n // world
        part = [kds.Column(c.sqltype, c.values[lo:hi], c.isnull[lo:hi]) for c in cols]
        rc, v, isn = oracle.gpupreagg(HASHED_SPEC, kds.build_kds("column", part), 7)
        assert rc == 0
        targets = codegen_gpupreagg(HASHED_SPEC).targets
        mv, mn = parallel.reduce_scatter_partial_rows(targets, v, isn)
        # every rank reports its partition; rank 0 checks union == one reduction, partitions disjoint
        shares = [None] * world
        dist.all_gather_object(shares, (mv, mn))
        # integer sums that cannot be added: every rank gets CpuReCheck
        big = v.copy()
        for t, (kind, oid) in enumerate(targets):
            if kind == 3 and oid not in (700, 701):
                big[:, t] = np.uint64(6 * 2**63 // 10)
        try:
            parallel.reduce_scatter_partial_rows(targets, big, isn)
            refused = False
        except runtime.StromError as e:
            refused = (e.errcode == 2)
        if rank == 0:
            rc, wv, wn = oracle.gpupreagg(HASHED_SPEC, kds.build_kds("column", cols), 7)
            gv = np.concatenate([s_[0] for s_ in shares])
            gn = np.concatenate([s_[1] for s_ in shares])
            ok = refused and all(len(s_[0]) > 0 for s_ in shares) and len(gv) == len(wv)
            owners = [parallel.owner_of_partial_rows(targets, s_[0], s_[1], world) for s_ in shares]
            ok = ok and all((o == r).all() for r, o in enumerate(owners))
            def order(v_, n_):
                return np.lexsort((v_[:, 1], n_[:, 1], v_[:, 0], n_[:, 0]))
            og, ow = order(gv, gn), order(wv, wn)
            gv, gn, wv, wn = gv[og], gn[og], wv[ow], wn[ow]
            ok = ok and np.array_equal(gn, wn)
            for t, (kind, oid) in enumerate(targets):
                if not ok:
                    break
                live = ~wn[:, t]
                if oid in (700, 701) and kind == 3:
                    ok &= np.allclose(gv[:, t].view(np.float64)[live], wv[:, t].view(np.float64)[live], rtol=1e-12, atol=0)
                else:
                    ok &= np.array_equal(gv[:, t][live], wv[:, t][live])
            outq.put(bool(ok))
        else:
            assert refused
    finally:
        dist.destroy_process_group()


def test_two_rank_hash_partitioned_merge_of_partial_rows():
    """the reduce-scatter of hashed sessions (strom_gpupreagg_reduce_scatter; its C steps run among
    sessions of one GPU in tests/test_exchange_gpu.py): every group lands on the rank its key
    hashes to, the ranks' shares are disjoint, their union is the single reduction; sums that
    could leave int8 together are refused on both ranks"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    outq = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 11
    procs = [ctx.Process(target=_scatter_worker, args=(r, 2, port, outq)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    assert all(p.exitcode == 0 for p in procs)
    assert outq.get(timeout=5) is True


def test_three_rank_hash_partitioned_merge_of_partial_rows():
    """the same exchange among three ranks: the owner of a group is its remixed key hash modulo the
    world size, which must also hold when that is not a power of two"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    outq = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 17
    procs = [ctx.Process(target=_scatter_worker, args=(r, 3, port, outq)) for r in range(3)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    assert all(p.exitcode == 0 for p in procs)
    assert outq.get(timeout=5) is True


def test_merge_partial_rows_orders_nan_like_postgresql():
    """float8 pmin/pmax across ranks: PostgreSQL sorts NaN above every number
    (float8_cmp_internal), so min(NaN, 1.0) = 1.0, min(NaN) = NaN, max(NaN, 1.0) = NaN"""
    from pg_strom_amd import parallel
    targets = [(parallel.KIND_KEY, 23), (parallel.KIND_PMIN, 701), (parallel.KIND_PMAX, 701)]

    def rows(keys, mins, maxs):
        v = np.zeros((len(keys), 3), dtype=np.uint64)
        v[:, 0] = np.array(keys, dtype=np.int64).view(np.uint64)
        v[:, 1] = np.array(mins, dtype=np.float64).view(np.uint64)
        v[:, 2] = np.array(maxs, dtype=np.float64).view(np.uint64)
        return v, np.zeros((len(keys), 3), dtype=bool)

    nan = float("nan")
    a = rows([1, 2, 3], [nan, nan, 5.0], [nan, 2.0, 5.0])
    b = rows([1, 2, 3], [1.0, nan, 7.0], [1.0, nan, 7.0])
    v, n = parallel.merge_partial_rows(targets, [a, b])
    order = np.argsort(v[:, 0].view(np.int64))
    mins = v[order, 1].view(np.float64)
    maxs = v[order, 2].view(np.float64)
    assert not n.any()
    assert mins[0] == 1.0 and np.isnan(mins[1]) and mins[2] == 5.0
    assert np.isnan(maxs[0]) and np.isnan(maxs[1]) and maxs[2] == 7.0
