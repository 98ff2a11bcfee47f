"""
GpuPreAgg parity, HIP path (needs an MI355X: -m gpu).  All device work goes
through the C ABI (strom_gpupreagg_create / strom_submit_gpupreagg /
strom_gpupreagg_fetch).

Bars: group keys, nrows, int8 psum, pmin/pmax -- bit-exact against the CPU
oracle; float8 psum -- relative 1e-12 (the reference's own tests compare
float8 after dropping 3 digits, extra_float_digits=-3; summation order on
the device differs from row order); float4 results relative 1e-3.
"""
import os

import numpy as np
import pytest

import agg_golden
import oracle_binding as oracle
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpupreagg import GpuPreAgg, domain_of, KIND_NROWS, KIND_KEY

pytestmark = pytest.mark.gpu


def partial_rows_as_raw8(pr):
    """device PartialRows -> uint64 images with floats widened to float64"""
    n, nt = pr.values.shape
    out = np.zeros((n, nt), dtype=np.uint64)
    for t, (kind, oid) in enumerate(pr.targets):
        v, _ = pr.column(t)
        if v.dtype.kind == "f":
            out[:, t] = v.astype(np.float64).view(np.uint64)
        else:
            out[:, t] = v.astype(np.int64).view(np.uint64)
    return out, pr.isnull.copy()


def superset_plan(plan):
    """One device program per (column, query shape) instead of one per
    query: every aggregate of the suite over a column derives from the same
    partial columns, so the device computes the superset once and each query
    picks its targets.  Returns (superset plan, column picks)."""
    import re
    from pg_strom_amd import aggregate
    attno = [int(a) for a in re.findall(r"\(var (\d+) ", plan["spec"]) if int(a) != 2]
    typ = plan["type"]
    var = "(var %d %s)" % (attno[0], typ)
    wanted = []
    for func in ("count", "avg", "sum", "min", "max", "stddev"):
        rw = aggregate.rewrite(func, typ, var)
        if rw:
            for t in rw[0]:
                if t not in wanted:
                    wanted.append(t)
    head = []
    if "(qual" in plan["spec"]:
        head.append("(qual (int4eq (var 2 int4) (const int4 1)))")
    ntargets = len(wanted)
    if plan["grouped"]:
        head.append("(key (var 2 int4))")
        ntargets += 1
    sup = dict(plan)
    sup["spec"] = "(gpupreagg " + " ".join(head + wanted) + ")"
    sup["ntargets"] = ntargets
    # which superset column feeds each target of the original plan
    mine = re.findall(r"\((?:nrows|psum_x2|psum|pmin|pmax)(?: \((?:[^()]|\([^()]*\))*\))*\)", plan["spec"])
    base = 1 if plan["grouped"] else 0
    picks = ([0] if plan["grouped"] else []) + [base + wanted.index(t) for t in mine]
    return sup, picks


def hip_runner(chunks, fmt):
    cache = {}

    def run(plan):
        sup, picks = superset_plan(plan)
        key = (sup["spec"], plan["table"])
        if key not in cache:
            vals, nulls = [], []
            agg = GpuPreAgg(sup["spec"])
            # key column is attno 2 (key int4, 1..30 or NULL)
            agg.begin([(1, 30)] if plan["grouped"] else [])
            try:
                for buf, rows in chunks[plan["table"]]:
                    status, _ = agg.fold(buf)
                    if status == 2:
                        assert plan["type"] == "int8"       # int8 partial sum overflow
                        v, n = agg_golden.cpu_fallback_rows(sup, rows)
                        vals.append(v)
                        nulls.append(n)
                v, n = partial_rows_as_raw8(agg.fetch())
                vals.append(v)
                nulls.append(n)
            finally:
                agg.end()
            cache[key] = (np.concatenate(vals), np.concatenate(nulls))
        v, n = cache[key]
        return v[:, picks], n[:, picks]
    return run


@pytest.mark.parametrize("fmt,nchunks", [("column", 1), ("column", 4), ("row", 2), ("tupslot", 1)])
def test_reference_regression_suites_on_device(fmt, nchunks):
    """the reference's own regression answers (PostgreSQL output) through HIP"""
    chunks = {"gpupreagg_test": agg_golden.fixture_chunks(fmt, nchunks),
              "gpupreagg_zero_test": agg_golden.fixture_chunks(fmt, 1, empty=True)}
    run = hip_runner(chunks, fmt)
    total = 0
    for suite in ("nogrp_agg", "group_agg", "where_agg", "zero_agg"):
        checked, _ = agg_golden.check_suite(suite, run)
        total += checked
    assert total >= 200


def compare_with_oracle(spec, bufs, domain, ext=(), float_tol=1e-12, compact=False, hashed=False,
                        resident=False, pfms=None):
    agg = GpuPreAgg(spec)
    if hashed:
        agg.begin_hashed(ext_params=ext)
    else:
        agg.begin(domain, ext_params=ext)
    nt = len(agg.targets)
    try:
        if compact:
            for b in bufs:
                agg.census(b)
            agg.compact()
        for b in bufs:
            ds = runtime.DeviceStore.upload(b) if resident else None
            status, pfm = agg.fold(ds if resident else b)
            if ds is not None:
                ds.release()
            assert status == 0
            if pfms is not None:
                pfms.append(pfm)
        pr = agg.fetch()
    finally:
        agg.end()
    assert_matches_oracle(spec, agg, bufs, pr, ext, float_tol)


def assert_matches_oracle(spec, agg, bufs, pr, ext=(), float_tol=1e-12, row_maps=None):
    """device partial rows (one per group over all of 'bufs') against the oracle's
    per-chunk partial rows merged with the associative rules"""
    nt = len(agg.targets)
    merged = {}
    for bi, b in enumerate(bufs):
        rm = None if row_maps is None else row_maps[bi]
        rc, v, n = oracle.gpupreagg(spec, b, nt, ext, row_map=rm) if rm is not None \
            else oracle.gpupreagg(spec, b, nt, ext)
        assert rc == 0
        for i in range(len(v)):
            key = tuple((None if n[i, t] else int(v[i, t].view(np.int64)))
                        for t, (k, _) in enumerate(agg.targets) if k == KIND_KEY)
            merged.setdefault(key, []).append((v[i], n[i]))
    got_v, got_n = partial_rows_as_raw8(pr)
    assert len(got_v) == len(merged)
    for i in range(len(got_v)):
        key = tuple((None if got_n[i, t] else int(got_v[i, t].view(np.int64)))
                    for t, (k, _) in enumerate(agg.targets) if k == KIND_KEY)
        assert key in merged
        rows = merged[key]
        for t, (kind, oid) in enumerate(agg.targets):
            if kind == KIND_KEY:
                continue
            parts = [(r[0][t], r[1][t]) for r in rows]
            isfloat = oid in (700, 701)
            vals = [np.array([p[0]], dtype=np.uint64).view(np.float64 if isfloat else np.int64)[0]
                    for p in parts if not p[1]]
            if kind == KIND_NROWS:
                want = sum(int(x) for x in vals)
                assert int(got_v[i, t].view(np.int64)) == want
                continue
            if not vals:
                assert got_n[i, t]
                continue
            assert not got_n[i, t]
            g = got_v[i, t].view(np.float64 if isfloat else np.int64)
            if kind == 3:       # psum
                want = sum(vals) if not isfloat else float(np.sum(np.array(vals)))
                if isfloat:
                    tol = 2e-3 if oid == 700 else float_tol
                    if np.isnan(want):
                        assert np.isnan(g), (key, t, g, want)
                    else:
                        assert abs(g - want) <= tol * max(abs(want), 1e-300), (key, t, g, want)
                else:
                    assert int(g) == int(want)
            else:
                if isfloat:
                    arr = np.array(vals)
                    if kind == 4:
                        nn = arr[~np.isnan(arr)]
                        want = nn.min() if len(nn) else np.nan
                    else:
                        want = np.nan if np.isnan(arr).any() else arr.max()
                    assert (np.isnan(g) and np.isnan(want)) or g == want, (key, t, g, want)
                else:
                    assert int(g) == (min(vals) if kind == 4 else max(vals))


def random_table(n, seed, ngroups=50, nulls=0.03):
    rng = np.random.default_rng(seed)
    g = rng.integers(-7, ngroups - 7, n).astype(np.int32)
    h = rng.integers(0, 3, n).astype(np.int16)
    x = rng.integers(-10**6, 10**6, n).astype(np.int32)
    y = rng.random(n) * 100
    y[::53] = np.nan
    z = (rng.random(n) * 10 - 5).astype(np.float32)
    cols = [kds.Column("int4", g, rng.random(n) < nulls), kds.Column("int2", h),
            kds.Column("int4", x, rng.random(n) < nulls), kds.Column("float8", y, rng.random(n) < nulls),
            kds.Column("float4", z, rng.random(n) < nulls)]
    return cols


SPEC_ALL = ("(gpupreagg (key (var 1 int4)) (key (var 2 int2)) (nrows) (nrows (isnotnull (var 3 int4)))"
            " (psum (int8 (var 3 int4))) (pmin (var 3 int4)) (pmax (var 3 int4))"
            " (psum (var 4 float8)) (pmin (var 4 float8)) (pmax (var 4 float8))"
            " (psum (var 5 float4)) (psum_x2 (float8 (var 5 float4))))")


@pytest.mark.parametrize("fmt", ["column", "row", "row_flat"])
def test_two_keys_all_partial_kinds(fmt):
    cols = random_table(60000, 5)
    bufs = [kds.build_kds(fmt, [kds.Column(c.sqltype, c.values[i::2], None if c.isnull is None else c.isnull[i::2])
                                for c in cols]) for i in range(2)]
    compare_with_oracle(SPEC_ALL, bufs, [(-7, 50), (0, 3)])


def test_qual_pullup_and_params():
    cols = random_table(40000, 9)
    buf = kds.build_kds("column", cols)
    spec = ("(gpupreagg (qual (and (int4gt (var 3 int4) (param 0 int4)) (float8lt (var 4 float8) (const float8 50))))"
            " (key (var 2 int2)) (nrows) (psum (var 4 float8)) (pmax (var 3 int4)))")
    compare_with_oracle(spec, [buf], [(0, 3)], ext=[np.int32(-500000)])


def test_id_range_split_over_workgroup_roles(monkeypatch):
    """state larger than the LDS budget: the dense id range is split over
    work-group roles (forced here with a tiny budget)"""
    monkeypatch.setenv("STROM_GPUPREAGG_LDS_BUDGET", "6000")
    cols = random_table(80000, 13, ngroups=400)
    buf = kds.build_kds("column", cols)
    spec = "(gpupreagg (key (var 1 int4)) (nrows) (psum (int8 (var 3 int4))) (psum (var 4 float8)))"
    compare_with_oracle(spec, [buf], domain_of([buf], [0]))


def c4_table(n, seed, ngroups, xlo=-10**6, xhi=10**6, nulls=None):
    rng = np.random.default_rng(seed)
    g = rng.integers(0, ngroups, n).astype(np.int32)
    g[:ngroups] = np.arange(ngroups, dtype=np.int32)           # every group occurs
    x = rng.integers(xlo, xhi, n).astype(np.int32)
    y = rng.random(n) * 100
    return [kds.Column("int4", g), kds.Column("int4", x, None if nulls is None else rng.random(n) < nulls),
            kds.Column("float8", y)]


C4_SPEC = "(gpupreagg (key (var 1 int4)) (nrows) (psum (int8 (var 2 int4))) (psum (var 3 float8)))"


@pytest.mark.parametrize("ngroups,resident", [(10000, True), (10000, False), (25000, True), (9000, True)])
def test_packed_accumulators_c4_shape(ngroups, resident):
    """BASELINE configs[3]: 1e4 groups do not fit the standard 21 B/group LDS image in one
    work-group; count(*) and the zone-map bounded integer sum share ONE 64-bit word
    (gpupreagg_packed_column, 16 B/group) -- bit-exact counts and integer sums, float8 1e-12.
    25000 groups: packed with id-range roles.  num_kern_prep marks the packed path."""
    bufs = [kds.build_kds("column", c4_table(150003, 31 + i, ngroups, xlo=-10**6 + 5 * i)) for i in range(2)]
    pfms = []
    compare_with_oracle(C4_SPEC, bufs, [(0, ngroups)], resident=resident, pfms=pfms)
    assert all(p["num_kern_prep"] == 1 for p in pfms)


def test_packed_accumulators_with_zipf_keys():
    """BASELINE configs[3] with its Zipf-1.0 variant: P(key k) ~ 1 / (k + 1) over 1e4
    keys, the hottest key holds a tenth of the rows -- every lane of a wave adds to the same few LDS words.
    Exact counts and integer sums, float sums at the usual tolerance, and the packed path
    (profiles/r03_c4_zipf.txt: 261 us per 1e8 rows against 247 us with uniform keys)."""
    rng = np.random.default_rng(404)
    n, ngroups = 3_000_017, 10000
    cdf = np.cumsum(1.0 / np.arange(1, ngroups + 1))
    g = np.minimum(np.searchsorted(cdf, rng.random(n) * cdf[-1]), ngroups - 1).astype(np.int32)
    g = rng.permutation(ngroups).astype(np.int32)[g]                # (the hot keys anywhere in the id range)
    cols = [kds.Column("int4", g), kds.Column("int4", rng.integers(-10**6, 10**6, n).astype(np.int32)),
            kds.Column("float8", rng.random(n) * 100)]
    assert np.bincount(g).max() > 0.08 * n
    pfms = []
    compare_with_oracle(C4_SPEC, [kds.build_kds("column", cols)], [(0, ngroups)], resident=True, pfms=pfms,
                        float_tol=1e-11)
    assert pfms[0]["num_kern_prep"] == 1


def test_packed_accumulators_with_qual_two_keys_and_wide_values():
    """a qual, two keys, an int8 column and a float4 -> float8 cast through the packed path;
    then a value range too wide for one word next to the count: the standard path"""
    rng = np.random.default_rng(77)
    n = 120001
    k1 = rng.integers(-50, 150, n).astype(np.int32)
    k2 = rng.integers(0, 60, n).astype(np.int16)
    w = rng.integers(-2**13, 2**13, n).astype(np.int64)        # 15 + 7 value bits + 3 x 14 count bits <= 64
    f = (rng.random(n) * 8 - 4).astype(np.float32)
    v = rng.integers(0, 100, n).astype(np.int32)
    cols = [kds.Column("int4", k1), kds.Column("int2", k2), kds.Column("int8", w), kds.Column("float4", f),
            kds.Column("int4", v)]
    buf = kds.build_kds("column", cols)
    spec = ("(gpupreagg (qual (int4lt (var 5 int4) (const int4 70))) (key (var 1 int4)) (key (var 2 int2)) (nrows)"
            " (psum (var 3 int8)) (psum (float8 (var 4 float4))) (psum (int8 (var 5 int4))))")
    pfms = []
    compare_with_oracle(spec, [buf], [(-50, 200), (0, 60)], resident=True, pfms=pfms, float_tol=1e-9)
    assert pfms[0]["num_kern_prep"] == 1
    wide = [kds.Column("int4", k1), kds.Column("int2", k2),
            kds.Column("int8", rng.integers(-2**50, 2**50, n).astype(np.int64)), kds.Column("float4", f),
            kds.Column("int4", v)]
    pfms = []
    compare_with_oracle(spec, [kds.build_kds("column", wide)], [(-50, 200), (0, 60)], resident=True, pfms=pfms,
                        float_tol=1e-9)
    assert pfms[0]["num_kern_prep"] == 0


@pytest.mark.parametrize("hot", [0.0, 0.4])
def test_packed_accumulators_spill_groups_when_the_fields_are_narrow(monkeypatch, hot):
    """the sum of a full-range int4 column next to the count needs more than 64 bits for all the
    rows of a work-group: narrow fields, the adds return the old word, and the add that brings a
    group's count to a quarter of its field moves the group to the slab (gpupreagg_packed_spill).
    Forced on a modest input by capping the count field at 15 bits (spill at 8192 rows).
    hot = 0: uniform keys, no group ever gets there -- the fold costs what the wide fields cost;
    hot = 0.4: one group holds 40 % of the rows, ~30000 per work-group: it is moved out three
    or four times per work-group while the others keep adding to it."""
    monkeypatch.setenv("STROM_GPUPREAGG_PACK_COUNT_BITS", "15")
    bufs = []
    for i in range(2):
        cols = c4_table(20_000_003 if i == 0 else 1_000_003, 51 + i, 10000, xlo=-2**31, xhi=2**31 - 1)
        if hot:
            g = cols[0].values
            g[np.random.default_rng(7 + i).random(len(g)) < hot] = 4242
        bufs.append(kds.build_kds("column", cols))
    pfms = []
    compare_with_oracle(C4_SPEC, bufs, [(0, 10000)], resident=True, pfms=pfms)
    assert all(p["num_kern_prep"] == 1 for p in pfms)


def test_packed_path_is_left_when_an_input_column_has_nulls_and_catches_a_wrong_zone_map():
    bufs = [kds.build_kds("column", c4_table(100003, 41, 10000, nulls=0.02))]
    pfms = []
    compare_with_oracle(C4_SPEC, bufs, [(0, 10000)], resident=True, pfms=pfms)
    assert pfms[0]["num_kern_prep"] == 0                        # has-value flags needed: standard image
    # a zone map that does not bound the column: the packed fields would overflow into each other
    buf = kds.build_kds("column", c4_table(100003, 43, 10000))
    head = kds.KdsHead(buf)
    off = ((48 + 8 * head.ncols + 15) & ~15) + 32 * 1            # coldir of column 2
    buf[off + 16:off + 32] = np.array([0, 1000], dtype=np.int64).view(np.uint8)
    agg = GpuPreAgg(C4_SPEC).begin([(0, 10000)])
    try:
        with pytest.raises(runtime.StromError) as ei:
            agg.fold(runtime.DeviceStore.upload(buf))
        assert ei.value.errcode == 300                          # StromError_DataStoreCorruption
    finally:
        agg.end()


def test_zone_map_domain_and_empty_chunks():
    cols = random_table(5000, 17)
    buf = kds.build_kds("column", cols)
    dom = domain_of([buf], [0, 1])
    assert dom[0] == (-7, 50) and dom[1] == (0, 3)
    empty = kds.build_kds("column", [kds.Column(c.sqltype, c.values[:0], None) for c in cols])
    compare_with_oracle("(gpupreagg (key (var 1 int4)) (nrows) (psum (var 4 float8)))",
                        [empty, buf, empty], dom[:1])


def test_recheck_chunk_is_not_folded():
    x = np.array([1, 2**31 - 1, 3] * 100, dtype=np.int32)
    bad = kds.build_kds("column", [kds.Column("int4", x)])
    good = kds.build_kds("column", [kds.Column("int4", np.arange(10, dtype=np.int32))])
    spec = "(gpupreagg (nrows) (psum (int8 (int4pl (var 1 int4) (const int4 1)))))"
    agg = GpuPreAgg(spec).begin([])
    assert agg.fold(good)[0] == 0
    assert agg.fold(bad)[0] == 2            # CpuReCheck: contributes nothing
    assert agg.fold(good)[0] == 0
    pr = agg.fetch()
    agg.end()
    assert len(pr) == 1
    assert int(pr.column(0)[0][0]) == 20 and int(pr.column(1)[0][0]) == 2 * sum(range(1, 11))


def test_row_map_input():
    cols = random_table(30000, 23)
    buf = kds.build_kds("row", cols)
    rmap = np.random.default_rng(1).permutation(30000)[:9000].astype(np.int32)
    spec = "(gpupreagg (key (var 2 int2)) (nrows) (psum (int8 (var 3 int4))))"
    agg = GpuPreAgg(spec).begin([(0, 3)])
    assert agg.fold(buf, row_map=rmap)[0] == 0
    got_v, got_n = partial_rows_as_raw8(agg.fetch())
    agg.end()
    rc, v, n = oracle.gpupreagg(spec, buf, 3, row_map=rmap)
    order_g = np.argsort(got_v[:, 0].view(np.int64))
    order_o = np.argsort(v[:, 0].view(np.int64))
    assert np.array_equal(got_v[order_g], v[order_o])


def test_c4_shape_full_size_properties():
    """BASELINE configs[3] shape on one GPU at 1e8 rows: GROUP BY g (1e4 groups)
    COUNT(*), SUM(x int4), AVG(y float8).  Integer results exact against an
    independent numpy reduction, float sums relative 1e-12."""
    n, ngroups = 100_000_000, 10_000
    rng = np.random.default_rng(99)
    g = rng.integers(0, ngroups, n, dtype=np.int64).astype(np.int32)
    x = rng.integers(-10**6, 10**6, n, dtype=np.int64).astype(np.int32)
    y = rng.random(n) * 100
    half = n // 2
    chunks = [runtime.DeviceStore.upload(kds.build_kds("column", [
        kds.Column("int4", g[s]), kds.Column("int4", x[s]), kds.Column("float8", y[s])]))
        for s in (slice(0, half), slice(half, n))]
    spec = "(gpupreagg (key (var 1 int4)) (nrows) (psum (int8 (var 2 int4))) (psum (var 3 float8)))"
    agg = GpuPreAgg(spec).begin([(0, ngroups)])
    for ds in chunks:
        assert agg.fold(ds)[0] == 0
    pr = agg.fetch()
    agg.end()
    for ds in chunks:
        ds.release()
    assert len(pr) == ngroups
    keys = pr.column(0)[0]
    order = np.argsort(keys)
    assert np.array_equal(keys[order], np.arange(ngroups))
    cnt = np.bincount(g, minlength=ngroups)
    sx = np.bincount(g, weights=x.astype(np.float64), minlength=ngroups)   # exact: |sum| < 2^53
    sy = np.bincount(g, weights=y, minlength=ngroups)
    assert np.array_equal(pr.column(1)[0][order], cnt)
    assert np.array_equal(pr.column(2)[0][order], sx.astype(np.int64))
    assert np.allclose(pr.column(3)[0][order], sy, rtol=1e-12, atol=0)


# ---- group-slot agreement: census + compact ----------
SPARSE_SPEC = ("(gpupreagg (qual (int4gt (var 3 int4) (const int4 -900000)))"
               " (key (var 1 int4)) (key (var 2 int2)) (nrows)"
               " (psum (int8 (var 3 int4))) (psum (var 4 float8)) (pmax (var 3 int4)))")


def sparse_table(n, seed):
    """two keys with wide ranges, a dozen combinations that occur"""
    rng = np.random.default_rng(seed)
    combos = [(int(a), int(b)) for a, b in zip(rng.integers(-500, 500, 12), rng.integers(0, 200, 12))]
    pick = rng.integers(0, len(combos), n)
    g = np.array([combos[i][0] for i in pick], dtype=np.int32)
    h = np.array([combos[i][1] for i in pick], dtype=np.int16)
    gn = rng.random(n) < 0.02
    x = rng.integers(-10**6, 10**6, n).astype(np.int32)
    y = rng.random(n)
    return [kds.Column("int4", g, gn), kds.Column("int2", h), kds.Column("int4", x), kds.Column("float8", y)]


@pytest.mark.parametrize("fmt", ["column", "row", "tupslot"])
def test_compacted_slots_give_the_same_partial_rows(fmt):
    cols = sparse_table(60000, 77)
    bufs = [kds.build_kds(fmt, [kds.Column(c.sqltype, c.values[s], None if c.isnull is None else c.isnull[s])
                                for c in cols]) for s in (slice(0, 25000), slice(25000, 60000))]
    domain = [(-500, 1000), (0, 200)]
    compare_with_oracle(SPARSE_SPEC, bufs, domain, compact=True)


def test_census_bitmap_is_the_set_of_dense_ids_that_pass_the_qual():
    cols = sparse_table(40000, 78)
    buf = kds.build_kds("column", cols)
    domain = [(-500, 1000), (0, 200)]
    agg = GpuPreAgg(SPARSE_SPEC).begin(domain)
    try:
        bitmap = agg.census(buf)
        g, h, x = cols[0], cols[1], cols[2]
        ok = x.values > -900000
        off_g = np.where(g.isnull != 0, 1000, g.values.astype(np.int64) + 500)
        dense = (off_g + (h.values.astype(np.int64) - 0) * 1001)[ok]
        want = np.zeros(len(bitmap) * 32, dtype=bool)
        want[np.unique(dense)] = True
        got = np.unpackbits(bitmap.view(np.uint8), bitorder="little").astype(bool)
        assert np.array_equal(got, want)
        ngroups = agg.compact()
        assert ngroups == len(np.unique(dense))
        # the compacted table is small enough for the lane-private kernel
        status, _ = agg.fold(buf)
        assert status == 0
        pr = agg.fetch()
        assert len(pr.column(0)[0]) == ngroups
    finally:
        agg.end()


def test_unmarked_combination_fails_its_chunk_after_compact():
    cols = sparse_table(20000, 79)
    first = kds.build_kds("column", cols)
    agg = GpuPreAgg(SPARSE_SPEC).begin([(-500, 1000), (0, 200)])
    try:
        agg.census(first)
        agg.compact()
        assert agg.fold(first)[0] == 0
        other = [kds.Column("int4", np.array([499], dtype=np.int32)), kds.Column("int2", np.array([199], dtype=np.int16)),
                 kds.Column("int4", np.array([5], dtype=np.int32)), kds.Column("float8", np.array([0.5]))]
        with pytest.raises(runtime.StromError) as ei:
            agg.fold(kds.build_kds("column", other))
        assert ei.value.errcode == 302          # StromError_DataStoreOutOfRange
        # census / compact are planning-time calls: refused once folding began
        with pytest.raises(runtime.StromError):
            agg.compact()
    finally:
        agg.end()


# ---------------------------------------------------------------------------
# hashed GROUP BY (strom_gpupreagg_create_hashed): keys of any type and spread
# ---------------------------------------------------------------------------
def any_key_table(n, seed, nulls=0.03):
    rng = np.random.default_rng(seed)
    f = rng.integers(-4, 5, n).astype(np.float64) / 4            # float8 key incl. -0.0 / +0.0
    f[rng.random(n) < 0.05] = -0.0
    f[rng.random(n) < 0.03] = np.nan                              # NaN is ONE group
    big = (rng.integers(0, 7, n).astype(np.int64) - 3) * (2**40 + 12345)   # int8 key, sparse
    num = kds.numeric_from_scaled(rng.integers(-3, 4, n) * 250, 3, rng.random(n) < nulls)   # numeric key
    x = rng.integers(-10**6, 10**6, n).astype(np.int32)
    y = rng.random(n) * 100
    z = rng.integers(-2, 3, n).astype(np.float32) * 0.5           # float4 key
    return [kds.Column("float8", f, rng.random(n) < nulls), kds.Column("int8", big, rng.random(n) < nulls),
            num, kds.Column("int4", x, rng.random(n) < nulls), kds.Column("float8", y, rng.random(n) < nulls),
            kds.Column("float4", z)]


SPEC_ANY_KEYS = ("(gpupreagg (key (var 1 float8)) (key (var 2 int8)) (nrows) (nrows (isnotnull (var 4 int4)))"
                 " (psum (int8 (var 4 int4))) (pmin (var 4 int4)) (pmax (var 4 int4))"
                 " (psum (var 5 float8)) (pmin (var 5 float8)) (pmax (var 5 float8)))")


@pytest.mark.parametrize("fmt", ["column", "row", "tupslot"])
def test_hashed_float_and_sparse_int8_keys(fmt):
    cols = any_key_table(40000, 31)
    bufs = [kds.build_kds(fmt, [kds.Column(c.sqltype, c.values[i::2], None if c.isnull is None else c.isnull[i::2])
                                for c in cols]) for i in range(2)]
    compare_with_oracle(SPEC_ANY_KEYS, bufs, None, hashed=True)


def test_hashed_numeric_and_float4_keys_with_a_qual():
    cols = any_key_table(30000, 37)
    buf = kds.build_kds("column", cols)
    spec = ("(gpupreagg (qual (int4gt (var 4 int4) (param 0 int4)))"
            " (key (var 3 numeric)) (key (var 6 float4)) (nrows) (psum (int8 (var 4 int4))) (pmax (var 5 float8)))")
    compare_with_oracle(spec, [buf], None, ext=[np.int32(-250000)], hashed=True)


def test_hashed_numeric_key_groups_by_value_not_by_image():
    """1.50 and 1.5 are one group: numeric keys are stripped before hashing"""
    n = 6000
    rng = np.random.default_rng(41)
    k = rng.integers(1, 6, n)
    # the same values in two images: mantissa*10 with exponent-1 (unnormalised)
    norm = kds.numeric_from_scaled(k * 5, 1).values
    mant = norm & np.uint64((1 << 57) - 1)
    expo = (norm.view(np.int64) >> 58)
    loose = (((expo - 1) & 0x3f).astype(np.uint64) << np.uint64(58)) | (mant * np.uint64(10))
    img = np.where(rng.random(n) < 0.5, norm, loose)
    buf = kds.build_kds("column", [kds.Column("numeric", img), kds.Column("int4", np.ones(n, dtype=np.int32))])
    agg = GpuPreAgg("(gpupreagg (key (var 1 numeric)) (nrows))").begin_hashed()
    assert agg.fold(buf)[0] == 0
    pr = agg.fetch()
    agg.end()
    assert len(pr) == 5
    keys, _ = pr.column(0)
    cnt, _ = pr.column(1)
    want = {int(v): int((k == kk).sum()) for kk, v in zip(range(1, 6), kds.numeric_from_scaled(np.arange(1, 6) * 5, 1).values)}
    assert {int(a): int(b) for a, b in zip(keys, cnt)} == want


def test_hashed_table_grows_with_the_group_count():
    """4M distinct int8 keys: more than the first table's fill limit, so rows of
    new groups are deferred, the table grows (device re-insert) and they are
    folded again; three chunks, the same keys again in the last one"""
    rng = np.random.default_rng(43)
    nkeys = 4000000
    universe = rng.permutation(np.arange(nkeys, dtype=np.int64) * 1000003 - 7 * 10**10)
    chunks = [universe[:1500000], universe[1500000:], universe[::3]]
    agg = GpuPreAgg("(gpupreagg (key (var 1 int8)) (nrows) (psum (int8 (var 2 int4))) (pmax (var 2 int4)))").begin_hashed()
    vals = []
    for c in chunks:
        v = rng.integers(-1000, 1000, len(c)).astype(np.int32)
        vals.append(v)
        assert agg.fold(kds.build_kds("column", [kds.Column("int8", c), kds.Column("int4", v)]))[0] == 0
    assert agg.num_groups() == nkeys
    pr = agg.fetch()
    agg.end()
    assert len(pr) == nkeys
    allk = np.concatenate(chunks)
    allv = np.concatenate(vals).astype(np.int64)
    uk, inv = np.unique(allk, return_inverse=True)
    want_cnt = np.bincount(inv)
    want_sum = np.bincount(inv, weights=allv).astype(np.int64)
    want_max = np.full(len(uk), -2**31, dtype=np.int64)
    np.maximum.at(want_max, inv, allv)
    k, _ = pr.column(0)
    order = np.argsort(k)
    assert np.array_equal(k[order], uk)
    assert np.array_equal(pr.column(1)[0][order], want_cnt)
    assert np.array_equal(pr.column(2)[0][order], want_sum)
    assert np.array_equal(pr.column(3)[0][order].astype(np.int64), want_max)


def test_hashed_recheck_chunk_is_not_folded_and_reset():
    x = np.array([1, 2**31 - 1, 3] * 100, dtype=np.int32)
    k = np.arange(300, dtype=np.float64) % 7
    bad = kds.build_kds("column", [kds.Column("int4", x), kds.Column("float8", k)])
    good = kds.build_kds("column", [kds.Column("int4", np.arange(70, dtype=np.int32)),
                                    kds.Column("float8", np.arange(70, dtype=np.float64) % 7)])
    spec = "(gpupreagg (key (var 2 float8)) (nrows) (psum (int8 (int4pl (var 1 int4) (const int4 1)))))"
    agg = GpuPreAgg(spec).begin_hashed()
    assert agg.fold(good)[0] == 0
    assert agg.fold(bad)[0] == 2            # CpuReCheck: contributes nothing, claims no slot
    assert agg.fold(good)[0] == 0
    pr = agg.fetch()
    assert len(pr) == 7 and agg.num_groups() == 7
    assert int(pr.column(1)[0].sum()) == 140 and int(pr.column(2)[0].sum()) == 2 * sum(range(1, 71))
    agg.reset()
    assert agg.num_groups() == 0 and len(agg.fetch()) == 0
    assert agg.fold(good)[0] == 0
    pr = agg.fetch()
    agg.end()
    assert len(pr) == 7 and int(pr.column(1)[0].sum()) == 70


def test_hashed_row_map_inputs_host_and_device():
    from pg_strom_amd.gpuscan import GpuScan
    cols = any_key_table(30000, 47)
    spec = "(gpupreagg (key (var 1 float8)) (nrows) (psum (int8 (var 4 int4))))"
    # host row map over a ROW chunk
    buf = kds.build_kds("row", cols)
    rmap = np.random.default_rng(2).permutation(30000)[:9000].astype(np.int32)
    agg = GpuPreAgg(spec).begin_hashed()
    assert agg.fold(buf, row_map=rmap)[0] == 0
    got_v, got_n = partial_rows_as_raw8(agg.fetch())
    agg.end()
    rc, v, n = oracle.gpupreagg(spec, buf, 3, row_map=rmap)
    assert rc == 0
    order_g = np.lexsort((got_v[:, 0], got_n[:, 0]))
    order_o = np.lexsort((v[:, 0], n[:, 0]))
    assert np.array_equal(got_v[order_g], v[order_o]) and np.array_equal(got_n[order_g], n[order_o])
    # device row map: GpuScan -> GpuPreAgg without leaving HBM
    colbuf = kds.build_kds("column", cols)
    store = runtime.DeviceStore.upload(colbuf)
    qual = "(int4gt (var 4 int4) (const int4 0))"
    scan = GpuScan(qual).begin()
    rowmap, _ = scan.scan_to_rowmap(store)
    agg = GpuPreAgg(spec).begin_hashed()
    assert agg.fold(store, row_map=rowmap)[0] == 0
    got_v, got_n = partial_rows_as_raw8(agg.fetch())
    agg.end()
    scan.end()
    rowmap.release()
    store.release()
    rc, sel = oracle.gpuscan(qual, colbuf, [])
    rc, v, n = oracle.gpupreagg(spec, colbuf, 3, row_map=(np.sort(sel) - 1).astype(np.int32))   # results[] are 1-based
    assert rc == 0
    order_g = np.lexsort((got_v[:, 0], got_n[:, 0]))
    order_o = np.lexsort((v[:, 0], n[:, 0]))
    assert np.array_equal(got_v[order_g], v[order_o]) and np.array_equal(got_n[order_g], n[order_o])


def test_hashed_session_refuses_the_dense_table_calls():
    agg = GpuPreAgg("(gpupreagg (key (var 1 float8)) (nrows))").begin_hashed()
    from pg_strom_amd._lib import lib
    assert lib.strom_gpupreagg_table_length(agg.session) == 0
    assert lib.strom_gpupreagg_dense_groups(agg.session) == 0
    with pytest.raises(runtime.StromError):
        agg.compact()
    assert len(agg.fetch()) == 0            # nothing folded yet
    agg.end()
    # and the dense path still refuses what only the hashed one can group
    with pytest.raises(runtime.StromError):
        GpuPreAgg("(gpupreagg (key (var 1 float8)) (nrows))").begin([(0, 10)])


@pytest.mark.parametrize("hint,qual", [(0, False), (5000, False), (5000, True)])
def test_hashed_roles_split_the_groups_over_lds_tables(hint, qual):
    """5000 float8 keys: with the hint (or, without it, from the second chunk on,
    when the group count is known) the work-groups take hash roles so that a
    role's groups fit its LDS table; the result must not depend on it.  With roles the
    check pass leaves a one-byte role map per row (rows the qual drops: no role) and the
    roles scan that instead of the key column."""
    rng = np.random.default_rng(53)
    n = 400000
    key = rng.integers(0, 5000, n).astype(np.float64) * 0.25 - 300.0
    x = rng.integers(-1000, 1000, n).astype(np.int32)
    isn = rng.random(n) < 0.02
    agg = GpuPreAgg("(gpupreagg " + ("(qual (int4gt (var 2 int4) (const int4 -600))) " if qual else "") +
                    "(key (var 1 float8)) (nrows) (psum (int8 (var 2 int4))) (pmin (var 2 int4)))")
    agg.begin_hashed(ngroups_hint=hint)
    for i in range(2):
        sl = slice(i * n // 2, (i + 1) * n // 2)
        buf = kds.build_kds("column", [kds.Column("float8", key[sl]), kds.Column("int4", x[sl], isn[sl])])
        assert agg.fold(buf)[0] == 0
    pr = agg.fetch()
    agg.end()
    if qual:
        keep = (~isn) & (x > -600)                  # a NULL x makes the qual NULL: row dropped
        key, x, isn = key[keep], x[keep], isn[keep]
    uk, inv = np.unique(key, return_inverse=True)
    assert len(pr) == len(uk)
    k, _ = pr.column(0)
    order = np.argsort(k)
    assert np.array_equal(k[order], uk)
    assert np.array_equal(pr.column(1)[0][order], np.bincount(inv))
    xs = np.where(isn, 0, x).astype(np.int64)
    assert np.array_equal(pr.column(2)[0][order], np.bincount(inv, weights=xs).astype(np.int64))
    want_min = np.full(len(uk), 2**31 - 1, dtype=np.int64)
    np.minimum.at(want_min, inv[~isn], x[~isn].astype(np.int64))
    got_min, got_null = pr.column(3)
    has = want_min != 2**31 - 1
    assert np.array_equal(got_null[order], ~has)
    assert np.array_equal(got_min[order][has].astype(np.int64), want_min[has])


# ---------------------------------------------------------------------------
# hashed GROUP BY over hash partitions (gpupreagg_hash_check_parts .. _fold_parts): the plan
# for more groups than the hash roles' LDS tables take.  STROM_GPUPREAGG_HASH_PARTS_MIN=0
# sends every chunk through it, whatever the group count; num_kern_prep reports it.
# ---------------------------------------------------------------------------
@pytest.fixture
def partitions(monkeypatch):
    monkeypatch.setenv("STROM_GPUPREAGG_HASH_PARTS_MIN", "0")


@pytest.mark.parametrize("fmt", ["column", "row", "tupslot"])
def test_partition_plan_any_keys_every_partial_kind(partitions, fmt):
    cols = any_key_table(40000, 61)
    bufs = [kds.build_kds(fmt, [kds.Column(c.sqltype, c.values[i::2], None if c.isnull is None else c.isnull[i::2])
                                for c in cols]) for i in range(2)]
    pfms = []
    compare_with_oracle(SPEC_ANY_KEYS, bufs, None, hashed=True, pfms=pfms)
    assert all(p["num_kern_prep"] == 1 for p in pfms)
    spec = ("(gpupreagg (qual (int4gt (var 4 int4) (param 0 int4)))"
            " (key (var 3 numeric)) (key (var 6 float4)) (nrows) (psum (int8 (var 4 int4))) (pmax (var 5 float8)))")
    if fmt == "column":
        compare_with_oracle(spec, bufs, None, ext=[np.int32(-250000)], hashed=True, resident=True)


@pytest.mark.parametrize("lds_scatter", [True, False])
def test_partition_plan_claims_new_groups_and_grows_the_table(partitions, monkeypatch, lds_scatter):
    """4M distinct keys from an empty table: units with a new group beyond the fill limit come back
    on the redo list, the table grows; 256 partitions of ~6000 new keys each overflow the units' LDS
    tables, so the second pass over a unit (records the LDS table does not know) runs too.  Also
    with the direct scatter (records too long for an LDS tile take it)."""
    if not lds_scatter:
        monkeypatch.setenv("STROM_GPUPREAGG_HASH_NO_LDS_SCATTER", "1")
    test_hashed_table_grows_with_the_group_count()


def test_partition_plan_row_maps_and_recheck(partitions):
    test_hashed_row_map_inputs_host_and_device()
    test_hashed_recheck_chunk_is_not_folded_and_reset()


@pytest.mark.parametrize("unit_rows", [1024, 32768])
def test_partition_plan_cuts_a_heavy_key_into_units(partitions, monkeypatch, unit_rows):
    """60 % of the rows carry one key: its partition is cut into units of unit_rows records,
    several work-groups fold it and meet in the global table"""
    monkeypatch.setenv("STROM_GPUPREAGG_HASH_UNIT_ROWS", str(unit_rows))
    rng = np.random.default_rng(67)
    n = 600000
    key = (rng.integers(0, 50000, n).astype(np.int64) - 25000) * (2**33 + 7)
    key[rng.random(n) < 0.6] = 123456789012345
    knull = rng.random(n) < 0.01                              # NULL is a group of its own
    x = rng.integers(-1000, 1000, n).astype(np.int32)
    y = rng.random(n)
    spec = "(gpupreagg (key (var 1 int8)) (nrows) (psum (int8 (var 2 int4))) (pmin (var 3 float8)) (psum (var 3 float8)))"
    agg = GpuPreAgg(spec).begin_hashed()
    for i in range(3):
        sl = slice(i * n // 3, (i + 1) * n // 3)
        st, pfm = agg.fold(kds.build_kds("column", [kds.Column("int8", key[sl], knull[sl]), kds.Column("int4", x[sl]),
                                                    kds.Column("float8", y[sl])]))
        assert st == 0 and pfm["num_kern_prep"] == 1
    pr = agg.fetch()
    agg.end()
    kk = np.where(knull, np.int64(2**62), key)               # (no real key has this value)
    uk, inv = np.unique(kk, return_inverse=True)
    k, kn = pr.column(0)
    k = np.where(kn, np.int64(2**62), k)
    order = np.argsort(k)
    assert len(pr) == len(uk) and np.array_equal(k[order], uk)
    assert np.array_equal(pr.column(1)[0][order], np.bincount(inv))
    assert np.array_equal(pr.column(2)[0][order], np.bincount(inv, weights=x.astype(np.float64)).astype(np.int64))
    want_min = np.full(len(uk), np.inf)
    np.minimum.at(want_min, inv, y)
    assert np.array_equal(pr.column(3)[0][order], want_min)
    assert np.allclose(pr.column(4)[0][order], np.bincount(inv, weights=y), rtol=1e-11)


def test_partition_plan_for_a_large_chunk_of_unknown_group_count():
    """no hint, first chunk, 4.5M rows: the plan whose cost does not depend on the group count;
    the second chunk knows there are 300 groups and goes back to the LDS table"""
    rng = np.random.default_rng(73)
    n = 4_500_000
    key = (rng.integers(0, 300, n).astype(np.int64) - 150) * (2**35 + 3)
    x = rng.integers(-1000, 1000, n).astype(np.int32)
    buf = kds.build_kds("column", [kds.Column("int8", key), kds.Column("int4", x)])
    agg = GpuPreAgg("(gpupreagg (key (var 1 int8)) (nrows) (psum (int8 (var 2 int4))) (pmin (var 2 int4)))").begin_hashed()
    try:
        st, pfm = agg.fold(buf)
        assert st == 0 and pfm["num_kern_prep"] == 1
        st, pfm = agg.fold(buf)
        assert st == 0 and pfm["num_kern_prep"] == 0
        pr = agg.fetch()
    finally:
        agg.end()
    uk, inv = np.unique(key, return_inverse=True)
    order = np.argsort(pr.column(0)[0])
    assert len(pr) == 300 and np.array_equal(pr.column(0)[0][order], uk)
    assert np.array_equal(pr.column(1)[0][order], 2 * np.bincount(inv))
    assert np.array_equal(pr.column(2)[0][order], 2 * np.bincount(inv, weights=x.astype(np.float64)).astype(np.int64))


def test_partition_plan_is_the_default_beyond_the_roles(monkeypatch):
    """a hint of 30000 groups: the partition plan without any knob; the same partial rows as
    the global-table path gives (integers bit for bit)"""
    rng = np.random.default_rng(71)
    n = 500000
    key = rng.integers(0, 30000, n).astype(np.float64) * 0.5 - 7000.0
    x = rng.integers(-1000, 1000, n).astype(np.int32)
    buf = kds.build_kds("column", [kds.Column("float8", key), kds.Column("int4", x, rng.random(n) < 0.02)])
    spec = "(gpupreagg (key (var 1 float8)) (nrows) (psum (int8 (var 2 int4))) (pmax (var 2 int4)))"
    rows = []
    for parts in (True, False):
        if not parts:
            monkeypatch.setenv("STROM_GPUPREAGG_HASH_NO_PARTS", "1")
        agg = GpuPreAgg(spec).begin_hashed(ngroups_hint=30000)
        st, pfm = agg.fold(buf)
        assert st == 0 and pfm["num_kern_prep"] == (1 if parts else 0)
        v, isn = partial_rows_as_raw8(agg.fetch())
        agg.end()
        o = np.argsort(v[:, 0].view(np.float64))
        rows.append((v[o], isn[o]))
    assert len(rows[0][0]) == len(np.unique(key))
    assert np.array_equal(rows[0][0], rows[1][0]) and np.array_equal(rows[0][1], rows[1][1])


# ---------------------------------------------------------------------------
# merge of hashed sessions: the groups travel (export -> import), on one device here; between
# GPUs strom_gpupreagg_allreduce all-gathers the exported records and runs the same import
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("plan", ["table", "partitions"])
def test_hashed_sessions_merge_on_the_device(plan, monkeypatch):
    """two sessions fold two chunks whose key sets overlap in part (every partial kind, NULL keys,
    NaN keys, groups whose inputs are all NULL on one side); a.merge_from(b) == one session that
    folded both chunks == the oracle; b is unchanged; merging into an EMPTY session copies"""
    if plan == "partitions":
        monkeypatch.setenv("STROM_GPUPREAGG_HASH_PARTS_MIN", "0")
    cols = any_key_table(60000, 83)
    half = [kds.build_kds("column", [kds.Column(c.sqltype, c.values[sl], None if c.isnull is None else c.isnull[sl])
                                      for c in cols]) for sl in (slice(0, 35000), slice(25000, 60000))]
    a = GpuPreAgg(SPEC_ANY_KEYS).begin_hashed()
    b = GpuPreAgg(SPEC_ANY_KEYS).begin_hashed()
    empty = GpuPreAgg(SPEC_ANY_KEYS).begin_hashed()
    try:
        assert a.fold(half[0])[0] == 0 and b.fold(half[1])[0] == 0
        nb = b.num_groups()
        a.merge_from(b)
        assert b.num_groups() == nb
        assert_matches_oracle(SPEC_ANY_KEYS, a, half, a.fetch())
        assert_matches_oracle(SPEC_ANY_KEYS, b, half[1:], b.fetch())
        empty.merge_from(b)
        assert_matches_oracle(SPEC_ANY_KEYS, empty, half[1:], empty.fetch())
        with pytest.raises(runtime.StromError):
            a.merge_from(a)
    finally:
        a.end(); b.end(); empty.end()


def test_hashed_merge_of_many_groups_grows_the_table():
    """3e5 + 3e5 sparse int8 keys, half of them shared: the destination table grows before the
    import; counts and sums are numpy's"""
    rng = np.random.default_rng(89)
    universe = rng.permutation(np.arange(450000, dtype=np.int64) * 1000003 - 5 * 10**10)
    parts = [universe[:300000], universe[150000:]]
    spec = "(gpupreagg (key (var 1 int8)) (nrows) (psum (int8 (var 2 int4))) (pmin (var 2 int4)))"
    sess, vals = [], []
    try:
        for p in parts:
            v = rng.integers(-1000, 1000, len(p)).astype(np.int32)
            vals.append(v)
            s_ = GpuPreAgg(spec).begin_hashed()
            sess.append(s_)
            assert s_.fold(kds.build_kds("column", [kds.Column("int8", p), kds.Column("int4", v)]))[0] == 0
        sess[0].merge_from(sess[1])
        pr = sess[0].fetch()
    finally:
        for s_ in sess:
            s_.end()
    allk, allv = np.concatenate(parts), np.concatenate(vals).astype(np.int64)
    uk, inv = np.unique(allk, return_inverse=True)
    k, _ = pr.column(0)
    order = np.argsort(k)
    assert len(pr) == 450000 and np.array_equal(k[order], uk)
    assert np.array_equal(pr.column(1)[0][order], np.bincount(inv))
    assert np.array_equal(pr.column(2)[0][order], np.bincount(inv, weights=allv).astype(np.int64))
    want_min = np.full(len(uk), 2**31, dtype=np.int64)
    np.minimum.at(want_min, inv, allv)
    assert np.array_equal(pr.column(3)[0][order].astype(np.int64), want_min)


# ---------------------------------------------------------------------------
# the reference's per-chunk message (strom_submit_gpupreagg_chunk)
# ---------------------------------------------------------------------------
def chunk_runner(chunks):
    """like hip_runner, but every chunk is ONE pgstrom_gpupreagg message: partial rows come
    back per chunk in the caller's kds_dest and are concatenated -- the reference's Agg
    node adds them up (pg_strom--1.0.sql:247-401)"""
    cache = {}

    def run(plan):
        sup, picks = superset_plan(plan)
        key = (sup["spec"], plan["table"])
        if key not in cache:
            vals, nulls = [], []
            agg = GpuPreAgg(sup["spec"])
            nt = len(agg.targets)
            for buf, rows in chunks[plan["table"]]:
                status, pr = agg.collect_chunk(agg.submit_chunk(buf))
                if status == 2:
                    assert plan["type"] == "int8"       # int8 partial sum overflow
                    v, n = agg_golden.cpu_fallback_rows(sup, rows)
                else:
                    v, n = partial_rows_as_raw8(pr)
                vals.append(v.reshape(-1, nt))
                nulls.append(n.reshape(-1, nt))
            cache[key] = (np.concatenate(vals), np.concatenate(nulls))
        v, n = cache[key]
        return v[:, picks], n[:, picks]
    return run


@pytest.mark.parametrize("fmt,nchunks", [("row", 3), ("column", 2)])
def test_reference_regression_suites_through_the_chunk_message(fmt, nchunks):
    """the reference's own regression answers with every chunk sent as the reference's own
    message {kern_gpupreagg, pds, pds_dest}: no session, no domain from the caller"""
    chunks = {"gpupreagg_test": agg_golden.fixture_chunks(fmt, nchunks),
              "gpupreagg_zero_test": agg_golden.fixture_chunks(fmt, 1, empty=True)}
    run = chunk_runner(chunks)
    total = 0
    for suite in ("nogrp_agg", "group_agg", "where_agg", "zero_agg"):
        checked, _ = agg_golden.check_suite(suite, run)
        total += checked
    assert total >= 200


@pytest.mark.parametrize("fmt", ["column", "row_flat", "tupslot"])
def test_chunk_message_matches_oracle_per_chunk(fmt):
    """two keys (one with NULLs and negative values), every partial kind; the library finds
    the key ranges itself (gpupreagg_keyrange)"""
    cols = random_table(50000, 77)
    agg = GpuPreAgg(SPEC_ALL)
    for i in range(2):
        buf = kds.build_kds(fmt, [kds.Column(c.sqltype, c.values[i::2],
                                              None if c.isnull is None else c.isnull[i::2]) for c in cols])
        assert agg.chunk_domain(buf) == [(-7, 50), (0, 3)]
        status, pr = agg.collect_chunk(agg.submit_chunk(buf))
        assert status == 0
        assert_matches_oracle(SPEC_ALL, agg, [buf], pr)


def test_chunk_message_row_map_resident_chunk_and_callback():
    import threading
    from pg_strom_amd._lib import DONE_CB
    cols = random_table(30000, 3)
    buf = kds.build_kds("column", cols)
    spec = ("(gpupreagg (qual (int4gt (var 3 int4) (param 0 int4)))"
            " (key (var 2 int2)) (nrows) (psum (var 4 float8)) (pmax (var 3 int4)))")
    agg = GpuPreAgg(spec)
    rm = np.arange(0, 30000, 3, dtype=np.int32)
    ds = runtime.DeviceStore.upload(buf)
    seen = []
    fired = threading.Event()

    def on_done(arg, errcode, pfm):
        seen.append((errcode, threading.get_ident()))
        fired.set()
    cb = DONE_CB(on_done)
    try:
        pending = agg.submit_chunk(ds, row_map=rm, ext_params=[np.int32(0)], done=cb)
        assert fired.wait(60)
        status, pr = agg.collect_chunk(pending)
    finally:
        ds.release()
    assert status == 0 and seen == [(0, seen[0][1])] and seen[0][1] != threading.get_ident()
    assert_matches_oracle(spec, agg, [buf], pr, ext=[np.int32(0)], row_maps=[rm])


def test_chunk_message_keys_without_dense_ids_take_the_hashed_table():
    """float8 key and a sparse int8 key: no dense ids -> the library goes to the hashed
    GROUP BY by itself; results are the oracle's"""
    rng = np.random.default_rng(8)
    n = 40000
    f = rng.integers(0, 40, n).astype(np.float64) / 4
    f[::97] = np.nan
    big = rng.integers(0, 50, n).astype(np.int64) * (1 << 40) - (1 << 45)
    x = rng.integers(-1000, 1000, n).astype(np.int32)
    buf = kds.build_kds("column", [kds.Column("float8", f), kds.Column("int8", big), kds.Column("int4", x)])
    for spec in ("(gpupreagg (key (var 1 float8)) (nrows) (psum (int8 (var 3 int4))))",
                 "(gpupreagg (key (var 2 int8)) (nrows) (pmin (var 3 int4)) (pmax (var 3 int4)))"):
        agg = GpuPreAgg(spec)
        with pytest.raises(runtime.StromError) as ei:
            agg.chunk_domain(buf)
        assert ei.value.errcode == 302          # DataStoreOutOfRange
        status, pr = agg.collect_chunk(agg.submit_chunk(buf, num_groups=64))
        assert status == 0
        rc, v, nn = oracle.gpupreagg(spec, buf, len(agg.targets))
        got_v, got_n = partial_rows_as_raw8(pr)
        assert rc == 0 and len(got_v) == len(v)
        want = sorted((tuple(int(a) for a in v[i]), tuple(bool(b) for b in nn[i])) for i in range(len(v)))
        got = sorted((tuple(int(a) for a in got_v[i]), tuple(bool(b) for b in got_n[i])) for i in range(len(v)))
        assert got == want


def test_chunk_message_with_very_many_dense_ids_takes_the_partition_plan(monkeypatch):
    """two int4 keys whose ranges multiply to 1e6 ids: the dense kernels would split them over
    dozens of id-range roles, each reading every row; the message goes to the hashed GROUP BY's
    partition plan instead (num_kern_prep reports it) -- and gives the dense path's partial rows"""
    rng = np.random.default_rng(15)
    n = 700000
    k1 = rng.integers(-500, 500, n).astype(np.int32)
    k2 = rng.integers(0, 1000, n).astype(np.int32)
    x = rng.integers(-1000, 1000, n).astype(np.int32)
    buf = kds.build_kds("column", [kds.Column("int4", k1), kds.Column("int4", k2), kds.Column("int4", x, rng.random(n) < 0.02)])
    spec = "(gpupreagg (key (var 1 int4)) (key (var 2 int4)) (nrows) (psum (int8 (var 3 int4))) (pmax (var 3 int4)))"
    rows = []
    for dense_only in (False, True):
        if dense_only:
            monkeypatch.setenv("STROM_GPUPREAGG_CHUNK_DENSE_ONLY", "1")
        agg = GpuPreAgg(spec)
        pending = agg.submit_chunk(buf, dest_rooms=n)
        status, pr = agg.collect_chunk(pending)
        assert status == 0
        v, isn = partial_rows_as_raw8(pr)
        o = np.lexsort((v[:, 1].view(np.int64), v[:, 0].view(np.int64)))
        rows.append((v[o], isn[o]))
    assert np.array_equal(rows[0][0], rows[1][0]) and np.array_equal(rows[0][1], rows[1][1])
    uk, inv = np.unique(k1.astype(np.int64) * 4096 + k2, return_inverse=True)
    assert len(rows[0][0]) == len(uk) and np.array_equal(rows[0][0][:, 2].view(np.int64), np.bincount(inv))


def test_chunk_message_recheck_and_no_space():
    n = 5000
    g = (np.arange(n) % 100).astype(np.int32)
    x = np.full(n, (1 << 62), dtype=np.int64)           # int8 sum overflows -> CpuReCheck
    buf = kds.build_kds("row", [kds.Column("int4", g), kds.Column("int8", x)])
    spec = "(gpupreagg (key (var 1 int4)) (nrows) (psum (var 2 int8)))"
    agg = GpuPreAgg(spec)
    status, pr = agg.collect_chunk(agg.submit_chunk(buf))
    assert status == 2 and pr is None
    # 100 groups do not fit a 10-row kds_dest
    spec2 = "(gpupreagg (key (var 1 int4)) (nrows))"
    agg2 = GpuPreAgg(spec2)
    with pytest.raises(runtime.StromError) as ei:
        agg2.collect_chunk(agg2.submit_chunk(buf, dest_rooms=10))
    assert ei.value.errcode == 301              # DataStoreNoSpace
    status, pr = agg2.collect_chunk(agg2.submit_chunk(buf, dest_rooms=100))
    assert status == 0 and len(pr) == 100


@pytest.mark.parametrize("formal", [False, True])
def test_hashed_claims_and_probes_race_across_the_chip(monkeypatch, formal):
    """the global table's publish protocol under contention (strom_common.h: STROM_PUBLISH_STATE /
    STROM_PROBE_STATE): every work-group of the chip finds or claims the SAME few thousand keys at
    the same time -- a 64-slot LDS table in front leaves nearly every row to the global path, no
    partition plan, no roles.  A probe that saw 'ready' with stale keys would claim a second slot
    for a key: more groups than keys, sums split.  Both forms must give exactly the distinct keys
    and exact sums: the fast one (payload stores acknowledged, then the state) and the formal
    RELEASE / ACQUIRE one (STROM_FORMAL_PUBLISH=1, a program of its own)."""
    monkeypatch.setenv("STROM_GPUPREAGG_HASH_LDS_SLOTS", "64")
    monkeypatch.setenv("STROM_GPUPREAGG_HASH_NO_PARTS", "1")
    monkeypatch.setenv("STROM_GPUPREAGG_HASH_ROLES", "1")
    if formal:
        monkeypatch.setenv("STROM_FORMAL_PUBLISH", "1")
    rng = np.random.default_rng(41)
    n, nkeys = 3000000, 3000
    keys = rng.integers(-2**62, 2**62, nkeys)
    k2 = rng.integers(0, 7, nkeys).astype(np.int32)
    pick = rng.integers(0, nkeys, n)
    x = rng.integers(-10**6, 10**6, n).astype(np.int32)
    buf = kds.build_kds("column", [kds.Column("int8", keys[pick]), kds.Column("int4", k2[pick]), kds.Column("int4", x)])
    spec = "(gpupreagg (key (var 1 int8)) (key (var 2 int4)) (nrows) (psum (int8 (var 3 int4))) (pmax (var 3 int4)))"
    agg = GpuPreAgg(spec).begin_hashed()
    try:
        for _ in range(3):
            assert agg.fold(buf)[0] == 0
        assert agg.num_groups() == nkeys
        pr = agg.fetch()
    finally:
        agg.end()
    assert len(pr) == nkeys
    order = np.argsort(pr.column(0)[0])
    kord = np.argsort(keys)
    assert np.array_equal(pr.column(0)[0][order], keys[kord]) and np.array_equal(pr.column(1)[0][order], k2[kord])
    cnt = np.bincount(pick, minlength=nkeys)
    sx = np.bincount(pick, weights=x.astype(np.float64), minlength=nkeys).astype(np.int64)
    mx = np.full(nkeys, -2**31, dtype=np.int64)
    np.maximum.at(mx, pick, x)
    assert np.array_equal(pr.column(2)[0][order], 3 * cnt[kord])
    assert np.array_equal(pr.column(3)[0][order], 3 * sx[kord])
    assert np.array_equal(pr.column(4)[0][order].astype(np.int64), mx[kord])


def test_hashed_merge_into_an_empty_session_then_fold_many_groups():
    """a table made by strom_gpupreagg_merge before the session's first fold has the initial
    65536 slots; the fold that follows brings 150000 new groups: it must grow the table first
    (round 2's fill limit -- slots * 7/8 minus the fold's headroom -- went below zero, unsigned,
    for such a table: never deferred, never grew, ended in DataStoreNoSpace)"""
    rng = np.random.default_rng(42)
    spec = "(gpupreagg (key (var 1 int8)) (nrows) (psum (int8 (var 2 int4))))"
    small_k = rng.integers(-2**60, 2**60, 500)
    small = kds.build_kds("column", [kds.Column("int8", small_k[rng.integers(0, 500, 20000)]),
                                     kds.Column("int4", np.ones(20000, dtype=np.int32))])
    many_k = rng.integers(-2**60, 2**60, 150000)
    many = kds.build_kds("column", [kds.Column("int8", many_k[np.arange(600000) % 150000]),
                                    kds.Column("int4", np.full(600000, 2, dtype=np.int32))])
    src = GpuPreAgg(spec).begin_hashed()
    dst = GpuPreAgg(spec).begin_hashed()
    try:
        assert src.fold(small)[0] == 0
        dst.merge_from(src)                      # the table is made here
        assert dst.num_groups() == len(np.unique(small_k))
        assert dst.fold(many)[0] == 0
        pr = dst.fetch()
    finally:
        src.end()
        dst.end()
    assert len(pr) == len(np.unique(np.concatenate([small_k, many_k])))
    assert int(pr.column(1)[0].sum()) == 620000 and int(pr.column(2)[0].sum()) == 20000 + 2 * 600000


@pytest.mark.parametrize("ngroups", [7, 300])
def test_counts_that_coincide_in_a_chunk_without_nulls_are_added_once(ngroups):
    """four nrows() -- three of them "column X is not NULL" -- are ONE counter while a chunk has no NULL
    bitmap at all (GPUPREAGG_COUNTALL_<a>, ROWFLAG_ALL_NOTNULL: one LDS atomic instead of four, the others
    copied when the slab is written); the session's second chunk HAS NULLs in two of the columns and its
    counters differ.  Both against the oracle, LDS-atomics and lane-private geometry."""
    rng = np.random.default_rng(88)
    spec = ("(gpupreagg (qual (int4gt (var 2 int4) (const int4 -900))) (key (var 1 int4))"
            " (nrows (isnotnull (var 2 int4))) (psum (int8 (var 2 int4))) (nrows (isnotnull (var 3 int8)))"
            " (nrows) (psum (var 3 int8)) (nrows (isnotnull (var 4 float8))) (nrows (isnotnull (var 2 int4)) (isnotnull (var 3 int8))))")
    assert "#define GPUPREAGG_COUNTALL_FIRST 0" in GpuPreAgg(spec).codegen.source
    bufs = []
    for i, nulls in enumerate((None, 0.1)):
        n = 200003
        g = rng.integers(0, ngroups, n).astype(np.int32)
        x = rng.integers(-1000, 1000, n).astype(np.int32)
        z = rng.integers(-10**12, 10**12, n).astype(np.int64)
        y = rng.random(n)
        bufs.append(kds.build_kds("column", [kds.Column("int4", g), kds.Column("int4", x),
                                             kds.Column("int8", z, None if nulls is None else rng.random(n) < nulls),
                                             kds.Column("float8", y, None if nulls is None else rng.random(n) < nulls)]))
    compare_with_oracle(spec, bufs, [(0, ngroups)], resident=True)
    compare_with_oracle(spec, bufs[:1], [(0, ngroups)])


def test_chunk_messages_of_heap_pages_reuse_the_key_domain_and_recover_when_it_is_wrong():
    """heap-page chunks have no zone maps: the per-chunk message tries the key domain the previous
    message of the program measured (a hint) before it measures again.  Chunk 2's keys lie outside
    chunk 1's range -- the fold answers DataStoreOutOfRange inside the request, the range is measured
    and the chunk folded again; chunk 3 fits the widened domain; the partial rows are the oracle's
    every time, whatever the hint was"""
    spec = "(gpupreagg (qual (int4ge (var 2 int4) (const int4 -500))) (key (var 1 int4)) (nrows) (psum (int8 (var 2 int4))) (pmin (var 2 int4)))"
    rng = np.random.default_rng(77)
    agg = GpuPreAgg(spec)
    for lo, hi in ((100, 140), (-3000, -2950), (-2990, 120), (100000, 100003)):
        n = 40000
        g = rng.integers(lo, hi, n).astype(np.int32)
        x = rng.integers(-1000, 1000, n).astype(np.int32)
        buf = kds.build_kds("row", [kds.Column("int4", g, rng.random(n) < 0.01), kds.Column("int4", x, rng.random(n) < 0.05)])
        status, pr = agg.collect_chunk(agg.submit_chunk(buf))
        assert status == 0
        assert_matches_oracle(spec, agg, [buf], pr)
