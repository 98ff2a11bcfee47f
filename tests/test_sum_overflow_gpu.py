"""
Integer partial sums never wrap silently (needs an MI355X: -m gpu).

Reference: CHECK_OVERFLOW_INT on every accumulate of GPUPREAGG_AGGCALC_PSUM_TEMPLATE
(opencl_gpupreagg.h:142-143, 933-948) -> StromError_CpuReCheck, the chunk is aggregated on the
CPU (gpupreagg.c:2507-2607, 2746-2750).  This build (strom_gpupreagg.h, "integer sums never
wrap"): a range proof (rows x largest input magnitude < 2^63) lets the unchecked kernels run;
a chunk that cannot be proven is folded again by the GPUPREAGG_CHECKED program, add by add;
the resident table keeps such sums 128 bits wide and the fetch hands a total beyond int8 out
as several partial rows.

Every case: device status == oracle status, and when both are 0 the device's partial rows add
up to Python's big-integer answer.  Inputs of one group share a sign where a sum overflows --
for mixed signs the reference's own answer depends on its reduction order.
"""
import numpy as np
import pytest

import oracle_binding as oracle
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpupreagg import GpuPreAgg, KIND_KEY, KIND_NROWS

pytestmark = pytest.mark.gpu

I64_MAX = (1 << 63) - 1
I64_MIN = -(1 << 63)
SPEC = "(gpupreagg (key (var 1 int4)) (nrows) (psum (var 2 int8)) (pmax (var 2 int8)))"
SPEC_NOKEY = "(gpupreagg (nrows) (psum (var 2 int8)) (pmin (var 2 int8)))"


def totals(prs, targets):
    """partial rows (several fetches, several rows per group) -> {key: [python ints per target]}:
    nrows and psum add up, pmin / pmax combine"""
    out = {}
    for pr in prs:
        cols = [pr.column(t) for t in range(len(targets))]
        for i in range(len(pr)):
            key = tuple((None if cols[t][1][i] else int(cols[t][0][i]))
                        for t, (k, _) in enumerate(targets) if k == KIND_KEY)
            acc = out.setdefault(key, [None] * len(targets))
            for t, (k, _) in enumerate(targets):
                if k == KIND_KEY or cols[t][1][i]:
                    continue
                v = int(cols[t][0][i])
                if acc[t] is None:
                    acc[t] = v
                elif k in (KIND_NROWS, 3):
                    acc[t] += v
                else:
                    acc[t] = min(acc[t], v) if k == 4 else max(acc[t], v)
    return out


def expected(g, x, with_key=True, minmax=max):
    want = {}
    for key, v in zip(g if with_key else [None] * len(x), x):
        k = (int(key),) if with_key else ()
        a = want.setdefault(k, [0, 0, None])
        a[0] += 1
        a[1] += int(v)
        a[2] = int(v) if a[2] is None else minmax(a[2], int(v))
    return want


def oracle_status(spec, buf, nt):
    rc, _, _ = oracle.gpupreagg(spec, buf, nt)
    return rc


def table(g, x, fmt):
    return kds.build_kds(fmt, [kds.Column("int4", np.asarray(g, dtype=np.int32)),
                               kds.Column("int8", np.asarray(x, dtype=np.int64))])


def run_dense(spec, bufs, domain, hashed=False):
    agg = GpuPreAgg(spec)
    if hashed:
        agg.begin_hashed()
    else:
        agg.begin(domain)
    try:
        statuses = [agg.fold(b)[0] for b in bufs]
        pr = agg.fetch()
        LAST["checked_folds"] = agg.checked_folds()
    finally:
        agg.end()
    return statuses, pr, agg.targets


LAST = {}       # which tier the last run_dense() took: folds by the GPUPREAGG_CHECKED program


# the shapes of the dense path: LDS atomics with and without replicas, lane-private LDS
# accumulators, one datum at a time over heap tuples, the hashed table
VARIANTS = [("column", 400, False), ("column", 5, False), ("row", 40, False), ("tupslot", 3000, False),
            ("column", 300, True), ("row_flat", 300, True)]


@pytest.mark.parametrize("fmt,ngroups,hashed", VARIANTS)
def test_sum_that_leaves_int8_is_cpu_recheck(fmt, ngroups, hashed):
    """five rows of 2^61 in ONE group (= 2^63 + 2^61) among ordinary rows"""
    rng = np.random.default_rng(3)
    n = 50000
    g = rng.integers(0, ngroups, n)
    x = rng.integers(-10**9, 10**9, n)
    hot = rng.choice(n, 5, replace=False)
    g[hot] = 1
    x[g == 1] = np.abs(x[g == 1])
    x[hot] = 1 << 61
    buf = table(g, x, fmt)
    assert oracle_status(SPEC, buf, 4) == 2
    statuses, _, _ = run_dense(SPEC, [buf], [(0, ngroups)], hashed)
    assert statuses == [2]
    # (inputs of 2^61: not even a work-group's own sums are covered by the proof -- tier 3; the
    # hashed table's bound over all groups fails likewise and the chunk is folded exactly)
    assert LAST["checked_folds"] == 1
    # 3 x 2^61 + 2^60 fit: the same chunk with two of the five rows made smaller is summed on
    # the device
    x[hot[0]] = 7
    x[hot[1]] = 1 << 60
    buf = table(g, x, fmt)
    assert oracle_status(SPEC, buf, 4) == 0
    statuses, pr, targets = run_dense(SPEC, [buf], [(0, ngroups)], hashed)
    # (hashed too: its bound over ALL groups at once -- rows x magnitude -- fails, the exact fold
    # into a scratch table with checked additions finds every group in range)
    assert statuses == [0] and LAST["checked_folds"] == 1
    assert totals([pr], targets) == {k: [None] + v for k, v in expected(g, x).items()}


def test_sum_without_group_by_is_checked_too():
    """no GROUP BY: register accumulators (gpupreagg_reg1_column)"""
    n = 30000
    g = np.zeros(n)
    x = np.full(n, 3, dtype=np.int64)
    x[[7, 77, 777, 7777, 17777]] = 1 << 61
    buf = table(g, x, "column")
    assert oracle_status(SPEC_NOKEY, buf, 3) == 2
    statuses, _, _ = run_dense(SPEC_NOKEY, [buf], [])
    assert statuses == [2]
    x[7] = -5                       # 3 x 2^61 + 2^60 + small change: fits
    x[77] = 1 << 60
    buf = table(g, x, "column")
    assert oracle_status(SPEC_NOKEY, buf, 3) == 0
    statuses, pr, targets = run_dense(SPEC_NOKEY, [buf], [])
    assert statuses == [0]
    want = expected(g, x, with_key=False, minmax=min)
    assert totals([pr], targets) == want


@pytest.mark.parametrize("fmt,ngroups", [("column", 400), ("column", 6), ("row", 50)])
def test_values_at_the_edges_of_int8_that_fit_are_exact(fmt, ngroups):
    """a group that IS int8's largest value, one that is its smallest, sums that end one short
    of the edge: nothing overflows, the oracle says Success, and so must the device (the range
    proof fails even for one work-group: tier 3, the checked program folds the chunk and finds
    every addition in range)"""
    rng = np.random.default_rng(4)
    n = 20000
    g = rng.integers(4, ngroups, n) if ngroups > 4 else np.full(n, 4)
    x = rng.integers(-10**12, 10**12, n)
    special = {0: [I64_MAX], 1: [I64_MIN], 2: [1 << 61, 1 << 61, 1 << 61, (1 << 61) - 1],
               3: [-(1 << 62), -(1 << 62)]}
    rows = rng.choice(n, sum(len(v) for v in special.values()), replace=False)
    i = 0
    for key, vals in special.items():
        for v in vals:
            g[rows[i]] = key
            x[rows[i]] = v
            i += 1
    buf = table(g, x, fmt)
    assert oracle_status(SPEC, buf, 4) == 0
    statuses, pr, targets = run_dense(SPEC, [buf], [(0, max(ngroups, 5))])
    assert statuses == [0]
    got = totals([pr], targets)
    assert got == {k: [None] + v for k, v in expected(g, x).items()}
    assert got[(0,)][2] == I64_MAX and got[(1,)][2] == I64_MIN and got[(3,)][2] == I64_MIN
    assert LAST["checked_folds"] == 1


@pytest.mark.parametrize("nrows,status", [(2000000, 2), (900000, 0)])
def test_many_rows_of_1e13(nrows, status):
    """2e6 rows of 1e13 in one group are 2e19 > 2^63: CpuReCheck, although no work-group's own
    sum overflows (the slabs do when they are added up); 9e5 rows are 9e18 and fit -- by less
    than the proof's margin, so this sum too is made by the checked program"""
    g = np.zeros(nrows) + 5
    x = np.full(nrows, 10**13, dtype=np.int64)
    g[::1000] = np.arange(len(g[::1000])) % 7 + 10          # a few other groups
    x[::1000] = -3
    buf = table(g, x, "column")
    assert oracle_status(SPEC, buf, 4) == status
    statuses, pr, targets = run_dense(SPEC, [buf], [(5, 20)])
    assert statuses == [status]
    # tier 2: a work-group's rows x 2^44 stay far below 2^63 -- the slabs are exact and the slab
    # check decides; the checked program is not needed
    assert LAST["checked_folds"] == 0
    if status == 0:
        assert totals([pr], targets) == {k: [None] + v for k, v in expected(g, x).items()}


@pytest.mark.parametrize("sign", [1, -1])
@pytest.mark.parametrize("n,value,nchunks", [(300000, 17 * 10**12, 6), (1000000, 15 * 10**12, 3)])
def test_sum_that_crosses_int8_only_when_a_later_chunk_is_folded(sign, n, value, nchunks):
    """each chunk's partial sums fit int8 (the oracle, like the reference, works chunk by
    chunk); the session's table adds them up over all chunks -- 128 bits wide, so 6 x 1.7e18
    (range proven per chunk: tier 1) and 3 x 5e18 (proven per work-group only: tier 2, the slab
    check adds the slabs up in 128 bits) are exact, and the fetch hands the total out as partial rows that each fit int8"""
    g = np.arange(n) % 3
    x = np.where(g == 0, sign * value, np.arange(n) - 7)
    bufs = [table(g, x, "column") for _ in range(nchunks)]
    assert oracle_status(SPEC, bufs[0], 4) == 0
    statuses, pr, targets = run_dense(SPEC, bufs, [(0, 3)])
    assert statuses == [0] * nchunks and LAST["checked_folds"] == 0     # tiers 1 and 2
    got = totals([pr], targets)
    want = expected(np.tile(g, nchunks), np.tile(x, nchunks))
    assert got == {k: [None] + v for k, v in want.items()}
    assert abs(got[(0,)][2]) > I64_MAX and len(pr) > 3       # more than one partial row for group 0
    # every partial row is a legal int8 partial: what crossed is their sum
    v, isn = pr.column(2)
    assert not isn.any() and v.dtype == np.int64


def test_per_chunk_message_rechecks_and_sums():
    """the reference's own message (strom_submit_gpupreagg_chunk): the status word says
    CpuReCheck for a chunk whose sum overflows, and edge values that fit come back exact"""
    rng = np.random.default_rng(5)
    n = 30000
    g = rng.integers(0, 60, n)
    x = rng.integers(0, 10**6, n)
    x[:3] = I64_MAX // 2 - 10**10        # two of them (and the group's other rows) fit, three do not
    g[:3] = 11
    buf = table(g, x, "row")
    assert oracle_status(SPEC, buf, 4) == 2
    agg = GpuPreAgg(SPEC)
    status, pr = agg.collect_chunk(agg.submit_chunk(buf))
    assert status == 2 and pr is None
    x[2] = 1
    g[2] = 12
    buf = table(g, x, "row")
    assert oracle_status(SPEC, buf, 4) == 0
    status, pr = agg.collect_chunk(agg.submit_chunk(buf))
    assert status == 0
    assert totals([pr], agg.targets) == {k: [None] + v for k, v in expected(g, x).items()}


def test_sums_of_int4_columns_need_no_measurement():
    """sum(int4) -- (psum (int8 (var N int4))) -- is bounded by its type: 2^31 x 2^32 rows < 2^63.
    The generated code says so (GPUPREAGG_SUMBITS) and the kernels carry no magnitude test;
    the extreme column sums exactly"""
    n = 400000
    g = np.arange(n) % 9
    x = np.where(g < 5, np.int64(-(1 << 31)), np.int64((1 << 31) - 1)).astype(np.int32)
    spec = "(gpupreagg (key (var 1 int4)) (nrows) (psum (int8 (var 2 int4))))"
    agg = GpuPreAgg(spec)
    assert "#define GPUPREAGG_SUMBITS_1 31" in agg.codegen.source
    buf = kds.build_kds("column", [kds.Column("int4", g.astype(np.int32)), kds.Column("int4", x)])
    agg.begin([(0, 9)])
    try:
        assert agg.fold(buf)[0] == 0 and agg.fold(buf)[0] == 0
        pr = agg.fetch()
    finally:
        agg.end()
    got = totals([pr], agg.targets)
    for k in range(9):
        cnt = 2 * int((g == k).sum())
        assert got[(k,)][1:] == [cnt, cnt * int(x[k])]


# ---------------------------------------------------------------------------------------------
# the hashed table, exactly.  Its running bound -- rows x largest magnitude, over all groups at
# once -- says little about any one group; where it fails the chunk is folded into a scratch
# session with checked additions and joins the table under a per-group check
# (csrc/gpupreagg.cpp: gpupreagg_hashed_exact).  Device == oracle == Python, never a CpuReCheck
# the reference would not have raised.
# ---------------------------------------------------------------------------------------------
HSPEC = "(gpupreagg (key (var 1 int4)) (nrows) (psum (var 2 int8)) (pmax (var 2 int8)))"


@pytest.mark.parametrize("fmt", ["column", "row"])
def test_hashed_two_million_rows_of_1e13_are_summed_not_sent_back(fmt):
    """2e6 rows of ~1e13 in 1000 groups: rows x magnitude is 2^65, every group's sum 2e16.
    (fmt row: 2e5 rows -- heap pages of 2e6 rows do not fit one chunk's 16-bit block index)"""
    n = 2_000_000 if fmt == "column" else 200_000
    rng = np.random.default_rng(8)
    g = rng.integers(0, 1000, n)
    x = rng.integers(10**13 - 10**6, 10**13 + 10**6, n) * rng.choice([-1, 1], n, p=[0.1, 0.9])
    if fmt == "row":
        x = x * 256                                # keep rows x magnitude beyond 2^63 at 2e5 rows
    buf = table(g, x, fmt)
    assert oracle_status(HSPEC, buf, 4) == 0
    statuses, pr, targets = run_dense(HSPEC, [buf, buf], None, hashed=True)
    assert statuses == [0, 0] and LAST["checked_folds"] >= 1
    want = {k: [None, 2 * v[0], 2 * v[1], v[2]] for k, v in expected(g, x).items()}
    assert totals([pr], targets) == want


def test_hashed_sum_that_crosses_int8_in_the_second_chunk_leaves_the_table_alone():
    """chunk 1 leaves 0.6 x 2^63 in group 7, chunk 2 brings as much again: inside chunk 2 every
    addition fits (the scratch fold succeeds), the group does not fit the table's -- the verify
    pass says so before anything is imported: CpuReCheck, table as it was.  A third chunk that
    takes the group back down is summed."""
    n = 40000
    rng = np.random.default_rng(9)
    g = rng.integers(0, 300, n)
    x = rng.integers(-10**9, 10**9, n)
    hot = np.where(g == 7)[0][:6]
    x[g == 7] = 0
    x[hot] = (6 * 2**63 // 10) // 6
    up = table(g, x, "column")
    x2 = x.copy()
    x2[hot] = -x2[hot]
    down = table(g, x2, "column")
    agg = GpuPreAgg(HSPEC).begin_hashed()
    try:
        assert agg.fold(up)[0] == 0
        before = totals([agg.fetch()], agg.targets)
        assert before[(7,)][2] == 6 * ((6 * 2**63 // 10) // 6) > 2**62
        assert agg.fold(up)[0] == 2                      # StromError_CpuReCheck
        assert totals([agg.fetch()], agg.targets) == before
        assert agg.fold(down)[0] == 0
        after = totals([agg.fetch()], agg.targets)
        assert after[(7,)][2] == 0 and after[(7,)][1] == 2 * before[(7,)][1]
        assert agg.checked_folds() >= 2
    finally:
        agg.end()


def test_hashed_values_at_the_edges_of_int8_that_fit_are_exact():
    """a group that IS int8's largest value, one that is its smallest, sums one short of the
    edge, among ordinary rows: Success in the oracle, Success and the same sums on the device"""
    rng = np.random.default_rng(10)
    n = 30000
    g = rng.integers(4, 500, n)
    x = rng.integers(-10**12, 10**12, n)
    special = {0: [I64_MAX], 1: [I64_MIN], 2: [1 << 61, 1 << 61, 1 << 61, (1 << 61) - 1],
               3: [-(1 << 62), -(1 << 62)]}
    rows = rng.choice(n, sum(len(v) for v in special.values()), replace=False)
    i = 0
    for key, vals in special.items():
        for v in vals:
            g[rows[i]] = key
            x[rows[i]] = v
            i += 1
    buf = table(g, x, "column")
    assert oracle_status(HSPEC, buf, 4) == 0
    statuses, pr, targets = run_dense(HSPEC, [buf], None, hashed=True)
    assert statuses == [0] and LAST["checked_folds"] == 1
    assert totals([pr], targets) == {k: [None] + v for k, v in expected(g, x).items()}
    # one more row on top of the largest value: that chunk is the CPU's, as in the reference
    g2, x2 = np.append(g, 0), np.append(x, 1)
    buf2 = table(g2, x2, "column")
    assert oracle_status(HSPEC, buf2, 4) == 2
    statuses, _, _ = run_dense(HSPEC, [buf2], None, hashed=True)
    assert statuses == [2]
