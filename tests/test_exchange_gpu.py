"""
The hash-partitioned exchange of hashed GROUP BY sessions, run among sessions of ONE GPU (needs an
MI355X: -m gpu).

strom_gpupreagg_allreduce / _reduce_scatter (csrc/parallel.cpp: hashed_exchange) move the groups of
hashed sessions between the ranks BY OWNER: a group belongs to the rank its key hashes to, every
rank packs its groups by owner, the ranks swap partitions pairwise and each merges its own.
strom_gpupreagg_exchange_local(sessions, n, gather) runs the same packing, the same decisions and
the same imports with session i standing for rank i and device copies for the collectives -- on one
GPU a communicator has one rank and the exchange is a no-op, so this is where a wrong owner, a group
sent twice, a lost partition or a dropped range check of the integer sums turns a test red.

Reference: none -- the reference has no collective; the oracle over the
union of the sessions' rows is the spec (what ONE backend's Agg node would have added up).
"""
import numpy as np
import pytest

import oracle_binding as oracle
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpupreagg import GpuPreAgg, PartialRows, KIND_KEY
from test_gpupreagg_gpu import assert_matches_oracle
from test_sum_overflow_gpu import totals

pytestmark = pytest.mark.gpu

# a float key (NaN is one group, -0 = +0, NULL is one group) and a sparse int8 key
SPEC = ("(gpupreagg (key (var 1 float8)) (key (var 2 int8)) (nrows) (psum (int8 (var 3 int4)))"
        " (psum (var 4 float8)) (pmin (var 3 int4)) (pmax (var 4 float8)) (psum (var 5 int8)))")


def side(seed, n, key_lo, key_hi):
    rng = np.random.default_rng(seed)
    f = rng.integers(key_lo, key_hi, n).astype(np.float64) / 2
    f[rng.random(n) < 0.02] = np.nan
    f[rng.random(n) < 0.02] = -0.0
    big = (rng.integers(0, 7, n).astype(np.int64) - 3) * (2**41 + 7)
    x = rng.integers(-10**6, 10**6, n).astype(np.int32)
    y = rng.normal(size=n) * 100
    z = rng.integers(-10**12, 10**12, n).astype(np.int64)
    return kds.build_kds("column", [kds.Column("float8", f, rng.random(n) < 0.02), kds.Column("int8", big, rng.random(n) < 0.02),
                                    kds.Column("int4", x, rng.random(n) < 0.05), kds.Column("float8", y, rng.random(n) < 0.05),
                                    kds.Column("int8", z)])


def keys_of(pr):
    v, n = pr.values, pr.isnull
    kcols = [t for t, (k, _) in enumerate(pr.targets) if k == KIND_KEY]
    return [tuple((None if n[i, t] else int(v[i, t])) for t in kcols) for i in range(len(pr))]


@pytest.mark.parametrize("nranks", [2, 3, 8])
@pytest.mark.parametrize("gather", [False, True])
def test_exchange_among_sessions_equals_the_oracle_over_all_rows(nranks, gather):
    """overlapping key ranges (rank r sees keys r*40 .. r*40+200: groups one rank alone saw, groups
    all of them saw), 2e4 rows each.  Reduce-scatter: the ranks' fetches are DISJOINT and their
    union is the oracle's answer over all rows; all-reduce: every rank's fetch is that answer."""
    bufs = [side(10 + r, 20000, r * 40, r * 40 + 200) for r in range(nranks)]
    sessions = [GpuPreAgg(SPEC).begin_hashed() for _ in range(nranks)]
    try:
        for s, b in zip(sessions, bufs):
            assert s.fold(b)[0] == 0
        before = [len(s.fetch()) for s in sessions]
        GpuPreAgg.exchange_local(sessions, gather_after=gather)
        prs = [s.fetch() for s in sessions]
        if gather:
            for s, pr in zip(sessions, prs):
                assert_matches_oracle(SPEC, s, bufs, pr)
            return
        owned = [set(keys_of(pr)) for pr in prs]
        assert sum(len(o) for o in owned) == len(set().union(*owned))          # no group on two ranks
        union = PartialRows(prs[0].targets, np.concatenate([pr.values for pr in prs]),
                            np.concatenate([pr.isnull for pr in prs]))
        assert_matches_oracle(SPEC, sessions[0], bufs, union)
        # the owner is a function of the key alone, and spreads the groups: every rank owns a share
        assert all(len(o) > len(union) // (4 * nranks) for o in owned)
        assert max(before) <= len(union)
        # exchanging once more moves nothing (every group already sits with its owner)
        GpuPreAgg.exchange_local(sessions, gather_after=False)
        assert [set(keys_of(s.fetch())) for s in sessions] == owned
    finally:
        for s in sessions:
            s.end()


def test_exchange_refuses_integer_sums_that_could_leave_int8():
    """two ranks hold 0.6 x 2^63 each in the same group: the merged sum does not fit int8.  The
    ranks' bounds travel with the counts, their sum reaches 2^63, every rank answers CpuReCheck
    BEFORE any group has moved -- the tables are as they were.  (A 64-bit atomic add of the two
    partials, what the import does, would have wrapped to a negative number without a word.)
    One such rank among empty-handed ones is fine: nothing is added to its sums."""
    spec = "(gpupreagg (key (var 1 int8)) (nrows) (psum (var 2 int8)))"
    # (the hashed table proves its sums' range from rows x largest magnitude: 1000 rows of 2^52 - 1,
    # folded twice, stay provable -- 2000 x 2^52 < 2^63 -- and leave 0.88 x 2^63 in group 0)
    n, per = 1000, 2**52 - 1
    g = np.where(np.arange(n) < 900, 0, 10**12).astype(np.int64)
    x = np.where(g == 0, per, 5).astype(np.int64)
    buf = kds.build_kds("column", [kds.Column("int8", g), kds.Column("int8", x)])
    small = kds.build_kds("column", [kds.Column("int8", g), kds.Column("int8", np.zeros(n, dtype=np.int64))])
    a, b, c = (GpuPreAgg(spec).begin_hashed() for _ in range(3))
    try:
        for s_ in (a, b):
            assert s_.fold(buf)[0] == 0 and s_.fold(buf)[0] == 0
        assert c.fold(small)[0] == 0
        want_a = totals([a.fetch()], a.targets)
        assert want_a[(0,)][2] == 1800 * per > 2**62
        for sessions in ([a, b], [a, c, b]):
            with pytest.raises(runtime.StromError) as ei:
                GpuPreAgg.exchange_local(sessions, gather_after=True)
            assert ei.value.errcode == 2                                       # StromError_CpuReCheck
            assert totals([a.fetch()], a.targets) == want_a and totals([b.fetch()], b.targets) == want_a
        with pytest.raises(runtime.StromError) as ei:
            a.merge_from(b)
        assert ei.value.errcode == 2 and totals([a.fetch()], a.targets) == want_a
        # the sums of 'c' are all zero: it adds nothing, the exchange goes through
        GpuPreAgg.exchange_local([a, c], gather_after=True)
        for s_ in (a, c):
            got = totals([s_.fetch()], s_.targets)
            assert got[(0,)] == [None, 1800 + 900, 1800 * per] and got[(10**12,)][1:] == [300, 1000]
    finally:
        for s in (a, b, c):
            s.end()


def test_exchange_refuses_what_is_not_a_set_of_matching_hashed_sessions():
    spec = "(gpupreagg (key (var 1 int8)) (nrows) (psum (var 2 int8)))"
    spec2 = "(gpupreagg (key (var 1 int8)) (nrows) (pmax (var 2 int8)))"
    buf = kds.build_kds("column", [kds.Column("int8", np.arange(1000, dtype=np.int64) % 7),
                                   kds.Column("int8", np.arange(1000, dtype=np.int64))])
    h1, h2 = GpuPreAgg(spec).begin_hashed(), GpuPreAgg(spec).begin_hashed()
    other = GpuPreAgg(spec2).begin_hashed()
    dense = GpuPreAgg(spec).begin([(0, 7)])
    try:
        for s in (h1, h2, other, dense):
            assert s.fold(buf)[0] == 0
        for sessions in ([h1, other], [h1, dense], [h1, h1], [dense, dense]):
            with pytest.raises(runtime.StromError) as ei:
                GpuPreAgg.exchange_local(sessions)
            assert ei.value.errcode == 101
        # one session alone owns everything: nothing moves
        GpuPreAgg.exchange_local([h1])
        assert totals([h1.fetch()], h1.targets) == totals([h2.fetch()], h2.targets)
        # and a session that never folded takes part with no groups
        empty = GpuPreAgg(spec).begin_hashed()
        try:
            GpuPreAgg.exchange_local([h1, empty], gather_after=True)
            assert totals([empty.fetch()], empty.targets) == totals([h2.fetch()], h2.targets)
        finally:
            empty.end()
    finally:
        for s in (h1, h2, other, dense):
            s.end()
