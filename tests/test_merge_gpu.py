"""
The multi-GPU merge's own steps, run with two DIFFERENT tables on one GPU (needs an MI355X: -m gpu).

strom_gpupreagg_allreduce (csrc/parallel.cpp) merges the ranks' resident tables lane by lane:
preagg_merge_prepare brings every table section into a form RCCL's SUM / MIN / MAX merge correctly
(identities for entries without a value, sign flips for the float keys, flags unpacked to bytes,
128-bit integer sums split into carry-free limbs), one all-reduce per lane, preagg_merge_finish
undoes the transforms (devlib/strom_merge.h).  On one GPU a communicator has one rank and every
all-reduce is the identity -- so strom_gpupreagg_merge(dst, src) of two dense sessions runs the
SAME prepare / lanes / finish with preagg_merge_apply standing where RCCL's operator stands between
two ranks.  A wrong identity, a missing sign flip or a lost carry turns these tests red.

Reference: none -- the reference has no collective; the oracle over the
union of both row ranges is the spec (what ONE backend's Agg node would have added up).
"""
import numpy as np
import pytest

import oracle_binding as oracle
from pg_strom_amd import kds, runtime
from pg_strom_amd.gpupreagg import GpuPreAgg, KIND_KEY
from test_gpupreagg_gpu import assert_matches_oracle
from test_sum_overflow_gpu import totals

pytestmark = pytest.mark.gpu

SPEC = ("(gpupreagg (key (var 1 int4)) (nrows) (nrows (isnotnull (var 2 int4))) (psum (int8 (var 2 int4)))"
        " (psum (var 6 float8)) (pmin (var 3 float8)) (pmax (var 3 float8)) (pmin (var 2 int4)) (pmax (var 2 int4))"
        " (pmin (var 5 int8)) (pmax (var 5 int8)) (psum (var 4 int8)))")


def side(seed, groups, n=60000, null_only_group=None, edge=False):
    """one rank's rows: keys from 'groups'; a float column with NaN, -0.0, +-inf for min / max
    (and a plain one for the sum); an int8 column at the edges of its type for min / max; a summed
    int8 column whose largest inputs defeat the range proof (edge: the checked program folds)"""
    rng = np.random.default_rng(seed)
    g = rng.choice(np.asarray(groups, dtype=np.int32), n)
    x = rng.integers(-2**31, 2**31, n).astype(np.int32)
    xn = rng.random(n) < 0.1
    if null_only_group is not None:
        xn |= (g == null_only_group)
    y = rng.normal(size=n) * 1e3
    y[rng.random(n) < 0.01] = np.nan
    y[rng.random(n) < 0.01] = -0.0
    y[rng.random(n) < 0.005] = np.inf
    y[rng.random(n) < 0.005] = -np.inf
    yn = rng.random(n) < 0.2
    z = rng.integers(-10**9, 10**9, n).astype(np.int64)
    w = rng.integers(-10**15, 10**15, n).astype(np.int64)
    f = rng.normal(size=n) * 50
    if edge:
        w[:3] = [2**63 - 1, -2**63, 0]
        g[:2] = groups[0], groups[1]
        z[g == groups[0]] = np.abs(z[g == groups[0]])           # same sign inside the two groups ...
        z[g == groups[1]] = -np.abs(z[g == groups[1]])
        z[0], z[1] = 2**62, -2**62                               # ... that hold a huge input each
    return kds.build_kds("column", [kds.Column("int4", g, rng.random(n) < 0.01), kds.Column("int4", x, xn),
                                    kds.Column("float8", y, yn), kds.Column("int8", z),
                                    kds.Column("int8", w, rng.random(n) < 0.05), kds.Column("float8", f, yn)])


def fold_into(spec, buf, domain, compact_bitmap=None):
    agg = GpuPreAgg(spec).begin(domain)
    if compact_bitmap is not None:
        agg.compact(compact_bitmap)
    assert agg.fold(buf)[0] == 0
    return agg


@pytest.mark.parametrize("compact", [False, True])
def test_merge_of_two_different_dense_tables_equals_the_oracle_over_both(compact):
    """groups only one side saw (left 0..59, right 40..99), a group with NULL-only inputs on one
    side and values on the other, float min / max over NaN, -0, +-inf, int min / max at the edges
    of int4 and int8, and -- compact -- census-compacted slots agreed on by both sides"""
    left = side(1, list(range(0, 60)), null_only_group=45, edge=True)
    right = side(2, list(range(40, 100)), null_only_group=50)
    domain = [(0, 100)]
    bitmap = None
    if compact:
        probe = GpuPreAgg(SPEC).begin(domain)
        try:
            probe.census(left)
            bitmap = probe.census(right)              # the union: what census_allreduce gives every rank
        finally:
            probe.end()
    a = fold_into(SPEC, left, domain, bitmap)
    b = fold_into(SPEC, right, domain, bitmap)
    try:
        before_b = b.fetch()
        a.merge_from(b)
        merged = a.fetch()
        after_b = b.fetch()
        assert_matches_oracle(SPEC, a, [left, right], merged)
        # the source is left as it was (partial rows come in no particular order: by key, NULL key last)
        ob = np.lexsort((before_b.values[:, 0].view(np.int64), before_b.isnull[:, 0]))
        oa = np.lexsort((after_b.values[:, 0].view(np.int64), after_b.isnull[:, 0]))
        assert np.array_equal(before_b.values[ob], after_b.values[oa]) and np.array_equal(before_b.isnull[ob], after_b.isnull[oa])
        # and merging is not idempotent: b once more doubles b's share of the counts
        a.merge_from(b)
        twice = a.fetch()
        n1 = {int(k): int(c) for k, c, kn in zip(merged.column(0)[0], merged.column(1)[0], merged.isnull[:, 0]) if not kn}
        n2 = {int(k): int(c) for k, c, kn in zip(twice.column(0)[0], twice.column(1)[0], twice.isnull[:, 0]) if not kn}
        nb = {int(k): int(c) for k, c, kn in zip(before_b.column(0)[0], before_b.column(1)[0], before_b.isnull[:, 0]) if not kn}
        assert all(n2[k] == n1[k] + nb.get(k, 0) for k in n1)
    finally:
        a.end()
        b.end()


def test_merge_adds_integer_sums_beyond_int8_without_wrapping():
    """each side's sum of group 0 is 0.9 x 2^63: the merged total does not fit int8.  The table
    keeps it in 128 bits, the merge moves it as three carry-free limbs (a plain 64-bit SUM -- the
    operator round 2 handed to RCCL -- wraps to a negative number here), the fetch hands it out as
    two partial rows"""
    spec = "(gpupreagg (key (var 1 int4)) (nrows) (psum (var 2 int8)) (pmax (var 2 int8)))"
    n = 90000
    g = (np.arange(n) % 3).astype(np.int32)
    big = (9 * 2**63 // 10) // (n // 3)
    tables = []
    for sign in (1, 1, -1):
        x = np.where(g == 0, sign * big, np.where(g == 1, -sign * big, 5)).astype(np.int64)
        tables.append(kds.build_kds("column", [kds.Column("int4", g), kds.Column("int8", x)]))
    a, b, c = (fold_into(spec, t, [(0, 3)]) for t in tables)
    try:
        a.merge_from(b)
        pr = a.fetch()
        got = totals([pr], a.targets)
        per = big * (n // 3)
        assert got == {(0,): [None, 2 * (n // 3), 2 * per, big], (1,): [None, 2 * (n // 3), -2 * per, -big],
                       (2,): [None, 2 * (n // 3), 10 * (n // 3), 5]}
        assert 2 * per > 2**63 and len(pr) == 5           # groups 0 and 1 leave as two rows each
        a.merge_from(c)                                    # ... and the third side takes it back into range
        pr = a.fetch()
        got = totals([pr], a.targets)
        assert got[(0,)][2] == per and got[(1,)][2] == -per and len(pr) == 3
    finally:
        a.end(), b.end(), c.end()


def test_merge_refuses_sessions_that_do_not_match():
    """different domain (slot i is another group), different program, dense into hashed: refused
    on the host -- the kernels would add unrelated slots, or read records of another length"""
    spec = "(gpupreagg (key (var 1 int4)) (nrows) (psum (int8 (var 2 int4))))"
    spec2 = "(gpupreagg (key (var 1 int4)) (nrows) (pmax (var 2 int4)))"
    buf = kds.build_kds("column", [kds.Column("int4", np.arange(1000, dtype=np.int32) % 7),
                                   kds.Column("int4", np.arange(1000, dtype=np.int32))])
    a = fold_into(spec, buf, [(0, 7)])
    b = fold_into(spec, buf, [(0, 8)])
    c = fold_into(spec2, buf, [(0, 7)])
    h = GpuPreAgg(spec).begin_hashed()
    h2 = GpuPreAgg(spec2).begin_hashed()
    try:
        assert h.fold(buf)[0] == 0 and h2.fold(buf)[0] == 0
        for dst, src in ((a, b), (a, c), (a, h), (h, a), (h, h2), (a, a)):
            with pytest.raises(runtime.StromError) as ei:
                dst.merge_from(src)
            assert ei.value.errcode == 101            # BadRequestMessage
    finally:
        for s in (a, b, c, h, h2):
            s.end()
