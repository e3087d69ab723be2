"""The kernels built ahead of time for one launch shape each (qe_static.hip) against the oracle, and
against the general interpreter kernels on the same plans.  Every case checks, through the launch
profile, that the path it means to test is the one that ran."""
import numpy as np
import pytest

from oracle import oracle as orc
from test_gpu_parity import C, mk, rng, run_agg, run_both

pytestmark = pytest.mark.gpu

SIZES = [1, 255, 1024, 4097, 100_003]
GT = ("gt", C(3), ("f64", 0.5))
FMA = ("add", C(1), ("mul", C(2), C(3)))


def kabc(n, seed, n_keys=10, key_dtype=np.int64):
    """k (skewed, no nulls), a, b in [-1, 1), c in [0, 1): the headline table's shape."""
    r = rng(seed)
    k = np.minimum(r.geometric(0.6, n) - 1, n_keys - 1).astype(key_dtype)
    t = orc.INT64 if key_dtype == np.int64 else orc.INT32
    return [orc.Array(t, k), orc.Array(orc.FLOAT64, r.uniform(-1, 1, n)), orc.Array(orc.FLOAT64, r.uniform(-1, 1, n)),
            orc.Array(orc.FLOAT64, r.uniform(0, 1, n))]


def both_paths(qe, ctx, fn, expect_kernel):
    """Runs fn() on the specialised path and on the interpreter path; returns both results."""
    ctx.profile(True)
    try:
        ctx.profile_read()
        a = fn()
        names = set(ctx.profile_read())
        # (the first pass of the high-cardinality mode is a specialised kernel of its own: qe_aggregate_spill_static<shape>)
        assert any(k.startswith(expect_kernel + "_static<") or k.startswith(expect_kernel + "_spill_static<") for k in names), \
            f"specialised {expect_kernel} kernel did not run: {names}"
        ctx.set_flags(qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP | qe.FLAG_NO_SPECIALIZED_KERNELS)
        b = fn()
        names = set(ctx.profile_read())
        assert not any("_static<" in k for k in names) and any(k == expect_kernel + "_kernel" for k in names), names
    finally:
        ctx.set_flags(qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP)
        ctx.profile(False)
    return a, b


@pytest.mark.parametrize("n", SIZES)
def test_project_and_filter_shapes(qe, ctx, n):
    cols = kabc(n, seed=n)
    proj = lambda m, scan: m.ProjectionExec([scan], [mk(m, FMA)])
    filt = lambda m, scan: m.ProjectionExec([m.SelectionExec([scan], mk(m, GT))], [mk(m, FMA)])
    filt2 = lambda m, scan: m.ProjectionExec([m.SelectionExec([scan], mk(m, GT))], [mk(m, C(0)), mk(m, FMA)])
    both_paths(qe, ctx, lambda: run_both(qe, ctx, [cols], proj, what=f"project n={n}"), "qe_project")
    both_paths(qe, ctx, lambda: run_both(qe, ctx, [cols], filt, what=f"filter n={n}"), "qe_filter")
    both_paths(qe, ctx, lambda: run_both(qe, ctx, [cols], filt2, what=f"filter k,fma n={n}"), "qe_filter")
    # same shape, other column positions: (a, b, c) without the key column
    abc = cols[1:]
    proj3 = lambda m, scan: m.ProjectionExec([scan], [mk(m, ("add", C(0), ("mul", C(1), C(2))))])
    both_paths(qe, ctx, lambda: run_both(qe, ctx, [abc], proj3, what=f"project abc n={n}"), "qe_project")


@pytest.mark.parametrize("n,n_keys", [(1, 1), (300, 3), (4097, 10), (100_003, 10), (100_003, 5000), (200_000, 150_000)])
def test_aggregate_shapes(qe, ctx, n, n_keys):
    cols = kabc(n, seed=n + n_keys, n_keys=n_keys)
    if n_keys > 10:
        cols[0] = orc.Array(orc.INT64, rng(n).integers(-n_keys // 2, n_keys // 2, n))
    sum_count = [("sum", C(3)), ("count", None)]
    four = [("sum", C(3)), ("min", C(3)), ("max", C(3)), ("avg", C(3))]
    both_paths(qe, ctx, lambda: run_agg(qe, ctx, [cols], [C(0)], sum_count, what=f"sum,count n={n} keys={n_keys}"), "qe_aggregate")
    both_paths(qe, ctx, lambda: run_agg(qe, ctx, [cols], [C(0)], four, what=f"sum,min,max,avg n={n} keys={n_keys}"), "qe_aggregate")
    # filter -> project -> group by: the projection is substituted into the aggregate's inputs
    def north_star(m, scan):
        sel = m.SelectionExec([scan], mk(m, GT))
        proj = m.ProjectionExec([sel], [mk(m, C(0)), mk(m, FMA)])
        args = {} if m is orc else {"group_capacity_hint": 2 * n_keys}
        return m.AggregateExec([proj], [mk(m, C(0))], [mk(m, ("sum", C(1))), mk(m, ("count", None))], **args)
    parts = [cols, kabc(777, seed=5, n_keys=min(n_keys, 10))]
    want = list(north_star(orc, orc.ScanExec([orc.Chunk(p) for p in parts])).execute())
    if not want:
        return

    def run():
        # float64 sums: tolerance 1e-12 x sum |v| per group, computed by run_agg's helper on the filtered input
        return run_agg(qe, ctx, parts, [C(0)], [("sum", FMA), ("count", None)], pred=GT, hint=2 * n_keys, what=f"north star n={n} keys={n_keys}")
    if any(len(w) for w in want):
        both_paths(qe, ctx, run, "qe_aggregate")
        got = list(north_star(qe, qe.ScanExec(ctx, [__import__("util").gpu_batch(qe, ctx, p) for p in parts])).execute())[0]
        assert got.num_rows == len(want[0])


def test_int32_key_and_partial_final(qe, ctx):
    cols = kabc(50_000, seed=3, key_dtype=np.int32)[:1] + [kabc(50_000, seed=3)[3]]
    both_paths(qe, ctx, lambda: run_agg(qe, ctx, [cols], [C(0)], [("sum", C(1)), ("count", None)], what="int32 key"), "qe_aggregate")
    # PARTIAL on two shards + FINAL merge, all three launches on specialised kernels
    full = kabc(60_000, seed=8)
    shards = [[orc.Array(a.type, a.values[s]) for a in full] for s in (slice(0, 25_000), slice(25_000, None))]
    util = __import__("util")

    def run():
        parts = []
        for sh in shards:
            sel = qe.SelectionExec([qe.ScanExec(ctx, [util.gpu_batch(qe, ctx, sh)])], mk(qe, GT))
            proj = qe.ProjectionExec([sel], [mk(qe, C(0)), mk(qe, FMA)])
            parts += list(qe.AggregateExec([proj], [mk(qe, C(0))], [mk(qe, ("sum", C(1))), mk(qe, ("count", None))], mode=qe.AGG_MODE_PARTIAL).execute())
        fin = qe.final_plan(qe.ScanExec(ctx, parts), 1, [mk(qe, ("sum", C(1))), mk(qe, ("count", None))])
        return list(fin.execute())[0].columns()
    a, b = both_paths(qe, ctx, run, "qe_aggregate")
    m = full[3].values > 0.5
    v = full[1].values + full[2].values * full[3].values
    for got in (a, b):
        order = np.argsort(got[0].values)
        keys = np.unique(full[0].values[m])
        assert np.array_equal(np.asarray(got[0].values)[order], keys)
        assert np.array_equal(np.asarray(got[2].values)[order], [int((m & (full[0].values == k)).sum()) for k in keys])
        for i, k in zip(order, keys):
            sel = m & (full[0].values == k)
            assert abs(got[1].values[i] - v[sel].sum()) <= 1e-12 * np.abs(v[sel]).sum()


def test_nullable_input_uses_interpreter(qe, ctx):
    """A validity bitmap on a cached column is outside what the specialised kernels handle."""
    cols = kabc(5000, seed=1)
    cols[3] = orc.Array(orc.FLOAT64, cols[3].values, rng(2).random(5000) > 0.2)
    ctx.profile(True)
    try:
        ctx.profile_read()
        run_both(qe, ctx, [cols], lambda m, scan: m.ProjectionExec([m.SelectionExec([scan], mk(m, GT))], [mk(m, FMA)]), what="nullable filter")
        run_agg(qe, ctx, [cols], [C(0)], [("sum", C(3)), ("count", None)], what="nullable aggregate")
        assert not any("_static<" in k for k in ctx.profile_read())
    finally:
        ctx.profile(False)


@pytest.mark.parametrize("n,n_keys,zipf", [(300_000, 100_000, False), (400_000, 1_000_000, True), (70_000, 200, False)])
def test_high_cardinality_mode(qe, ctx, n, n_keys, zipf):
    """A large group-count hint switches the aggregate to log-and-partition for the rows that miss the
    LDS tables.  Same results on the specialised and the general kernels, and the second phase ran."""
    r = rng(n)
    if zipf:
        k = (r.zipf(1.1, n) % n_keys).astype(np.int64) * 2654435761 % (2 ** 40)
    else:
        k = r.integers(-n_keys // 2, n_keys // 2, n).astype(np.int64)
    cols = [orc.Array(orc.INT64, k), orc.Array(orc.FLOAT64, r.uniform(-100, 100, n))]
    hint = 1 << 17
    sum_count = [("sum", C(1)), ("count", None)]
    four = [("sum", C(1)), ("min", C(1)), ("max", C(1)), ("avg", C(1))]
    both_paths(qe, ctx, lambda: run_agg(qe, ctx, [cols], [C(0)], sum_count, hint=hint, what=f"spill sum,count n={n}"), "qe_aggregate")
    both_paths(qe, ctx, lambda: run_agg(qe, ctx, [cols], [C(0)], four, hint=hint, what=f"spill four n={n}"), "qe_aggregate")
    ctx.profile(True)
    try:
        ctx.profile_read()
        # computed key and inputs, filter, two batches, int64 sum next to float64 min
        cols3 = cols + [orc.Array(orc.INT64, r.integers(-2 ** 40, 2 ** 40, n))]
        parts = [cols3, [orc.Array(a.type, a.values[:1000]) for a in cols3]]
        run_agg(qe, ctx, parts, [("add", C(0), ("i64", 1))], [("sum", C(2)), ("min", ("mul", C(1), ("f64", 2.0))), ("max", C(2)), ("count", None)],
                pred=("gt", C(1), ("f64", -50.0)), hint=hint, what="spill computed")
        names = set(ctx.profile_read())
    finally:
        ctx.profile(False)
    if n_keys >= 100_000:
        assert "qe_spill_count_kernel" in names, names  # the mode was on
    if n_keys >= 100_000 and not zipf:
        # uniform keys overflow every workgroup's LDS table (the skewed ones, spread over the workgroups, need not)
        assert "qe_aggregate_partitioned_kernel" in names, names


def test_lean_first_pass_of_the_high_cardinality_mode(qe):
    """With a hint of many groups the first pass is a kernel of its own (qe_aggregate_spill_static / _spill_jit) that
    also counts its logged rows per partition (no qe_spill_count_kernel).  Against the oracle: the pre-built shape, a
    run-time compiled shape with six accumulators over three inputs (folded accumulator by accumulator, not row by row),
    the key equal to the table's empty marker, NaN inputs of min / max, two batches into one table."""
    jit = qe.Context(0)
    jit.set_flags(qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP | qe.FLAG_JIT_SYNC)
    try:
        n, n_keys = 1_200_000, 300_000  # (a batch of at least 2^20 rows: smaller ones are not worth a run-time compile)
        r = rng(7)
        k = (r.zipf(1.2, n) % n_keys).astype(np.int64) * 2654435761 % (2 ** 40)
        k[::1000] = -1  # all ones: the marker of an empty table slot, kept in a slot of its own
        z = r.uniform(-100, 100, n)
        x = z.copy()
        x[5::997] = np.nan  # min / max skip NaN
        y = r.integers(-2 ** 40, 2 ** 40, n)
        jit.profile(True)
        jit.profile_read()
        hint = 400_000
        # the pre-built shape: (k, z) -> SUM, MIN, MAX, AVG of z, two batches into one table
        kz = [orc.Array(orc.INT64, k), orc.Array(orc.FLOAT64, z)]
        four = [("sum", C(1)), ("min", C(1)), ("max", C(1)), ("avg", C(1))]
        run_agg(qe, jit, [kz], [C(0)], four, hint=hint, what="lean four")
        names = set(jit.profile_read())
        assert any(k.startswith("qe_aggregate_spill_static<") for k in names), names
        assert "qe_spill_count_kernel" not in names and "qe_aggregate_partitioned_kernel" in names, names
        # (a second batch of the same shape goes into the first one's launch: one first pass for both)
        run_agg(qe, jit, [kz, [orc.Array(a.type, a.values[:50_000]) for a in kz]], [C(0)], four, hint=hint, what="lean four, two batches")
        prof = jit.profile_read()
        assert "qe_spill_count_kernel" not in prof and sum(v["launches"] for k, v in prof.items() if k.startswith("qe_aggregate_spill_static<")) == 1, prof
        # (a small batch first: too few workgroups for one per column of the cell table, so it has a launch of its own and
        # its rows are counted by the count pass)
        run_agg(qe, jit, [[orc.Array(a.type, a.values[:50_000]) for a in kz], kz], [C(0)], four, hint=hint, what="lean four, a small batch first")
        assert "qe_spill_count_kernel" in set(jit.profile_read())
        # a run-time compiled shape: seven aggregates over three inputs behind a filter
        cols = [orc.Array(orc.INT64, k), orc.Array(orc.FLOAT64, z), orc.Array(orc.FLOAT64, x), orc.Array(orc.INT64, y)]
        second = [orc.Array(a.type, a.values[:50_000]) for a in cols]
        seven = [("sum", C(1)), ("min", C(2)), ("max", C(3)), ("sum", C(3)), ("min", C(3)), ("max", C(2)), ("avg", C(1))]
        run_agg(qe, jit, [cols], [C(0)], seven, pred=("gt", C(3), ("i64", -2 ** 39)), hint=hint, what="lean seven")
        names = set(jit.profile_read())
        assert any(k.startswith("qe_aggregate_spill_jit<") for k in names), names
        assert "qe_spill_count_kernel" not in names, names
        run_agg(qe, jit, [cols, second], [C(0)], seven, pred=("gt", C(3), ("i64", -2 ** 39)), hint=hint, what="lean seven, two batches")
    finally:
        jit.profile(False)
        jit.close()


def test_many_groups_without_a_hint(qe, ctx):
    """A million groups and no group-count hint from the caller: the default table fills up, is given up at once
    (not walked slot by slot for every further key, which took minutes), and the second attempt - sized, and
    switched to the high-cardinality mode, by what the first taught - succeeds.  Counts against numpy."""
    import time
    n, groups = 8_000_000, 1_000_000
    r = np.random.default_rng(31)
    k = r.integers(0, groups, n)
    v = r.uniform(0, 1, n)
    batch = qe.Batch.from_numpy(ctx, [qe.Column(qe.TYPE_INT64, k), qe.Column(qe.TYPE_FLOAT64, v)]).to_device()
    C = qe.ColumnExpression
    t0 = time.perf_counter()
    out = list(qe.AggregateExec([qe.ScanExec(ctx, [batch])], [C(0)], [qe.CountExpression(None), qe.MaxExpression(C(1))]).execute())[0]
    cols = out.columns()
    seconds = time.perf_counter() - t0
    want = np.bincount(k, minlength=groups)
    order = np.argsort(np.asarray(cols[0].values))
    assert np.array_equal(np.asarray(cols[0].values)[order], np.nonzero(want)[0])
    assert np.array_equal(np.asarray(cols[1].values)[order], want[want > 0])
    mx = np.zeros(groups)
    np.maximum.at(mx, k, v)
    assert np.array_equal(np.asarray(cols[2].values)[order], mx[want > 0])
    assert seconds < 20, seconds  # milliseconds of device time; the bound only has to catch the slot-by-slot walk


@pytest.mark.parametrize("groups,hint", [(300, 0), (2000, 0), (2000, 2000), (30_000, 0), (60_000, 0), (700, 700), (5000, 64)])
def test_mid_cardinality_large_batches(qe, ctx, groups, hint):
    """Batches of a million rows and more whose groups may outnumber a workgroup's LDS table (no hint, or a hint
    beyond the table) go through the row log; a hint that undersells the groups stays on the tables.  SUM / MAX /
    COUNT against numpy; small-integer keys, whose hash clusters in the LDS table were once a cliff."""
    n = 3_000_000
    r = np.random.default_rng(groups + hint)
    k = r.integers(0, groups, n)
    v = r.uniform(-1, 1, n)
    batches = [qe.Batch.from_numpy(ctx, [qe.Column(qe.TYPE_INT64, k[s:e]), qe.Column(qe.TYPE_FLOAT64, v[s:e])]).to_device()
               for s, e in ((0, 1_700_000), (1_700_000, n))]
    E = qe.ColumnExpression
    out = list(qe.AggregateExec([qe.ScanExec(ctx, batches)], [E(0)], [qe.SumExpression(E(1)), qe.MaxExpression(E(1)), qe.CountExpression(None)],
                                group_capacity_hint=hint).execute())[0]
    cols = out.columns()
    order = np.argsort(np.asarray(cols[0].values))
    cnt = np.bincount(k, minlength=groups)
    assert np.array_equal(np.asarray(cols[0].values)[order], np.nonzero(cnt)[0])
    assert np.array_equal(np.asarray(cols[3].values)[order], cnt[cnt > 0])
    mx = np.full(groups, -np.inf)
    np.maximum.at(mx, k, v)
    assert np.array_equal(np.asarray(cols[2].values)[order], mx[cnt > 0])
    sums = np.bincount(k, weights=v, minlength=groups)
    mags = np.bincount(k, weights=np.abs(v), minlength=groups)
    assert (np.abs(np.asarray(cols[1].values)[order] - sums[cnt > 0]) <= 1e-12 * mags[cnt > 0] + 1e-300).all()


def test_null_outputs_of_a_large_table(qe, ctx):
    """Groups whose aggregate is null (every input of the group null) among 200 K groups: the output routine of tables
    beyond 64 K slots puts a wave's validity bits together before it writes them - rows valid and null mixed within a
    wave, runs of rows that start anywhere in a 32-row validity word."""
    n_keys = 200_000
    r = rng(41)
    k = np.tile(np.arange(n_keys, dtype=np.int64) * 7919 - 3, 3)
    r.shuffle(k)
    x = r.uniform(-10, 10, len(k))
    null_group = ((k + 3) // 7919) % 3 == 0  # a third of the groups see nulls only
    xv = ~null_group & (r.random(len(k)) > 0.2)
    y = r.integers(-1000, 1000, len(k)).astype(np.int64)
    yv = r.random(len(k)) > 0.6  # some groups lose all three of their rows by chance
    cols = [orc.Array(orc.INT64, k), orc.Array(orc.FLOAT64, x, xv), orc.Array(orc.INT64, y, yv)]
    aggs = [("sum", C(1)), ("min", C(1)), ("max", C(2)), ("avg", C(1)), ("count", C(2)), ("count", None)]
    got, want = run_agg(qe, ctx, [cols], [C(0)], aggs, hint=n_keys, what="null outputs, 200 K groups")
    sums = got[0].columns()[1]
    assert sums.validity is not None and 0.3 < 1 - sums.validity.mean() < 0.4
