"""Kernels compiled at run time for launch shapes that have no kernel built ahead of time: same
parity bar as the other paths, and the launch profile must show that the compiled kernel ran."""
import time

import numpy as np
import pytest

from oracle import oracle as orc
from test_gpu_parity import ALL_AGGS_F64, C, EXPRS, agg_columns, make_columns, mk, rng, run_agg, run_both
from util import assert_column_equal, gpu_batch

pytestmark = pytest.mark.gpu


@pytest.fixture()
def jit_ctx(qe, ctx):
    ctx.set_flags(qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP | qe.FLAG_JIT_SYNC)
    ctx.profile(True)
    ctx.profile_read()
    yield ctx
    ctx.profile(False)
    ctx.set_flags(qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP)


def jit_kernels(ctx):
    return [k for k in ctx.profile_read() if "_jit<" in k]


def test_expressions_compiled(qe, jit_ctx):
    """Every expression of the parity list (nulls, NaN, division by zero, MIN / -1, deep trees) through a
    compiled projection kernel, at a size with a ragged tail."""
    n = 5000 + 37
    cols = make_columns(n)
    chunk = orc.Chunk(cols)
    batch = gpu_batch(qe, jit_ctx, cols)
    for spec in EXPRS:
        want = mk(orc, spec).evaluate(chunk)
        got = mk(qe, spec).evaluate(batch)
        assert_column_equal(got.array.columns()[0], want.array, what=f"jit {spec}")
    assert len(jit_kernels(jit_ctx)) >= len(EXPRS) - 4  # shapes that coincide share a kernel


def test_filter_and_aggregate_compiled(qe, jit_ctx):
    cols = make_columns(20_000, seed=3)
    parts = [make_columns(n, seed=s) for n, s in ((5000, 1), (1, 2), (9000, 3))]

    def chain(m, scan):
        s1 = m.SelectionExec([scan], mk(m, ("gt", C(5), ("f64", 0.5))))
        p1 = m.ProjectionExec([s1], [mk(m, ("add", C(4), ("mul", C(5), C(5)))), mk(m, C(1)), mk(m, C(6)), mk(m, C(3))])
        s2 = m.SelectionExec([p1], mk(m, ("lt", C(1), ("i32", 25))))
        return m.ProjectionExec([s2], [mk(m, ("mul", C(0), ("f64", 2.0))), mk(m, C(2)), mk(m, ("add", C(3), ("i64", 1)))])
    run_both(qe, jit_ctx, parts, chain, what="jit selection/projection chain")
    run_both(qe, jit_ctx, [cols], lambda m, scan: m.SelectionExec([scan], mk(m, ("gt", C(4), ("f64", 0.0)))), what="jit null predicate, all columns")
    assert any(k.startswith("qe_filter_jit<") for k in jit_kernels(jit_ctx))
    # nullable outputs over many tiles: a wave's validity bits leave as whole words that neighbouring waves share, at
    # every alignment (few, half and nearly all rows kept; a ragged last tile)
    big = make_columns(300_017, seed=5)
    for lim in (0.03, 0.5, 0.97):
        run_both(qe, jit_ctx, [big], lambda m, scan: m.SelectionExec([scan], mk(m, ("lt", C(5), ("f64", lim)))), what=f"jit nullable outputs, {lim} kept")
    for n, n_keys in ((1, 1), (5000, 10), (70_000, 3000)):
        run_agg(qe, jit_ctx, [agg_columns(n, n_keys, seed=n + n_keys)], [C(0)], ALL_AGGS_F64, what=f"jit agg n={n} keys={n_keys}")
    acols = agg_columns(30_000, 40, seed=11, key_dtype=np.int32)
    run_agg(qe, jit_ctx, [acols], [C(0), C(1)], ALL_AGGS_F64, what="jit two int32 keys")
    run_agg(qe, jit_ctx, [acols], [("mod", C(1), ("i32", 2))], [("sum", ("add", C(2), ("mul", C(5), C(5)))), ("max", ("mul", C(4), C(1))), ("avg", C(3))],
            pred=("gt", C(5), ("f64", 0.5)), what="jit computed key, filter, int avg")
    assert any(k.startswith("qe_aggregate_jit<") for k in jit_kernels(jit_ctx))


def test_background_compilation(qe, ctx):
    """Default mode: the first launches of a new shape run on the general kernel while the compiler works
    on another thread; later launches pick the compiled kernel up.  Results agree."""
    n = 5_000_000  # above the row threshold of background compilation
    r = rng(4)
    a, b = r.uniform(-1, 1, n), r.uniform(0, 1, n)
    batch = gpu_batch(qe, ctx, [orc.Array(orc.FLOAT64, a), orc.Array(orc.FLOAT64, b)])
    expr = lambda: qe.SubExpression(qe.MulExpression(qe.ColumnExpression(0), qe.LiteralFloatExpression(3.25)), qe.ColumnExpression(1))
    want = a * 3.25 - b
    ctx.profile(True)
    try:
        ctx.profile_read()
        first = expr().evaluate(batch).array.columns()[0].values
        names = set(ctx.profile_read())
        assert np.array_equal(first, want)
        deadline = time.time() + 60
        while not any("_jit<" in k for k in names) and time.time() < deadline:
            time.sleep(0.2)
            got = expr().evaluate(batch).array.columns()[0].values
            names = set(ctx.profile_read())
        assert any(k.startswith("qe_project_jit<") for k in names), names
        assert np.array_equal(got, want)
    finally:
        ctx.profile(False)


def test_dense_key_bins(qe, jit_ctx):
    """Aggregates whose small keys go to the per-lane LDS bins (no nullable inputs, few groups): keys on
    both sides of the bin range and below zero in one batch, every accumulator kind, int32 and computed keys,
    a filter in front, several batches."""
    def cols(n, seed, lo, hi, key_dtype=np.int64):
        a = agg_columns(n, 1, seed=seed, key_dtype=key_dtype, nulls=False)
        a[0] = orc.Array(a[0].type, rng(seed + 1).integers(lo, hi, n).astype(key_dtype))
        return a
    aggs = [("max", C(2)), ("min", C(2)), ("sum", C(2)), ("count", None), ("avg", C(2)), ("sum", C(3)), ("min", C(4)), ("max", C(3))]
    for n, lo, hi in ((1, 3, 4), (4097, 0, 16), (50_000, -5, 40), (30_000, 14, 18)):
        run_agg(qe, jit_ctx, [cols(n, n, lo, hi)], [C(0)], aggs, hint=64, what=f"dense bins n={n} keys [{lo},{hi})")
    run_agg(qe, jit_ctx, [cols(20_000, 5, 0, 10, np.int32), cols(7, 6, 8, 20, np.int32), cols(9000, 7, -2, 12, np.int32)], [C(0)], aggs, hint=64,
            what="dense bins, int32 key, three batches")
    run_agg(qe, jit_ctx, [cols(40_000, 9, 0, 1000)], [("mod", C(0), ("i64", 7))], [("sum", ("mul", C(5), C(2))), ("count", None), ("max", C(5))],
            hint=64, pred=("gt", C(5), ("f64", 0.25)), what="dense bins, computed key, filter")
    run_agg(qe, jit_ctx, [cols(40_000, 10, 0, 3)], [("gt", C(5), ("f64", 0.5)), C(1)], [("sum", C(2)), ("avg", C(5))], hint=64,
            what="boolean + int32 key packed together")
    assert any(k.startswith("qe_aggregate_jit<") for k in jit_kernels(jit_ctx))


def test_skewed_keys_outside_the_bins(qe, jit_ctx):
    """A few keys that dominate every wave-step, none of them small enough for the dense bins: two sit in the
    hot-key registers, the next is reduced across the wave, the rest meet in the LDS table.  (Once the dominant
    key's fold was skipped when no other row was left in a wave-step.)"""
    n = 60_000
    r = rng(19)
    for shares in ([0.6, 0.25, 0.1, 0.05], [0.5, 0.5], [1.0], [0.34, 0.33, 0.33]):
        keys = np.array([10_000 + 1000 * i for i in range(len(shares))], dtype=np.int64)
        a = agg_columns(n, 1, seed=len(shares), nulls=False)
        a[0] = orc.Array(orc.INT64, r.choice(keys, size=n, p=shares))
        run_agg(qe, jit_ctx, [a, [orc.Array(x.type, x.values[:777]) for x in a]], [C(0)],
                [("sum", C(2)), ("min", C(2)), ("max", C(3)), ("count", None), ("avg", C(5))], hint=64, what=f"skewed keys {shares}")
        run_agg(qe, jit_ctx, [a], [("gt", C(5), ("f64", 0.5)), C(0)], [("sum", C(3)), ("count", None)], what=f"boolean next to skewed keys {shares}")
    assert any(k.startswith("qe_aggregate_jit<") for k in jit_kernels(jit_ctx))
