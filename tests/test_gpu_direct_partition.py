"""The many-groups aggregate whose first pass partitions its rows itself (no workgroup table, no log): chosen when the
first rows of a large batch nearly all miss the workgroups' tables - very many keys, none of them frequent.  Checked
against torch reductions over the same device columns (keys and counts exactly, minima / maxima bit for bit, sums
within 1e-12 of the magnitudes summed) and, on a prefix, against the oracle; the kernel that ran is read from the
context's launch profile.  AggregateExec::execute, physical_plan/mod.rs:292-489.
"""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

REL = 1e-12
GROUPS = 1_000_000


@pytest.fixture(scope="module")
def torch_ctx(qe):
    import torch
    torch.cuda.set_device(0)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    c = qe.Context(0, stream=stream.cuda_stream)
    c.set_flags(qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP | qe.FLAG_JIT_SYNC)
    yield torch, c
    torch.cuda.synchronize()
    c.close()
    torch.cuda.set_stream(torch.cuda.default_stream())


def columns(torch, ctx, n, seed=0):
    dev = torch.device("cuda", 0)
    k = torch.empty(n, dtype=torch.int64, device=dev)
    a, b, c = (torch.empty(n, dtype=torch.float64, device=dev) for _ in range(3))
    ctx.generate(1, 99 + seed, 0, n, 0, GROUPS, k.data_ptr())  # uniform keys
    ctx.generate(0, 44 + seed, 0, n, -1.0, 1.0, a.data_ptr())
    ctx.generate(0, 45 + seed, 0, n, -1.0, 1.0, b.data_ptr())
    ctx.generate(0, 46 + seed, 0, n, 0.0, 1.0, c.data_ptr())
    ctx.synchronize()
    return k, a, b, c


def by_key(cols):
    order = np.argsort(np.asarray(cols[0].values))
    return [np.asarray(c.values)[order] for c in cols]


def kernels_of(ctx):
    return set(ctx.profile_read().keys())


def headline_plan(qe, ctx, batches, aggs, hint=1_100_000, mode=None):
    C = qe.ColumnExpression
    sel = qe.SelectionExec([qe.ScanExec(ctx, batches)], qe.GtExpression(C(3), qe.LiteralFloatExpression(0.5)))
    proj = qe.ProjectionExec([sel], [C(0), qe.AddExpression(C(1), qe.MulExpression(C(2), C(3)))])
    return qe.AggregateExec([proj], [C(0)], aggs, group_capacity_hint=hint, **({"mode": mode} if mode is not None else {}))


def check_against_torch(torch, k, v, keep, got, with_min_max):
    """got: keys, sum, count, avg[, min, max] in key order; the reference: torch over the kept rows."""
    dev = k.device
    kk, vv = k[keep], v[keep]
    uk, inv, cnt = torch.unique(kk, return_inverse=True, return_counts=True)
    g = uk.shape[0]
    assert np.array_equal(got[0], uk.cpu().numpy()) and np.array_equal(got[2], cnt.cpu().numpy())
    ref_sum = torch.zeros(g, dtype=torch.float64, device=dev).scatter_add_(0, inv, vv).cpu().numpy()
    ref_mag = torch.zeros(g, dtype=torch.float64, device=dev).scatter_add_(0, inv, vv.abs()).cpu().numpy()
    assert (np.abs(got[1] - ref_sum) <= 2 * REL * ref_mag).all()
    assert (np.abs(got[3] - ref_sum / got[2]) <= 2 * REL * ref_mag / got[2]).all()
    if with_min_max:
        ref_min = torch.full((g,), 9.0, dtype=torch.float64, device=dev).scatter_reduce_(0, inv, vv, "amin").cpu().numpy()
        ref_max = torch.full((g,), -9.0, dtype=torch.float64, device=dev).scatter_reduce_(0, inv, vv, "amax").cpu().numpy()
        assert np.array_equal(got[4], ref_min) and np.array_equal(got[5], ref_max)


@pytest.mark.parametrize("n_batches", [1, 2, 3, 6])
def test_uniform_keys_take_the_partitioning_first_pass(qe, torch_ctx, n_batches):
    torch, ctx = torch_ctx
    n = 72_000_001  # (ragged: the last wave-step is partial, the batches differ in size)
    k, a, b, c = columns(torch, ctx, n)
    per = (n + n_batches - 1) // n_batches
    batches = [qe.Batch.from_torch(ctx, [t[s:s + per] for t in (k, a, b, c)]) for s in range(0, n, per)]
    C = qe.ColumnExpression
    aggs = [qe.SumExpression(C(1)), qe.CountExpression(None), qe.AvgExpression(C(1)), qe.MinExpression(C(1)), qe.MaxExpression(C(1))]
    ctx.profile(True)
    got = by_key(list(headline_plan(qe, ctx, batches, aggs).execute())[0].columns())
    names = kernels_of(ctx)
    ctx.profile(False)
    assert any(name.startswith("qe_aggregate_direct") for name in names), names
    check_against_torch(torch, k, a + b * c, c > 0.5, got, True)


def test_prefix_against_the_oracle(qe, torch_ctx):
    """The same plan over the first 40 M rows, of which the oracle sees what falls on 300 keys (a filter on the key
    keeps its literal aggregate tractable): those groups' rows of the engine's output are the oracle's."""
    torch, ctx = torch_ctx
    n = 40_000_000
    k, a, b, c = columns(torch, ctx, n, seed=3)
    C = qe.ColumnExpression
    aggs = [qe.SumExpression(C(1)), qe.CountExpression(None), qe.AvgExpression(C(1))]
    ctx.profile(True)
    got = by_key(list(headline_plan(qe, ctx, [qe.Batch.from_torch(ctx, [k, a, b, c])], aggs).execute())[0].columns())
    names = kernels_of(ctx)
    ctx.profile(False)
    assert any(name.startswith("qe_aggregate_direct") for name in names), names
    # the oracle's plan over the rows of 300 keys (a group's result depends on its own rows only)
    from test_gpu_parity import C as CS, mk
    pick = k < 300
    arrays = [orc.Array(orc.INT64, k[pick].cpu().numpy())] + [orc.Array(orc.FLOAT64, t[pick].cpu().numpy()) for t in (a, b, c)]
    sel = orc.SelectionExec([orc.ScanExec([orc.Chunk(arrays)])], mk(orc, ("gt", CS(3), ("f64", 0.5))))
    proj = orc.ProjectionExec([sel], [mk(orc, CS(0)), mk(orc, ("add", CS(1), ("mul", CS(2), CS(3))))])
    want = list(orc.AggregateExec([proj], [mk(orc, CS(0))], [mk(orc, ("sum", CS(1))), mk(orc, ("count", None)), mk(orc, ("avg", CS(1)))]).execute())[0]
    order = np.argsort(want.arrays[0].values)
    wk, ws, wc, wa = (want.arrays[i].values[order] for i in range(4))
    assert len(wk) == 300 and np.array_equal(got[0][:300], wk) and np.array_equal(got[2][:300], wc)
    kept = pick & (c > 0.5)
    mag = np.zeros(300)
    np.add.at(mag, k[kept].cpu().numpy(), np.abs((a + b * c)[kept].cpu().numpy()))
    assert (np.abs(got[1][:300] - ws) <= REL * mag).all() and (np.abs(got[3][:300] - wa) <= REL * mag / wc).all()


def test_keys_that_turn_skewed_start_over_through_the_log(qe, torch_ctx):
    """The first rows look uniform, everything after them holds 1 000 keys: the stretches of those keys' partitions
    overflow, the aggregation notices at the end and runs again through the first pass with a table and a log."""
    torch, ctx = torch_ctx
    n = 48_000_000
    k, a, b, c = columns(torch, ctx, n, seed=5)
    k[4_000_000:] %= 1000
    torch.cuda.synchronize()
    C = qe.ColumnExpression
    aggs = [qe.SumExpression(C(1)), qe.CountExpression(None), qe.AvgExpression(C(1)), qe.MinExpression(C(1)), qe.MaxExpression(C(1))]
    ctx.profile(True)
    got = by_key(list(headline_plan(qe, ctx, [qe.Batch.from_torch(ctx, [k, a, b, c])], aggs).execute())[0].columns())
    names = kernels_of(ctx)
    ctx.profile(False)
    assert any(name.startswith("qe_aggregate_direct") for name in names), names  # tried ...
    assert any(name.startswith("qe_aggregate_spill") for name in names), names   # ... and done again
    check_against_torch(torch, k, a + b * c, c > 0.5, got, True)


def test_streaming_protocol_and_partial_final(qe, torch_ctx):
    """The Accumulator protocol fed batch by batch (physical_expressions.rs:351-362), and PARTIAL aggregates of ragged
    parts merged by a FINAL one: both equal the single plan's output."""
    torch, ctx = torch_ctx
    n = 60_000_000
    k, a, b, c = columns(torch, ctx, n, seed=7)
    v = a + b * c
    C = qe.ColumnExpression
    aggs = lambda: [qe.SumExpression(C(1)), qe.CountExpression(None), qe.AvgExpression(C(1))]
    cuts = [0, 25_000_000, n]
    st = qe.AggregationState(ctx, [C(0)], aggs(), group_capacity_hint=1_100_000)
    ctx.profile(True)
    for lo, hi in zip(cuts, cuts[1:]):
        st.accumulate(qe.Batch.from_torch(ctx, [k[lo:hi], v[lo:hi]]))
    streamed = by_key(st.final_value().columns())
    names = kernels_of(ctx)
    ctx.profile(False)
    st.close()
    assert any(name.startswith("qe_aggregate_direct") for name in names), names
    everything = torch.ones_like(k, dtype=torch.bool)
    check_against_torch(torch, k, v, everything, streamed, False)
    parts = []
    for lo, hi in zip(cuts, cuts[1:]):
        parts += list(qe.AggregateExec([qe.ScanExec(ctx, [qe.Batch.from_torch(ctx, [k[lo:hi], v[lo:hi]])])], [C(0)], aggs(), mode=qe.AGG_MODE_PARTIAL,
                                       group_capacity_hint=1_100_000).execute())
    merged = by_key(list(qe.final_plan(qe.ScanExec(ctx, parts), 1, aggs(), group_capacity_hint=1_100_000).execute())[0].columns())
    assert np.array_equal(streamed[0], merged[0]) and np.array_equal(streamed[2], merged[2])
    mag = np.abs(streamed[1]) + 1.0
    assert (np.abs(streamed[1] - merged[1]) <= 1e-9 * mag).all()


@pytest.mark.parametrize("n_keys,skewed", [(2_400_000, False), (5_000_000, False), (2_400_000, True)])
def test_more_groups_than_the_partitions_tables_hold(qe, torch_ctx, n_keys, skewed):
    """Hints beyond what 512 partitions' LDS tables hold (1.3 M groups of this shape): the partitions stay 512 and each
    is folded in two or four passes by further hash bits - after the first pass that partitions directly (uniform
    keys) and after the one with a table and a log (zipf keys), over two batches.  Against torch, as above."""
    torch, ctx = torch_ctx
    n = 64_000_003
    dev = torch.device("cuda", 0)
    k = torch.empty(n, dtype=torch.int64, device=dev)
    a, b, c = (torch.empty(n, dtype=torch.float64, device=dev) for _ in range(3))
    if skewed:
        ctx.generate(2, 177, 0, n, n_keys, 1.1, k.data_ptr())
    else:
        ctx.generate(1, 199, 0, n, 0, n_keys, k.data_ptr())
    ctx.generate(0, 144, 0, n, -1.0, 1.0, a.data_ptr())
    ctx.generate(0, 145, 0, n, -1.0, 1.0, b.data_ptr())
    ctx.generate(0, 146, 0, n, 0.0, 1.0, c.data_ptr())
    ctx.synchronize()
    half = n // 2 + 7
    batches = [qe.Batch.from_torch(ctx, [t[s:e] for t in (k, a, b, c)]) for s, e in ((0, half), (half, n))]
    C = qe.ColumnExpression
    aggs = [qe.SumExpression(C(1)), qe.CountExpression(None), qe.AvgExpression(C(1)), qe.MinExpression(C(1)), qe.MaxExpression(C(1))]
    ctx.profile(True)
    got = by_key(list(headline_plan(qe, ctx, batches, aggs, hint=n_keys + n_keys // 10).execute())[0].columns())
    names = kernels_of(ctx)
    ctx.profile(False)
    assert any(name.startswith("qe_aggregate_partitioned") for name in names), names
    assert any(name.startswith("qe_aggregate_spill" if skewed else "qe_aggregate_direct") for name in names), names
    assert not any("level" in name for name in names), names  # (no two-level scatter: 512 partitions)
    v, keep = a + b * c, c > 0.5
    if not skewed:
        check_against_torch(torch, k, v, keep, got, True)
        return
    # Zipf keys: torch's scatter_add_ / scatter_reduce_ take a minute apiece over a key that holds a tenth of the rows
    # (atomics on one address).  Counts from torch.unique, sums by numpy's bincount on the host (one pass, in row order),
    # minima and maxima by sorting: values ascending, then a stable sort by key - a group's first row is its minimum,
    # its last its maximum.
    uk, inv, cnt = torch.unique(k[keep], return_inverse=True, return_counts=True)
    assert np.array_equal(got[0], uk.cpu().numpy()) and np.array_equal(got[2], cnt.cpu().numpy())
    inv_h, v_h = inv.cpu().numpy(), v[keep].cpu().numpy()
    ref_sum = np.bincount(inv_h, weights=v_h, minlength=len(got[0]))
    ref_mag = np.bincount(inv_h, weights=np.abs(v_h), minlength=len(got[0]))
    assert (np.abs(got[1] - ref_sum) <= 2 * REL * ref_mag).all()
    assert (np.abs(got[3] - ref_sum / got[2]) <= 2 * REL * ref_mag / got[2]).all()
    vv, order = torch.sort(v[keep], stable=True)
    kk, order2 = torch.sort(k[keep][order], stable=True)
    vv = vv[order2]
    first = torch.ones_like(kk, dtype=torch.bool)
    first[1:] = kk[1:] != kk[:-1]
    last = torch.ones_like(first)
    last[:-1] = first[1:]
    assert np.array_equal(got[4], vv[first].cpu().numpy()) and np.array_equal(got[5], vv[last].cpu().numpy())


def test_several_value_columns_and_more_partitions_than_512(qe, torch_ctx):
    """Five accumulators over two value columns and a million groups: the log carries two value columns (no pairs, so
    neither the one-pass scatter nor the direct first pass, and no folding of a partition in passes either): 1024
    partitions through the two scatter levels.  Sums against torch, minima and maxima bit for bit."""
    torch, ctx = torch_ctx
    n = 24_000_001
    k, a, b, _ = columns(torch, ctx, n, seed=11)
    C = qe.ColumnExpression
    aggs = [qe.SumExpression(C(1)), qe.SumExpression(C(2)), qe.MinExpression(C(1)), qe.MaxExpression(C(2)), qe.AvgExpression(C(2)), qe.CountExpression(None)]
    half = n // 2 - 5
    batches = [qe.Batch.from_torch(ctx, [t[s:e] for t in (k, a, b)]) for s, e in ((0, half), (half, n))]
    ctx.profile(True)
    got = by_key(list(qe.AggregateExec([qe.ScanExec(ctx, batches)], [C(0)], aggs, group_capacity_hint=1_100_000).execute())[0].columns())
    names = kernels_of(ctx)
    ctx.profile(False)
    assert any(name.startswith("qe_aggregate_partitioned") for name in names), names
    dev = k.device
    uk, inv, cnt = torch.unique(k, return_inverse=True, return_counts=True)
    g = uk.shape[0]
    assert np.array_equal(got[0], uk.cpu().numpy()) and np.array_equal(got[6], cnt.cpu().numpy())
    for col, t in ((1, a), (2, b)):
        ref = torch.zeros(g, dtype=torch.float64, device=dev).scatter_add_(0, inv, t).cpu().numpy()
        mag = torch.zeros(g, dtype=torch.float64, device=dev).scatter_add_(0, inv, t.abs()).cpu().numpy()
        assert (np.abs(got[col] - ref) <= 2 * REL * mag).all()
        if col == 2:
            assert (np.abs(got[5] - ref / got[6]) <= 2 * REL * mag / got[6]).all()
    ref_min = torch.full((g,), 9.0, dtype=torch.float64, device=dev).scatter_reduce_(0, inv, a, "amin").cpu().numpy()
    ref_max = torch.full((g,), -9.0, dtype=torch.float64, device=dev).scatter_reduce_(0, inv, b, "amax").cpu().numpy()
    assert np.array_equal(got[3], ref_min) and np.array_equal(got[4], ref_max)
