"""The configurations at their full sizes (100 M / 500 M / 1 B rows, and one batch past 2^31 rows),
checked through properties that do not need the oracle to chew through them: counts against
torch reductions over the same device columns, stable compaction as "the output of a prefix is a
prefix of the output" (that prefix checked bit for bit against the oracle), the aggregate of the
whole against the merge of the aggregates of its parts, sums within 1e-12 of the magnitude summed.
"""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

REL = 1e-12  # float64 SUM / AVG tolerance, relative to the sum of magnitudes


@pytest.fixture(scope="module")
def torch_ctx(qe):
    """Engine and torch on one stream, so that torch reductions see what the engine wrote and vice versa."""
    import torch
    torch.cuda.set_device(0)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    c = qe.Context(0, stream=stream.cuda_stream)
    yield torch, c
    torch.cuda.synchronize()
    c.close()
    torch.cuda.set_stream(torch.cuda.default_stream())


def table(torch, ctx, n, key_kind=3, key_lo=0, key_hi=0):
    dev = torch.device("cuda", 0)
    k = torch.empty(n, dtype=torch.int64, device=dev)
    a, b, c = (torch.empty(n, dtype=torch.float64, device=dev) for _ in range(3))
    ctx.generate(key_kind, 43, 0, n, key_lo, key_hi, k.data_ptr())
    ctx.generate(0, 44, 0, n, -1.0, 1.0, a.data_ptr())
    ctx.generate(0, 45, 0, n, -1.0, 1.0, b.data_ptr())
    ctx.generate(0, 46, 0, n, 0.0, 1.0, c.data_ptr())
    ctx.synchronize()
    return k, a, b, c


def by_key(cols):
    order = np.argsort(np.asarray(cols[0].values))
    return [np.asarray(c.values)[order] for c in cols]


def test_c2_filter_project_100m(qe, torch_ctx):
    torch, ctx = torch_ctx
    n = 100_000_000
    _, a, b, c = table(torch, ctx, n)
    C = qe.ColumnExpression
    pred = lambda m: m.GtExpression(m.ColumnExpression(2), m.LiteralFloatExpression(0.5))
    expr = lambda m: m.AddExpression(m.ColumnExpression(0), m.MulExpression(m.ColumnExpression(1), m.ColumnExpression(2)))
    batch = qe.Batch.from_torch(ctx, [a, b, c])
    out = list(qe.ProjectionExec([qe.SelectionExec([qe.ScanExec(ctx, [batch])], pred(qe))], [expr(qe)]).execute())[0]
    keep = c > 0.5
    assert out.num_rows == int(keep.sum())
    desc = out.column_desc(0)
    from bench import DevArray
    got = torch.as_tensor(DevArray(desc.values, desc.length, "<f8"), device="cuda")
    # the whole output, bit for bit: torch evaluates a + b*c with two roundings as well
    want = (a + b * c)[keep]
    assert torch.equal(got.view(torch.int64), want.view(torch.int64))
    # and the oracle on a prefix: stable compaction makes its output a prefix of the output
    m = 1_000_000
    chunk = orc.Chunk([orc.Array(orc.FLOAT64, t[:m].cpu().numpy()) for t in (a, b, c)])
    ref = list(orc.ProjectionExec([orc.SelectionExec([orc.ScanExec([chunk])], pred(orc))], [expr(orc)]).execute())[0].arrays[0].values
    assert np.array_equal(got[:len(ref)].cpu().numpy().view(np.uint64), ref.view(np.uint64))


def test_c3_groupby_500m(qe, torch_ctx):
    torch, ctx = torch_ctx
    n = 500_000_000
    k, _, _, c = table(torch, ctx, n)
    C = qe.ColumnExpression
    step = 1 << 28
    batches = [qe.Batch.from_torch(ctx, [k[s:s + step], c[s:s + step]]) for s in range(0, n, step)]
    out = list(qe.AggregateExec([qe.ScanExec(ctx, batches)], [C(0)], [qe.SumExpression(C(1)), qe.CountExpression(None), qe.MinExpression(C(1)),
                                                                        qe.MaxExpression(C(1))], group_capacity_hint=64).execute())[0]
    keys, sums, counts, mins, maxs = by_key(out.columns())
    want_counts = torch.bincount(k).cpu().numpy()
    assert np.array_equal(keys, np.nonzero(want_counts)[0]) and np.array_equal(counts, want_counts[want_counts > 0])
    assert counts.sum() == n
    for key, total, lo, hi in zip(keys, sums, mins, maxs):
        mine = k == int(key)
        want = float(torch.where(mine, c, torch.zeros((), dtype=torch.float64, device="cuda")).sum())  # torch.sum reduces as a tree
        assert abs(total - want) <= REL * want  # values are positive: the sum of magnitudes is the sum
        sel = c[mine]
        assert lo == float(sel.min()) and hi == float(sel.max())


def test_c5_whole_equals_merge_of_parts_1b(qe, torch_ctx):
    torch, ctx = torch_ctx
    n = 1_000_000_000
    k, a, b, c = table(torch, ctx, n)
    C = qe.ColumnExpression
    aggs = lambda: [qe.SumExpression(C(1)), qe.CountExpression(None), qe.AvgExpression(C(1))]

    def plan(lo, hi, mode):
        step = 1 << 28
        batches = [qe.Batch.from_torch(ctx, [t[s:min(hi, s + step)] for t in (k, a, b, c)]) for s in range(lo, hi, step)]
        sel = qe.SelectionExec([qe.ScanExec(ctx, batches)], qe.GtExpression(C(3), qe.LiteralFloatExpression(0.5)))
        proj = qe.ProjectionExec([sel], [C(0), qe.AddExpression(C(1), qe.MulExpression(C(2), C(3)))])
        return qe.AggregateExec([proj], [C(0)], aggs(), mode=mode, group_capacity_hint=64)
    whole = by_key(list(plan(0, n, qe.AGG_MODE_SINGLE).execute())[0].columns())
    cuts = [0, 123_456_789, 600_000_001, n]  # ragged shards
    parts = []
    for lo, hi in zip(cuts, cuts[1:]):
        parts += list(plan(lo, hi, qe.AGG_MODE_PARTIAL).execute())
    merged = by_key(list(qe.final_plan(qe.ScanExec(ctx, parts), 1, aggs()).execute())[0].columns())
    assert np.array_equal(whole[0], merged[0]) and np.array_equal(whole[2], merged[2])
    keep = c > 0.5
    assert whole[2].sum() == int(keep.sum())
    v = a + b * c
    zero = torch.zeros((), dtype=torch.float64, device="cuda")
    for i, key in enumerate(whole[0]):
        mine = keep & (k == int(key))
        ref = float(torch.where(mine, v, zero).sum())  # torch.sum reduces as a tree: accurate far below the tolerance
        mag = float(torch.where(mine, v.abs(), zero).sum())
        assert abs(whole[1][i] - ref) <= REL * mag and abs(merged[1][i] - ref) <= REL * mag
        assert abs(whole[3][i] - ref / whole[2][i]) <= REL * mag / whole[2][i]


def test_deterministic_sum_same_bits_at_headline_size(qe, torch_ctx):
    """QE_FLAG_DETERMINISTIC_SUM on the headline's shape: 800 M rows in three batches of 2^28, generated on the device.
    Two runs give the same bits group by group (rows put in key order: the table's order changes from run to run, as the
    reference's HashMap order does, physical_plan/mod.rs:375-376), and the sums lie within 1e-12 of the magnitudes summed
    of a torch tree reduction.  The reference folds in one fixed order (physical_plan/mod.rs:340-372); what is promised
    here is identity from run to run."""
    torch, ctx = torch_ctx
    n = 3 << 28
    k, a, b, c = table(torch, ctx, n)
    C = qe.ColumnExpression
    step = 1 << 28
    batches = [qe.Batch.from_torch(ctx, [t[s:s + step] for t in (k, a, b, c)]) for s in range(0, n, step)]
    assert len(batches) == 3

    def run():
        sel = qe.SelectionExec([qe.ScanExec(ctx, batches)], qe.GtExpression(C(3), qe.LiteralFloatExpression(0.5)))
        proj = qe.ProjectionExec([sel], [C(0), qe.AddExpression(C(1), qe.MulExpression(C(2), C(3)))])
        plan = qe.AggregateExec([proj], [C(0)], [qe.SumExpression(C(1)), qe.CountExpression(None), qe.AvgExpression(C(1))], group_capacity_hint=64)
        return by_key(list(plan.execute())[0].columns())
    saved = ctx.flags
    ctx.set_flags(qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP | qe.FLAG_DETERMINISTIC_SUM | qe.FLAG_JIT_SYNC)
    try:
        one, two, three = run(), run(), run()
    finally:
        ctx.set_flags(saved)
    for other in (two, three):
        assert np.array_equal(one[0], other[0]) and np.array_equal(one[2], other[2])
        assert np.array_equal(one[1].view(np.uint64), other[1].view(np.uint64)), "SUM differs in its bits between two runs"
        assert np.array_equal(one[3].view(np.uint64), other[3].view(np.uint64)), "AVG differs in its bits between two runs"
    keep = c > 0.5
    assert one[2].sum() == int(keep.sum())
    v = a + b * c
    zero = torch.zeros((), dtype=torch.float64, device="cuda")
    for i, key in enumerate(one[0]):
        mine = keep & (k == int(key))
        ref = float(torch.where(mine, v, zero).sum())  # torch.sum reduces as a tree
        mag = float(torch.where(mine, v.abs(), zero).sum())
        assert abs(one[1][i] - ref) <= REL * mag
        assert abs(one[3][i] - ref / one[2][i]) <= REL * mag / one[2][i]


def test_one_batch_past_2_to_31_rows(qe, torch_ctx):
    """2.4 G rows in a single batch: row positions and counts are 64-bit all the way."""
    torch, ctx = torch_ctx
    n = 2_400_000_000
    half = 1_200_000_000
    dev = torch.device("cuda", 0)
    k = torch.empty(n, dtype=torch.int64, device=dev)
    c = torch.empty(n, dtype=torch.float64, device=dev)
    ctx.generate(3, 43, 0, n, 0, 0, k.data_ptr())
    ctx.generate(0, 46, 0, n, 0.0, 1.0, c.data_ptr())
    ctx.synchronize()
    C = qe.ColumnExpression

    def count_by_key(lo, hi):
        b = qe.Batch.from_torch(ctx, [k[lo:hi], c[lo:hi]])
        out = list(qe.AggregateExec([qe.ScanExec(ctx, [b])], [C(0)], [qe.CountExpression(None), qe.MaxExpression(C(1))], group_capacity_hint=64).execute())[0]
        return by_key(out.columns())
    whole, first, second = count_by_key(0, n), count_by_key(0, half), count_by_key(half, n)
    assert whole[1].sum() == n
    assert np.array_equal(whole[0], first[0]) and np.array_equal(whole[1], first[1] + second[1])
    assert np.array_equal(whole[2], np.maximum(first[2], second[2]))

    # stable compaction across the 2^31 boundary: the output of the whole is the outputs of the halves, in order
    def select(lo, hi):
        b = qe.Batch.from_torch(ctx, [k[lo:hi], c[lo:hi]])
        out = list(qe.SelectionExec([qe.ScanExec(ctx, [b])], qe.GtExpression(C(1), qe.LiteralFloatExpression(0.999))).execute())[0]
        cols = out.columns()
        return np.asarray(cols[0].values), np.asarray(cols[1].values)
    wk, wc = select(0, n)
    fk, fc = select(0, half)
    sk, sc = select(half, n)
    assert len(wc) == len(fc) + len(sc) == int((c > 0.999).sum())
    assert np.array_equal(wk, np.concatenate([fk, sk])) and np.array_equal(wc.view(np.uint64), np.concatenate([fc, sc]).view(np.uint64))


def test_c4_zipf_1m_keys_500m(qe, torch_ctx):
    """C4 at its full size: 500 M rows, 1 M zipf(1.1) Int64 keys, SUM / MIN / MAX / AVG / COUNT of one float64 column
    through the high-cardinality path.  Keys and counts against torch.unique, sums against a torch scatter-add within
    1e-12 of the magnitudes summed, minima / maxima bit for bit against torch scatter-reduce, the whole against the
    merge of ragged parts, and the oracle on a prefix of the rows."""
    torch, ctx = torch_ctx
    n = 500_000_000
    dev = torch.device("cuda", 0)
    k = torch.empty(n, dtype=torch.int64, device=dev)
    v = torch.empty(n, dtype=torch.float64, device=dev)
    ctx.generate(2, 77, 0, n, 1_000_000, 1.1, k.data_ptr())
    ctx.generate(0, 46, 0, n, 0.0, 1.0, v.data_ptr())
    ctx.synchronize()
    C = qe.ColumnExpression
    aggs = lambda: [qe.SumExpression(C(1)), qe.MinExpression(C(1)), qe.MaxExpression(C(1)), qe.AvgExpression(C(1)), qe.CountExpression(None)]
    step = 1 << 28

    def plan(lo, hi, mode):
        batches = [qe.Batch.from_torch(ctx, [k[s:min(hi, s + step)], v[s:min(hi, s + step)]]) for s in range(lo, hi, step)]
        return qe.AggregateExec([qe.ScanExec(ctx, batches)], [C(0)], aggs(), mode=mode, group_capacity_hint=1_100_000)
    print("c4: table generated", flush=True)
    whole = by_key(list(plan(0, n, qe.AGG_MODE_SINGLE).execute())[0].columns())
    keys, sums, mins, maxs, avgs, counts = whole
    print("c4: aggregated", len(keys), flush=True)
    uk, inv, cnt = torch.unique(k, return_inverse=True, return_counts=True)  # sorted keys
    assert np.array_equal(keys, uk.cpu().numpy()) and np.array_equal(counts, cnt.cpu().numpy())
    assert counts.sum() == n and len(keys) > 900_000
    print("c4: torch.unique done", flush=True)
    g = uk.shape[0]
    # References.  torch's scatter reductions are one atomic per row on the group's address (a compare-swap loop for
    # min / max): on the heaviest keys - tens of millions of rows on one address - they take minutes, and the rounding
    # error of such a serial float64 sum is itself of the size of the tolerance.  So the 64 heaviest keys are checked
    # against tree reductions over a mask (torch.sum / min / max), all other keys against the scatter reductions.
    heavy = np.argsort(counts)[-64:]
    zero = torch.zeros((), dtype=torch.float64, device=dev)
    for i in heavy:
        mine = k == int(keys[i])
        tree = float(torch.where(mine, v, zero).sum())  # values are positive: the sum of magnitudes is the sum
        assert abs(sums[i] - tree) <= REL * tree and abs(avgs[i] - tree / counts[i]) <= REL * tree / counts[i]
        assert mins[i] == float(torch.where(mine, v, zero + 2.0).min()) and maxs[i] == float(torch.where(mine, v, zero - 1.0).max())
    print("c4: heavy keys checked", flush=True)
    is_heavy = torch.zeros(g, dtype=torch.bool, device=dev)
    is_heavy[torch.from_numpy(heavy).to(dev)] = True
    rows = ~is_heavy[inv]
    inv_l, v_l = inv[rows], v[rows]
    del rows
    ref_sum = torch.zeros(g, dtype=torch.float64, device=dev).scatter_add_(0, inv_l, v_l).cpu().numpy()
    ref_min = torch.full((g,), 2.0, dtype=torch.float64, device=dev).scatter_reduce_(0, inv_l, v_l, "amin").cpu().numpy()
    ref_max = torch.full((g,), -1.0, dtype=torch.float64, device=dev).scatter_reduce_(0, inv_l, v_l, "amax").cpu().numpy()
    light = np.ones(g, np.bool_)
    light[heavy] = False
    assert (np.abs(sums - ref_sum) <= 2 * REL * ref_sum)[light].all()  # scatter_add's order differs from ours: both within the bound
    assert (np.abs(avgs - ref_sum / counts) <= 2 * REL * ref_sum / counts)[light].all()
    assert np.array_equal(mins[light], ref_min[light]) and np.array_equal(maxs[light], ref_max[light])
    del inv_l, v_l
    del inv, uk, cnt
    print("c4: sums, minima, maxima checked", flush=True)
    # the whole = the FINAL merge of PARTIAL aggregates of ragged parts (PARTIAL splits AVG into sum and count)
    cuts = [0, 99_999_999, 310_000_001, n]
    parts = []
    for lo, hi in zip(cuts, cuts[1:]):
        parts += list(plan(lo, hi, qe.AGG_MODE_PARTIAL).execute())
    merged = by_key(list(qe.final_plan(qe.ScanExec(ctx, parts), 1, aggs(), group_capacity_hint=1_100_000).execute())[0].columns())
    assert np.array_equal(merged[0], keys) and np.array_equal(merged[5], counts)
    assert np.array_equal(merged[2], mins) and np.array_equal(merged[3], maxs)
    assert (np.abs(merged[1] - sums) <= 2 * REL * sums).all() and (np.abs(merged[4] - avgs) <= 2 * REL * avgs).all()
    print("c4: merge of parts checked", flush=True)
    # the oracle on a prefix (its single-pass aggregate): 2 M rows through the same high-cardinality path
    m = 2_000_000
    kh, vh = k[:m].cpu().numpy(), v[:m].cpu().numpy()
    Co = orc.ColumnExpression
    want = list(orc.AggregateExec([orc.ScanExec([orc.Chunk([orc.Array(orc.INT64, kh), orc.Array(orc.FLOAT64, vh)])])], [Co(0)],
                                  [orc.SumExpression(Co(1)), orc.MinExpression(Co(1)), orc.MaxExpression(Co(1)), orc.AvgExpression(Co(1)),
                                   orc.CountExpression(None)]).execute())[0].arrays
    wo = np.argsort(want[0].values)
    b = qe.Batch.from_torch(ctx, [k[:m], v[:m]])
    got = by_key(list(qe.AggregateExec([qe.ScanExec(ctx, [b])], [C(0)], aggs(), group_capacity_hint=1_100_000).execute())[0].columns())
    # oracle columns: key, SUM, MIN, MAX, AVG, COUNT - the same order as ours
    assert np.array_equal(got[0], want[0].values[wo]) and np.array_equal(got[5], want[5].values[wo])
    assert np.array_equal(got[2], want[2].values[wo]) and np.array_equal(got[3], want[3].values[wo])
    osum = want[1].values[wo]
    assert (np.abs(got[1] - osum) <= REL * osum).all()
    assert (np.abs(got[4] - want[4].values[wo]) <= REL * osum / got[5]).all()
