"""QE_FLAG_DETERMINISTIC_SUM: Float64 SUM / AVG as exact integer sums of fixed-point images.  Two runs give the same
bits (the default path folds with atomic adds in a changing order), the bits do not depend on the kernels' launch
geometry or on which kernel variant ran, and the values stay within the tolerance of the default path (1e-12 of the
group's sum of magnitudes) of the oracle - in fact within a few units in the last place of the exactly rounded sum
(math.fsum).  The reference folds row by row in input order (physical_plan/mod.rs:340-372): one fixed order, its bits
are not reproduced by any parallel fold; what is promised here is run-to-run identity."""
import math

import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

REL = 1e-12


@pytest.fixture()
def det_ctx(qe):
    c = qe.Context(0)
    c.set_flags(qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP | qe.FLAG_DETERMINISTIC_SUM | qe.FLAG_JIT_SYNC)
    yield c
    c.close()


def _plan(m, scan, mode=None):
    C = m.ColumnExpression
    sel = m.SelectionExec([scan], m.GtExpression(C(3), m.LiteralFloatExpression(0.5)))
    proj = m.ProjectionExec([sel], [C(0), m.AddExpression(C(1), m.MulExpression(C(2), C(3)))])
    aggs = [m.SumExpression(C(1)), m.CountExpression(None), m.AvgExpression(C(1)), m.MinExpression(C(1))]
    if mode is None:
        return m.AggregateExec([proj], [C(0)], aggs)
    return m.AggregateExec([proj], [C(0)], aggs, mode=mode)


def _table(n, n_keys, seed, spread=1.0):
    r = np.random.default_rng(seed)
    k = r.integers(0, n_keys, n).astype(np.int64) * 5 - 7
    a = r.uniform(-1, 1, n) * np.exp(r.uniform(0, spread, n))
    return k, a, r.uniform(-1, 1, n), r.uniform(0, 1, n)


def _sorted_columns(batch):
    cols = batch.columns()
    order = np.argsort(np.asarray(cols[0].values))
    return [np.asarray(c.values)[order] for c in cols]


def _exact_sums(k, a, b, c):
    keep = c > 0.5
    v = a + b * c
    out = {}
    for key, x in zip(k[keep].tolist(), v[keep].tolist()):
        out.setdefault(key, []).append(x)
    return {key: (math.fsum(xs), math.fsum(abs(x) for x in xs), len(xs)) for key, xs in out.items()}


@pytest.mark.parametrize("n,n_keys,spread", [(400_000, 7, 1.0), (300_000, 900, 30.0), (1_000_000, 120_000, 5.0)])
def test_two_runs_same_bits_and_close_to_exact(qe, det_ctx, n, n_keys, spread):
    k, a, b, c = _table(n, n_keys, seed=11, spread=spread)
    runs = []
    for _ in range(2):
        batch = qe.Batch.from_numpy(det_ctx, [k, a, b, c])
        runs.append(_sorted_columns(list(_plan(qe, qe.ScanExec(det_ctx, [batch])).execute())[0]))
    for x, y in zip(runs[0], runs[1]):
        assert np.array_equal(x.view(np.uint64) if x.dtype == np.float64 else x, y.view(np.uint64) if y.dtype == np.float64 else y)
    exact = _exact_sums(k, a, b, c)
    keys, sums, counts, avgs = runs[0][0], runs[0][1], runs[0][2], runs[0][3]
    assert sorted(exact) == keys.tolist()
    for key, s, cnt, avg in zip(keys.tolist(), sums.tolist(), counts.tolist(), avgs.tolist()):
        e, mag, m = exact[key]
        assert cnt == m
        # the fixed-point image keeps 96 - 8 bits below the largest value of the batch: exact to the last place or two
        # unless the group's values are tiny next to the batch's largest
        assert abs(s - e) <= max(4 * np.spacing(abs(e)), REL * mag * 1e-3), (key, s, e)
        assert abs(avg - e / m) <= max(8 * np.spacing(abs(e / m)), REL * mag / m)


def test_against_the_oracle(qe, det_ctx):
    k, a, b, c = _table(200_000, 300, seed=5)
    batch = qe.Batch.from_numpy(det_ctx, [k, a, b, c])
    got = _sorted_columns(list(_plan(qe, qe.ScanExec(det_ctx, [batch])).execute())[0])
    ref = list(_plan(orc, orc.ScanExec([orc.Chunk([k, a, b, c])])).execute())[0].arrays
    order = np.argsort(ref[0].values)
    want = [np.asarray(x.values)[order] for x in ref]
    keep = c > 0.5
    uk, inv = np.unique(k[keep], return_inverse=True)
    mag = np.bincount(inv, weights=np.abs((a + b * c)[keep]))
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2])
    assert np.array_equal(got[4], want[4])  # MIN: bit exact
    assert (np.abs(got[1] - want[1]) <= REL * mag).all()
    assert (np.abs(got[3] - want[3]) <= REL * mag / want[2]).all()


def test_same_bits_from_the_general_kernels(qe, det_ctx):
    """The shape-specialised kernel and the interpreter kernel fold the rows in different groupings; the integer limbs
    do not care."""
    k, a, b, c = _table(300_000, 50, seed=8, spread=12.0)
    got = []
    for extra in (0, qe.FLAG_NO_SPECIALIZED_KERNELS):
        det_ctx.set_flags(qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP | qe.FLAG_DETERMINISTIC_SUM | qe.FLAG_JIT_SYNC | extra)
        batch = qe.Batch.from_numpy(det_ctx, [k, a, b, c])
        got.append(_sorted_columns(list(_plan(qe, qe.ScanExec(det_ctx, [batch])).execute())[0]))
    assert np.array_equal(got[0][1].view(np.uint64), got[1][1].view(np.uint64))
    assert np.array_equal(got[0][3].view(np.uint64), got[1][3].view(np.uint64))


def test_batches_with_growing_values_move_the_scale(qe, det_ctx):
    """Later batches hold larger values: the sums already in the table are moved to the larger scale.  Same bits on a
    second run, and the result close to the exact sum of everything."""
    r = np.random.default_rng(2)
    parts = []
    for i, scale in enumerate([1e-3, 1.0, 1e6, 1e2, 1e15]):
        n = 50_000 + 1000 * i
        parts.append((r.integers(0, 40, n).astype(np.int64), r.uniform(-1, 1, n) * scale))
    C = qe.ColumnExpression

    def run():
        batches = [qe.Batch.from_numpy(det_ctx, [k, v]) for k, v in parts]
        plan = qe.AggregateExec([qe.ScanExec(det_ctx, batches)], [C(0)], [qe.SumExpression(C(1)), qe.AvgExpression(C(1))])
        return _sorted_columns(list(plan.execute())[0])

    one, two = run(), run()
    assert np.array_equal(one[1].view(np.uint64), two[1].view(np.uint64))
    assert np.array_equal(one[2].view(np.uint64), two[2].view(np.uint64))
    k = np.concatenate([p[0] for p in parts])
    v = np.concatenate([p[1] for p in parts])
    for key, s in zip(one[0].tolist(), one[1].tolist()):
        xs = v[k == key].tolist()
        e, mag = math.fsum(xs), math.fsum(abs(x) for x in xs)
        assert abs(s - e) <= REL * mag * 1e-3


def test_infinities_nan_nulls_and_empty_groups(qe, det_ctx):
    n = 10_000
    r = np.random.default_rng(4)
    k = (np.arange(n) % 6).astype(np.int64)
    v = r.uniform(-5, 5, n)
    v[k == 1] = np.where(np.arange(n)[k == 1] % 50 == 1, np.inf, v[k == 1])       # +inf among finite values
    v[6 * 10 + 2] = -np.inf                                                          # one -inf
    v[6 * 20 + 3] = np.inf
    v[6 * 21 + 3] = -np.inf                                                          # both: NaN
    v[6 * 30 + 4] = np.nan
    valid = np.ones(n, dtype=bool)
    valid[k == 5] = False                                                            # a group of nulls only
    valid[::7] = False
    C = qe.ColumnExpression
    batch = qe.Batch.from_numpy(det_ctx, [k, (v, valid)])
    plan = qe.AggregateExec([qe.ScanExec(det_ctx, [batch])], [C(0)], [qe.SumExpression(C(1)), qe.AvgExpression(C(1))])
    out = list(plan.execute())[0].columns()
    order = np.argsort(np.asarray(out[0].values))
    sums = np.asarray(out[1].values)[order]
    sum_valid = np.ones(6, bool) if out[1].validity is None else np.asarray(out[1].validity)[order]
    ref = list(orc.AggregateExec([orc.ScanExec([orc.Chunk([k, (v, valid)])])], [orc.ColumnExpression(0)],
                                 [orc.SumExpression(orc.ColumnExpression(1)), orc.AvgExpression(orc.ColumnExpression(1))]).execute())[0].arrays
    rorder = np.argsort(ref[0].values)
    want = np.asarray(ref[1].values)[rorder]
    want_valid = np.ones(6, bool) if ref[1].validity is None else np.asarray(ref[1].validity)[rorder]
    assert np.array_equal(sum_valid, want_valid)
    assert sums[1] == np.inf and sums[2] == -np.inf and np.isnan(sums[3]) and np.isnan(sums[4])
    for g in range(6):
        if not want_valid[g]:
            continue
        if np.isfinite(want[g]):
            assert abs(sums[g] - want[g]) <= REL * np.abs(v[(k == g) & valid]).sum()
        else:
            assert (np.isnan(want[g]) and np.isnan(sums[g])) or want[g] == sums[g]


def test_partial_then_final(qe, det_ctx):
    k, a, b, c = _table(500_000, 2_000, seed=9, spread=3.0)
    cut = 210_000

    def run():
        partials = []
        for lo, hi in ((0, cut), (cut, len(k))):
            batch = qe.Batch.from_numpy(det_ctx, [k[lo:hi], a[lo:hi], b[lo:hi], c[lo:hi]])
            partials.append(list(_plan(qe, qe.ScanExec(det_ctx, [batch]), qe.AGG_MODE_PARTIAL).execute())[0])
        aggs = [qe.SumExpression(qe.ColumnExpression(1)), qe.CountExpression(None), qe.AvgExpression(qe.ColumnExpression(1)),
                qe.MinExpression(qe.ColumnExpression(1))]
        merged = list(qe.final_plan(qe.ScanExec(det_ctx, partials), 1, aggs).execute())[0]
        return _sorted_columns(merged)

    one, two = run(), run()
    assert np.array_equal(one[1].view(np.uint64), two[1].view(np.uint64))
    assert np.array_equal(one[3].view(np.uint64), two[3].view(np.uint64))
    exact = _exact_sums(k, a, b, c)
    for key, s, avg in zip(one[0].tolist(), one[1].tolist(), one[3].tolist()):
        e, mag, m = exact[key]
        assert abs(s - e) <= REL * mag and abs(avg - e / m) <= REL * mag / m


def test_a_value_beyond_the_scale_the_first_rows_suggest(qe, det_ctx):
    """The scale comes from the first 2^22 rows of the first batch (with 8 bits to spare); here the rows after them are
    2^70 times larger.  The kernels notice, the aggregation starts over with a scale pass per batch, and the result is
    the exact sum all the same - with the same bits on a second run."""
    n = 4_600_000
    r = np.random.default_rng(31)
    k = (np.arange(n) % 7).astype(np.int64)
    v = r.uniform(-1e-3, 1e-3, n)
    v[4_400_000:] = r.uniform(-1, 1, n - 4_400_000) * 1e18
    C = qe.ColumnExpression

    def run():
        batch = qe.Batch.from_numpy(det_ctx, [k, v])
        plan = qe.AggregateExec([qe.ScanExec(det_ctx, [batch])], [C(0)], [qe.SumExpression(C(1)), qe.CountExpression(None)])
        return _sorted_columns(list(plan.execute())[0])
    one, two = run(), run()
    assert np.array_equal(one[1].view(np.uint64), two[1].view(np.uint64))
    for key, s, cnt in zip(one[0].tolist(), one[1].tolist(), one[2].tolist()):
        xs = v[k == key]
        assert cnt == len(xs)
        e, mag = math.fsum(xs.tolist()), float(np.abs(xs).sum())
        assert abs(s - e) <= REL * mag * 1e-3, (key, s, e)


def test_streaming_protocol_starts_over_too(qe, det_ctx):
    """qe_agg_update / qe_agg_finalize with batches whose values grow past the first batch's scale: finalize folds the
    batches in again with exact scales."""
    r = np.random.default_rng(37)
    parts = [(r.integers(0, 20, 30_000).astype(np.int64), r.uniform(-1, 1, 30_000) * s) for s in (1.0, 1e30, 1e-5)]
    C = qe.ColumnExpression
    st = qe.AggregationState(det_ctx, [C(0)], [qe.SumExpression(C(1)), qe.AvgExpression(C(1))])
    for kk, vv in parts:
        st.accumulate(qe.Batch.from_numpy(det_ctx, [kk, vv]))
    out = _sorted_columns(st.final_value())
    st.close()
    k = np.concatenate([p[0] for p in parts])
    v = np.concatenate([p[1] for p in parts])
    for key, s in zip(out[0].tolist(), out[1].tolist()):
        xs = v[k == key].tolist()
        e, mag = math.fsum(xs), math.fsum(abs(x) for x in xs)
        assert abs(s - e) <= REL * mag * 1e-3
