"""PARTIAL aggregate -> exchange over RCCL (qe_exchange_all_gather / qe_exchange_by_key, a one-rank communicator: the
calls, buffers and kernels are those of any world size) -> FINAL aggregate, on device buffers end to end, against the
oracle's aggregate of the whole table.  The reference has no counterpart (single process, physical_plan/mod.rs:58);
the two-rank layouts are covered on CPU by test_distributed.py."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

REL = 1e-12


def _table(n, n_keys, seed):
    r = np.random.default_rng(seed)
    k = (np.minimum(r.geometric(0.002, n), n_keys) * 7 - 3).astype(np.int64)
    return k, r.uniform(-1, 1, n), r.uniform(-1, 1, n), r.uniform(0, 1, n)


def _aggs(m):
    C = m.ColumnExpression
    return [m.SumExpression(C(1)), m.CountExpression(None), m.MinExpression(C(1)), m.MaxExpression(C(1)), m.AvgExpression(C(1))]


def _local_plan(m, scan, mode=None):
    C = m.ColumnExpression
    sel = m.SelectionExec([scan], m.GtExpression(C(3), m.LiteralFloatExpression(0.5)))
    proj = m.ProjectionExec([sel], [C(0), m.AddExpression(C(1), m.MulExpression(C(2), C(3)))])
    if mode is None:
        return m.AggregateExec([proj], [C(0)], _aggs(m))
    return m.AggregateExec([proj], [C(0)], _aggs(m), mode=mode)


def _check(qe, got_batch, k, a, b, c):
    want = list(_local_plan(orc, orc.ScanExec([orc.Chunk([k, a, b, c])])).execute())[0].arrays
    got = got_batch.columns()
    og, ow = np.argsort(got[0].values), np.argsort(want[0].values)
    assert np.array_equal(np.asarray(got[0].values)[og], want[0].values[ow])            # keys
    assert np.array_equal(np.asarray(got[2].values)[og], want[2].values[ow])            # COUNT(*)
    assert np.array_equal(np.asarray(got[3].values)[og], want[3].values[ow])            # MIN, bit exact
    assert np.array_equal(np.asarray(got[4].values)[og], want[4].values[ow])            # MAX
    keep = c > 0.5
    v = a + b * c
    uk, inv = np.unique(k[keep], return_inverse=True)
    mag = np.bincount(inv, weights=np.abs(v[keep]))
    assert (np.abs(np.asarray(got[1].values)[og] - want[1].values[ow]) <= REL * mag).all()  # SUM
    assert (np.abs(np.asarray(got[5].values)[og] - want[5].values[ow]) <= REL * mag / want[2].values[ow]).all()  # AVG


@pytest.fixture(scope="module")
def comm(qe, ctx):
    c = qe.Comm(ctx, 0, 1, qe.Comm.unique_id())
    assert c.rccl_version > 0
    yield c
    c.close()


@pytest.mark.parametrize("n,n_keys", [(200_000, 300), (50, 5), (1_000_000, 40_000)])
def test_partial_all_gather_final(qe, ctx, comm, n, n_keys):
    k, a, b, c = _table(n, n_keys, seed=3)
    batch = qe.Batch.from_numpy(ctx, [k, a, b, c])
    partial = list(_local_plan(qe, qe.ScanExec(ctx, [batch]), qe.AGG_MODE_PARTIAL).execute())[0]
    padded = (partial.num_rows + 31) // 32 * 32 + 64
    for _ in range(2):  # pool buffers are reused between steps
        gathered = comm.all_gather(partial, padded)
    assert gathered.num_rows == padded and gathered.num_columns == partial.num_columns
    first, _ = qe.partial_layout(1, _aggs(qe))
    count_col = first[1]
    real = qe.SelectionExec([qe.ScanExec(ctx, [gathered])], qe.GtExpression(qe.ColumnExpression(count_col), qe.LiteralInt64Expression(0)))
    merged = list(qe.final_plan(real, 1, _aggs(qe)).execute())[0]
    _check(qe, merged, k, a, b, c)
    with pytest.raises(qe.QeError):
        comm.all_gather(partial, partial.num_rows - 1)  # padded_rows below the table's rows


def test_all_gather_of_a_table_that_has_the_padded_size(qe, ctx, comm):
    """padded_rows == the table's rows (what the deferred mode's one-row-per-slot block gives): the columns are sent from
    where they lie, no copy into place first.  With the deferred flag on, as bench.py uses it."""
    k, a, b, c = _table(150_000, 40, seed=6)
    batch = qe.Batch.from_numpy(ctx, [k, a, b, c])
    ctx.set_flags(qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP | qe.FLAG_DEFERRED_GROUP_COUNT)
    try:
        C = qe.ColumnExpression
        sel = qe.SelectionExec([qe.ScanExec(ctx, [batch])], qe.GtExpression(C(3), qe.LiteralFloatExpression(0.5)))
        proj = qe.ProjectionExec([sel], [C(0), qe.AddExpression(C(1), qe.MulExpression(C(2), C(3)))])
        aggs = [qe.SumExpression(C(1)), qe.CountExpression(None)]
        partial = list(qe.AggregateExec([proj], [C(0)], aggs, mode=qe.AGG_MODE_PARTIAL, group_capacity_hint=64).execute())[0]
        assert partial.num_rows > 40  # one row per table slot
        for _ in range(2):
            gathered = comm.all_gather(partial, partial.num_rows)
        real = qe.SelectionExec([qe.ScanExec(ctx, [gathered])], qe.GtExpression(C(2), qe.LiteralInt64Expression(0)))
        merged = list(qe.final_plan(real, 1, aggs, group_capacity_hint=64).execute())[0]
        ctx.synchronize()
    finally:
        ctx.set_flags(qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP)
    got = merged.columns()
    real_rows = np.asarray(got[2].values) > 0
    keep = c > 0.5
    v = a + b * c
    uk, inv = np.unique(k[keep], return_inverse=True)
    order = np.argsort(np.asarray(got[0].values)[real_rows])
    assert np.array_equal(np.asarray(got[0].values)[real_rows][order], uk)
    assert np.array_equal(np.asarray(got[2].values)[real_rows][order], np.bincount(inv))
    want = np.bincount(inv, weights=v[keep])
    assert (np.abs(np.asarray(got[1].values)[real_rows][order] - want) <= REL * np.bincount(inv, weights=np.abs(v[keep]))).all()


@pytest.mark.parametrize("n,n_keys", [(300_000, 2_000), (2_000_000, 150_000)])
def test_partial_exchange_by_key_final(qe, ctx, comm, n, n_keys):
    k, a, b, c = _table(n, n_keys, seed=4)
    batch = qe.Batch.from_numpy(ctx, [k, a, b, c])
    partial = list(_local_plan(qe, qe.ScanExec(ctx, [batch]), qe.AGG_MODE_PARTIAL).execute())[0]
    for _ in range(2):
        mine = comm.exchange_by_key(partial, 1, n_keys)
    ctx.synchronize()  # would report a peer block that overflowed
    first, _ = qe.partial_layout(1, _aggs(qe))
    real = qe.SelectionExec([qe.ScanExec(ctx, [mine])], qe.GtExpression(qe.ColumnExpression(first[1]), qe.LiteralInt64Expression(0)))
    merged = list(qe.final_plan(real, 1, _aggs(qe), group_capacity_hint=n_keys).execute())[0]
    _check(qe, merged, k, a, b, c)


def test_exchange_by_key_checks_its_bound(qe, ctx, comm):
    k, a, b, c = _table(100_000, 20_000, seed=5)
    batch = qe.Batch.from_numpy(ctx, [k, a, b, c])
    partial = list(_local_plan(qe, qe.ScanExec(ctx, [batch]), qe.AGG_MODE_PARTIAL).execute())[0]
    with pytest.raises(qe.QeError) as e:
        comm.exchange_by_key(partial, 1, partial.num_rows - 1)  # the bound must hold for every rank's table
    assert e.value.name == "InvalidArgument"
    with pytest.raises(qe.QeError):
        comm.exchange_by_key(partial, 0, partial.num_rows)
    out = comm.exchange_by_key(partial, 1, partial.num_rows)
    ctx.synchronize()  # nothing overflowed: passes
    counts = np.asarray(out.columns()[2].values)
    assert int((counts > 0).sum()) == partial.num_rows
