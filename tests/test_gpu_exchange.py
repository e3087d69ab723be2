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


# ---------------------------------------------------------------------------------------------------------------
# The layouts of more than one rank, on one GPU: a loopback communicator (qe_comm_init_loopback) lays every exchange
# out as rank r of W ranks does - the owner hash and scatter kernel run over W owners, the all-gather puts the table at
# r * padded_rows - and moves the blocks by device copies.  Every simulated rank runs PARTIAL over its row range; what a
# rank would receive is put together from the simulated ranks' outputs on the host and merged by a FINAL aggregate.
# ---------------------------------------------------------------------------------------------------------------
def _mix64(x):
    x = x.astype(np.uint64)
    with np.errstate(over="ignore"):
        x ^= x >> np.uint64(33)
        x *= np.uint64(0xff51afd7ed558ccd)
        x ^= x >> np.uint64(33)
        x *= np.uint64(0xc4ceb9fe1a85ec53)
        x ^= x >> np.uint64(33)
    return x


def _owner(keys, world):
    """Replica of the library's owner hash for one Int64 key column: mix64(key) >> 17 modulo the ranks."""
    return ((_mix64(np.asarray(keys).view(np.uint64)) >> np.uint64(17)) % np.uint64(world)).astype(np.int64)


def _host_columns(batch):
    return [(np.asarray(c.values), None if c.validity is None else np.asarray(c.validity)) for c in batch.columns()]


def _partials(qe, ctx, k, a, b, c, world):
    n = len(k)
    out = []
    for r in range(world):
        lo, hi = r * n // world, (r + 1) * n // world
        batch = qe.Batch.from_numpy(ctx, [k[lo:hi], a[lo:hi], b[lo:hi], c[lo:hi]])
        out.append(list(_local_plan(qe, qe.ScanExec(ctx, [batch]), qe.AGG_MODE_PARTIAL).execute())[0])
    return out


def _final_of(qe, ctx, cols, hint):
    """FINAL aggregate over host columns [(values, validity)] of exchanged partial tables (unfilled rows: COUNT(*) = 0)."""
    batch = qe.Batch.from_numpy(ctx, [v if m is None else (v, m) for v, m in cols])
    first, _ = qe.partial_layout(1, _aggs(qe))
    real = qe.SelectionExec([qe.ScanExec(ctx, [batch])], qe.GtExpression(qe.ColumnExpression(first[1]), qe.LiteralInt64Expression(0)))
    return list(qe.final_plan(real, 1, _aggs(qe), group_capacity_hint=hint).execute())[0]


@pytest.mark.parametrize("world", [2, 8])
def test_loopback_all_gather_layout_of_every_rank(qe, ctx, world):
    k, a, b, c = _table(240_000, 300, seed=13)
    partials = _partials(qe, ctx, k, a, b, c, world)
    padded = (max(p.num_rows for p in partials) + 31) // 32 * 32 + 32
    union = None
    for r, partial in enumerate(partials):
        comm = qe.Comm(ctx, r, world, loopback=True)
        assert comm.rccl_version == 0
        for _ in range(2):  # pool buffers are reused between steps
            got = comm.all_gather(partial, padded)
        comm.close()
        assert got.num_rows == world * padded and got.num_columns == partial.num_columns
        mine, cols = _host_columns(partial), _host_columns(got)
        rows = partial.num_rows
        assert any(m is not None for _, m in mine)  # MIN / MAX of a Float64 column can be null: their bits travel too
        for (v, m), (pv, pm) in zip(cols, mine):
            # this rank's table at rank * padded_rows, bit for bit; zero everywhere else
            block = v[r * padded:r * padded + rows]
            assert np.array_equal(block.view(np.uint64) if block.dtype == np.float64 else block, pv.view(np.uint64) if pv.dtype == np.float64 else pv)
            outside = np.ones(len(v), bool)
            outside[r * padded:r * padded + rows] = False
            assert not v[outside].view(np.uint64 if v.dtype.itemsize == 8 else np.uint32).any()
            assert (m is None) == (pm is None)
            if m is not None:
                assert np.array_equal(m[r * padded:r * padded + rows], pm) and not m[outside].any()
        if union is None:
            union = [(v.copy(), None if m is None else m.copy()) for v, m in cols]
        else:
            for (uv, um), (v, m) in zip(union, cols):
                w = np.uint64 if uv.dtype.itemsize == 8 else np.uint32
                uv.view(w)[:] |= v.view(w)  # blocks of different ranks never overlap: OR puts them side by side
                if um is not None:
                    um |= m
    # what a real all-gather leaves on every rank -> FINAL = the aggregate of the whole table
    _check(qe, _final_of(qe, ctx, union, 4096), k, a, b, c)


@pytest.mark.parametrize("world,n,n_keys", [(2, 300_000, 3_000), (8, 1_200_000, 90_000)])
def test_loopback_exchange_by_key_layout_of_every_rank(qe, ctx, world, n, n_keys):
    k, a, b, c = _table(n, n_keys, seed=17)
    partials = _partials(qe, ctx, k, a, b, c, world)
    bound = max(p.num_rows for p in partials)
    first, _ = qe.partial_layout(1, _aggs(qe))
    count_col = first[1]
    outs = []
    for r, partial in enumerate(partials):
        comm = qe.Comm(ctx, r, world, loopback=True)
        for _ in range(2):
            got = comm.exchange_by_key(partial, 1, bound)
        ctx.synchronize()  # would report a peer block that overflowed
        comm.close()
        cols = _host_columns(got)
        cap = got.num_rows // world
        assert cap * world == got.num_rows and cap % 32 == 0
        keys, counts = cols[0][0], cols[count_col][0]
        real = counts > 0
        # every row in the block of the rank that owns its key; the blocks' rows sum to the table
        assert int(real.sum()) == partial.num_rows
        block_of_row = np.arange(got.num_rows) // cap
        assert np.array_equal(_owner(keys[real], world), block_of_row[real])
        # rows are filled from the front of their block
        for p in range(world):
            filled = int(real[p * cap:(p + 1) * cap].sum())
            assert real[p * cap:p * cap + filled].all()
        # every column's values and validity bits landed with their rows; unfilled rows are zero
        mine = _host_columns(partial)
        order_in = np.argsort(mine[0][0], kind="stable")
        order_out = np.argsort(keys[real], kind="stable")
        for (v, m), (pv, pm) in zip(cols, mine):
            w = np.uint64 if v.dtype.itemsize == 8 else np.uint32
            assert np.array_equal(v[real][order_out].view(w), pv[order_in].view(w))
            assert not v[~real].view(w).any()
            assert (m is None) == (pm is None)
            if m is not None:
                assert np.array_equal(m[real][order_out], pm[order_in]) and not m[~real].any()
        outs.append((cols, cap))
    # rank q receives block q of every rank's send buffer; its FINAL aggregate gives its share of the groups
    shares = []
    for q in range(world):
        received = []
        for ci in range(len(outs[0][0])):
            vals = np.concatenate([cols[ci][0][q * cap:(q + 1) * cap] for cols, cap in outs])
            masks = None if outs[0][0][ci][1] is None else np.concatenate([cols[ci][1][q * cap:(q + 1) * cap] for cols, cap in outs])
            received.append((vals, masks))
        if not (received[count_col][0] > 0).any():
            continue
        share = _final_of(qe, ctx, received, max(64, 2 * n_keys // world)).columns()
        assert (_owner(np.asarray(share[0].values), world) == q).all()
        shares.append(share)
    # the result is the concatenation over the ranks: no key on two ranks, every group there
    all_keys = np.concatenate([np.asarray(s[0].values) for s in shares])
    assert len(np.unique(all_keys)) == len(all_keys)

    class _Cat:
        def __init__(self, cols):
            self._cols = cols

        def columns(self):
            return self._cols
    cat = []
    for ci in range(len(shares[0])):
        col = qe.Column(shares[0][ci].type, np.concatenate([np.asarray(s[ci].values) for s in shares]))
        cat.append(col)
    _check(qe, _Cat(cat), k, a, b, c)


def test_loopback_exchange_by_key_reports_a_block_that_overflows(qe, ctx):
    """Keys that all belong to one rank: its block (an even share plus a quarter) cannot hold them, which the next
    synchronise reports (AggregateCapacity) - nothing in the exchange itself waits for the device."""
    r = np.random.default_rng(23)
    cand = np.unique(r.integers(0, 1 << 40, 200_000).astype(np.int64))
    keys = cand[_owner(cand, 2) == 0][:40_000]
    assert len(keys) == 40_000
    n = len(keys) * 3
    k = np.tile(keys, 3)
    a, b, c = r.uniform(-1, 1, n), r.uniform(-1, 1, n), r.uniform(0.6, 1, n)
    batch = qe.Batch.from_numpy(ctx, [k, a, b, c])
    partial = list(_local_plan(qe, qe.ScanExec(ctx, [batch]), qe.AGG_MODE_PARTIAL).execute())[0]
    assert partial.num_rows == 40_000
    comm = qe.Comm(ctx, 1, 2, loopback=True)
    try:
        comm.exchange_by_key(partial, 1, partial.num_rows)  # returns at once
        with pytest.raises(qe.QeError) as e:
            ctx.synchronize()
        assert e.value.name == "AggregateCapacity"
        ctx.synchronize()  # reported once
        # with room for the skew it goes through, and every row is in rank 0's block
        out = comm.exchange_by_key(partial, 1, 4 * partial.num_rows)
        ctx.synchronize()
        counts = np.asarray(out.columns()[2].values)
        cap = out.num_rows // 2
        assert int((counts[:cap] > 0).sum()) == partial.num_rows and not (counts[cap:] > 0).any()
    finally:
        comm.close()


def test_loopback_communicator_checks_its_arguments(qe, ctx):
    for rank, world in ((2, 2), (-1, 2), (0, 0), (0, 65)):
        with pytest.raises(qe.QeError) as e:
            qe.Comm(ctx, rank, world, loopback=True)
        assert e.value.name == "InvalidArgument"
    assert qe.Comm.library_rccl_version() > 0
