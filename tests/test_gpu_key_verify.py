"""Group keys that are identified by a 64-bit hash of the tuple (strings, wide tuples) are checked for equality: once
every batch is in, each row's key columns are compared with those of its group's representative row.  The reference
identifies groups by hash alone and never compares keys (physical_plan/mod.rs:310-358): two distinct keys with one
hash merge silently there.  QE_FLAG_DEBUG_WEAK_KEY_HASH takes the hash modulo 5, so that collisions happen and the
check has something to find; QE_FLAG_HASH_ONLY_GROUP_IDENTITY leaves the check out (the reference's behaviour)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ctx(qe, extra):
    c = qe.Context(0)
    c.set_flags(qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP | extra)
    return c


def _table(n, n_keys, seed):
    r = np.random.default_rng(seed)
    ids = r.integers(0, n_keys, n)
    strings = ["country-%d" % i for i in ids]
    return ids.astype(np.int64), strings, r.uniform(0, 1, n)


def _plan(qe, ctx, batches, key_cols):
    C = qe.ColumnExpression
    return qe.AggregateExec([qe.ScanExec(ctx, batches)], [C(k) for k in key_cols], [qe.CountExpression(None), qe.MaxExpression(C(2))])


@pytest.mark.parametrize("key_cols", [[1], [1, 0], [0, 1, 0]])
def test_a_collision_is_found(qe, key_cols):
    ctx = _ctx(qe, qe.FLAG_DEBUG_WEAK_KEY_HASH)
    try:
        ids, strings, v = _table(20_000, 40, seed=3)
        batches = [qe.Batch.from_numpy(ctx, [ids[:12_000], strings[:12_000], v[:12_000]]), qe.Batch.from_numpy(ctx, [ids[12_000:], strings[12_000:], v[12_000:]])]
        with pytest.raises(qe.QeError) as e:
            list(_plan(qe, ctx, batches, key_cols).execute())
        assert e.value.name == "Internal" and "share one 64-bit hash" in e.value.message
    finally:
        ctx.close()


def test_no_false_alarm_when_the_weak_hash_separates_the_keys(qe):
    """One distinct key per residue at most: the weakened hash still tells them apart, and the check agrees."""
    ctx = _ctx(qe, qe.FLAG_DEBUG_WEAK_KEY_HASH)
    try:
        n = 5_000
        strings = ["only"] * n
        ids = np.zeros(n, np.int64)
        v = np.random.default_rng(1).uniform(0, 1, n)
        out = list(_plan(qe, ctx, [qe.Batch.from_numpy(ctx, [ids, strings, v])], [1]).execute())[0].columns()
        assert out[0].values == [b"only"] and out[1].values.tolist() == [n] and out[2].values[0] == v.max()
    finally:
        ctx.close()


def test_hash_only_identity_merges_silently_like_the_reference(qe):
    ctx = _ctx(qe, qe.FLAG_DEBUG_WEAK_KEY_HASH | qe.FLAG_HASH_ONLY_GROUP_IDENTITY)
    try:
        ids, strings, v = _table(20_000, 40, seed=3)
        out = list(_plan(qe, ctx, [qe.Batch.from_numpy(ctx, [ids, strings, v])], [1]).execute())[0]
        assert out.num_rows <= 9 and int(np.asarray(out.columns()[1].values).sum()) == 20_000  # hash classes (signed remainders), not keys
    finally:
        ctx.close()


def test_the_check_passes_on_real_hashes(qe):
    """The full 64-bit hash over many keys, several batches, a predicate and null keys: no collision, no false alarm,
    and the result is the exact group-by."""
    ctx = _ctx(qe, 0)
    try:
        n = 400_000
        r = np.random.default_rng(9)
        ids = r.integers(0, 150_000, n).astype(np.int64)
        strings = [None if i % 97 == 0 else "k%07d" % i for i in ids]
        v = r.uniform(0, 1, n)
        C = qe.ColumnExpression
        cuts = [0, 130_000, 130_001, n]
        batches = [qe.Batch.from_numpy(ctx, [ids[a:b], strings[a:b], v[a:b]]) for a, b in zip(cuts, cuts[1:])]
        sel = qe.SelectionExec([qe.ScanExec(ctx, batches)], qe.GtExpression(C(2), qe.LiteralFloatExpression(0.25)))
        out = list(qe.AggregateExec([sel], [C(1), C(0)], [qe.CountExpression(None)], group_capacity_hint=200_000).execute())[0].columns()
        keep = (v > 0.25) & (ids % 97 != 0)
        uk, cnt = np.unique(ids[keep], return_counts=True)
        order = np.argsort(np.asarray(out[1].values))
        assert np.array_equal(np.asarray(out[1].values)[order], uk) and np.array_equal(np.asarray(out[2].values)[order], cnt)
        assert [out[0].values[i] for i in order[:3]] == [b"k%07d" % i for i in uk[:3]]
    finally:
        ctx.close()
