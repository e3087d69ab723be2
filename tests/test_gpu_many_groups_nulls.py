"""The many-groups aggregate over accumulator inputs that hold nulls: a null is logged as a value no input can hold (a
NaN payload for Float64, INT64_MIN for a sign-extended Int32) and left out of the accumulators by the partition pass,
while the row still counts for its group; whether a group saw a value at all is told by a COUNT accumulator.  Against
the oracle's aggregate of the same batches (null propagation of the accumulators: physical_expressions.rs:364-504 -
Max / Min skip nulls, an aggregate over no value is null), and at size against torch.
"""
import numpy as np
import pytest

from oracle import oracle as orc
from test_gpu_parity import C, rng, run_agg

pytestmark = pytest.mark.gpu

HINT = 200_000


def launches(prof, prefix):
    return sum(v["launches"] for k, v in prof.items() if k.startswith(prefix))


@pytest.fixture()
def jit(qe):
    c = qe.Context(0)
    c.set_flags(qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP | qe.FLAG_JIT_SYNC)
    c.profile(True)
    c.profile_read()
    yield c
    c.profile(False)
    c.close()


def columns(n, seed, n_keys=150_000, value="f64", null_share=0.15, all_null_keys=2000):
    """0: Int64 key, 1: the value column with nulls (every row of the first `all_null_keys` keys is null)."""
    r = rng(seed)
    k = r.integers(0, n_keys, n).astype(np.int64)
    valid = (r.random(n) > null_share) & (k >= all_null_keys)
    if value == "f64":
        v = r.uniform(-100, 100, n)
        return [orc.Array(orc.INT64, k), orc.Array(orc.FLOAT64, v, valid)]
    if value == "i32":
        v = r.integers(-2 ** 31, 2 ** 31, n, dtype=np.int64).astype(np.int32)
        v[r.random(n) < 0.01] = -2 ** 31
        return [orc.Array(orc.INT64, k), orc.Array(orc.INT32, v, valid)]
    v = r.integers(-2 ** 62, 2 ** 62, n, dtype=np.int64)
    return [orc.Array(orc.INT64, k), orc.Array(orc.INT64, v, valid)]


ALL = [("sum", C(1)), ("min", C(1)), ("max", C(1)), ("avg", C(1)), ("count", C(1)), ("count", None)]


@pytest.mark.parametrize("value", ["f64", "i32", "i64"])
def test_nullable_input_against_the_oracle(qe, jit, value):
    """Three batches (a small one in between), every aggregate over the nullable column.  Float64 and Int32 inputs go
    through the log and the partition passes; an Int64 input has no bit pattern to spare for a null and stays on the
    path of the global table."""
    batches = [columns(n, seed=200 + i, value=value) for i, n in enumerate([600_000, 30_001, 500_000])]
    run_agg(qe, jit, batches, [C(0)], ALL, hint=HINT, what=f"nullable {value} input, many groups")
    prof = jit.profile_read()
    if value == "i64":
        assert launches(prof, "qe_aggregate_partitioned_kernel") == 0, prof
    else:
        assert launches(prof, "qe_aggregate_partitioned_kernel") >= 1, prof


def test_nans_among_the_values_one_of_them_the_null_mark_itself(qe, jit):
    """Valid Float64 values that are NaN - some with the very bits that mark a null in the log: NaNs like any other (min
    and max skip them, they count as values, their groups' sums are NaN)."""
    from test_gpu_parity import mk
    from util import gpu_batch
    cols = columns(400_000, seed=77)
    mark = np.array([0x7FF8000000006E75], dtype=np.uint64).view(np.float64)[0]
    v, valid = cols[1].values.copy(), cols[1].validity.copy()
    idx = np.flatnonzero(valid)
    v[idx[::997]] = mark
    v[idx[5::1409]] = np.nan
    cols[1] = orc.Array(orc.FLOAT64, v, valid)
    run_agg(qe, jit, [cols], [C(0)], [("min", C(1)), ("max", C(1)), ("count", C(1)), ("count", None)], hint=HINT, what="NaNs, one kind with the null mark's payload")
    got = list(qe.AggregateExec([qe.ScanExec(jit, [gpu_batch(qe, jit, cols)])], [mk(qe, C(0))], [mk(qe, ("sum", C(1)))], group_capacity_hint=HINT).execute())[0].columns()
    order = np.argsort(np.asarray(got[0].values))
    keys, sums = np.asarray(got[0].values)[order], np.asarray(got[1].values)[order]
    sum_valid = np.ones(len(keys), bool) if got[1].validity is None else np.asarray(got[1].validity, dtype=bool)[order]
    k = cols[0].values
    uk, inv = np.unique(k, return_inverse=True)
    assert np.array_equal(keys, uk)
    has_nan = np.bincount(inv, weights=(valid & np.isnan(v)).astype(np.float64), minlength=len(uk)) > 0
    has_val = np.bincount(inv, weights=valid.astype(np.float64), minlength=len(uk)) > 0
    assert np.array_equal(sum_valid, has_val)
    assert np.array_equal(np.isnan(sums) & sum_valid, has_nan)
    clean = valid & ~np.isnan(v)
    ref = np.bincount(inv, weights=np.where(clean, v, 0.0), minlength=len(uk))
    mag = np.bincount(inv, weights=np.where(clean, np.abs(v), 0.0), minlength=len(uk))
    ok = has_val & ~has_nan
    assert (np.abs(sums[ok] - ref[ok]) <= 1e-12 * mag[ok] + 1e-300).all()


def test_two_nullable_value_columns_and_a_filter(qe, jit):
    """Accumulators over two nullable columns (a log of several value columns, each with its own mark), behind a filter
    on one of them (a null predicate drops the row)."""
    r = rng(5)
    n = 700_000
    k = r.integers(0, 120_000, n).astype(np.int64)
    a = orc.Array(orc.FLOAT64, r.uniform(-1, 1, n), r.random(n) > 0.2)
    b = orc.Array(orc.INT32, r.integers(-1000, 1000, n).astype(np.int32), r.random(n) > 0.3)
    aggs = [("sum", C(1)), ("max", C(1)), ("sum", C(2)), ("min", C(2)), ("avg", C(2)), ("count", C(1)), ("count", None)]
    run_agg(qe, jit, [[orc.Array(orc.INT64, k), a, b]], [C(0)], aggs, hint=HINT, pred=("gt", C(1), ("f64", -0.9)), what="two nullable value columns")


def test_nullable_input_at_size_against_torch(qe):
    """64 M rows, a million uniform keys, a fifth of the values null (whole bytes of the bitmap and single bits): the
    first pass that partitions directly takes the plan; sums, counts, minima and maxima of the valid values against
    torch."""
    import torch
    torch.cuda.set_device(0)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx = qe.Context(0, stream=stream.cuda_stream)
    ctx.set_flags(qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP | qe.FLAG_JIT_SYNC)
    try:
        n, n_keys = 64_000_000, 1_000_000
        dev = torch.device("cuda", 0)
        k = torch.empty(n, dtype=torch.int64, device=dev)
        v = torch.empty(n, dtype=torch.float64, device=dev)
        ctx.generate(1, 299, 0, n, 0, n_keys, k.data_ptr())
        ctx.generate(0, 244, 0, n, -1.0, 1.0, v.data_ptr())
        ctx.synchronize()
        g = torch.Generator(device=dev)
        g.manual_seed(3)
        bits = torch.randint(0, 256, (n // 8 + 64,), dtype=torch.uint8, device=dev, generator=g)
        bits |= torch.randint(0, 256, (n // 8 + 64,), dtype=torch.uint8, device=dev, generator=g)  # a bit is set three times in four
        bits[torch.rand(n // 8 + 64, device=dev, generator=g) < 0.05] = 0                           # and whole bytes are null
        valid = ((bits[:n // 8, None] >> torch.arange(8, device=dev, dtype=torch.uint8)[None, :]) & 1).reshape(-1).bool()
        cols = [{"length": n, "type": qe.TYPE_INT64, "values": k.data_ptr()},
                {"length": n, "type": qe.TYPE_FLOAT64, "values": v.data_ptr(), "validity": bits.data_ptr(), "null_count": -1}]
        batch = qe.Batch.from_device_columns(ctx, cols, keepalive=[k, v, bits])
        Cx = qe.ColumnExpression
        aggs = [qe.SumExpression(Cx(1)), qe.CountExpression(Cx(1)), qe.CountExpression(None), qe.MinExpression(Cx(1)), qe.MaxExpression(Cx(1))]
        ctx.profile(True)
        out = list(qe.AggregateExec([qe.ScanExec(ctx, [batch])], [Cx(0)], aggs, group_capacity_hint=1_100_000).execute())[0].columns()
        names = set(ctx.profile_read().keys())
        ctx.profile(False)
        assert any(nm.startswith("qe_aggregate_partitioned") for nm in names), names
        order = np.argsort(np.asarray(out[0].values))
        got = [np.asarray(c.values)[order] for c in out]
        uk, inv, cnt = torch.unique(k, return_inverse=True, return_counts=True)
        ng = uk.shape[0]
        assert np.array_equal(got[0], uk.cpu().numpy()) and np.array_equal(got[3], cnt.cpu().numpy())
        vz = torch.where(valid, v, torch.zeros_like(v))
        ref_sum = torch.zeros(ng, dtype=torch.float64, device=dev).scatter_add_(0, inv, vz).cpu().numpy()
        ref_mag = torch.zeros(ng, dtype=torch.float64, device=dev).scatter_add_(0, inv, vz.abs()).cpu().numpy()
        ref_cnt = torch.zeros(ng, dtype=torch.int64, device=dev).scatter_add_(0, inv, valid.to(torch.int64)).cpu().numpy()
        assert np.array_equal(got[2], ref_cnt)
        some = ref_cnt > 0
        sum_valid = np.asarray(out[1].validity)[order] if out[1].validity is not None else np.ones(ng, bool)
        assert np.array_equal(np.asarray(sum_valid, dtype=bool), some)
        assert (np.abs(got[1][some] - ref_sum[some]) <= 2e-12 * ref_mag[some]).all()
        ref_min = torch.full((ng,), 9.0, dtype=torch.float64, device=dev).scatter_reduce_(0, inv, torch.where(valid, v, torch.full_like(v, 9.0)), "amin").cpu().numpy()
        ref_max = torch.full((ng,), -9.0, dtype=torch.float64, device=dev).scatter_reduce_(0, inv, torch.where(valid, v, torch.full_like(v, -9.0)), "amax").cpu().numpy()
        assert np.array_equal(got[4][some], ref_min[some]) and np.array_equal(got[5][some], ref_max[some])
    finally:
        torch.cuda.synchronize()
        ctx.close()
        torch.cuda.set_stream(torch.cuda.default_stream())


def test_partial_aggregates_of_nullable_inputs_merged_by_a_final_one(qe, jit):
    """PARTIAL aggregates of two parts (their partial sums are null for groups that saw no value) merged by a FINAL one,
    all three in the many-groups mode: the same groups, counts, minima and maxima as the single aggregate over both
    parts, sums within 1e-12 of the magnitudes."""
    from test_gpu_parity import mk
    from util import gpu_batch
    parts_in = [columns(500_000, seed=301), columns(450_000, seed=302)]
    aggs = lambda: [mk(qe, a) for a in [("sum", C(1)), ("count", C(1)), ("count", None), ("min", C(1)), ("max", C(1)), ("avg", C(1))]]
    whole = list(qe.AggregateExec([qe.ScanExec(jit, [gpu_batch(qe, jit, p) for p in parts_in])], [mk(qe, C(0))], aggs(), group_capacity_hint=HINT).execute())[0].columns()
    partials = []
    for p in parts_in:
        partials += list(qe.AggregateExec([qe.ScanExec(jit, [gpu_batch(qe, jit, p)])], [mk(qe, C(0))], aggs(), mode=qe.AGG_MODE_PARTIAL, group_capacity_hint=HINT).execute())
    merged = list(qe.final_plan(qe.ScanExec(jit, partials), 1, aggs(), group_capacity_hint=HINT).execute())[0].columns()
    assert len(merged) == len(whole)
    ow, om = np.argsort(np.asarray(whole[0].values)), np.argsort(np.asarray(merged[0].values))
    for i, (w, m) in enumerate(zip(whole, merged)):
        wv, mv = np.asarray(w.values)[ow], np.asarray(m.values)[om]
        wval = np.ones(len(wv), bool) if w.validity is None else np.asarray(w.validity, dtype=bool)[ow]
        mval = np.ones(len(mv), bool) if m.validity is None else np.asarray(m.validity, dtype=bool)[om]
        assert np.array_equal(wval, mval), i
        if i in (1, 6):  # sum, avg
            assert (np.abs(wv[wval] - mv[wval]) <= 1e-12 * 100 * 50).all(), i
        else:
            assert np.array_equal(wv[wval], mv[wval]), i
    null_keys = np.asarray(whole[0].values)[ow] < 2000  # (the keys whose every value is null)
    assert null_keys.any() and not np.asarray(whole[1].validity, dtype=bool)[ow][null_keys].any()


def test_ten_sums_over_ten_nullable_columns(qe, jit):
    """Each SUM brings a COUNT accumulator along in this mode: twenty accumulators are more than one kernel takes, so the
    aggregate list is split and the parts' outputs joined - the same columns as the oracle's."""
    r = rng(9)
    n = 150_000
    k = r.integers(0, 70_000, n).astype(np.int64)
    cols = [orc.Array(orc.INT64, k)] + [orc.Array(orc.FLOAT64, r.uniform(-1, 1, n), r.random(n) > 0.1 * (i + 1) / 2) for i in range(10)]
    run_agg(qe, jit, [cols], [C(0)], [("sum", C(1 + i)) for i in range(10)] + [("count", None)], hint=HINT, what="ten sums over ten nullable columns")
