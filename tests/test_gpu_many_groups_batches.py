"""The many-groups aggregate over plans of many batches: first passes that take several batches in one launch, the
logs of several launches partitioned and aggregated together, and what keeps a batch out of either (a batch too small
for a workgroup per compute unit, a batch with nulls where the others have none, more batches than a group of
launches holds).  Every case against the oracle's aggregate of the same batches; the launches made are read from the
context's profile.  AggregateExec::execute's loop over batches, physical_plan/mod.rs:298-374.
"""
import numpy as np
import pytest

from oracle import oracle as orc
from test_gpu_parity import C, rng, run_agg

pytestmark = pytest.mark.gpu

HINT = 400_000
FOUR = [("sum", C(1)), ("min", C(1)), ("max", C(1)), ("avg", C(1))]


def key_value(n, seed, n_keys=300_000, skew=True, nulls=None):
    """nulls: None = no validity buffer, False = a validity buffer that says every row is valid, True = one with nulls."""
    r = rng(seed)
    k = ((r.zipf(1.2, n) % n_keys) if skew else r.integers(0, n_keys, n)).astype(np.int64) * 2654435761 % (2 ** 40)
    v = r.uniform(-100, 100, n)
    validity = None if nulls is None else (r.random(n) > 0.1) if nulls else np.ones(n, np.bool_)
    return [orc.Array(orc.INT64, k), orc.Array(orc.FLOAT64, v, validity)]


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


def test_six_batches_two_launches_one_partition_pass(qe, jit):
    """Six batches of one shape, ragged sizes: four go into the first launch, two into the second, and the logs of both
    are partitioned into one copy and aggregated by one partition pass."""
    sizes = [1_150_000, 1_100_001, 1_234_567, 1_099_999, 1_300_000, 1_111_111]
    batches = [key_value(n, seed=10 + i) for i, n in enumerate(sizes)]
    run_agg(qe, jit, batches, [C(0)], FOUR, hint=HINT, what="six batches")
    prof = jit.profile_read()
    assert launches(prof, "qe_aggregate_spill_static<") == 2, prof
    assert launches(prof, "qe_aggregate_partitioned_kernel") == 1 and launches(prof, "qe_spill_scatter_onepass_kernel") == 2, prof


def test_predicate_and_a_small_batch_in_between(qe, jit):
    """A filter in front, and a batch too small to lead a launch of its own shape between large ones: it rides along
    when it follows a large batch, and the results do not depend on who rode with whom."""
    sizes = [1_200_000, 40_000, 1_100_000, 3, 1_150_000]
    batches = [key_value(n, seed=30 + i) for i, n in enumerate(sizes)]
    run_agg(qe, jit, batches, [C(0)], FOUR, hint=HINT, pred=("gt", C(1), ("f64", -20.0)), what="filter, small batches in between")
    prof = jit.profile_read()
    assert launches(prof, "qe_aggregate_partitioned_kernel") >= 1, prof
    # the small batch first: it has a launch to itself (too few workgroups for one per column of the cell table)
    run_agg(qe, jit, [batches[1], batches[0], batches[2]], [C(0)], FOUR, hint=HINT, what="a small batch first")


def test_a_batch_with_nulls_among_batches_without(qe, jit):
    """A nullable value column of which only the third batch holds nulls (the others' validity buffers say "all valid" and
    are left aside): another kernel shape, so that batch cannot share the launch of the batches before it - and an
    aggregate over a nullable input needs validity tracking, which the earlier batches' table lacks: the plan starts over
    with every validity buffer taken at face value (physical_plan/mod.rs:340-372 has one accumulator object per group
    whatever the nulls)."""
    batches = [key_value(1_100_000, seed=50, nulls=False), key_value(1_100_000, seed=51, nulls=False), key_value(1_100_000, seed=52, nulls=True),
               key_value(1_100_000, seed=53, nulls=False)]
    run_agg(qe, jit, batches, [C(0)], FOUR, hint=HINT, what="a nullable batch among non-nullable ones")


def test_more_batches_than_a_group_of_launches_holds(qe, jit):
    """Nine batches each led by a row count that keeps them apart (a small batch between any two large ones cannot ride
    with a predecessor that is itself small): more launches than the cell table has column blocks for, so the partition
    passes run more than once."""
    batches = []
    for i in range(9):
        batches.append(key_value(1_100_000 + 1000 * i, seed=70 + i))
    # uniform keys this time (nearly every row is logged)
    batches = [key_value(len(b[0]), seed=90 + i, skew=False, n_keys=350_000) for i, b in enumerate(batches)]
    run_agg(qe, jit, batches, [C(0)], [("sum", C(1)), ("count", None), ("avg", C(1))], hint=HINT, what="nine batches, uniform keys")
    prof = jit.profile_read()
    assert launches(prof, "qe_aggregate_partitioned_kernel") >= 1, prof


def test_streaming_protocol_takes_a_batch_at_a_time(qe, jit):
    """The Accumulator protocol (physical_expressions.rs:351-362) sees one batch per call: a launch each, the logs still
    partitioned together when the result is asked for."""
    from test_gpu_parity import mk
    from util import gpu_batch
    sizes = [1_100_000, 1_200_000, 1_150_000]
    batches = [key_value(n, seed=110 + i) for i, n in enumerate(sizes)]
    st = qe.AggregationState(jit, [mk(qe, C(0))], [mk(qe, a) for a in FOUR], group_capacity_hint=HINT)
    for b in batches:
        st.accumulate(gpu_batch(qe, jit, b))
    got = st.final_value()
    prof = jit.profile_read()
    st.close()
    assert launches(prof, "qe_aggregate_spill_static<") == 3 and launches(prof, "qe_aggregate_partitioned_kernel") == 1, prof
    want = list(orc.AggregateExec([orc.ScanExec([orc.Chunk(b) for b in batches])], [mk(orc, C(0))], [mk(orc, a) for a in FOUR]).execute())[0]
    gc = got.columns()
    go, wo = np.argsort(np.asarray(gc[0].values)), np.argsort(want.arrays[0].values)
    assert np.array_equal(np.asarray(gc[0].values)[go], want.arrays[0].values[wo])
    for i in (2, 3):  # min, max: bit for bit
        assert np.array_equal(np.asarray(gc[i].values)[go], want.arrays[i].values[wo])
    # sum, avg: within 1e-12 of the magnitudes summed (|v| <= 100)
    keys, counts = np.unique(np.concatenate([b[0].values for b in batches]), return_counts=True)
    assert np.array_equal(keys, want.arrays[0].values[wo])
    assert (np.abs(np.asarray(gc[1].values)[go] - want.arrays[1].values[wo]) <= 1e-12 * 100 * counts).all()
    assert (np.abs(np.asarray(gc[4].values)[go] - want.arrays[4].values[wo]) <= 1e-12 * 100).all()
