"""QE_FLAG_DEFERRED_ROW_COUNT: a Selection hands its output batch back without waiting for the device to say how many
rows it kept (the reference streams batch after batch, SelectionIterator::next, physical_plan/mod.rs:205-229); the
batch takes its final shape on first use.  Same results as without the flag - against the oracle, bit for bit - for
fixed-width outputs, nullable and boolean ones, string columns (the reference's test_filter, tests/mod.rs:138-188),
selections that keep nothing or everything, several batches queued before any is looked at, and the shapes that fall
back to waiting (a string literal as an output, more outputs than one launch takes)."""
import numpy as np
import pytest

from oracle import oracle as orc
from test_gpu_parity import COL, C, make_columns, mk, rng, run_both
from util import gpu_batch, load_userdata

pytestmark = pytest.mark.gpu


@pytest.fixture()
def dctx(qe):
    c = qe.Context(0)
    c.set_flags(qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP | qe.FLAG_DEFERRED_ROW_COUNT)
    yield c
    c.close()


@pytest.mark.parametrize("n", [1, 4095, 4097, 3 * 8192 + 5, 300_000])
@pytest.mark.parametrize("sel", [0.0, 0.01, 0.5, 1.0])
def test_selection_all_columns(qe, dctx, n, sel):
    cols = make_columns(n, seed=n + int(sel * 100))
    cols[5] = orc.Array(orc.FLOAT64, rng(n).uniform(0, 1, n))
    pred = ("lt", C(5), ("f64", sel)) if sel < 1.0 else ("lteq", C(5), ("f64", 1.0))
    got, _ = run_both(qe, dctx, [cols], lambda m, scan: m.SelectionExec([scan], mk(m, pred)), what=f"deferred selection n={n} sel={sel}")
    assert got[0].num_rows == (int((cols[5].values < sel).sum()) if sel < 1.0 else n)


def test_batches_queued_before_any_is_looked_at(qe, dctx):
    parts = [make_columns(n, seed=s) for n, s in ((50_000, 1), (1, 2), (90_000, 3), (4096, 4), (70_001, 5))]

    def chain(m, scan):
        s1 = m.SelectionExec([scan], mk(m, ("gt", C(5), ("f64", 0.5))))
        return m.ProjectionExec([s1], [mk(m, ("add", C(4), ("mul", C(5), C(5)))), mk(m, C(1)), mk(m, C(6)), mk(m, C(3)), mk(m, ("gt", C(4), ("f64", 0.0)))])
    got, want = run_both(qe, dctx, parts, chain, what="deferred chain over five batches")
    assert [g.num_rows for g in got] == [len(w) for w in want]
    # read in reverse: a batch waits for its own count only
    batches = [gpu_batch(qe, dctx, a) for a in parts]
    outs = list(chain(qe, qe.ScanExec(dctx, batches)).execute())
    for g, w in reversed(list(zip(outs, want))):
        assert g.num_rows == len(w)


def test_reference_test_filter_with_strings(qe, dctx):
    """tests/mod.rs:138-188: gender == "Female" over all 13 columns, 10 of them Utf8."""
    arrays, _ = load_userdata()
    pred = ("eq", C(COL["gender"]), ("str", "Female"))
    got, _ = run_both(qe, dctx, [arrays, arrays, arrays], lambda m, scan: m.SelectionExec([scan], mk(m, pred)), what="deferred test_filter")
    out = got[0].columns()
    assert got[0].num_rows == 482 and len(out) == 13
    assert (out[2].values[0], out[3].values[0], out[2].values[1], out[3].values[1]) == (b"Amanda", b"Jordan", b"Evelyn", b"Morgan")
    run_both(qe, dctx, [arrays], lambda m, scan: m.SelectionExec([scan], mk(m, ("eq", C(COL["gender"]), ("str", "nobody")))), what="deferred, nothing kept")
    run_both(qe, dctx, [arrays], lambda m, scan: m.SelectionExec([scan], mk(m, ("neq", C(COL["gender"]), ("str", "nobody")))), what="deferred, everything kept")


def test_string_columns_at_size(qe, dctx):
    """Every kind of wave of the byte copy, more tiles than workgroups, nulls, empty strings - with the row count read on
    the device by both gather kernels and the outputs sized by their sources."""
    n = 1_300_000
    r = np.random.default_rng(29)
    words = [b"", b"a", b"Female", b"x" * 16, b"y" * 17, b"p" * 33, b"q" * 49, b"z" * 64, b"w" * 65, b"v" * 300]
    p = np.array([0.1, 0.2, 0.3, 0.1, 0.1, 0.08, 0.06, 0.056, 0.002, 0.002])
    pick = r.choice(len(words), n, p=p / p.sum())
    lens = np.array([len(w) for w in words], np.int64)[pick]
    offs = np.zeros(n + 1, np.int32)
    np.cumsum(lens, out=offs[1:])
    data = np.frombuffer(b"".join(words[i] for i in pick), np.uint8).copy()
    flag = (r.random(n) < 0.7).astype(np.int32)
    cols = [orc.Array(orc.INT32, flag), orc.Array(orc.UTF8, (offs, data), r.random(n) > 0.1), orc.Array(orc.UTF8, (offs, data), None),
            orc.Array(orc.FLOAT64, r.uniform(0, 1, n), r.random(n) > 0.3)]
    got, _ = run_both(qe, dctx, [cols], lambda m, scan: m.SelectionExec([scan], mk(m, ("eq", C(0), ("i32", 1)))), what="deferred string gather")
    assert got[0].num_rows == int(flag.sum())


def test_shapes_that_wait_all_the_same(qe, dctx):
    cols = make_columns(30_000, seed=9)
    # a string literal as an output column: its broadcast is sized by the row count
    got, _ = run_both(qe, dctx, [cols], lambda m, scan: m.ProjectionExec([m.SelectionExec([scan], mk(m, ("gt", C(5), ("f64", 0.3))))],
                                                                            [mk(m, C(1)), mk(m, ("str", "tag"))]), what="string literal output")
    assert got[0].columns()[1].values[0] == b"tag"
    # more outputs than one launch takes: several passes, each checked against the first
    many = [("add", C(1), ("i32", j)) for j in range(15)]
    run_both(qe, dctx, [cols], lambda m, scan: m.ProjectionExec([m.SelectionExec([scan], mk(m, ("gt", C(5), ("f64", 0.3))))],
                                                                  [mk(m, e) for e in many]), what="fifteen outputs")


def test_a_pending_batch_as_input(qe, dctx):
    """The output of one plan, not looked at, as the scan of the next and as the input of an aggregate and of
    qe_batch_to_host: each use settles it first."""
    cols = make_columns(120_000, seed=12, nulls=False)
    batch = gpu_batch(qe, dctx, cols)
    Cx = qe.ColumnExpression
    first = list(qe.SelectionExec([qe.ScanExec(dctx, [batch])], qe.GtExpression(Cx(5), qe.LiteralFloatExpression(0.25))).execute())[0]
    second = list(qe.SelectionExec([qe.ScanExec(dctx, [first])], qe.LtExpression(Cx(5), qe.LiteralFloatExpression(0.75))).execute())[0]
    v = cols[5].values
    keep = (v > 0.25) & (v < 0.75)
    assert second.num_rows == int(keep.sum())
    assert np.array_equal(np.asarray(second.columns()[1].values), cols[1].values[keep])
    third = list(qe.SelectionExec([qe.ScanExec(dctx, [batch])], qe.GtExpression(Cx(5), qe.LiteralFloatExpression(0.25))).execute())[0]
    agg = list(qe.AggregateExec([qe.ScanExec(dctx, [third])], [Cx(1)], [qe.CountExpression(None)]).execute())[0].columns()
    uk, cnt = np.unique(cols[1].values[v > 0.25], return_counts=True)
    order = np.argsort(np.asarray(agg[0].values))
    assert np.array_equal(np.asarray(agg[0].values)[order], uk) and np.array_equal(np.asarray(agg[1].values)[order], cnt)
    fourth = list(qe.SelectionExec([qe.ScanExec(dctx, [batch])], qe.GtExpression(Cx(5), qe.LiteralFloatExpression(0.25))).execute())[0]
    host = fourth.to_host()
    assert host.num_rows == int((v > 0.25).sum())
