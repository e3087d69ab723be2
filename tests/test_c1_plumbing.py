"""The CPU-runnable configuration: 1 M rows, one batch, Scan(memory) -> Selection(c > 0.5) -> Projection(a + b*c).
On the CPU it exercises the oracle's operator model end to end (API shape, stable row order, float64 results
bit for bit: two roundings, no fused multiply-add); on a GPU the same plan through the C ABI must reproduce
it exactly."""
import numpy as np
import pytest

from oracle import oracle as orc

N = 1_000_000


def table(seed=42):
    r = np.random.default_rng(seed)
    return r.uniform(-1, 1, N), r.uniform(-1, 1, N), r.uniform(0, 1, N)


def plan(m, scan):
    C = m.ColumnExpression
    sel = m.SelectionExec([scan], m.GtExpression(C(2), m.LiteralFloatExpression(0.5)))
    return m.ProjectionExec([sel], [m.AddExpression(C(0), m.MulExpression(C(1), C(2))), C(2)])


def expected(a, b, c):
    keep = c > 0.5
    t = b[keep] * c[keep]          # rounded once ...
    return a[keep] + t, c[keep]    # ... and once more: what two separate passes give


def test_c1_oracle():
    a, b, c = table()
    chunk = orc.Chunk([orc.Array(orc.FLOAT64, a), orc.Array(orc.FLOAT64, b), orc.Array(orc.FLOAT64, c)])
    out = list(plan(orc, orc.ScanExec([chunk])).execute())
    assert len(out) == 1 and len(out[0].arrays) == 2
    v, cc = expected(a, b, c)
    assert len(out[0]) == len(v) and abs(len(v) - N // 2) < 5000
    assert np.array_equal(out[0].arrays[0].values, v)   # bit for bit
    assert np.array_equal(out[0].arrays[1].values, cc)  # rows keep their input order
    assert out[0].arrays[0].validity is None


@pytest.mark.gpu
def test_c1_device(qe, ctx):
    from util import gpu_batch
    a, b, c = table()
    batch = gpu_batch(qe, ctx, [orc.Array(orc.FLOAT64, a), orc.Array(orc.FLOAT64, b), orc.Array(orc.FLOAT64, c)])
    out = list(plan(qe, qe.ScanExec(ctx, [batch])).execute())
    assert len(out) == 1
    cols = out[0].columns()
    v, cc = expected(a, b, c)
    assert out[0].num_rows == len(v)
    assert np.array_equal(np.asarray(cols[0].values), v)
    assert np.array_equal(np.asarray(cols[1].values), cc)
