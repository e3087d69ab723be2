"""The layer above the physical path — DataFrame builder, logical expressions, planner — through the C ABI,
with the reference's own tests (tests/mod.rs:28-207) re-expressed on the fixture data of userdata.parquet
(decoded once into tests/golden/userdata.npz; the Parquet reader itself is outside the device path)."""
import numpy as np
import pytest

from oracle import oracle as orc
from util import USERDATA_COLUMNS, gpu_batch, load_userdata

pytestmark = pytest.mark.gpu


@pytest.fixture()
def userdata(qe, ctx):
    arrays, exp = load_userdata()
    df = qe.DataFrame.from_batches(ctx, USERDATA_COLUMNS, [gpu_batch(qe, ctx, arrays)], label="src/tests/userdata.parquet")
    return df, arrays, exp


def strings(col, n):
    return [v.decode() for v in col.values[:n]]


def test_schema(qe, userdata):  # tests/mod.rs:28-32
    df, _, exp = userdata
    got = [(f["name"], f["data_type"], f["is_nullable"]) for f in df.schema()]
    want = [(name, "Int64" if t.startswith("Timestamp") else t, True) for name, t in exp["schema"]]  # timestamps travel as Int64
    assert got == want and len(got) == 13


def test_scan(qe, userdata):  # tests/mod.rs:34-84
    df, _, _ = userdata
    result = df.execute()
    cols = result[0].columns()
    assert strings(cols[2], 2) == ["Amanda", "Albert"] and strings(cols[3], 2) == ["Jordan", "Freeman"]


def test_projection(qe, userdata):  # tests/mod.rs:86-136
    df, _, _ = userdata
    P = qe.prelude
    result = df.project([P.col("email"), P.col("country")]).execute()
    cols = result[0].columns()
    assert len(cols) == 2
    assert strings(cols[0], 2) == ["ajordan0@com.com", "afreeman1@is.gd"] and strings(cols[1], 2) == ["Indonesia", "Canada"]


def test_filter(qe, userdata):  # tests/mod.rs:138-188
    df, _, exp = userdata
    P = qe.prelude
    result = df.filter(P.col("gender").eq(P.lit_string("Female"))).execute()
    cols = result[0].columns()
    assert len(cols) == 13 and result[0].num_rows == len(exp["female_row_indices"]) == 482
    assert strings(cols[2], 2) == ["Amanda", "Evelyn"] and strings(cols[3], 2) == ["Jordan", "Morgan"]
    assert np.array_equal(np.asarray(cols[1].values), np.asarray(exp["female_row_indices"]) + 1)  # id = row + 1: order kept


def test_max(qe, userdata):  # tests/mod.rs:190-207: aggregate([country], [max(salary)])
    df, _, exp = userdata
    P = qe.prelude
    out = df.aggregate([P.col("country")], [P.max(P.col("salary"))])
    assert [(f["name"], f["data_type"]) for f in out.schema()] == [("country", "Utf8"), ("max", "Float64")]
    cols = out.execute()[0].columns()
    got = {k.decode(): (None if (cols[1].validity is not None and not cols[1].validity[i]) else float(cols[1].values[i]))
           for i, k in enumerate(cols[0].values)}
    assert got == {k: v["max"] for k, v in exp["country_stats"].items()}


def test_plan_display_optimizer_and_errors(qe, userdata):
    df, arrays, _ = userdata
    P = qe.prelude
    q = df.filter(P.col("gender").eq(P.lit_string("Female"))).project([P.col("id"), P.col("salary") * P.lit_float(2.0)])
    assert q.logical_plan() == ("Projection: #id, #salary * '2',  \n"
                                " \tSelection: #gender == 'Female',  \n"
                                " \t \tScan: src/tests/userdata.parquet; projection=None \n")
    assert [(f["name"], f["data_type"], f["is_nullable"]) for f in q.schema()] == [("id", "Int32", True), ("mul", "Float64", False)]
    # projection push-down: the scan keeps only what the operators above it name
    opt = df.project([P.col("email"), P.col("country")]).optimize()
    assert opt.logical_plan().splitlines()[1].startswith(" \tScan: src/tests/userdata.parquet; projection=country, email, ")
    a, b = opt.execute()[0].columns(), df.project([P.col("email"), P.col("country")]).execute()[0].columns()
    assert [x.values for x in a] == [x.values for x in b]
    agg = df.filter(P.col("salary").gt(P.lit_float(100000.0))).aggregate([P.col("gender")], [P.count(P.col("id")), P.avg(P.col("salary"))])
    r1, r2 = agg.execute()[0].columns(), agg.optimize().execute()[0].columns()
    d1, d2 = dict(zip(r1[0].values, zip(r1[1].values, r1[2].values))), dict(zip(r2[0].values, zip(r2[1].values, r2[2].values)))
    assert set(d1) == set(d2) and all(d1[g][0] == d2[g][0] and abs(d1[g][1] - d2[g][1]) <= 1e-12 * abs(d1[g][1]) for g in d1)  # AVG: float sum order
    # an unknown column is an error value, not a panic
    with pytest.raises(qe.QeError) as e:
        df.project([P.col("nope")]).schema()
    assert e.value.name == "NoFieldInLogicalPlan"
    with pytest.raises(qe.QeError) as e:
        df.project([P.max(P.col("salary"))]).execute()
    assert e.value.name == "PhysicalExpressionNotSuported"
    with pytest.raises(qe.QeError) as e:
        df.aggregate([P.col("country")], [P.col("salary")]).execute()
    assert e.value.name == "PhysicalExpressionNotSuported"


def test_optimizer_keeps_what_the_root_outputs(qe, userdata):
    """Projection push-down prunes to what the operators above the scan need; a root that is the scan itself or a
    Selection outputs every column (the reference's pass, optimizer.rs:20-57, would leave such a scan the
    predicate's columns only, or none)."""
    df, _, _ = userdata
    P = qe.prelude
    female = lambda: P.col("gender").eq(P.lit_string("Female"))
    assert len(df.optimize().execute()[0].columns()) == 13
    out = df.filter(female()).optimize().execute()[0]
    assert len(out.columns()) == 13 and out.num_rows == 482
    q = df.filter(female()).project([P.col("email")]).optimize()
    assert "projection=email, gender" in q.logical_plan()  # not every column the Selection passes through
    out = q.execute()[0]
    assert len(out.columns()) == 1 and out.num_rows == 482 and out.columns()[0].values[0] == b"ajordan0@com.com"
    # names above a Projection are its outputs' names, not the scan's: `id` is not read for them
    q = df.project([P.col("salary") * P.lit_float(2.0), P.col("country")]).aggregate([P.col("country")], [P.max(P.col("mul"))]).optimize()
    assert "projection=country, salary" in q.logical_plan()
    assert q.optimize().logical_plan() == q.logical_plan()  # a second pass changes nothing


def test_dataframe_query_matches_physical_oracle(qe, ctx):
    """filter -> project -> group by through the DataFrame API equals the oracle's physical plan with
    the indices the planner must have resolved."""
    r = np.random.default_rng(12)
    n = 50_000
    k = r.integers(0, 40, n).astype(np.int64)
    a, b, c = r.uniform(-1, 1, n), r.uniform(-1, 1, n), r.uniform(0, 1, n)
    arrays = [orc.Array(orc.INT64, k), orc.Array(orc.FLOAT64, a), orc.Array(orc.FLOAT64, b), orc.Array(orc.FLOAT64, c)]
    P = qe.prelude
    df = qe.DataFrame.from_batches(ctx, ["k", "a", "b", "c"], [gpu_batch(qe, ctx, arrays)])
    # `add` and `k` are the field names the projection's schema gives its outputs (to_field)
    q = (df.filter(P.col("c").gt(P.lit_float(0.5)))
           .project([P.col("k"), P.col("a") + P.col("b") * P.col("c")])
           .aggregate([P.col("k")], [P.sum(P.col("add")), P.count(P.col("k")), P.min(P.col("add"))]))
    assert [f["name"] for f in q.schema()] == ["k", "sum", "count", "min"]
    got = q.execute()[0].columns()
    C = orc.ColumnExpression
    sel = orc.SelectionExec([orc.ScanExec([orc.Chunk(arrays)])], orc.GtExpression(C(3), orc.LiteralFloatExpression(0.5)))
    proj = orc.ProjectionExec([sel], [C(0), orc.AddExpression(C(1), orc.MulExpression(C(2), C(3)))])
    want = list(orc.AggregateExec([proj], [C(0)], [orc.SumExpression(C(1)), orc.CountExpression(C(0)), orc.MinExpression(C(1))]).execute())[0]
    og, ow = np.argsort(got[0].values), np.argsort(want.arrays[0].values)
    assert np.array_equal(np.asarray(got[0].values)[og], want.arrays[0].values[ow])
    assert np.array_equal(np.asarray(got[2].values)[og], want.arrays[2].values[ow])
    assert np.array_equal(np.asarray(got[3].values)[og], want.arrays[3].values[ow])
    v = a + b * c
    tol = np.array([1e-12 * np.abs(v[(c > 0.5) & (k == key)]).sum() for key in want.arrays[0].values[ow]])
    assert (np.abs(np.asarray(got[1].values)[og] - want.arrays[1].values[ow]) <= tol).all()
