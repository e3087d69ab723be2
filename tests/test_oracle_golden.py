"""Pins the CPU oracle (oracle/) against what the reference's own tests assert
(/root/reference/query-engine/src/tests/mod.rs) on the data file they read, carried here as
tests/golden/userdata.npz, and against independent numpy / pyarrow cross-checks for the parts no
reference test covers ("parity unpinned": arithmetic, numeric comparison, grouping)."""
import numpy as np
import pytest

from oracle import oracle as orc
from util import USERDATA_COLUMNS, load_userdata

COL = {n: i for i, n in enumerate(USERDATA_COLUMNS)}


@pytest.fixture(scope="module")
def userdata():
    arrays, exp = load_userdata()
    return orc.Chunk(arrays), exp


def test_schema(userdata):
    # tests/mod.rs:28-32: 13 fields; id Int32, salary Float64, registration_dttm Timestamp(ns), 10 x Utf8
    chunk, exp = userdata
    assert [n for n, _ in exp["schema"]] == USERDATA_COLUMNS
    types = dict(exp["schema"])
    assert types["registration_dttm"] == "Timestamp(Nanosecond, None)"
    assert types["id"] == "Int32" and types["salary"] == "Float64"
    assert sum(1 for _, t in exp["schema"] if t == "Utf8") == 10
    assert exp["rows"] == 1000 and exp["row_groups"] == 1 and len(chunk) == 1000


def test_scan(userdata):
    # tests/mod.rs:34-84
    chunk, _ = userdata
    out = list(orc.ScanExec([chunk]).execute())
    assert len(out) == 1
    assert out[0].arrays[2].strings()[0] == b"Amanda" and out[0].arrays[3].strings()[0] == b"Jordan"
    assert out[0].arrays[2].strings()[1] == b"Albert" and out[0].arrays[3].strings()[1] == b"Freeman"


def test_projection(userdata):
    # tests/mod.rs:86-136: project(email, country) keeps expression order
    chunk, _ = userdata
    plan = orc.ProjectionExec([orc.ScanExec([chunk])], [orc.ColumnExpression(COL["email"]), orc.ColumnExpression(COL["country"])])
    out = list(plan.execute())[0]
    assert len(out.arrays) == 2
    assert out.arrays[0].strings()[:2] == [b"ajordan0@com.com", b"afreeman1@is.gd"]
    assert out.arrays[1].strings()[:2] == [b"Indonesia", b"Canada"]


def test_filter(userdata):
    # tests/mod.rs:138-188: gender == "Female" keeps all 13 columns, in input order
    chunk, exp = userdata
    pred = orc.EqExpression(orc.ColumnExpression(COL["gender"]), orc.LiteralStringExpression("Female"))
    out = list(orc.SelectionExec([orc.ScanExec([chunk])], pred).execute())[0]
    assert len(out.arrays) == 13
    assert out.arrays[2].strings()[0] == b"Amanda" and out.arrays[3].strings()[0] == b"Jordan"
    assert out.arrays[2].strings()[1] == b"Evelyn" and out.arrays[3].strings()[1] == b"Morgan"
    # independent cross-check: exactly the rows pyarrow sees as Female, in order
    idx = np.array(exp["female_row_indices"])
    assert len(out) == len(idx) == 482
    assert np.array_equal(out.arrays[COL["id"]].values, chunk.arrays[COL["id"]].values[idx])
    sal = chunk.arrays[COL["salary"]]
    assert np.array_equal(out.arrays[COL["salary"]].validity, sal.validity[idx])
    # the scalar-on-the-left arm gives the same mask (physical_expressions.rs:199-203)
    pred2 = orc.EqExpression(orc.LiteralStringExpression("Female"), orc.ColumnExpression(COL["gender"]))
    out2 = list(orc.SelectionExec([orc.ScanExec([chunk])], pred2).execute())[0]
    assert np.array_equal(out2.arrays[COL["id"]].values, out.arrays[COL["id"]].values)


def test_max_call_shape(userdata):
    # tests/mod.rs:190-207 asserts an impossible value (WIP in the reference) but fixes the call shape
    # aggregate([country], [max(salary)]).  Utf8 keys are not lowered by this build yet; the literal
    # aggregate is exercised on Int32 `id % 7`-style keys instead and cross-checked with numpy.
    chunk, exp = userdata
    ids = chunk.arrays[COL["id"]].values
    key = orc.Array(orc.INT32, (ids % 7).astype(np.int32))
    sal = chunk.arrays[COL["salary"]]
    scan = orc.ScanExec([orc.Chunk([key, sal])])
    lit = list(orc.AggregateExec([scan], [orc.ColumnExpression(0)], [orc.MaxExpression(orc.ColumnExpression(1)),
                                                                      orc.MinExpression(orc.ColumnExpression(1))], literal=True).execute())[0]
    sp = list(orc.AggregateExec([scan], [orc.ColumnExpression(0)], [orc.MaxExpression(orc.ColumnExpression(1)),
                                                                     orc.MinExpression(orc.ColumnExpression(1))]).execute())[0]
    for out in (lit, sp):
        assert len(out) == 7
        for g in range(7):
            k = out.arrays[0].values[g]
            m = (key.values == k) & sal.validity
            assert out.arrays[1].values[g] == sal.values[m].max()
            assert out.arrays[2].values[g] == sal.values[m].min()


def test_country_stats_cross_check(userdata):
    # per-country salary statistics against the pyarrow-derived table (dictionary-encoded key on the host)
    chunk, exp = userdata
    names = chunk.arrays[COL["country"]].strings()
    uniq = sorted(set(names))
    code = np.array([uniq.index(s) for s in names], dtype=np.int32)
    sal = chunk.arrays[COL["salary"]]
    scan = orc.ScanExec([orc.Chunk([orc.Array(orc.INT32, code), sal])])
    c1 = orc.ColumnExpression(1)
    out = list(orc.AggregateExec([scan], [orc.ColumnExpression(0)], [orc.MaxExpression(c1), orc.MinExpression(c1), orc.SumExpression(c1),
                                                                      orc.CountExpression(c1), orc.CountExpression(None), orc.AvgExpression(c1)]).execute())[0]
    assert len(out) == 120
    for g in range(len(out)):
        st = exp["country_stats"][uniq[out.arrays[0].values[g]].decode()]
        assert out.arrays[5].values[g] == st["rows"] and out.arrays[4].values[g] == st["nn"]
        if st["nn"]:
            assert out.arrays[1].values[g] == st["max"] and out.arrays[2].values[g] == st["min"]
            assert abs(out.arrays[3].values[g] - st["sum"]) <= 1e-12 * abs(st["sum"])
            assert abs(out.arrays[6].values[g] - st["sum"] / st["nn"]) <= 1e-12 * abs(st["sum"] / st["nn"])
        else:
            assert not out.arrays[1].validity[g] and not out.arrays[3].validity[g] and not out.arrays[6].validity[g]
    assert exp["country_stats"]["Indonesia"]["max"] == 284608.82 and exp["country_stats"]["Indonesia"]["nn"] == 88
    assert int(chunk.arrays[COL["id"]].values.astype(np.int64).sum()) == exp["id_sum"] == 500500
    assert exp["salary_nulls"] == 68


# ----------------------------------------------------------- unpinned semantics, cross-checked
def _rng():
    return np.random.default_rng(1234)


@pytest.mark.parametrize("dtype", [np.int32, np.int64, np.float64])
def test_arithmetic_matches_numpy(dtype):
    r = _rng()
    n = 1000
    if dtype == np.float64:
        a, b = r.uniform(-1e6, 1e6, n), r.uniform(-1e3, 1e3, n)
    else:
        info = np.iinfo(dtype)
        a = r.integers(info.min, info.max, n, dtype=dtype, endpoint=True)
        b = r.integers(info.min, info.max, n, dtype=dtype, endpoint=True)
        b[::7] = 0
        a[5], b[5] = info.min, -1
    va = r.random(n) > 0.1
    chunk = orc.Chunk([orc.Array.from_python(a, va), orc.Array.from_python(b)])
    c0, c1 = orc.ColumnExpression(0), orc.ColumnExpression(1)
    with np.errstate(all="ignore"):
        want = {"add": a + b, "sub": a - b, "mul": a * b}
    for name, cls in (("add", orc.AddExpression), ("sub", orc.SubExpression), ("mul", orc.MulExpression)):
        out = cls(c0, c1).evaluate(chunk).array
        assert np.array_equal(out.values.view(np.uint8), want[name].view(np.uint8)), name
        assert np.array_equal(out.validity, va)
    out = orc.DivExpression(c0, c1).evaluate(chunk).array
    if dtype == np.float64:
        with np.errstate(all="ignore"):
            assert np.array_equal(out.values, a / b, equal_nan=True)
    else:
        info = np.iinfo(dtype)
        bad = (b == 0) | ((a == info.min) & (b == -1))
        assert np.array_equal(out.validity, va & ~bad)
        ok = ~bad
        q = np.trunc(a[ok].astype(object) / 1).astype(object)  # exact python ints
        want_q = np.array([int(abs(int(x)) // abs(int(y))) * (1 if (x >= 0) == (y >= 0) else -1) for x, y in zip(a[ok], b[ok])], dtype=object)
        assert [int(v) for v in out.values[ok]] == [int(v) for v in want_q]
        rem = orc.ModExpression(c0, c1).evaluate(chunk).array
        assert [int(v) for v in rem.values[ok]] == [int(x) - int(qq) * int(y) for x, qq, y in zip(a[ok], want_q, b[ok])]
        del q


def test_f64_mul_then_add_is_not_fused():
    # a + b*c must round twice (two arrow2 passes in the reference: physical_expressions.rs:249-251)
    a = np.array([1.0, -1.0 + 2 ** -53])
    b = np.array([1.0 + 2 ** -30, 1.0 + 2 ** -52])
    c = np.array([1.0 - 2 ** -30, 1.0 - 2 ** -53])
    chunk = orc.Chunk([a, b, c])
    e = orc.AddExpression(orc.ColumnExpression(0), orc.MulExpression(orc.ColumnExpression(1), orc.ColumnExpression(2)))
    out = e.evaluate(chunk).array.values
    assert np.array_equal(out, a + (b * c))


def test_scalar_array_swap_quirk():
    # physical_expressions.rs:264-268: `lit - col` evaluates `col - lit`
    col = np.array([10, 20, 30], dtype=np.int32)
    chunk = orc.Chunk([col])
    out = orc.SubExpression(orc.LiteralIntegerExpression(1), orc.ColumnExpression(0)).evaluate(chunk).array.values
    assert out.tolist() == [9, 19, 29]
    out = orc.DivExpression(orc.LiteralIntegerExpression(2), orc.ColumnExpression(0)).evaluate(chunk).array.values
    assert out.tolist() == [5, 10, 15]
    out = orc.AddExpression(orc.LiteralIntegerExpression(1), orc.ColumnExpression(0)).evaluate(chunk).array.values
    assert out.tolist() == [11, 21, 31]


def test_scalar_scalar_arms():
    chunk = orc.Chunk([np.zeros(3, np.int32)])
    v = orc.AddExpression(orc.LiteralIntegerExpression(2 ** 31 - 1), orc.LiteralIntegerExpression(1)).evaluate(chunk)
    assert v.is_scalar and v.scalar.value == -2 ** 31  # wrapping (Rust release)
    v = orc.MulExpression(orc.LiteralFloatExpression(1.5), orc.LiteralFloatExpression(2.0)).evaluate(chunk)
    assert v.is_scalar and v.scalar.value == 3.0
    with pytest.raises(orc.OracleError) as e:
        orc.AddExpression(orc.LiteralIntegerExpression(1), orc.LiteralFloatExpression(1.0)).evaluate(chunk)
    assert e.value.name == "PhysicalTypeNotSuported"  # physical_expressions.rs:316-319
    v = orc.EqExpression(orc.LiteralIntegerExpression(1), orc.LiteralFloatExpression(1.0)).evaluate(chunk)
    assert v.is_scalar and v.scalar.value is False  # dyn Scalar PartialEq across types
    # a scalar projected through ProjectionExec is broadcast (physical_plan/mod.rs:146-148)
    out = list(orc.ProjectionExec([orc.ScanExec([chunk])], [orc.LiteralFloatExpression(2.5)]).execute())[0]
    assert out.arrays[0].values.tolist() == [2.5, 2.5, 2.5]


def test_errors():
    chunk = orc.Chunk([np.zeros(3, np.int32), np.zeros(3, np.float64)])
    with pytest.raises(orc.OracleError) as e:
        orc.ColumnExpression(7).evaluate(chunk)
    assert e.value.name == "PrimitiveTypeNotSuported"  # physical_expressions.rs:57-60
    with pytest.raises(orc.OracleError) as e:
        list(orc.SelectionExec([orc.ScanExec([chunk])], orc.ColumnExpression(0)).execute())
    assert e.value.name == "NoBooleanArrayForFilter"  # physical_plan/mod.rs:215-218
    with pytest.raises(orc.OracleError) as e:
        list(orc.ProjectionExec([], [orc.ColumnExpression(0)]).execute())
    assert e.value.name == "MissingInputPhysicalPlan"  # physical_plan/mod.rs:168-170
    with pytest.raises(orc.OracleError) as e:
        list(orc.AggregateExec([orc.ScanExec([])], [orc.ColumnExpression(0)], [orc.MaxExpression(orc.ColumnExpression(1))]).execute())
    assert e.value.name == "EmptyHashmapForAggregate"  # physical_plan/mod.rs:377


def test_comparisons_match_numpy_and_null_rows_drop():
    r = _rng()
    n = 777
    a = r.integers(-5, 5, n).astype(np.int64)
    b = r.integers(-5, 5, n).astype(np.int64)
    va = r.random(n) > 0.2
    chunk = orc.Chunk([orc.Array.from_python(a, va), b])
    c0, c1 = orc.ColumnExpression(0), orc.ColumnExpression(1)
    for cls, fn in ((orc.EqExpression, np.equal), (orc.NeqExpression, np.not_equal), (orc.GtExpression, np.greater),
                    (orc.GtEqExpression, np.greater_equal), (orc.LtExpression, np.less), (orc.LtEqExpression, np.less_equal)):
        out = cls(c0, c1).evaluate(chunk).array
        assert np.array_equal(out.values, fn(a, b)) and np.array_equal(out.validity, va)
        out = cls(orc.LiteralInt64Expression(1), c0).evaluate(chunk).array  # scalar on the left keeps its meaning
        assert np.array_equal(out.values, fn(1, a))
    sel = list(orc.SelectionExec([orc.ScanExec([chunk])], orc.GtExpression(c0, c1)).execute())[0]
    keep = va & (a > b)
    assert np.array_equal(sel.arrays[1].values, b[keep])  # null predicate rows are dropped, order kept
    f = np.array([1.0, np.nan, 3.0])
    out = orc.EqExpression(orc.ColumnExpression(0), orc.ColumnExpression(0)).evaluate(orc.Chunk([f])).array
    assert out.values.tolist() == [True, False, True]  # IEEE: NaN != NaN


def test_literal_aggregate_matches_single_pass_and_quirks():
    r = _rng()
    n = 2000
    k1 = r.integers(0, 6, n).astype(np.int32)
    k2 = r.integers(0, 6, n).astype(np.int32)
    kv = r.random(n) > 0.05
    v = r.uniform(-100, 100, n)
    vv = r.random(n) > 0.1
    chunks = [orc.Chunk([orc.Array(orc.INT32, k1[s], kv[s]), orc.Array(orc.INT32, k2[s]), orc.Array(orc.FLOAT64, v[s], vv[s])])
              for s in (slice(0, 700), slice(700, 1500), slice(1500, n))]
    aggs = [orc.MaxExpression(orc.ColumnExpression(2)), orc.MinExpression(orc.ColumnExpression(2))]
    # one key column: hash identity == key identity, literal and single-pass agree
    a = list(orc.AggregateExec([orc.ScanExec(chunks)], [orc.ColumnExpression(0)], aggs, literal=True).execute())[0]
    b = list(orc.AggregateExec([orc.ScanExec(chunks)], [orc.ColumnExpression(0)], aggs).execute())[0]
    pa, pb = np.argsort(a.arrays[0].values), np.argsort(b.arrays[0].values)
    for x, y in zip(a.arrays, b.arrays):
        assert np.array_equal(x.values[pa], y.values[pb])
    assert len(a) == 6  # rows with a null key are dropped, not grouped (physical_plan/mod.rs:367)
    # two key columns of one type: the commutative hash sum merges (x, y) with (y, x)
    # (physical_plan/mod.rs:310-320) — the literal form shows fewer groups than full key equality
    a2 = list(orc.AggregateExec([orc.ScanExec(chunks)], [orc.ColumnExpression(0), orc.ColumnExpression(1)], aggs, literal=True).execute())[0]
    b2 = list(orc.AggregateExec([orc.ScanExec(chunks)], [orc.ColumnExpression(0), orc.ColumnExpression(1)], aggs).execute())[0]
    assert len(b2) == 36 and len(a2) == 21


# ----------------------------------------------------------------- Parquet scan oracle
def test_parquet_oracle_decodes_the_reference_fixture():
    """oracle/parquet_oracle.py on the reference's userdata.parquet: the schema the reference pins
    (tests/mod.rs:31), the cells its tests assert (tests/mod.rs:40-83), and every column against the
    dump pyarrow made of the same file (tests/golden/userdata.npz)."""
    import os
    from oracle import parquet_oracle as porc
    here = os.path.dirname(os.path.abspath(__file__))
    f = porc.ParquetFile(os.path.join(here, "golden", "userdata.parquet"))
    schema = f.schema()
    assert schema[0] == ("registration_dttm", "Timestamp(Nanosecond, None)", True)
    assert schema[1] == ("id", "Int32", True) and schema[10] == ("salary", "Float64", True)
    assert all(t == "Utf8" and nullable for _, t, nullable in schema[2:10] + schema[11:])
    chunks = list(f.scan())
    assert len(chunks) == 1
    arrays = chunks[0].arrays
    assert arrays[2].strings()[:2] == [b"Amanda", b"Albert"] and arrays[3].strings()[:2] == [b"Jordan", b"Freeman"]
    g = np.load(os.path.join(here, "golden", "userdata.npz"))
    for (name, _, _), a in zip(schema, arrays):
        assert np.array_equal(a.validity, g[name + ".valid"]), name
        if a.type == orc.UTF8:
            assert np.array_equal(a.values[0], g[name + ".offsets"]) and np.array_equal(a.values[1], g[name + ".bytes"]), name
        else:
            assert np.array_equal(a.values, g[name + ".values"]), name
    # projection: named columns, file order (data_source.rs:49-62)
    proj = list(f.scan(["salary", "id"]))[0].arrays
    assert [a.type for a in proj] == [orc.INT32, orc.FLOAT64]


def test_parquet_schema_through_the_abi(qe):
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    s = qe.parquet_schema(os.path.join(here, "golden", "userdata.parquet"))
    assert s["rows"] == 1000 and s["row_groups"] == 1 and len(s["fields"]) == 13
    assert s["fields"][0]["data_type"] == "Timestamp(Nanosecond, None)" and s["fields"][10]["data_type"] == "Float64"
    with pytest.raises(qe.QeError) as e:
        qe.parquet_schema(os.path.join(here, "golden", "missing.parquet"))
    assert e.value.name == "IoError"
