"""Host lowering without a GPU: qe_lowering_describe, and that the generated Spec header is current."""
import importlib
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_describe_north_star(qe):
    C, F = qe.ColumnExpression, qe.LiteralFloatExpression
    d = qe.describe_lowering([qe.TYPE_INT64] + [qe.TYPE_FLOAT64] * 3, predicate=qe.GtExpression(C(3), F(0.5)),
                             exprs=[C(0), qe.AddExpression(C(1), qe.MulExpression(C(2), C(3)))], group_exprs=[C(0)],
                             agg_exprs=[qe.SumExpression(C(1)), qe.CountExpression(None)])
    (launch,) = d["launches"]
    assert launch["kind"] == "aggregate"
    assert [t["role"] for t in launch["tasks"]] == [1, 2, 3, 4]  # predicate, key, lookup, one accumulator (COUNT(*) = row count)
    assert launch["keys"] == [{"type": qe.TYPE_INT64, "shift": 0, "bits": 64}]
    assert launch["accs"] == [{"op": 2, "type": qe.TYPE_FLOAT64, "track": 0, "has_input": 1}]
    assert len(launch["cache_kind"]) == 4  # k, a, b, c each loaded once per wave-step
    ops = [w & 0xff for w in launch["code"]]
    assert ops.count(0) == 3  # three expressions, each closed by OP_END


def test_describe_filter_and_project_and_errors(qe):
    C, F = qe.ColumnExpression, qe.LiteralFloatExpression
    f64 = [qe.TYPE_FLOAT64] * 3
    fma = lambda: qe.AddExpression(C(0), qe.MulExpression(C(1), C(2)))
    (p,) = qe.describe_lowering(f64, exprs=[fma()])["launches"]
    assert p["kind"] == "project" and len(p["outs"]) == 1 and p["outs"][0]["nullable"] == 0
    (f,) = qe.describe_lowering(f64, predicate=qe.GtExpression(C(2), F(0.5)), exprs=[fma()])["launches"]
    assert f["kind"] == "filter" and f["pred_cache_mask"] == 1 and f["out_cache_mask"] == 7
    (fn,) = qe.describe_lowering(f64, [True, False, False], predicate=qe.GtExpression(C(2), F(0.5)), exprs=[fma()])["launches"]
    assert fn["outs"][0]["nullable"] == 1 and fn["sig"] != f["sig"]
    # more expressions than one launch holds -> several launches
    many = qe.describe_lowering(f64, exprs=[qe.AddExpression(C(0), F(float(i))) for i in range(30)])["launches"]
    assert len(many) == 3 and sum(len(l["outs"]) for l in many) == 30
    # pass-through columns need no kernel
    assert qe.describe_lowering(f64, exprs=[C(1)])["launches"] == []
    try:
        qe.describe_lowering(f64, exprs=[qe.AddExpression(C(0), qe.ColumnExpression(7))])
        assert False
    except qe.QeError as e:
        assert e.name == "PrimitiveTypeNotSuported"


def test_generated_specs_are_current(tmp_path):
    """qe_static_specs.h must equal what the generator writes from today's lowering."""
    spec = importlib.util.spec_from_file_location("gen_static_specs", os.path.join(ROOT, "tools", "gen_static_specs.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    committed = open(gen.OUT).read()
    gen.OUT = str(tmp_path / "specs.h")
    gen.main()
    assert open(gen.OUT).read() == committed, "run tools/gen_static_specs.py and rebuild"


def test_runtime_compilation_offline(qe):
    """The run-time compiler builds a kernel for shapes that were not built ahead of time (no GPU needed):
    nullable inputs, operand swaps, division, casts, more columns than the cache holds."""
    C, F, I = qe.ColumnExpression, qe.LiteralFloatExpression, qe.LiteralIntegerExpression
    i32, i64, f64, boolean, utf8 = qe.TYPE_INT32, qe.TYPE_INT64, qe.TYPE_FLOAT64, qe.TYPE_BOOLEAN, qe.TYPE_UTF8
    nk, nb = qe.compile_lowering([i64, f64, f64, f64], [False, True, False, True], predicate=qe.LtExpression(C(3), F(0.25)),
                                 exprs=[C(0), qe.SubExpression(C(1), qe.MulExpression(C(2), C(3)))], group_exprs=[C(0)],
                                 agg_exprs=[qe.SumExpression(C(1)), qe.MinExpression(C(1)), qe.CountExpression(C(1)), qe.AvgExpression(C(1))])
    assert nk == 1 and nb > 4096
    nk, _ = qe.compile_lowering([i32, f64, boolean], [True, False, False], predicate=qe.AndExpression(C(2), qe.GtExpression(C(0), I(3))),
                                exprs=[qe.AddExpression(C(0), C(0)), qe.MulExpression(C(1), F(2.0)), C(2)])
    assert nk == 1
    nk, _ = qe.compile_lowering([i32, f64], exprs=[qe.DivExpression(C(1), F(2.0)), qe.ModExpression(C(0), I(7)), qe.DivExpression(I(100), C(0))])
    assert nk == 1
    nk, _ = qe.compile_lowering([i32] * 9, exprs=[qe.AddExpression(C(i), C(8 - i)) for i in range(4)])  # 8 columns, 6 cached
    assert nk == 1
    # a key tuple wider than 64 bits: grouped by a hash of the tuple (multiply, add, OP_MIX64), still a compiled kernel;
    # with a string in the tuple the hash of the string needs the general kernels
    nk, _ = qe.compile_lowering([i64, i64, i32, f64], group_exprs=[C(0), C(1), C(2)], agg_exprs=[qe.SumExpression(C(3))])
    assert nk == 1
    d = qe.describe_lowering([i64, i64, f64], group_exprs=[C(0), C(1)], agg_exprs=[qe.SumExpression(C(2))])
    assert d["launches"][0]["specialisable"] == 1
    assert qe.compile_lowering([utf8, i32, f64], group_exprs=[C(0), C(1)], agg_exprs=[qe.SumExpression(C(2))]) == (0, 0)
    # utf8 comparisons run on the general kernels only
    assert qe.compile_lowering([utf8, i32], exprs=[qe.EqExpression(C(0), C(0))]) == (0, 0)
    d = qe.describe_lowering([utf8, i32], exprs=[qe.EqExpression(C(0), C(0))])
    assert d["launches"][0]["specialisable"] == 0


def test_order_independent_sum_lowering(qe):
    """QE_FLAG_DETERMINISTIC_SUM: a Float64 SUM becomes four integer sums (three limbs of the value's fixed-point image
    and its infinity / NaN marks), AVG the same plus its count; integer sums stay as they are; the run-time compiler
    builds the shape."""
    C, F = qe.ColumnExpression, qe.LiteralFloatExpression
    i64, f64 = qe.TYPE_INT64, qe.TYPE_FLOAT64
    flags = qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP | qe.FLAG_DETERMINISTIC_SUM
    plan = dict(predicate=qe.GtExpression(C(3), F(0.5)), exprs=[C(0), qe.AddExpression(C(1), qe.MulExpression(C(2), C(3)))],
                group_exprs=[C(0)], agg_exprs=[qe.SumExpression(C(1)), qe.CountExpression(None)])
    (launch,) = qe.describe_lowering([i64, f64, f64, f64], flags=flags, **plan)["launches"]
    assert launch["accs"] == [{"op": 1, "type": i64, "track": 0, "has_input": 1}] * 4
    assert launch["specialisable"] == 1
    (plain,) = qe.describe_lowering([i64, f64, f64, f64], **plan)["launches"]
    assert len(plain["accs"]) == 1 and plain["sig"] != launch["sig"]
    (ints,) = qe.describe_lowering([i64, i64], flags=flags, group_exprs=[C(0)], agg_exprs=[qe.SumExpression(C(1))])["launches"]
    assert ints["accs"] == [{"op": 1, "type": i64, "track": 0, "has_input": 1}]
    (avg,) = qe.describe_lowering([i64, f64], [False, True], flags=flags, group_exprs=[C(0)], agg_exprs=[qe.AvgExpression(C(1))])["launches"]
    assert [a["op"] for a in avg["accs"]] == [1, 1, 1, 1, 5]  # limbs, marks, COUNT(x)
    nk, nb = qe.compile_lowering([i64, f64, f64, f64], flags=flags, **plan)
    assert nk == 1 and nb > 4096


def test_logical_expression_display(qe):
    """impl Display for LogicalExpression (logical_expression.rs:68-96, 123-127, 276-280, 378-382)."""
    P = qe.prelude
    assert str(P.col("gender").eq(P.lit_string("Female"))) == "#gender == 'Female'"
    assert str(P.max(P.col("salary"))) == "max (#salary)"
    assert str(P.col("a") + P.col("b") * P.lit_float(0.5)) == "#a + #b * '0.5'"
    assert str(P.lit_int(7).lteq(P.col("x")).and_(P.lit_bool(True))) == "'7' <= #x && 'true'"
    assert str(P.count(P.col("id") % P.lit_int64(3))) == "count (#id % '3')"
