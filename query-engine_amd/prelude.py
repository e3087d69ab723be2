"""prelude.rs:7-33: col, lit_string, lit_int, lit_float, lit_bool, max, min — plus lit_int64, sum,
count and avg, which the reference defines only as logical expressions (logical_expression.rs:386-423)."""
from .dataframe import LogicalExpression as _L


def _agg(kind):
    import importlib
    qe = importlib.import_module(__package__)
    return {"max": qe.AGG_MAX, "min": qe.AGG_MIN, "sum": qe.AGG_SUM, "count": qe.AGG_COUNT, "avg": qe.AGG_AVG}[kind]


def col(name): return _L("col", name)
def lit_string(value): return _L("string", value)
def lit_int(value): return _L("int", int(value))
def lit_int64(value): return _L("int64", int(value))
def lit_float(value): return _L("float", float(value))
def lit_bool(value): return _L("bool", bool(value))
def max(expr): return _L("aggregate", op=_agg("max"), left=expr)      # noqa: A001
def min(expr): return _L("aggregate", op=_agg("min"), left=expr)      # noqa: A001
def sum(expr): return _L("aggregate", op=_agg("sum"), left=expr)      # noqa: A001
def count(expr): return _L("aggregate", op=_agg("count"), left=expr)
def avg(expr): return _L("aggregate", op=_agg("avg"), left=expr)
