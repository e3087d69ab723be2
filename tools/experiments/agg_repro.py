import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, importlib
qe = importlib.import_module("query-engine_amd")
from oracle import oracle as orc
from test_gpu_parity import C, agg_columns, run_agg, mk
from util import gpu_batch, sort_by_first_columns
ctx = qe.Context(0); ctx.set_flags(qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP)
case = sys.argv[1]
if case == "2":
    cols = [agg_columns(30000, 3000, seed=1, key_dtype=np.int32, nulls=True), agg_columns(7, 3000, seed=2, key_dtype=np.int32, nulls=True)]
    x = ("add", ("sub", ("add", C(2), ("f64", 1.0)), ("sub", ("f64", 0.5), C(2))), C(2))
    for groups in ([C(1), ("mod", C(0), ("i32", 7)), C(0)], [C(1), C(0)], [C(0)]):
        for aggs in ([("max", x), ("min", x)], [("max", C(2))]):
            for hint in (0, 200000):
                try:
                    run_agg(qe, ctx, cols, groups, aggs, hint=hint, what="x")
                    print("ok  ", len(groups), aggs[0][1] == x, hint)
                except AssertionError as e:
                    print("FAIL", len(groups), aggs[0][1] == x, hint, str(e)[:200])
if case == "1":
    cols = [agg_columns(4097, 16, seed=1, key_dtype=np.int64, nulls=False)]
    groups = [("mod", C(0), ("i64", 7)), ("mod", C(4), ("i32", 5)), C(0), ("gt", C(5), ("f64", 0.5)), C(1)]
    lit = [("sum", ("f64", 2.0)), ("max", ("f64", 2.0)), ("avg", ("f64", 2.0))]
    base = [("avg", C(2)), ("min", ("mul", C(4), ("i32", 3))), ("max", ("sub", C(3), ("i64", 7))), ("min", ("add", C(5), ("f64", 1.5)))]
    for g in (groups, [C(0)]):
        for aggs in (lit, base, base * 4, base * 4 + lit, lit + base * 4):
            for hint in (0, 70000):
                try:
                    run_agg(qe, ctx, cols, g, aggs, hint=hint, what="x")
                    print("ok  ", len(g), len(aggs), hint)
                except BaseException as e:
                    print("FAIL", len(g), len(aggs), hint, repr(e)[:300])
