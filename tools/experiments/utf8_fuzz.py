"""Seeded random plans over tables with string columns: utf8 predicates (column against literal, column against
column) combined with numeric ones, string columns and literals among the outputs, string group keys alone and
in tuples - against the oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import importlib

qe = importlib.import_module("query-engine_amd")
from oracle import oracle as orc
from test_gpu_parity import C, _rows_by_key, mk, run_both
from util import gpu_batch

seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ctx = qe.Context(0)
ctx.set_flags(qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP | (qe.FLAG_DEFERRED_ROW_COUNT if os.environ.get("QE_FUZZ_DEFER_ROWS") == "1" else 0) |
              (qe.FLAG_JIT_SYNC if os.environ.get("QE_FUZZ_JIT") == "1" else 0))
r = np.random.default_rng(seed0)
WORDS = ["", "a", "ab", "abc", "Ünïcode", "a longer string with spaces", "z" * 70, "Female", "Male", "x" * 300]
bad = 0


def table(n, seed):
    q = np.random.default_rng(seed)
    s1 = [None if x == 0 else WORDS[x % len(WORDS)] for x in q.integers(0, 40, n)]
    s2 = [WORDS[x] for x in q.integers(0, 4, n)]
    s3 = [None if x % 11 == 0 else "k%d" % (x % 500) for x in q.integers(0, 5000, n)]
    return [orc.Array.from_python(s1), orc.Array.from_python(s2), orc.Array.from_python(s3),
            orc.Array(orc.INT32, q.integers(-3, 4, n).astype(np.int32), q.random(n) > 0.05),
            orc.Array(orc.INT64, q.integers(-100, 100, n)), orc.Array(orc.FLOAT64, q.uniform(-10, 10, n), q.random(n) > 0.1),
            orc.Array(orc.BOOLEAN, q.random(n) < 0.5)]


def random_pred(depth):
    p = r.random()
    if depth == 0 or p < 0.5:
        k = int(r.integers(0, 7))
        if k == 0:
            return ("eq", C(int(r.integers(0, 3))), ("str", WORDS[r.integers(0, len(WORDS))]))
        if k == 1:
            return ("neq", ("str", WORDS[r.integers(0, len(WORDS))]), C(int(r.integers(0, 3))))
        if k == 2:
            return ("eq", C(0), C(1))
        if k == 3:
            return ("neq", C(1), C(0))
        if k == 4:
            return ("gt", C(5), ("f64", float(r.integers(-8, 8))))
        if k == 5:
            return ("lteq", C(3), ("i32", int(r.integers(-3, 4))))
        return C(6)
    return (["and", "or"][r.integers(0, 2)], random_pred(depth - 1), random_pred(depth - 1))


for it in range(rounds):
    sizes = [int(r.choice([1, 2, 300, 4097, 20_000])) for _ in range(int(r.integers(1, 4)))]
    parts = [table(n, seed0 * 1000 + it * 10 + j) for j, n in enumerate(sizes)]
    pred = random_pred(int(r.integers(0, 4)))
    what = f"it={it} sizes={sizes} pred={pred}"
    try:
        outs_pool = [C(0), C(1), C(2), C(3), C(4), C(5), C(6), ("str", "lit"), ("add", C(4), ("i64", 1)), ("mul", C(5), C(5)), ("i32", 5)]
        outs = [outs_pool[i] for i in r.choice(len(outs_pool), size=int(r.integers(1, 8)), replace=False)]
        shape = int(r.integers(0, 3))
        if shape == 0:
            run_both(qe, ctx, parts, lambda m, scan: m.SelectionExec([scan], mk(m, pred)), what=what + " selection")
        elif shape == 1:
            run_both(qe, ctx, parts, lambda m, scan: m.ProjectionExec([m.SelectionExec([scan], mk(m, pred))], [mk(m, o) for o in outs]), what=what + f" outs={outs}")
        else:
            keysets = [[C(0)], [C(2)], [C(0), C(3)], [C(1), C(0)], [C(2), C(4), C(6)], [C(3), C(1)], [("mod", C(4), ("i64", 3)), C(1)]]
            groups = keysets[r.integers(0, len(keysets))]
            aggs = [("count", None), ("max", C(4)), ("min", C(5)), ("count", C(5)), ("sum", C(4)), ("max", C(3))][: int(r.integers(1, 7))]

            def build(m, scan):
                return m.AggregateExec([m.SelectionExec([scan], mk(m, pred))], [mk(m, g) for g in groups], [mk(m, a) for a in aggs])
            try:
                want = _rows_by_key(list(build(orc, orc.ScanExec([orc.Chunk(p) for p in parts])).execute())[0].arrays, len(groups), True)
            except orc.OracleError as e:
                try:
                    list(build(qe, qe.ScanExec(ctx, [gpu_batch(qe, ctx, p) for p in parts])).execute())
                    raise AssertionError(f"oracle raised {e.name}, the product did not")
                except qe.QeError as e2:
                    assert e2.name == e.name, (e.name, e2.name)
                continue
            got = _rows_by_key(list(build(qe, qe.ScanExec(ctx, [gpu_batch(qe, ctx, p) for p in parts])).execute())[0].columns(), len(groups), False)
            assert got == want, f"{what} groups={groups} aggs={aggs}: {len(got)} vs {len(want)}; {next(((k, got.get(k), want[k]) for k in want if got.get(k) != want[k]), None)}"
    except BaseException as e:  # noqa: BLE001
        bad += 1
        print("FAIL", repr(e)[:600], what, flush=True)
        if bad > 5:
            sys.exit(1)
    if it % 20 == 19:
        print("iteration", it + 1, "failures", bad, flush=True)
print("done; failures:", bad, flush=True)
sys.exit(1 if bad else 0)
