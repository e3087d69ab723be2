"""Longer run of the seeded random expression trees of tests/test_gpu_trees.py (other seeds, more trees)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401  (before the engine: one HIP runtime)
import importlib

qe = importlib.import_module("query-engine_amd")
from oracle import oracle as orc
import test_gpu_trees as T
from test_gpu_parity import make_columns, run_both, mk, C
from util import gpu_batch

seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
ctx = qe.Context(0)
ctx.set_flags(qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP | (qe.FLAG_DEFERRED_ROW_COUNT if os.environ.get("QE_FUZZ_DEFER_ROWS") == "1" else 0) |
              (qe.FLAG_JIT_SYNC if os.environ.get("QE_FUZZ_JIT") == "1" else 0))
bad = 0
for rnd in range(4):
    cols = make_columns(700 + 311 * rnd, seed=100 + rnd + seed0)
    chunk, batch = orc.Chunk(cols), gpu_batch(qe, ctx, cols)
    for i, spec in enumerate(T.random_trees(seed0 * 10 + rnd, count // 4, 8)):
        try:
            T.check_tree(qe, ctx, chunk, batch, spec, f"seed {seed0} round {rnd} tree {i}")
        except BaseException as e:  # noqa: BLE001
            bad += 1
            print("FAIL", repr(e)[:300], flush=True)
            if bad > 5:
                sys.exit(1)
    print("round", rnd, "done", flush=True)
r = np.random.default_rng(seed0)
cols = make_columns(5000, seed=seed0)
done = 0
while done < 60:
    pred = T.random_boolean(r, int(r.integers(2, 7)))
    outs = [T.random_tree(r, int(r.integers(1, 7))) for _ in range(4)]
    if not T.has_column(pred) or not all(T.has_column(o) for o in outs):
        continue
    try:
        mk(orc, pred).evaluate(orc.Chunk(cols))
        for o in outs:
            mk(orc, o).evaluate(orc.Chunk(cols))
    except orc.OracleError:
        continue
    try:
        run_both(qe, ctx, [cols], lambda m, scan: m.ProjectionExec([m.SelectionExec([scan], mk(m, pred))], [mk(m, o) for o in outs] + [mk(m, C(2))]),
                 what=f"plan {done}")
    except BaseException as e:  # noqa: BLE001
        bad += 1
        print("FAIL plan", pred, outs, repr(e)[:300], flush=True)
        if bad > 5:
            sys.exit(1)
    done += 1
print("plans done; failures:", bad, flush=True)
sys.exit(1 if bad else 0)
