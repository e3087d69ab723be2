"""Seeded random aggregations through the other ways to run them: PARTIAL per shard -> FINAL, the deferred
group count (padded output, no host synchronisation), the streaming accumulator protocol - against the oracle."""
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
from test_gpu_parity import C, agg_columns, check_partial_final, mk
from util import gpu_batch

seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ctx = qe.Context(0)
BASE = qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP | (qe.FLAG_DETERMINISTIC_SUM if os.environ.get("QE_FUZZ_DET") == "1" else 0)
ctx.set_flags(BASE)
r = np.random.default_rng(seed0)
bad = 0


def exact_compare(got_cols, want_arrays, nk, what):
    """Bit-exact comparison of aggregate outputs (no float sums among them), rows matched by the key columns."""
    def rows(cols, oracle):
        lists = []
        for c in cols:
            vals = np.asarray(c.values)
            valid = np.ones(len(vals), bool) if c.validity is None else np.asarray(c.validity, dtype=bool)
            lists.append([v.item() if ok else None for v, ok in zip(vals, valid)])
        return {tuple(row[:nk]): row[nk:] for row in zip(*lists)}
    g, w = rows(got_cols, False), rows(want_arrays, True)
    assert g == w, f"{what}: {len(g)} vs {len(w)} groups; first difference {next(((k, g.get(k), w[k]) for k in w if g.get(k) != w[k]), None)}"


class Col:  # a filtered device column as arrays
    def __init__(self, values, validity):
        self.values, self.validity = values, validity


for it in range(rounds):
    n_keys = int(r.choice([1, 3, 10, 40, 700, 3000]))
    nulls = bool(r.integers(0, 2))
    key_dtype = np.int32 if r.random() < 0.5 else np.int64
    inputs = [C(2), C(3), C(4), C(5), ("mul", C(2), C(5)), ("sub", C(3), ("i64", 7)), ("mul", C(4), ("i32", 3))]
    na = int(r.choice([1, 3, 6, 12, 20]))
    kinds_all = ["sum", "max", "min", "avg", "count"]
    aggs = []
    for _ in range(na):
        kind = kinds_all[r.integers(0, 5)]
        aggs.append((kind, None) if kind == "count" and r.random() < 0.4 else (kind, inputs[r.integers(0, len(inputs))]))
    what = f"it={it} keys={n_keys} nulls={nulls} {key_dtype.__name__} aggs={aggs}"
    try:
        cols = agg_columns(60_000, n_keys, seed=seed0 * 1000 + it, key_dtype=key_dtype, nulls=nulls, zipf=bool(r.integers(0, 2)))
        check_partial_final(qe, ctx, cols, aggs)
        # exact aggregates only from here on (row order of float sums is not the point)
        exact = [a for a in aggs if a[0] in ("max", "min", "count") or (a[0] == "sum" and a[1] in (C(3), C(4), ("sub", C(3), ("i64", 7)), ("mul", C(4), ("i32", 3))))]
        exact = exact + [("count", None)]
        groups = [C(0)] if r.random() < 0.6 else [C(0), C(1)]
        parts = [agg_columns(n, n_keys, seed=seed0 * 1000 + it + 500 + j, key_dtype=key_dtype, nulls=nulls) for j, n in enumerate((20_000, 3, 9000))]
        want = list(orc.AggregateExec([orc.ScanExec([orc.Chunk(p) for p in parts])], [mk(orc, g) for g in groups], [mk(orc, a) for a in exact]).execute())[0].arrays
        batches = [gpu_batch(qe, ctx, p) for p in parts]
        # deferred group count
        ctx.set_flags(BASE | qe.FLAG_DEFERRED_GROUP_COUNT)
        hint = int(r.choice([0, 8, 4000]))
        cols_d = None
        try:
            out = list(qe.AggregateExec([qe.ScanExec(ctx, batches)], [mk(qe, g) for g in groups], [mk(qe, a) for a in exact], group_capacity_hint=hint).execute())[0]
            ctx.synchronize()
            cols_d = out.columns()
        except qe.QeError as e:
            # a deferred aggregate cannot grow its table: the overflow report is the contract when the hint was too small
            assert e.name == "AggregateCapacity" and len(want[0]) > max(hint, 512), (e.name, hint, len(want[0]))
        finally:
            ctx.set_flags(BASE)
        if cols_d is not None:
            keep = np.asarray(cols_d[-1].values) > 0
            filtered = [Col(np.asarray(c.values)[keep], None if c.validity is None else np.asarray(c.validity, dtype=bool)[keep]) for c in cols_d]
            exact_compare(filtered, want, len(groups), what + " deferred")
        # streaming protocol
        state = qe.AggregationState(ctx, [mk(qe, g) for g in groups], [mk(qe, a) for a in exact], group_capacity_hint=int(r.choice([0, 8])))
        for b in batches:
            state.accumulate(b)
            if r.random() < 0.3:
                state.final_value()
        got = state.final_value().columns()
        state.close()
        exact_compare(got, want, len(groups), what + " streaming")
    except BaseException as e:  # noqa: BLE001
        bad += 1
        print("FAIL", repr(e)[:500], what, flush=True)
        if bad > 5:
            sys.exit(1)
    if it % 10 == 9:
        print("iteration", it + 1, "failures", bad, flush=True)
print("done; failures:", bad, flush=True)
sys.exit(1 if bad else 0)
