"""Expression trees of any shape and size.  The reference evaluates a tree node by node, one array
per node (physical_expressions.rs:243-268), so nothing limits its size; the device evaluates a tree
inside one kernel and cuts a tree that exceeds a kernel's program into several launches.  Both the
hand-made oversized trees and the seeded random ones are compared with the oracle, bit for bit.
"""
import numpy as np
import pytest

from oracle import oracle as orc
from test_gpu_parity import C, agg_columns, check_partial_final, make_columns, mk, rng, run_agg, run_both
from util import assert_column_equal, gpu_batch

pytestmark = pytest.mark.gpu


def balanced(op_a, op_b, leaves):
    """Balanced binary tree over `leaves`, alternating two operators by level: needs log2(n)+1 stack slots."""
    level, ops = list(leaves), (op_a, op_b)
    d = 0
    while len(level) > 1:
        level = [(ops[d % 2], level[i], level[i + 1]) if i + 1 < len(level) else level[i] for i in range(0, len(level), 2)]
        d += 1
    return level[0]


def chain(op, leaves):
    t = leaves[0]
    for x in leaves[1:]:
        t = (op, t, x)
    return t


def oversized_trees():
    f = [C(4), C(5)]
    return {
        "32 leaves, six stack slots": balanced("add", "mul", [f[i % 2] for i in range(32)]),
        "64 leaves with literals": balanced("sub", "mul", [f[i % 2] if i % 3 else ("f64", 0.25 + i) for i in range(64)]),
        "50 distinct literals": chain("add", [C(4)] + [("f64", 1.0 + i / 8) for i in range(50)]),
        "200-node chain": chain("add", [f[i % 2] for i in range(100)]),
        "int64 wrap, deep": balanced("mul", "add", [C(2) if i % 2 else C(3) for i in range(32)]),
        "int32 division by zero inside": balanced("add", "div", [C(0) if i % 2 else C(1) for i in range(32)]),
        "boolean over every column": ("and", balanced("or", "and", [("gt", C(0), C(1)), ("lt", C(2), C(3)), ("gteq", C(4), C(5)), ("eq", C(6), C(7)),
                                                                     ("neq", C(1), ("i32", 3)), ("lt", C(3), ("i64", 0)), ("gt", C(5), ("f64", 0.5)), C(6),
                                                                     ("lteq", C(0), ("i32", 0)), ("gt", C(2), ("i64", 5)), C(7), ("neq", C(4), C(5)),
                                                                     ("eq", C(1), C(1)), ("lt", C(4), ("f64", 0.0)), ("gteq", C(3), C(3)), ("neq", C(6), C(7))]),
                                      ("gt", balanced("add", "mul", [f[i % 2] for i in range(16)]), ("f64", 0.1))),
    }


@pytest.mark.parametrize("n", [1, 257, 5000])
def test_oversized_expressions(qe, ctx, n):
    cols = make_columns(n, seed=n)
    chunk, batch = orc.Chunk(cols), gpu_batch(qe, ctx, cols)
    for name, spec in oversized_trees().items():
        want = mk(orc, spec).evaluate(chunk)
        got = mk(qe, spec).evaluate(batch)
        assert_column_equal(got.array.columns()[0], want.array, what=f"n={n} {name}")


def test_oversized_under_selection_and_aggregate(qe, ctx):
    trees = oversized_trees()
    cols = make_columns(20_000, seed=5)
    pred = trees["boolean over every column"]
    outs = [trees["32 leaves, six stack slots"], C(1), trees["50 distinct literals"], trees["int64 wrap, deep"], C(6),
            chain("add", [C(4), C(5)] * 9)]  # 35 nodes: fits a kernel alone, not next to the predicate

    def plan(m, scan):
        return m.ProjectionExec([m.SelectionExec([scan], mk(m, pred))], [mk(m, e) for e in outs])
    run_both(qe, ctx, [cols, make_columns(3, seed=6), make_columns(4097, seed=7)], plan, what="oversized predicate and outputs")
    run_both(qe, ctx, [cols], lambda m, scan: m.SelectionExec([scan], mk(m, pred)), what="oversized predicate, every column compacted")

    acols = [agg_columns(n, 40, seed=s, key_dtype=np.int32) for n, s in ((30_000, 11), (5, 12), (9000, 13))]
    v, f = C(2), C(5)
    deep_f64 = balanced("add", "mul", [v if i % 2 else f for i in range(32)])
    deep_key = ("mod", balanced("add", "mul", [C(1) if i % 2 else C(4) for i in range(32)]), ("i32", 7))
    deep_pred = ("gt", balanced("add", "mul", [f] * 32), ("f64", 1.0))
    run_agg(qe, ctx, acols, [deep_key], [("sum", deep_f64), ("max", deep_f64), ("avg", deep_f64), ("count", deep_f64), ("count", None)], pred=deep_pred,
            what="oversized key, inputs and predicate")
    # many aggregates, each small: together beyond one program (fifteen accumulators; a kernel carries sixteen)
    many = [(k, ("add", ("mul", v, ("f64", 1.0 + i)), ("mul", f, ("f64", 2.0 + i)))) for i, k in enumerate(["sum", "max", "min", "avg"] * 3)]
    run_agg(qe, ctx, acols, [C(0), C(1)], many, pred=("gt", f, ("f64", 0.25)), what="twelve aggregates over computed inputs")


def test_more_columns_than_one_kernel_reads(qe, ctx):
    n, k = 3000, 40
    r = rng(8)
    cols = [orc.Array(orc.FLOAT64, r.uniform(-1, 1, n), (r.random(n) > 0.02) if i % 5 == 0 else None) for i in range(k)]
    chunk, batch = orc.Chunk(cols), gpu_batch(qe, ctx, cols)
    for name, spec in (("chain", chain("add", [C(i) for i in range(k)])), ("balanced", balanced("add", "mul", [C(i) for i in range(k)]))):
        want = mk(orc, spec).evaluate(chunk)
        got = mk(qe, spec).evaluate(batch)
        assert_column_equal(got.array.columns()[0], want.array, what=f"{k} columns, {name}")
    spec = chain("add", [C(i) for i in range(k)])
    run_both(qe, ctx, [cols], lambda m, scan: m.ProjectionExec([m.SelectionExec([scan], mk(m, ("gt", spec, ("f64", 0.0))))], [mk(m, spec), mk(m, C(39))]),
             what="forty columns in predicate and output")


# ------------------------------------------------------------------ seeded random trees
NUM = {"i32": (0, 1), "i64": (2, 3), "f64": (4, 5)}
LIT = {"i32": lambda r: ("i32", int(r.integers(-5, 6))), "i64": lambda r: ("i64", int(r.integers(-5, 6))),
       "f64": lambda r: ("f64", float(r.integers(-8, 9)) / 4)}


def random_numeric(r, t, depth):
    p = r.random()
    if depth == 0 or p < 0.15:
        return C(int(NUM[t][r.integers(0, 2)])) if r.random() < 0.75 else LIT[t](r)
    ops = ["add", "sub", "mul", "div"] + ([] if t == "f64" else ["mod"])
    return (ops[r.integers(0, len(ops))], random_numeric(r, t, depth - 1), random_numeric(r, t, depth - 1))


def random_boolean(r, depth):
    p = r.random()
    if depth == 0 or p < 0.1:
        return C(int(r.integers(6, 8))) if r.random() < 0.85 else ("bool", bool(r.integers(0, 2)))
    if p < 0.55:
        t = ["i32", "i64", "f64"][r.integers(0, 3)]
        op = ["eq", "neq", "gt", "gteq", "lt", "lteq"][r.integers(0, 6)]
        return (op, random_numeric(r, t, depth - 1), random_numeric(r, t, depth - 1))
    op = ["and", "or", "eq", "neq"][r.integers(0, 4)]
    return (op, random_boolean(r, depth - 1), random_boolean(r, depth - 1))


def random_tree(r, depth):
    kind = r.integers(0, 4)
    return random_boolean(r, depth) if kind == 3 else random_numeric(r, ["i32", "i64", "f64"][kind], depth)


def has_column(spec):
    return spec[0] == "col" or (len(spec) == 3 and isinstance(spec[1], tuple) and (has_column(spec[1]) or has_column(spec[2])))


def random_trees(seed, count, max_depth):
    r = np.random.default_rng(seed)
    out = []
    while len(out) < count:
        t = random_tree(r, int(r.integers(1, max_depth + 1)))
        if has_column(t):
            out.append(t)
    return out


def check_tree(qe, ctx, chunk, batch, spec, what):
    try:
        want = mk(orc, spec).evaluate(chunk)
    except orc.OracleError as err:  # e.g. a null scalar (1 / 0 folded) meeting an array in a way the reference rejects
        with pytest.raises(qe.QeError) as e:
            mk(qe, spec).evaluate(batch)
        assert e.value.name == err.name, f"{what}: {spec}"
        return
    got = mk(qe, spec).evaluate(batch)
    assert_column_equal(got.array.columns()[0], want.array, what=f"{what}: {spec}")


def test_random_trees(qe, ctx):
    cols = make_columns(1500, seed=21)
    chunk, batch = orc.Chunk(cols), gpu_batch(qe, ctx, cols)
    for i, spec in enumerate(random_trees(1234, 150, 7)):
        check_tree(qe, ctx, chunk, batch, spec, f"random tree {i}")


def test_random_trees_compiled(qe, ctx):
    """A sample of the random trees through kernels compiled for them at run time."""
    cols = make_columns(1500 + 37, seed=22)
    chunk = orc.Chunk(cols)
    ctx.set_flags(qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP | qe.FLAG_JIT_SYNC)
    ctx.profile(True)
    try:
        ctx.profile_read()
        batch = gpu_batch(qe, ctx, cols)
        for i, spec in enumerate(random_trees(99, 25, 5)):
            check_tree(qe, ctx, chunk, batch, spec, f"compiled random tree {i}")
        assert len({k for k in ctx.profile_read() if "_jit<" in k}) >= 15  # trees of one shape share a kernel
    finally:
        ctx.profile(False)
        ctx.set_flags(qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP)


def test_random_trees_as_predicates_and_aggregate_inputs(qe, ctx):
    cols = make_columns(6000, seed=23)
    r = np.random.default_rng(5)
    done = 0
    while done < 12:
        pred = random_boolean(r, int(r.integers(2, 6)))
        outs = [random_tree(r, int(r.integers(1, 6))) for _ in range(3)]
        if not has_column(pred) or not all(has_column(o) for o in outs):
            continue
        try:
            mk(orc, pred).evaluate(orc.Chunk(cols))
            for o in outs:
                mk(orc, o).evaluate(orc.Chunk(cols))
        except orc.OracleError:
            continue
        run_both(qe, ctx, [cols], lambda m, scan: m.ProjectionExec([m.SelectionExec([scan], mk(m, pred))], [mk(m, o) for o in outs] + [mk(m, C(2))]),
                 what=f"random plan {done}: {pred} -> {outs}")
        done += 1
    acols = [agg_columns(8000, 30, seed=31, key_dtype=np.int32)]
    fnum = lambda d: random_f64_over_agg_columns(r, d)
    for j in range(6):
        ins = [fnum(int(r.integers(1, 6))) for _ in range(3)]
        run_agg(qe, ctx, acols, [C(0)], [("sum", ins[0]), ("max", ins[1]), ("avg", ins[2]), ("count", ins[0]), ("count", None)],
                pred=("gt", C(5), ("f64", 0.1 * j)), what=f"random aggregate inputs {j}: {ins}")


def random_f64_over_agg_columns(r, depth):
    """float64 trees over agg_columns' float columns (2: nullable values, 5: uniform [0,1)); no division, so sums stay finite."""
    if depth == 0 or r.random() < 0.15:
        return C(2) if r.random() < 0.4 else C(5) if r.random() < 0.7 else ("f64", float(r.integers(1, 9)) / 4)
    op = ["add", "sub", "mul"][r.integers(0, 3)]
    l, rr = random_f64_over_agg_columns(r, depth - 1), random_f64_over_agg_columns(r, depth - 1)
    if not has_column(l) and not has_column(rr):
        l = C(5)
    return (op, l, rr)


def test_more_aggregates_than_one_kernel_accumulates(qe, ctx):
    """Forty aggregates (a kernel carries sixteen accumulators): the list is split, the parts' outputs are put side
    by side.  Nullable inputs, computed inputs, a predicate, several batches, packed and hashed keys, no key at all."""
    acols = [agg_columns(n, 40, seed=s, key_dtype=np.int32) for n, s in ((30_000, 41), (5, 42), (9000, 43))]
    v, iv, i32v, f = C(2), C(3), C(4), C(5)
    kinds = ["sum", "max", "min", "avg", "count"]
    inputs = [v, iv, i32v, f, ("mul", v, f), ("add", f, ("f64", 1.5)), ("sub", iv, ("i64", 7)), ("mul", i32v, ("i32", 3))]
    many = [(kinds[i % 5], inputs[(i // 5 + i) % 8]) for i in range(39)] + [("count", None)]
    run_agg(qe, ctx, acols, [C(0), C(1)], many, pred=("gt", f, ("f64", 0.25)), what="forty aggregates, two packed keys")
    run_agg(qe, ctx, acols, [], many, what="forty aggregates, one global group")
    run_agg(qe, ctx, acols, [C(0), C(1), ("mod", C(4), ("i32", 3)), ("gt", f, ("f64", 0.5)), ("mod", C(1), ("i32", 2))], many[:25],
            what="twenty-five aggregates, five keys (hashed tuple)")
    run_agg(qe, ctx, [agg_columns(150_000, 100_000, seed=44, nulls=False)], [C(0)], many[:20], hint=64, what="split aggregation, table growth")
    # PARTIAL per shard -> FINAL, both split (an AVG travels as a sum and a count)
    check_partial_final(qe, ctx, agg_columns(60_000, 700, seed=45), many[:30])
    # the streaming accumulator protocol: batches fed one by one to a split state, a running result in between
    exact = [(k, x) for k, x in many if k in ("max", "min", "count")][:20]
    batches = [gpu_batch(qe, ctx, a) for a in acols]
    state = qe.AggregationState(ctx, [mk(qe, C(0))], [mk(qe, a) for a in exact], group_capacity_hint=8)
    state.accumulate(batches[0])
    assert state.final_value().num_rows == 40
    for b in batches[1:]:
        state.accumulate(b)
    got = state.final_value().columns()
    state.close()
    want = list(orc.AggregateExec([orc.ScanExec([orc.Chunk(a) for a in acols])], [mk(orc, C(0))], [mk(orc, a) for a in exact]).execute())[0].arrays
    og, ow = np.argsort(np.asarray(got[0].values)), np.argsort(want[0].values)
    assert len(got) == len(want) == 21
    for g, w in zip(got, want):
        wv = np.ones(len(w), np.bool_) if w.validity is None else w.validity
        gv = np.ones(len(w), np.bool_) if g.validity is None else np.asarray(g.validity, dtype=bool)
        assert np.array_equal(gv[og], wv[ow])
        assert np.array_equal(np.asarray(g.values)[og][wv[ow]], w.values[ow][wv[ow]])
