"""Looks for cliffs along two more axes: the selectivity of a filter (Selection -> Projection), and skewed group keys
(zipf over a varying number of keys), with and without a hint."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, importlib
qe = importlib.import_module("query-engine_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64_000_000
torch.cuda.set_device(0)
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctx = qe.Context(0, stream=s.cuda_stream)
dev = torch.device("cuda", 0)
k = torch.empty(n, dtype=torch.int64, device=dev)
a, b, c = (torch.empty(n, dtype=torch.float64, device=dev) for _ in range(3))
ctx.generate(0, 44, 0, n, -1.0, 1.0, a.data_ptr()); ctx.generate(0, 45, 0, n, -1.0, 1.0, b.data_ptr()); ctx.generate(0, 46, 0, n, 0.0, 1.0, c.data_ptr())
ctx.synchronize()
C = qe.ColumnExpression


def best(run):
    t = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); out = run(); ctx.synchronize(); t = min(t, (time.perf_counter() - t0) * 1e3)
    return t, out


batch = qe.Batch.from_torch(ctx, [a, b, c])
for sel in (0.0, 1e-6, 0.001, 0.1, 0.5, 0.9, 0.999, 1.0):
    plan = lambda: list(qe.ProjectionExec([qe.SelectionExec([qe.ScanExec(ctx, [batch])], qe.LtExpression(C(2), qe.LiteralFloatExpression(sel)))],
                                          [qe.AddExpression(C(0), qe.MulExpression(C(1), C(2))), C(0)]).execute())[0]
    t, out = best(plan)
    print(f"selectivity {sel:<8} {t:8.2f} ms  rows out {out.num_rows}", flush=True)
for keys in (100, 3000, 100_000, 1_000_000, 50_000_000):
    for s_ in (1.1, 2.0):
        ctx.generate(2, 77, 0, n, keys, s_, k.data_ptr()); ctx.synchronize()
        kb = qe.Batch.from_torch(ctx, [k, c])
        line = f"zipf s={s_} over {keys:>9} keys:"
        for hint in (keys, 0):
            t, out = best(lambda: list(qe.AggregateExec([qe.ScanExec(ctx, [kb])], [C(0)], [qe.SumExpression(C(1)), qe.MinExpression(C(1)), qe.CountExpression(None)],
                                                        group_capacity_hint=hint).execute())[0])
            line += f"   hint {hint:>9}: {t:8.2f} ms ({out.num_rows} groups)"
        print(line, flush=True)
