#!/usr/bin/env python3
"""Times the headline plan over zipf keys with more groups than 512 partitions' tables hold, and torch's reductions over
the same rows (the reference the tests check against): which of the two a slow test waits for."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

qe = importlib.import_module("query-engine_amd")
n, n_keys = 64_000_003, int(sys.argv[1]) if len(sys.argv) > 1 else 2_400_000
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
ctx = qe.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.set_flags(qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP | qe.FLAG_JIT_SYNC)
k = torch.empty(n, dtype=torch.int64, device=dev)
a, b, c = (torch.empty(n, dtype=torch.float64, device=dev) for _ in range(3))
ctx.generate(2, 177, 0, n, n_keys, 1.1, k.data_ptr())
ctx.generate(0, 144, 0, n, -1.0, 1.0, a.data_ptr())
ctx.generate(0, 145, 0, n, -1.0, 1.0, b.data_ptr())
ctx.generate(0, 146, 0, n, 0.0, 1.0, c.data_ptr())
ctx.synchronize()
C = qe.ColumnExpression
for rep in range(2):
    aggs = [qe.SumExpression(C(1)), qe.CountExpression(None), qe.AvgExpression(C(1)), qe.MinExpression(C(1)), qe.MaxExpression(C(1))]
    sel = qe.SelectionExec([qe.ScanExec(ctx, [qe.Batch.from_torch(ctx, [k, a, b, c])])], qe.GtExpression(C(3), qe.LiteralFloatExpression(0.5)))
    proj = qe.ProjectionExec([sel], [C(0), qe.AddExpression(C(1), qe.MulExpression(C(2), C(3)))])
    ctx.profile(True)
    t0 = time.perf_counter()
    out = list(qe.AggregateExec([proj], [C(0)], aggs, group_capacity_hint=n_keys + n_keys // 10).execute())[0]
    torch.cuda.synchronize()
    print("engine", round((time.perf_counter() - t0) * 1e3, 2), "ms", out.num_rows, {kk: round(v["ms"], 2) for kk, v in ctx.profile_read().items()}, flush=True)
    ctx.profile(False)
keep = c > 0.5
kk, vv = k[keep], (a + b * c)[keep]
t0 = time.perf_counter()
uk, inv, cnt = torch.unique(kk, return_inverse=True, return_counts=True)
torch.cuda.synchronize()
print("torch unique", round(time.perf_counter() - t0, 2), "s", flush=True)
t0 = time.perf_counter()
s = torch.zeros(uk.shape[0], dtype=torch.float64, device=dev).scatter_add_(0, inv, vv)
torch.cuda.synchronize()
print("torch scatter_add", round(time.perf_counter() - t0, 2), "s", flush=True)
