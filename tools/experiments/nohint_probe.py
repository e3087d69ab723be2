"""Time of a million-group aggregation with and without a group-count hint from the caller."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, importlib
qe = importlib.import_module("query-engine_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256_000_000
torch.cuda.set_device(0)
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctx = qe.Context(0, stream=s.cuda_stream)
dev = torch.device("cuda", 0)
k = torch.empty(n, dtype=torch.int64, device=dev); v = torch.empty(n, dtype=torch.float64, device=dev)
ctx.generate(2, 43, 0, n, 1_000_000, 1.1, k.data_ptr())   # zipf keys: a million of them, s = 1.1
ctx.generate(0, 44, 0, n, 0.0, 1.0, v.data_ptr())
ctx.synchronize()
C = qe.ColumnExpression
b = qe.Batch.from_torch(ctx, [k, v])
for hint in (2_000_000, 0, 2_000_000, 0):
    t0 = time.perf_counter()
    out = list(qe.AggregateExec([qe.ScanExec(ctx, [b])], [C(0)], [qe.SumExpression(C(1)), qe.MinExpression(C(1)), qe.MaxExpression(C(1)), qe.AvgExpression(C(1))], group_capacity_hint=hint).execute())[0]
    ctx.synchronize()
    print("hint", hint, "groups", out.num_rows, "ms", round((time.perf_counter() - t0) * 1e3, 2), flush=True)
