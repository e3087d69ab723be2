"""Heavily skewed keys (zipf s = 2: three fifths of the rows on one key) on the general and on the compiled kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, importlib
qe = importlib.import_module("query-engine_amd")
n = 64_000_000
torch.cuda.set_device(0)
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctx = qe.Context(0, stream=s.cuda_stream)
dev = torch.device("cuda", 0)
k = torch.empty(n, dtype=torch.int64, device=dev); v = torch.empty(n, dtype=torch.float64, device=dev)
ctx.generate(0, 46, 0, n, 0.0, 1.0, v.data_ptr())
C = qe.ColumnExpression
for keys, sk in ((100, 2.0), (100, 1.1), (3000, 2.0)):
    ctx.generate(2, 77, 0, n, keys, sk, k.data_ptr()); ctx.synchronize()
    top = torch.unique(k, return_counts=True)[1].sort(descending=True).values[:5].cpu().numpy() / n
    b = qe.Batch.from_torch(ctx, [k, v])
    for rep in range(2):
        ctx.profile(1); ctx.profile_read()
        out = list(qe.AggregateExec([qe.ScanExec(ctx, [b])], [C(0)], [qe.SumExpression(C(1)), qe.MinExpression(C(1)), qe.CountExpression(None)], group_capacity_hint=keys).execute())[0]
        ctx.synchronize()
        prof = ctx.profile_read(); ctx.profile(0)
    print(f"keys {keys} s={sk} top shares {top.round(3)} min key {int(k.min())}", {kn: round(r["ms"], 3) for kn, r in prof.items() if "aggregate" in kn}, flush=True)
