"""The low-cardinality group-by (ten skewed groups, SUM and COUNT) with the keys as they are (0..9: dense bins) and
shifted out of the bins' range (hot-key registers and LDS table instead)."""
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
ctx.generate(3, 43, 0, n, 0, 0, k.data_ptr()); ctx.generate(0, 46, 0, n, 0.0, 1.0, v.data_ptr()); ctx.synchronize()
C = qe.ColumnExpression
for shift in (0, 1000, 1 << 40):
    kk = k + shift
    b = qe.Batch.from_torch(ctx, [kk, v])
    for rep in range(3):
        ctx.profile(1); ctx.profile_read()
        out = list(qe.AggregateExec([qe.ScanExec(ctx, [b])], [C(0)], [qe.SumExpression(C(1)), qe.CountExpression(None)], group_capacity_hint=64).execute())[0]
        ctx.synchronize()
        prof = ctx.profile_read(); ctx.profile(0)
    print("key shift", shift, {kname: round(r["ms"], 3) for kname, r in prof.items() if "aggregate" in kname}, "groups", out.num_rows, flush=True)
