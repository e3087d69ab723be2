"""One aggregation with the launch profile printed: which kernels ran, for how long."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, importlib
qe = importlib.import_module("query-engine_amd")
n, groups, hint = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
torch.cuda.set_device(0)
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctx = qe.Context(0, stream=s.cuda_stream)
dev = torch.device("cuda", 0)
k = torch.empty(n, dtype=torch.int64, device=dev); v = torch.empty(n, dtype=torch.float64, device=dev)
ctx.generate(0, 44, 0, n, 0.0, 1.0, v.data_ptr())
lo = int(os.environ.get('KEY_LO', '0'))
ctx.generate(1, 43, 0, n, lo, lo + groups, k.data_ptr())
if os.environ.get('KEY_MUL'):
    k.mul_(int(os.environ['KEY_MUL']))
ctx.synchronize()
C = qe.ColumnExpression
b = qe.Batch.from_torch(ctx, [k, v])
for rep in range(2):
    ctx.profile(1); ctx.profile_read()
    t0 = time.perf_counter()
    out = list(qe.AggregateExec([qe.ScanExec(ctx, [b])], [C(0)], [qe.SumExpression(C(1)), qe.MaxExpression(C(1)), qe.CountExpression(C(1))], group_capacity_hint=hint).execute())[0]
    ctx.synchronize()
    print("total ms", round((time.perf_counter() - t0) * 1e3, 2), "groups", out.num_rows)
    for name, rec in ctx.profile_read().items():
        print("   ", name, rec)
    ctx.profile(0)
