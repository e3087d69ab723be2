"""Aggregation time against the number of groups, with the caller's hint and without: looks for cliffs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, importlib
qe = importlib.import_module("query-engine_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64_000_000
nullable = len(sys.argv) > 2 and sys.argv[2] == "nullable"
torch.cuda.set_device(0)
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctx = qe.Context(0, stream=s.cuda_stream)
dev = torch.device("cuda", 0)
k = torch.empty(n, dtype=torch.int64, device=dev); v = torch.empty(n, dtype=torch.float64, device=dev)
ctx.generate(0, 44, 0, n, 0.0, 1.0, v.data_ptr())
valid = (torch.rand(n, device=dev) > 0.1) if nullable else None
C = qe.ColumnExpression
for groups in [int(x) for x in os.environ.get('SWEEP_GROUPS', '10,1000,30000,60000,70000,300000,1000000,10000000').split(',')]:
    ctx.generate(1, 43, 0, n, 0, groups, k.data_ptr())
    ctx.synchronize()
    if nullable:
        import numpy as np
        cols = [qe.Column(qe.TYPE_INT64, k.cpu().numpy()), qe.Column(qe.TYPE_FLOAT64, v.cpu().numpy(), valid.cpu().numpy())]
        b = qe.Batch.from_numpy(ctx, cols).to_device()
    else:
        b = qe.Batch.from_torch(ctx, [k, v])
    line = f"groups {groups:>9}"
    for hint in [int(x) for x in os.environ.get('SWEEP_HINTS', '-1,0').split(',')]:
        hint = groups if hint < 0 else hint
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            out = list(qe.AggregateExec([qe.ScanExec(ctx, [b])], [C(0)], [qe.SumExpression(C(1)), qe.MaxExpression(C(1)), qe.CountExpression(C(1))], group_capacity_hint=hint).execute())[0]
            ctx.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3)
        line += f"   hint {hint:>9}: {best:8.2f} ms ({out.num_rows} groups)"
    print(line, flush=True)
