#!/usr/bin/env python3
"""What a nullable accumulator input costs with very many groups: the many-groups mode takes non-nullable inputs only,
so such an aggregate goes to the global table row by row.  SUM / MIN / MAX of one Float64 column by a million keys,
with and without a validity bitmap (2 % nulls)."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

qe = importlib.import_module("query-engine_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128_000_000
zipf = len(sys.argv) > 2 and sys.argv[2] == "zipf"
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
ctx = qe.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.set_flags(qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP | qe.FLAG_JIT_SYNC)
k = torch.empty(n, dtype=torch.int64, device=dev)
v = torch.empty(n, dtype=torch.float64, device=dev)
if zipf:
    ctx.generate(2, 77, 0, n, 1_000_000, 1.1, k.data_ptr())
else:
    ctx.generate(1, 99, 0, n, 0, 1_000_000, k.data_ptr())
ctx.generate(0, 44, 0, n, -1.0, 1.0, v.data_ptr())
ctx.synchronize()
bits = (torch.rand(n // 8 + 64, device=dev) < 0.85).to(torch.uint8) * 0xFF  # whole bytes of nulls: 15 % of the bytes ...
bits |= torch.randint(0, 256, (n // 8 + 64,), dtype=torch.uint8, device=dev) | 0x7F  # ... with one null in the others now and then
C = qe.ColumnExpression
for nullable in (False, True):
    cols = [{"length": n, "type": qe.TYPE_INT64, "values": k.data_ptr()},
            {"length": n, "type": qe.TYPE_FLOAT64, "values": v.data_ptr()}]
    if nullable:
        cols[1].update(validity=bits.data_ptr(), null_count=-1)
    batch = qe.Batch.from_device_columns(ctx, cols, keepalive=[k, v, bits])
    for rep in range(3):
        ctx.profile(True)
        t0 = time.perf_counter()
        out = list(qe.AggregateExec([qe.ScanExec(ctx, [batch])], [C(0)], [qe.SumExpression(C(1)), qe.MinExpression(C(1)), qe.MaxExpression(C(1)), qe.CountExpression(None)],
                                    group_capacity_hint=1_100_000).execute())[0]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        prof = ctx.profile_read()
        ctx.profile(False)
    print("nullable" if nullable else "all valid", "zipf" if zipf else "uniform", n, "rows:", round(dt * 1e3, 2), "ms", out.num_rows, "groups", {kk: round(x["ms"], 2) for kk, x in prof.items()}, flush=True)
