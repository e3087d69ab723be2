#!/usr/bin/env python3
"""Scan after scan of the benchmark's Parquet file (16 M rows, two row groups), each timed by itself: how much a scan's
time varies on a box, and whether the first scans after the file was written are the slow ones."""
import importlib
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

qe = importlib.import_module("query-engine_amd")
import pyarrow as pa  # noqa: E402
import pyarrow.parquet as pq  # noqa: E402

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
ctx = qe.Context(0, stream=torch.cuda.current_stream().cuda_stream)
n = 16_000_000
k = torch.randint(0, 10, (n,), dtype=torch.int64, device=dev)
a, b, c = (torch.rand(n, dtype=torch.float64, device=dev) for _ in range(3))
tbl = pa.table({name: pa.array(t.cpu().numpy()) for name, t in (("k", k), ("a", a), ("b", b), ("c", c))})
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "headline.parquet")
    pq.write_table(tbl, path, compression="none", use_dictionary=["k"], row_group_size=8_000_000)
    times = []
    for i in range(40):
        t0 = time.perf_counter()
        rows = 0
        for batch in qe.ParquetScanExec(ctx, path).execute():
            rows += batch.num_rows
        ctx.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    print("scan ms:", " ".join(f"{t:.1f}" for t in times))
    s = sorted(times[5:])
    print(f"after the first five: min {s[0]:.2f}  median {s[len(s) // 2]:.2f}  mean {sum(s) / len(s):.2f}  max {s[-1]:.2f}")
