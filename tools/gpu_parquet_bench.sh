#!/bin/bash
# The Parquet leg of the benchmark alone (16 M rows of the headline table, two row groups), and the probe beside it.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pqb
mkdir -p $O
timeout -k 10 300 python3 -u - > $O/bench_parquet.log 2>&1 <<PY || { tail -20 $O/bench_parquet.log; exit 1; }
import importlib, json, sys
sys.path.insert(0, "$R")
import torch
import bench
qe = importlib.import_module("query-engine_amd")
torch.cuda.set_device(0)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
ctx = qe.Context(0, stream=stream.cuda_stream)
n = 16_000_000
dev = torch.device("cuda", 0)
k = torch.randint(0, 10, (n,), dtype=torch.int64, device=dev)
a, b, c = (torch.rand(n, dtype=torch.float64, device=dev) for _ in range(3))
for _ in range(3):
    print(json.dumps(bench.parquet_config(qe, ctx, torch, k, a, b, c, n)), flush=True)
PY
grep -v amdgpu.ids $O/bench_parquet.log
timeout -k 10 300 python3 -u $R/tools/parquet_probe.py 32000000 3 > $O/probe.log 2>&1 || { tail -20 $O/probe.log; exit 1; }
grep -E "^scan k,a,b,c|^headline" $O/probe.log | sed -E 's/\{.*//'
