#!/bin/bash
# The bench's test_filter-shaped entry under the values of a tuning knob: <ENV_NAME> <value>...
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/utf8v
mkdir -p $O
NAME=$1; shift
for v in "$@"; do
  env $NAME=$v timeout -k 10 300 python3 -u - > $O/bench_$v.log 2>&1 <<PY || { tail -20 $O/bench_$v.log; exit 1; }
import sys, importlib, json
sys.path.insert(0, "$R")
import torch
import bench
qe = importlib.import_module("query-engine_amd")
ctx = qe.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.set_flags(qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP | qe.FLAG_JIT_SYNC)
torch.cuda.set_device(0)
for rep in range(2):
    d = bench.utf8_filter_config(qe, ctx, torch, torch.device("cuda", 0))
    print("$NAME=$v", round(d['ms_per_step'], 2), {k: round(x, 2) for k, x in d['kernels_ms_per_step'].items()})
PY
  grep "^$NAME" $O/bench_$v.log
done
