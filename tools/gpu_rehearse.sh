#!/bin/bash
# GPU-box rehearsal of the bench's N = 2 code path on one GPU (ranks share the device, gloo exchange).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
QE_BENCH_BACKEND=gloo timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  $R/bench.py --gpus 2 --steps 3 --warmup 1 --rows 400000000 > $R/gpurun_out/bench_n2_gloo.json 2> $R/gpurun_out/bench_n2_gloo.err || { tail -30 $R/gpurun_out/bench_n2_gloo.err; exit 1; }
cat $R/gpurun_out/bench_n2_gloo.json
timeout -k 10 300 python3 -u $R/bench.py --steps 3 --warmup 1 --rows 400000000 --no-extra --no-cpu > $R/gpurun_out/bench_n1_400m.json 2>> $R/gpurun_out/bench_n2_gloo.err && cat $R/gpurun_out/bench_n1_400m.json
