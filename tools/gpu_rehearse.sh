#!/bin/bash
# GPU-box rehearsal of the bench's N = 2 code paths on one GPU (ranks share the device, gloo exchange).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
for G in 10 1000000; do
QE_BENCH_BACKEND=gloo timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  $R/bench.py --gpus 2 --steps 3 --warmup 1 --rows 400000000 --groups $G > $R/gpurun_out/bench_n2_gloo_$G.json 2> $R/gpurun_out/bench_n2_gloo_$G.err || { tail -30 $R/gpurun_out/bench_n2_gloo_$G.err; exit 1; }
python3 -c "
import json,sys
d=json.loads(open('$R/gpurun_out/bench_n2_gloo_$G.json').read().strip().splitlines()[-1])
print('N=2 G=$G', d['value'], d['ms_per_step'], d['result_groups'], d['result_count_total'], d['config']['merge'])"
timeout -k 10 300 python3 -u $R/bench.py --steps 3 --warmup 1 --rows 400000000 --groups $G --no-extra --no-cpu > $R/gpurun_out/bench_n1_$G.json 2>> $R/gpurun_out/bench_n2_gloo_$G.err || exit 1
python3 -c "
import json,sys
d=json.loads(open('$R/gpurun_out/bench_n1_$G.json').read().strip().splitlines()[-1])
print('N=1 G=$G', d['value'], d['ms_per_step'], d['result_groups'], d['result_count_total'])"
done
# the same code paths over real RCCL with a single rank (QE_BENCH_FORCE_DIST): block all-gather and key-partitioned all-to-all
for G in 10 1000000; do
QE_BENCH_FORCE_DIST=1 timeout -k 10 300 python3 -u $R/bench.py --steps 3 --warmup 1 --rows 400000000 --groups $G --no-extra --no-cpu > $R/gpurun_out/bench_rccl1_$G.json 2> $R/gpurun_out/bench_rccl1_$G.err || { tail -30 $R/gpurun_out/bench_rccl1_$G.err; exit 1; }
python3 -c "
import json
d=json.loads(open('$R/gpurun_out/bench_rccl1_$G.json').read().strip().splitlines()[-1])
print('RCCL world=1 G=$G', d['value'], d['ms_per_step'], d['result_groups'], d['result_count_total'], d['config']['merge'])"
done
