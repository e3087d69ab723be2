#!/bin/bash
# A/B of two builds of the library on one box: headline bench alternated, three rounds.
# usage: gpu_ab.sh <old.so> ; the tree's own build is B.  Extra environment for B in B_ENV.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; : > $R/gpurun_out/ab.log
run() { # label, env...
  local label=$1; shift
  env "$@" timeout -k 10 200 python3 -u $R/bench.py --no-cpu --no-extra > $R/gpurun_out/ab_tmp.json 2> $R/gpurun_out/ab_tmp.err || { tail -5 $R/gpurun_out/ab_tmp.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('$R/gpurun_out/ab_tmp.json').read().strip().splitlines()[-1])
print('$label', round(d['roofline']['avg_launch_ms'],4), round(d['ms_per_step'],3))" >> $R/gpurun_out/ab.log
}
for i in 1 2 3; do
  run old QE_GPU_LIBRARY=$R/$1
  run new X=1
  run new_nobins QE_AGGREGATE_DENSE_BINS=2
done
cat $R/gpurun_out/ab.log
