#!/bin/bash
# A/B of two builds of the library on one box over the bench's configurations: alternated, three rounds.
# usage: gpu_ab2.sh <old.so> ; the tree's own build is B.  Variants of B: "label:ENV=value" words in B_VARIANTS.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; : > $R/gpurun_out/ab2.log
run() { # label, env...
  local label=$1; shift
  env "$@" timeout -k 10 300 python3 -u $R/bench.py --no-cpu > $R/gpurun_out/ab_tmp.json 2> $R/gpurun_out/ab_tmp.err || { tail -5 $R/gpurun_out/ab_tmp.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('$R/gpurun_out/ab_tmp.json').read().strip().splitlines()[-1])
o=d.get('other_configs',{})
print('$label', 'C5', round(d['roofline']['avg_launch_ms'],4), ' '.join(k.split('_')[0]+('i32' if 'int32' in k else '')+('1M' if '1M' in k else '')+' '+str(round(v.get('avg_launch_ms',0),4)) for k,v in o.items() if 'avg_launch_ms' in v))" >> $R/gpurun_out/ab2.log
}
for i in 1 2 3; do
  run old QE_GPU_LIBRARY=$R/$1
  run new X=1
  for v in $B_VARIANTS; do run ${v%%:*} ${v#*:}; done
done
cat $R/gpurun_out/ab2.log
