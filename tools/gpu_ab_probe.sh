#!/bin/bash
# A/B of two builds of the library on one box with tools/probe.py: <old.so> <config> <rows>, alternated three times.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/abp
mkdir -p $O; : > $O/ab.log
run() { # label, env...
  local label=$1; shift
  echo "== $label" >> $O/ab.log
  env "$@" timeout -k 10 120 python3 -u $R/tools/probe.py $CFG $ROWS 5 2>&1 | grep -v amdgpu.ids >> $O/ab.log || { tail -5 $O/ab.log; exit 1; }
}
OLD=$1; CFG=${2:-c2}; ROWS=${3:-256000000}
for i in 1 2 3; do
  run old QE_GPU_LIBRARY=$R/$OLD
  run new X=1
done
python3 - <<PY
import re,ast
for blk in open("$O/ab.log").read().split("== ")[1:]:
    head,rest=blk.split("\n",1)
    m=re.search(r"\{.*\}",rest)
    d=ast.literal_eval(m.group(0))
    ms=re.search(r"ms ([0-9.]+)",rest).group(1)
    print(head, "total=%.3f"%float(ms), " ".join(f"{k.split('qe_')[-1][:28]}={v['ms']/v['launches']:.4f}" for k,v in d.items() if 'init' not in k))
PY
