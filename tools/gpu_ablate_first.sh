#!/bin/bash
# Where the many-groups first pass spends its time: a library built with QE_BUILD_ABLATE=1 (ab/libqe_ablate.so) with parts of
# the kernel switched off (QE_SPILL_ABLATE_FIRST: 16 no log stores, 32 no look at the table, 64 no partition histogram).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/ablate_first
mkdir -p $O; : > $O/log.txt
for cfg in c5m c4; do
  for mask in ${MASKS:-0 16 32 64 48 112 0}; do
    echo "== $cfg mask $mask" >> $O/log.txt
    QE_GPU_LIBRARY=$R/ab/libqe_ablate.so QE_SPILL_ABLATE_FIRST=$mask timeout -k 10 120 python3 -u $R/tools/probe.py $cfg 256000000 5 2>&1 | grep -v amdgpu.ids >> $O/log.txt || { tail -5 $O/log.txt; exit 1; }
  done
done
python3 - <<PY
import re,ast
for blk in open("$O/log.txt").read().split("== ")[1:]:
    head,rest=blk.split("\n",1)
    m=re.search(r"\{.*\}",rest)
    d=ast.literal_eval(m.group(0))
    print(head, " ".join(f"{k.split('qe_')[-1][:28]}={v['ms']/v['launches']:.4f}" for k,v in d.items() if 'spill_static' in k))
PY
