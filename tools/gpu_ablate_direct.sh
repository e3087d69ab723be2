#!/bin/bash
# Where the many-groups first pass that partitions directly spends its time: a library built with QE_BUILD_ABLATE=1
# (ab/libqe_ablate.so) with parts of the kernel switched off (QE_DIRECT_ABLATE: 0x200 = 512 the rows go nowhere, 0x400 = 1024
# complete lines are not written to memory, 0x800 = 2048 no rounds either: columns and expressions only).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/ablate_direct
mkdir -p $O; : > $O/log.txt
for mask in ${MASKS:-0 1024 512 2048 0}; do
  echo "== c5m mask $mask" >> $O/log.txt
  QE_GPU_LIBRARY=$R/ab/libqe_ablate.so QE_DIRECT_ABLATE=$mask timeout -k 10 120 python3 -u $R/tools/probe.py c5m 256000000 5 2>&1 | grep -v amdgpu.ids >> $O/log.txt || { tail -5 $O/log.txt; exit 1; }
done
python3 - <<PY
import re,ast
for blk in open("$O/log.txt").read().split("== ")[1:]:
    head,rest=blk.split("\n",1)
    m=re.search(r"\{.*\}",rest)
    d=ast.literal_eval(m.group(0))
    print(head, " ".join(f"{k.split('qe_')[-1][:28]}={v['ms']/v['launches']:.4f}" for k,v in d.items() if 'direct' in k or 'partitioned' in k))
PY
