#!/bin/bash
# Variants of the first pass that partitions directly, one probe run each (a million uniform keys, 256 M rows).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/direct_variants
mkdir -p $O; : > $O/log.txt
run() { echo "== $*" >> $O/log.txt; env $* timeout -k 10 120 python3 -u $R/tools/probe.py c5m ${ROWS:-256000000} 5 2>&1 | grep -v amdgpu.ids >> $O/log.txt || { tail -5 $O/log.txt; exit 1; }; }
for v in "$@"; do run $v; done
python3 - <<PY
import re,ast
for blk in open("$O/log.txt").read().split("== ")[1:]:
    head,rest=blk.split("\n",1)
    m=re.search(r"\{.*\}",rest)
    d=ast.literal_eval(m.group(0))
    ms=re.search(r" ms ([0-9.]+)",rest).group(1)
    print(head, "step", ms[:6], " ".join(f"{k.split('qe_')[-1][:28]}={v['ms']/v['launches']:.4f}" for k,v in d.items() if 'direct' in k or 'partitioned' in k))
PY
