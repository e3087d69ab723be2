#!/bin/bash
# Timing experiments on the many-groups path: settings given as VAR=value words, one run per argument.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/c4
mkdir -p $O; rm -f $O/ablate.log
run() { echo "== $*" >> $O/ablate.log; env $* timeout -k 10 120 python -u $R/tools/probe.py ${PROBE_CFG:-c4} 256000000 3 2>&1 | grep -v amdgpu.ids >> $O/ablate.log || { tail -5 $O/ablate.log; exit 1; }; }
for a in "$@"; do run $a; done
python3 - <<PY
import re,ast
for blk in open("$O/ablate.log").read().split("== ")[1:]:
    head,rest=blk.split("\n",1)
    m=re.search(r"\{.*\}",rest)
    d=ast.literal_eval(m.group(0))
    ms=re.search(r"ms ([0-9.]+)",rest).group(1)
    print(head, "total=%.2f"%float(ms), " ".join(f"{k.split('qe_')[-1][:24]}={v['ms']/v['launches']:.3f}" for k,v in d.items() if 'init' not in k))
PY
