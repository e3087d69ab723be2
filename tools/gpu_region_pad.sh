#!/bin/bash
# The stride between the waves' regions of the many-groups log: whole steps (pad 0) against strides that spread the waves'
# current positions over the memory channels.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/region_pad
mkdir -p $O; : > $O/log.txt
for cfg in c5m c4; do
  for rows in 256000000 268435456; do
    for pad in 0 32 0 32 96 544 2080; do
      echo "== $cfg rows $rows pad $pad" >> $O/log.txt
      QE_SPILL_REGION_PAD=$pad timeout -k 10 120 python3 -u $R/tools/probe.py $cfg $rows 5 2>&1 | grep -v amdgpu.ids >> $O/log.txt || { tail -5 $O/log.txt; exit 1; }
    done
  done
done
python3 - <<PY
import re,ast
for blk in open("$O/log.txt").read().split("== ")[1:]:
    head,rest=blk.split("\n",1)
    m=re.search(r"\{.*\}",rest)
    d=ast.literal_eval(m.group(0))
    ms=re.search(r"ms ([0-9.]+)",rest).group(1)
    print(head, "total=%.3f"%float(ms), " ".join(f"{k.split('qe_')[-1][:22]}={v['ms']/v['launches']:.4f}" for k,v in d.items() if 'init' not in k and 'final' not in k))
PY
