#!/bin/bash
# A/B of one launch over a plan's batches against a launch per batch for the few-groups aggregate kernels (experiment:
# ab/libqe_new.so built with the batch loop in qe_aggregate_static, QE_MERGE_ALL=1 turns the merge on).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/ab_merge
mkdir -p $O; : > $O/log.txt
for i in 1 2 3 4 5 6; do
  for v in "old ab/libqe_old.so 0" "new ab/libqe_new.so 0" "newmerged ab/libqe_new.so 1"; do
    set -- $v
    for cfg in "c5 1000000000 4" "c3 500000000 2"; do
      set -- $v $cfg
      if [ "$3" = 1 ]; then export QE_MERGE_ALL=1; else unset QE_MERGE_ALL; fi
      echo -n "$1 $4 " >> $O/log.txt
      QE_GPU_LIBRARY=$R/$2 timeout -k 10 120 python3 -u $R/tools/probe.py $4 $5 5 $6 2>&1 | grep -v amdgpu | awk '{print $5}' >> $O/log.txt || exit 1
    done
  done
done
python3 - <<PY
import collections
d=collections.defaultdict(list)
for l in open("$O/log.txt"):
    a=l.split()
    if len(a)==3: d[(a[0],a[1])].append(float(a[2]))
for k,v in sorted(d.items()):
    v.sort()
    print(k, "median %.4f min %.4f max %.4f n %d"%(v[len(v)//2], v[0], v[-1], len(v)))
PY
