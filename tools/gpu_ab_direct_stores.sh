#!/bin/bash
# The direct first pass's lines as streaming stores against ordinary ones, alternated on one box, with the headline
# kernel's launch time beside them (sessions differ in it, and the uniform-keys plan differs the other way).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/ab_direct_stores
mkdir -p $O
timeout -k 10 200 python3 -u $R/tools/probe.py c5 500000000 5 2 > $O/c5.log 2>&1 || { tail $O/c5.log; exit 1; }
grep "^c5" $O/c5.log | cut -c1-200
for i in 1 2 3; do
  for v in 1 0; do
    QE_DIRECT_STREAMING_STORES=$v timeout -k 10 200 python3 -u $R/tools/probe.py c5m 500000000 5 2 > $O/c5m_$v.log 2>&1 || { tail $O/c5m_$v.log; exit 1; }
    echo "streaming stores $v: $(grep '^c5m' $O/c5m_$v.log | cut -c1-200)"
  done
done
