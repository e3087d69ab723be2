#!/bin/bash
# The C5 plan over more uniform keys than 512 partitions are sized for: the partitions the hint asks for (1024, two-level
# scatter) against 512 with what does not fit a partition's LDS table going to the global table.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/bitscap
mkdir -p $O
for setting in "1500000 1650000" "2000000 2200000" "1000000 2000000" "2500000 2750000"; do
  set -- $setting
  for cap in 0 9; do
    QE_PROBE_KEYS=$1 QE_PROBE_HINT=$2 QE_SPILL_BITS_CAP=$cap timeout -k 10 200 python3 -u $R/tools/probe.py c5m 500000000 3 2 > $O/k$1_h$2_cap$cap.log 2>&1 || { tail -20 $O/k$1_h$2_cap$cap.log; exit 1; }
    echo "keys $1 hint $2 cap $cap: $(grep '^c5m' $O/k$1_h$2_cap$cap.log | cut -c1-100)"
  done
done
