#!/bin/bash
# Parquet scan against the number of reader threads (QE_IO_THREADS), with the CPU allowance of the box beside it
# (cpu.max, and how often the group was throttled during each run).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pqt
mkdir -p $O
{ echo "nproc $(nproc)"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; grep -i cpus_allowed_list /proc/self/status; } > $O/cpu.txt 2>&1
cat $O/cpu.txt
for t in ${THREADS:-16 12 8 14 10 24}; do
  a=$(grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' ')
  QE_IO_THREADS=$t timeout -k 10 200 python3 -u $R/tools/parquet_probe.py 32000000 3 > $O/probe_$t.log 2>&1 || { tail -20 $O/probe_$t.log; exit 1; }
  b=$(grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' ')
  echo "threads $t: $(grep -E 'scan k,a,b,c|headline plan' $O/probe_$t.log | sed -E 's/\{.*//' | tr '\n' '|')"
  echo "   throttle before: $a after: $b"
done
