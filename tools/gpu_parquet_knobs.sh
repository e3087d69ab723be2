#!/bin/bash
# The benchmark's Parquet leg under a few settings (slice size, reader threads, where the pinned memory lies).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() {
  echo "$1"
  env $1 bash $R/tools/gpu_parquet_bench.sh 2>&1 | sed -E 's/.*"ms_per_step": ([0-9.]+).*"scan_only_ms": ([0-9.]+), "scan_only_file_GBps": ([0-9.]+).*/   plan \1 ms, scan \2 ms = \3 GB\/s/' || exit 1
}
numactl -H 2>/dev/null | sed -n "1,12p"
cat /sys/class/drm/card*/device/numa_node 2>/dev/null | tr '\n' ' '; echo
for setting in ${SETTINGS:-QE_X=0}; do run "$setting"; done
