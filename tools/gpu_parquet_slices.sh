#!/bin/bash
# The benchmark's Parquet leg against the size of a scan's first slices.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
for kb in ${SLICES:-4096 1024 512 2048 256}; do
  echo "first slices of $kb KiB"
  QE_PARQUET_FIRST_SLICE_KB=$kb bash $R/tools/gpu_parquet_bench.sh 2>&1 | sed -E 's/.*"ms_per_step": ([0-9.]+).*"scan_only_ms": ([0-9.]+), "scan_only_file_GBps": ([0-9.]+).*/   plan \1 ms, scan \2 ms = \3 GB\/s/' || exit 1
done
