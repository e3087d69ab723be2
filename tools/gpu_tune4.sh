#!/bin/bash
# Workgroups per compute unit for the two-column group-by with the dense bins on.
set -o pipefail
mkdir -p gpurun_out; : > gpurun_out/tune4.log
for b in ${BLOCKS:-2 3 4 5}; do
  echo "== blocks/CU $b" >> gpurun_out/tune4.log
  QE_AGGREGATE_BLOCKS_PER_CU=$b timeout -k 10 100 python3 -u tools/probe.py c3 256000000 10 2>&1 | grep -v amdgpu.ids | cut -c1-200 >> gpurun_out/tune4.log || exit 1
done
cat gpurun_out/tune4.log
