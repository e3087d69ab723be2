#!/bin/bash
# Workgroups per compute unit sweep for one probe configuration.  usage: CFG=c5 BLOCKS="2 3 4" gpu_tune4.sh
set -o pipefail
mkdir -p gpurun_out; : > gpurun_out/tune4.log
for round in 1 2; do
for b in ${BLOCKS:-2 3 4 5}; do
  echo "== blocks/CU $b" >> gpurun_out/tune4.log
  QE_AGGREGATE_BLOCKS_PER_CU=$b timeout -k 10 100 python3 -u tools/probe.py ${CFG:-c3} 256000000 10 2>&1 | grep -v amdgpu.ids | cut -c1-200 >> gpurun_out/tune4.log || exit 1
done
done
cat gpurun_out/tune4.log
