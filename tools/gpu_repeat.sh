#!/bin/bash
set -o pipefail
mkdir -p gpurun_out; : > gpurun_out/repeat.log
for i in 1 2 3; do for c in $CFGS; do timeout -k 10 100 python -u tools/probe.py $c 256000000 10 2>&1 | grep -v amdgpu.ids | cut -c1-200 >> gpurun_out/repeat.log || exit 1; done; done
cat gpurun_out/repeat.log
