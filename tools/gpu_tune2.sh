#!/bin/bash
set -o pipefail
mkdir -p gpurun_out; : > gpurun_out/tune.log
run() { echo "== $*" >> gpurun_out/tune.log; env "$@" timeout -k 10 100 python -u tools/probe.py $CFG 256000000 10 2>&1 | grep -v amdgpu.ids | cut -c1-170 >> gpurun_out/tune.log || exit 1; }
for g in 512 640 768 896 960 1000 1016 1024 1032 1088 1152 1280 1536 1792 2048; do CFG=c5 run QE_AGGREGATE_GRID=$g; done
cat gpurun_out/tune.log
