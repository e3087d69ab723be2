#!/bin/bash
set -o pipefail
mkdir -p gpurun_out; : > gpurun_out/tune.log
run() { echo "== $*" >> gpurun_out/tune.log; env "$@" timeout -k 10 100 python -u tools/probe.py $CFG 256000000 3 2>&1 | grep -v amdgpu.ids | grep -o "'qe_spill_[a-z]*_kernel': {[^}]*}" >> gpurun_out/tune.log || exit 1; }
for b in 32 64 96 128 192 256; do CFG=c4 run QE_SPILL_BLOCKS=$b; done
cat gpurun_out/tune.log
