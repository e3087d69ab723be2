#!/bin/bash
set -o pipefail
mkdir -p gpurun_out; : > gpurun_out/tune.log
timeout -k 10 600 python -u -m pytest tests/ -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/pytest_gpu.log
run() { echo "== $*" >> gpurun_out/tune.log; env "$@" timeout -k 10 100 python -u tools/probe.py $CFG 256000000 5 2>&1 | grep -v amdgpu.ids >> gpurun_out/tune.log || exit 1; }
for c in ${CFGS:-c3 c5 c4}; do CFG=$c run X=1; done
cat gpurun_out/tune.log
