#!/bin/bash
# GPU-box check: parity tests, then one probe per configuration.  Every GPU step has its own time limit.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/ -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -5 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
for cfg in p c2 c3 c5 c4; do
  timeout -k 10 120 python -u tools/probe.py $cfg ${PROBE_ROWS:-256000000} 3 >> gpurun_out/probe.log 2>&1 || { tail -5 gpurun_out/probe.log; exit 1; }
done
cat gpurun_out/probe.log
