#!/bin/bash
# Start-of-session check on a GPU box: the parity tests, then the default bench line summarised.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s4
mkdir -p $O
timeout -k 10 900 python -u -m pytest $R/tests/ -m gpu -x -q --timeout 120 --timeout-method thread > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
bash $R/tools/gpu_bench_only.sh
