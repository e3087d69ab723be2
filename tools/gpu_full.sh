#!/bin/bash
# GPU-box run at the end of a work block: tests, bench, rocprofv3 stats of the bench, PMC traffic of C5.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
timeout -k 10 600 python -u -m pytest $R/tests/ -m gpu -x -q --timeout 120 --timeout-method thread > $R/gpurun_out/pytest_gpu.log 2>&1 || { tail -30 $R/gpurun_out/pytest_gpu.log; exit 1; }
tail -2 $R/gpurun_out/pytest_gpu.log
timeout -k 10 500 python3 -u $R/bench.py > $R/gpurun_out/bench.json 2> $R/gpurun_out/bench.err || { tail -20 $R/gpurun_out/bench.err; exit 1; }
bash $R/tools/gpu_prof.sh
