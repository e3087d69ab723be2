#!/bin/bash
# Check C: the deferred row count's tests, then the selection tests of the parity suite (unchanged paths).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c
mkdir -p $O
timeout -k 10 400 python -u -m pytest $R/tests/test_gpu_deferred_rows.py -x -v --timeout 200 --timeout-method thread > $O/deferred.log 2>&1 || { tail -60 $O/deferred.log; exit 1; }
tail -2 $O/deferred.log
timeout -k 10 600 python -u -m pytest $R/tests/test_gpu_parity.py -x -q --timeout 200 --timeout-method thread > $O/parity.log 2>&1 || { tail -60 $O/parity.log; exit 1; }
tail -2 $O/parity.log
