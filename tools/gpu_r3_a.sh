#!/bin/bash
# Check A: the tests added or changed last, each file under a time limit of its own.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3a
mkdir -p $O
run() {  # name, limit, pytest arguments
  local name=$1 limit=$2
  shift 2
  timeout -k 10 $limit python -u -m pytest "$@" -x -v --timeout 300 --timeout-method thread > $O/$name.log 2>&1 || { tail -60 $O/$name.log; exit 1; }
  tail -2 $O/$name.log
}
run exchange 400 $R/tests/test_gpu_exchange.py
run noalloc 300 $R/tests/test_zz_no_alloc.py
run detsum 300 $R/tests/test_gpu_deterministic_sum.py
run detsum_full 400 $R/tests/test_gpu_full_size.py -k deterministic
