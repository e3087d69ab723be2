#!/bin/bash
# Check D: hashed group keys' equality check, then every test that groups by strings or tuples.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3d
mkdir -p $O
timeout -k 10 400 python -u -m pytest $R/tests/test_gpu_key_verify.py -x -v --timeout 200 --timeout-method thread > $O/verify.log 2>&1 || { tail -60 $O/verify.log; exit 1; }
tail -2 $O/verify.log
timeout -k 10 900 python -u -m pytest $R/tests/test_gpu_parity.py $R/tests/test_gpu_trees.py $R/tests/test_gpu_dataframe.py $R/tests/test_gpu_parquet.py -x -q --timeout 300 --timeout-method thread > $O/others.log 2>&1 || { tail -60 $O/others.log; exit 1; }
tail -2 $O/others.log
