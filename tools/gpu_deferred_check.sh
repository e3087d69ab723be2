#!/bin/bash
# The deferred-row-count selection: its tests, the fuzzers in that mode, then the string shape's probe and the bench line.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/defer
mkdir -p $O
timeout -k 10 600 python -u -m pytest $R/tests/test_gpu_deferred_rows.py $R/tests/test_gpu_parity.py $R/tests/test_gpu_dataframe.py $R/tests/test_gpu_jit.py $R/tests/test_zz_no_alloc.py -x -q --timeout 200 --timeout-method thread > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
QE_FUZZ_DEFER_ROWS=1 timeout -k 10 400 python3 -u $R/tools/experiments/utf8_fuzz.py 11 60 > $O/fuzz.log 2>&1 || { tail -20 $O/fuzz.log; exit 1; }
tail -1 $O/fuzz.log
QE_FUZZ_DEFER_ROWS=1 timeout -k 10 400 python3 -u $R/tools/experiments/tree_fuzz.py 12 80 > $O/tree.log 2>&1 || { tail -20 $O/tree.log; exit 1; }
tail -1 $O/tree.log
QE_FUZZ_DEFER_ROWS=1 QE_FUZZ_JIT=1 timeout -k 10 400 python3 -u $R/tools/experiments/agg_fuzz.py 13 30 > $O/agg.log 2>&1 || { tail -20 $O/agg.log; exit 1; }
tail -1 $O/agg.log
timeout -k 10 300 python3 -u $R/tools/probe.py utf8 > $O/utf8.log 2>&1 || { tail -20 $O/utf8.log; exit 1; }
grep "^utf8" $O/utf8.log
bash $R/tools/gpu_bench_only.sh
