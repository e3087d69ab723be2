#!/bin/bash
# Parquet scan leaf: its tests, the fuzzer, then the probe (host timings per chunk with QE_PARQUET_TRACE=1 in a second run).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pq
mkdir -p $O
timeout -k 10 500 python -u -m pytest $R/tests/test_gpu_parquet.py $R/tests/test_gpu_dataframe.py -x -q --timeout 200 --timeout-method thread > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 400 python3 -u $R/tools/experiments/parquet_fuzz.py 7 60 > $O/fuzz.log 2>&1 || { tail -20 $O/fuzz.log; exit 1; }
tail -1 $O/fuzz.log
timeout -k 10 400 python3 -u $R/tools/parquet_probe.py 32000000 3 > $O/probe.log 2>&1 || { tail -20 $O/probe.log; exit 1; }
grep -v amdgpu.ids $O/probe.log
QE_PARQUET_TRACE=1 timeout -k 10 400 python3 -u $R/tools/parquet_probe.py 32000000 1 > $O/trace.log 2>&1 || { tail -20 $O/trace.log; exit 1; }
grep "row group\|column a\|column b" $O/trace.log | sed -n "1,24p"
