#!/bin/bash
# GPU-box run for the many-groups path: its parity tests, then the C4 / C5-1M probes under a few settings.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/c4
mkdir -p $O
timeout -k 10 600 python -u -m pytest $R/tests/test_gpu_specialized.py $R/tests/test_gpu_trees.py $R/tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 400 python -u -m pytest $R/tests/test_gpu_full_size.py -k "c4 or c3" -x -v -s --timeout 300 --timeout-method thread > $O/pytest_full.log 2>&1 || { tail -40 $O/pytest_full.log; exit 1; }
tail -2 $O/pytest_full.log
run() { echo "== $*" >> $O/probe.log; env "$@" timeout -k 10 120 python -u $R/tools/probe.py c4 256000000 3 >> $O/probe.log 2>&1 || { tail -5 $O/probe.log; exit 1; }; }
run QE_X=0
run QE_AGGREGATE_WIDE=2
run QE_SPILL_BITS=10
run QE_SPILL_BITS=11
run QE_SPILL_BLOCKS=512
cat $O/probe.log
