#!/bin/bash
# Check F: the many-groups path's parity tests, then its probes with the one-pass scatter on and off.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3f
mkdir -p $O
timeout -k 10 600 python -u -m pytest $R/tests/test_gpu_specialized.py $R/tests/test_gpu_trees.py -x -q --timeout 200 --timeout-method thread > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 500 python -u -m pytest $R/tests/test_gpu_full_size.py -k "c4" -x -v -s --timeout 400 --timeout-method thread > $O/pytest_full.log 2>&1 || { tail -40 $O/pytest_full.log; exit 1; }
tail -2 $O/pytest_full.log
: > $O/probe.log
run() { echo "== $*" >> $O/probe.log; env "$@" timeout -k 10 120 python -u $R/tools/probe.py $CFG 256000000 3 >> $O/probe.log 2>&1 || { tail -5 $O/probe.log; exit 1; }; }
for CFG in c4 c5m; do
  run QE_SPILL_ONEPASS=2
  run QE_X=0
  run QE_SPILL_ONEPASS=2
  run QE_X=0
done
grep -v amdgpu.ids $O/probe.log
