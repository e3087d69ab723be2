#!/bin/bash
# The many-groups first pass that partitions directly: parity tests of the aggregate paths, then a million uniform keys
# and C4 at 500 M rows in two batches and in one, with the direct first pass and without (QE_AGGREGATE_DIRECT=2).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s4direct
mkdir -p $O
rm -f $O/probe.log
if [ "${SKIP_TESTS:-0}" != 1 ]; then
timeout -k 10 600 python -u -m pytest $R/tests/test_gpu_parity.py $R/tests/test_gpu_full_size.py $R/tests/test_gpu_specialized.py $R/tests/test_gpu_jit.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
fi
for cfg in c5m c4; do
  for nb in 2 1; do
    for d in 1 2; do
      echo "== $cfg batches=$nb QE_AGGREGATE_DIRECT=$d" >> $O/probe.log
      QE_SPILL_VERBOSE=${VERBOSE:-} QE_AGGREGATE_DIRECT=$d timeout -k 10 120 python -u $R/tools/probe.py $cfg 500000000 5 $nb >> $O/probe.log 2>&1 || { tail -5 $O/probe.log; exit 1; }
    done
  done
done
grep -v amdgpu.ids $O/probe.log
