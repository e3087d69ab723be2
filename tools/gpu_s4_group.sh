#!/bin/bash
# Many-groups mode with the logs of a plan's batches partitioned together (QE_SPILL_GROUP=0: batch by batch): parity
# tests of the aggregate paths, then C4 / a million uniform keys at 500 M rows in two batches both ways.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s4group
mkdir -p $O
timeout -k 10 600 python -u -m pytest $R/tests/test_gpu_parity.py $R/tests/test_gpu_full_size.py $R/tests/test_gpu_specialized.py $R/tests/test_gpu_jit.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for cfg in c4 c5m; do
  for g in 1 0; do
    echo "== $cfg QE_SPILL_GROUP=$g" >> $O/probe.log
    QE_SPILL_GROUP=$g timeout -k 10 120 python -u $R/tools/probe.py $cfg 500000000 5 2 >> $O/probe.log 2>&1 || { tail -5 $O/probe.log; exit 1; }
  done
done
cat $O/probe.log
