#!/bin/bash
# The many-groups aggregate: its tests, then C4 and the uniform-keys plan as the bench runs them.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/mg
mkdir -p $O
timeout -k 10 900 python -u -m pytest $R/tests/test_gpu_many_groups_batches.py $R/tests/test_gpu_direct_partition.py $R/tests/test_gpu_specialized.py $R/tests/test_gpu_parity.py $R/tests/test_gpu_key_verify.py $R/tests/test_gpu_trees.py $R/tests/test_gpu_full_size.py -x -q --timeout 200 --timeout-method thread > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for cfg in c4 c5m c4 c5m; do
  timeout -k 10 300 python3 -u $R/tools/probe.py $cfg 500000000 5 2 > $O/$cfg.log 2>&1 || { tail -20 $O/$cfg.log; exit 1; }
  grep "^$cfg" $O/$cfg.log | cut -c1-120
done
