#!/bin/bash
# String-path check: the tests that move strings (both row-count modes), the utf8 fuzzer, then the bench's test_filter-shaped entry.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/utf8r3
mkdir -p $O
timeout -k 10 400 python -u -m pytest $R/tests/test_gpu_parity.py $R/tests/test_gpu_deferred_rows.py $R/tests/test_gpu_dataframe.py $R/tests/test_gpu_parquet.py -x -q --timeout 200 --timeout-method thread > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
QE_FUZZ_JIT=1 timeout -k 10 400 python3 -u $R/tools/experiments/utf8_fuzz.py 11 80 > $O/fuzz.log 2>&1 || { tail -20 $O/fuzz.log; exit 1; }
tail -1 $O/fuzz.log
for flags in 0 32; do
  timeout -k 10 300 python3 -u - > $O/bench_$flags.log 2>&1 <<PY || { tail -20 $O/bench_$flags.log; exit 1; }
import sys, importlib, json
sys.path.insert(0, "$R")
import torch
import bench
qe = importlib.import_module("query-engine_amd")
ctx = qe.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.set_flags(qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP | qe.FLAG_JIT_SYNC | $flags)
torch.cuda.set_device(0)
print(json.dumps(bench.utf8_filter_config(qe, ctx, torch, torch.device("cuda", 0))))
PY
  python3 -c "
import json
d=json.loads(open('$O/bench_$flags.log').read().strip().splitlines()[-1])
print('flags $flags: utf8 test_filter shape, 100 M rows:', round(d['ms_per_step'],2), round(d['frac_hbm_peak_end_to_end'],3), {k:round(v,2) for k,v in d['kernels_ms_per_step'].items()})"
done
