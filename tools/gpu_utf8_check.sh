#!/bin/bash
# String selections, waiting and not: the tests that select string columns, the string fuzzer in both modes, then the
# string shape of the bench under both sets of flags.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/utf8chk
mkdir -p $O
timeout -k 10 600 python -u -m pytest $R/tests/test_gpu_parity.py $R/tests/test_gpu_dataframe.py $R/tests/test_gpu_parquet.py $R/tests/test_gpu_deferred_rows.py $R/tests/test_gpu_key_verify.py $R/tests/test_gpu_trees.py -x -q --timeout 200 --timeout-method thread > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 400 python3 -u $R/tools/experiments/utf8_fuzz.py 31 80 > $O/fuzz.log 2>&1 || { tail -20 $O/fuzz.log; exit 1; }
tail -1 $O/fuzz.log
QE_FUZZ_JIT=1 timeout -k 10 400 python3 -u $R/tools/experiments/utf8_fuzz.py 32 40 > $O/fuzz_jit.log 2>&1 || { tail -20 $O/fuzz_jit.log; exit 1; }
tail -1 $O/fuzz_jit.log
QE_FUZZ_DEFER_ROWS=1 timeout -k 10 400 python3 -u $R/tools/experiments/utf8_fuzz.py 33 60 > $O/fuzz_defer.log 2>&1 || { tail -20 $O/fuzz_defer.log; exit 1; }
tail -1 $O/fuzz_defer.log
timeout -k 10 400 python3 -u $R/tools/experiments/dataframe_fuzz.py 34 60 > $O/df.log 2>&1 || { tail -20 $O/df.log; exit 1; }
tail -1 $O/df.log
timeout -k 10 300 python3 -u - > $O/bench_utf8.log 2>&1 <<PY || { tail -20 $O/bench_utf8.log; exit 1; }
import importlib, json, sys
sys.path.insert(0, "$R")
import torch
import bench
qe = importlib.import_module("query-engine_amd")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
ctx = qe.Context(0, stream=torch.cuda.current_stream().cuda_stream)
for flags, name in ((qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP | qe.FLAG_JIT_SYNC, "default flags"), (qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP | qe.FLAG_JIT_SYNC | qe.FLAG_DEFERRED_ROW_COUNT, "deferred row count")):
    ctx.set_flags(flags)
    for _ in range(2):
        r = bench.utf8_filter_config(qe, ctx, torch, dev)
        print(name, round(r["ms_per_step"], 3), "ms", round(r["frac_hbm_peak_end_to_end"], 4), {k: round(v, 3) for k, v in r["kernels_ms_per_step"].items()}, flush=True)
PY
grep -v amdgpu.ids $O/bench_utf8.log
