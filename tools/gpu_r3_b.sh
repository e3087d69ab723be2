#!/bin/bash
# Check B: the order-independent sums' tests, the whole GPU suite, then the default bench line.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3b
mkdir -p $O
timeout -k 10 300 python -u -m pytest $R/tests/test_gpu_deterministic_sum.py -x -v --timeout 200 --timeout-method thread > $O/detsum.log 2>&1 || { tail -60 $O/detsum.log; exit 1; }
tail -2 $O/detsum.log
timeout -k 10 1000 python -u -m pytest $R/tests -m gpu -x -v --timeout 300 --timeout-method thread > $O/pytest_gpu.log 2>&1 || { tail -60 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
timeout -k 10 500 python3 -u $R/bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python3 - <<PY
import json
b=json.load(open("$O/bench.json"))
print("headline", b["value"], b["ms_per_step"], b["roofline"]["frac"], b["roofline"]["avg_launch_ms"])
for k,v in b.get("other_configs",{}).items():
    print(k, {x:v[x] for x in ("ms_per_step","frac_hbm_peak_end_to_end","frac_hbm_peak_main_kernel","same_bits_on_a_second_run") if x in v})
PY
