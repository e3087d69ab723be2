#!/bin/bash
# GPU-box: the bench without the CPU baseline, one line per configuration.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r2
mkdir -p $O
timeout -k 10 400 python3 -u $R/bench.py --no-cpu "$@" > $O/bench_b.json 2> $O/bench_b.err || { tail -20 $O/bench_b.err; exit 1; }
python3 - <<PY
import json
b=json.load(open("$O/bench_b.json"))
print("headline", b["ms_per_step"], b["roofline"]["frac"], b["roofline"]["avg_launch_ms"], b["gpu_state_after_timed_steps"])
for k,v in b.get("other_configs",{}).items():
    print(k, {x:round(v[x],4) for x in ("ms_per_step","frac_hbm_peak_end_to_end","frac_hbm_peak_main_kernel") if x in v})
PY
