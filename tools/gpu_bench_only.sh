#!/bin/bash
# The default bench line (no CPU baseline), summarised.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/benchonly
mkdir -p $O
timeout -k 10 500 python3 -u $R/bench.py --no-cpu > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python3 - <<PY
import json
b=json.load(open("$O/bench.json"))
print("headline", b["value"], b["ms_per_step"], b["roofline"]["frac"], b["roofline"]["avg_launch_ms"])
for k,v in b.get("other_configs",{}).items():
    print(k, {x:(round(v[x],4) if isinstance(v[x],float) else v[x]) for x in ("ms_per_step","frac_hbm_peak_end_to_end","frac_hbm_peak_main_kernel","same_bits_on_a_second_run") if x in v})
    if "kernels_ms_per_step" in v and ("C4" in k or "1M" in k or "utf8" in k): print("    ", {a:round(c,3) for a,c in v["kernels_ms_per_step"].items()})
PY
