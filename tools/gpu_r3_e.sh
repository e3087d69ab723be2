#!/bin/bash
# Experiment E: one scatter level against two (C4 and the C5 plan over 1 M uniform keys), then a bench line.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3e
mkdir -p $O
: > $O/probe.log
run() { echo "== $*" >> $O/probe.log; env "$@" timeout -k 10 120 python -u $R/tools/probe.py $CFG 256000000 3 >> $O/probe.log 2>&1 || { tail -5 $O/probe.log; exit 1; }; }
for CFG in c4 c5m; do
  run QE_X=0
  run QE_SPILL_LEVELS=1
  run QE_X=0
  run QE_SPILL_LEVELS=1
done
cat $O/probe.log
timeout -k 10 500 python3 -u $R/bench.py --no-cpu > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python3 - <<PY
import json
b=json.load(open("$O/bench.json"))
print("headline", b["value"], b["ms_per_step"], b["roofline"]["frac"], b["roofline"]["avg_launch_ms"])
print(json.dumps(b.get("gpu_state_during_timed_steps")))
for k,v in b.get("other_configs",{}).items():
    print(k, {x:v[x] for x in ("ms_per_step","frac_hbm_peak_end_to_end","frac_hbm_peak_main_kernel","same_bits_on_a_second_run") if x in v}, json.dumps(v.get("clocks_during_timed_steps",{}))[:300])
PY
