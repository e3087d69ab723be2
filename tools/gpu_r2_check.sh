#!/bin/bash
# Check: the no-alloc entry point's test alone, then the whole GPU suite, then the bench with the
# rocprofv3 kernel summary of the same command taken in the same session.  Every GPU step has its own time limit.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r2
mkdir -p $O
rocm-smi --showclocks --showpower > $O/smi_before.txt 2>&1
if [ -f $R/tests/test_zz_no_alloc.py ]; then
  timeout -k 10 300 python -u -m pytest $R/tests/test_zz_no_alloc.py -x -v --timeout 120 --timeout-method thread > $O/pytest_noalloc.log 2>&1 || { tail -40 $O/pytest_noalloc.log; exit 1; }
  tail -3 $O/pytest_noalloc.log
fi
timeout -k 10 900 python -u -m pytest $R/tests -m gpu -x -v -s --timeout 300 --timeout-method thread > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
timeout -k 10 500 python3 -u $R/bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 -u $R/bench.py --no-cpu --no-extra > $O/bench_under_prof.json 2> $O/bench_under_prof.err || { tail -20 $O/bench_under_prof.err; exit 1; }
find $O/prof -name "*kernel_trace*" -delete
rocm-smi --showclocks --showpower > $O/smi_after.txt 2>&1
python3 - <<PY
import json
b=json.load(open("$O/bench.json"))
print("headline", b["value"], b["ms_per_step"], b["roofline"]["frac"], b["roofline"]["avg_launch_ms"])
for k,v in b.get("other_configs",{}).items():
    print(k, {x:v[x] for x in ("ms_per_step","frac_hbm_peak_end_to_end","frac_hbm_peak_main_kernel") if x in v})
PY
