#!/bin/bash
# Check G: many-groups parity tests, then probes with one environment setting against the default (A/B on one box).
# usage: gpu_r3_g.sh VAR=value [cfgs...]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3g
mkdir -p $O
ALT=${1:-QE_X=0}
shift
CFGS=${*:-c4 c5m}
timeout -k 10 600 python -u -m pytest $R/tests/test_gpu_specialized.py $R/tests/test_gpu_trees.py -x -q --timeout 200 --timeout-method thread > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 500 python -u -m pytest $R/tests/test_gpu_full_size.py -k "c4" -x -v -s --timeout 400 --timeout-method thread > $O/pytest_full.log 2>&1 || { tail -40 $O/pytest_full.log; exit 1; }
tail -2 $O/pytest_full.log
: > $O/probe.log
run() { echo "== $*" >> $O/probe.log; env "$@" timeout -k 10 120 python -u $R/tools/probe.py $CFG 256000000 3 >> $O/probe.log 2>&1 || { tail -5 $O/probe.log; exit 1; }; }
for CFG in $CFGS; do
  run $ALT
  run QE_X=0
  run $ALT
  run QE_X=0
done
grep -v amdgpu.ids $O/probe.log | python3 -c "
import sys, ast, re
for line in sys.stdin:
    line = line.strip()
    if line.startswith('=='):
        print(line, end='  ')
        continue
    m = re.match(r'(\w+) rows (\d+) ms ([\d.]+) .*?out_rows (\d+) (\{.*\})', line)
    if not m:
        print(line)
        continue
    prof = ast.literal_eval(m.group(5))
    print(m.group(1), 'ms %.3f' % float(m.group(3)), ' '.join('%s=%.3f' % (k.replace('qe_', '').replace('_kernel', '')[:28], v['ms'] / v['launches']) for k, v in prof.items()))
"
