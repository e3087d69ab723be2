#!/bin/bash
# Dense-bin ordering sweep: auto / off / bins first / hot keys first, on the configurations of bench.py.
set -o pipefail
mkdir -p gpurun_out; : > gpurun_out/tune3.log
for mode in ${MODES:-1 2}; do
  echo "== QE_AGGREGATE_DENSE_BINS=$mode" >> gpurun_out/tune3.log
  QE_AGGREGATE_DENSE_BINS=$mode timeout -k 10 300 python3 -u bench.py --no-cpu > gpurun_out/bench_mode$mode.json 2> gpurun_out/bench_mode$mode.err || { tail -5 gpurun_out/bench_mode$mode.err; exit 1; }
  python3 - >> gpurun_out/tune3.log <<PY
import json
d=json.loads(open('gpurun_out/bench_mode$mode.json').read().strip().splitlines()[-1])
print('C5', round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],3), round(d['value']/1e9,1))
for k,v in d['other_configs'].items():
    if 'avg_launch_ms' in v: print(k, round(v['avg_launch_ms'],4), round(v['frac_hbm_peak'],3), round(v['rows_per_s']/1e9,1))
PY
done
cat gpurun_out/tune3.log
