#!/bin/bash
# GPU-box profiling: rocprofv3 kernel stats of the bench command, then PMC passes (HBM traffic) on one
# launch-sized probe of the same plan.  Each pass is its own run under its own time limit.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# headline only (--no-extra): the other configurations launch the same kernel on other data, which would
# blur the per-kernel average this summary is read for; the full run gets a summary of its own
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 -u $R/bench.py --no-cpu --no-extra > $O/bench_under_prof.json 2> $O/bench_under_prof.err || { tail -20 $O/bench_under_prof.err; exit 1; }
find $O/bench -name "*kernel_trace*" -delete
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_full -o bench -- python3 -u $R/bench.py --no-cpu > $O/bench_full_under_prof.json 2> $O/bench_full_under_prof.err || { tail -20 $O/bench_full_under_prof.err; exit 1; }
find $O/bench_full -name "*kernel_trace*" -delete
# the Parquet scan leaf: headline table written by the probe itself, scanned from the page cache
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/parquet -o parquet -- python3 -u $R/tools/parquet_probe.py 32000000 5 none 1 > $O/parquet_probe.log 2> $O/parquet_probe.err || { tail -20 $O/parquet_probe.err; exit 1; }
find $O/parquet -name "*kernel_trace*" -delete
for cfg in ${PMC_CFGS:-c5}; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 120 rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_${cfg}_$ctr -o pmc -- python3 -u $R/tools/probe.py $cfg 256000000 2 > $O/pmc_${cfg}_$ctr.log 2>&1 || { tail -20 $O/pmc_${cfg}_$ctr.log; exit 1; }
  done
done
find $O -type f | head -40
