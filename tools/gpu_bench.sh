#!/bin/bash
# GPU-box run of the headline bench: plain, then under rocprofv3 --kernel-trace --stats.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
timeout -k 10 500 python3 -u $R/bench.py > $R/gpurun_out/bench.json 2> $R/gpurun_out/bench.err || { tail -20 $R/gpurun_out/bench.err; exit 1; }
cat $R/gpurun_out/bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_bench
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench -o bench -- python3 -u $R/bench.py --no-cpu > $R/gpurun_out/bench_prof.json 2> $R/gpurun_out/bench_prof.err || { tail -20 $R/gpurun_out/bench_prof.err; exit 1; }
find $R/gpurun_out/prof_bench -name "*kernel_stats*" | head
# the trace itself is large: keep the summaries only
find $R/gpurun_out/prof_bench -name "*kernel_trace*" -delete
