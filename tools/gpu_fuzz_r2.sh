#!/bin/bash
# Seeded fuzzers against the oracle with the newer options switched on: order-independent sums (aggregations, the
# PARTIAL / FINAL / deferred modes), string plans with the run-time compiler.  Progress goes to gpurun_out/fuzz/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/fuzz
mkdir -p $O
SEED=${1:-7}; ROUNDS=${2:-60}
QE_FUZZ_DET=1 timeout -k 10 500 python3 -u $R/tools/experiments/agg_fuzz.py $SEED $ROUNDS > $O/agg_det.log 2>&1 || { tail -15 $O/agg_det.log; exit 1; }
tail -1 $O/agg_det.log
QE_FUZZ_DET=1 QE_FUZZ_JIT=1 timeout -k 10 500 python3 -u $R/tools/experiments/agg_fuzz.py $((SEED+1)) $((ROUNDS/2)) > $O/agg_det_jit.log 2>&1 || { tail -15 $O/agg_det_jit.log; exit 1; }
tail -1 $O/agg_det_jit.log
QE_FUZZ_DET=1 timeout -k 10 400 python3 -u $R/tools/experiments/mode_fuzz.py $SEED $ROUNDS > $O/mode_det.log 2>&1 || { tail -15 $O/mode_det.log; exit 1; }
tail -1 $O/mode_det.log
QE_FUZZ_JIT=1 timeout -k 10 400 python3 -u $R/tools/experiments/utf8_fuzz.py $SEED $ROUNDS > $O/utf8_jit.log 2>&1 || { tail -15 $O/utf8_jit.log; exit 1; }
tail -1 $O/utf8_jit.log
