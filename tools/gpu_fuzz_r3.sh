#!/bin/bash
# Seeded fuzzers against the oracle over what changed last: the many-groups log as (key, value) pairs, Selections that hand
# their batches back without waiting for the row count (string plans, expression trees, aggregations over them), the
# checked hashed group keys, the Parquet scan's read-ahead.  Progress goes to gpurun_out/fuzz3/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/fuzz3
mkdir -p $O
SEED=${1:-11}; ROUNDS=${2:-60}
timeout -k 10 500 python3 -u $R/tools/experiments/agg_fuzz.py $SEED $ROUNDS > $O/agg.log 2>&1 || { tail -15 $O/agg.log; exit 1; }
tail -1 $O/agg.log
QE_FUZZ_DEFER_ROWS=1 QE_FUZZ_JIT=1 timeout -k 10 500 python3 -u $R/tools/experiments/agg_fuzz.py $((SEED+1)) $((ROUNDS/2)) > $O/agg_defer_jit.log 2>&1 || { tail -15 $O/agg_defer_jit.log; exit 1; }
tail -1 $O/agg_defer_jit.log
QE_FUZZ_DEFER_ROWS=1 timeout -k 10 400 python3 -u $R/tools/experiments/utf8_fuzz.py $SEED $ROUNDS > $O/utf8_defer.log 2>&1 || { tail -15 $O/utf8_defer.log; exit 1; }
tail -1 $O/utf8_defer.log
QE_FUZZ_DEFER_ROWS=1 QE_FUZZ_JIT=1 timeout -k 10 400 python3 -u $R/tools/experiments/utf8_fuzz.py $((SEED+2)) $((ROUNDS/2)) > $O/utf8_defer_jit.log 2>&1 || { tail -15 $O/utf8_defer_jit.log; exit 1; }
tail -1 $O/utf8_defer_jit.log
QE_FUZZ_DEFER_ROWS=1 timeout -k 10 400 python3 -u $R/tools/experiments/tree_fuzz.py $SEED $ROUNDS > $O/tree_defer.log 2>&1 || { tail -15 $O/tree_defer.log; exit 1; }
tail -1 $O/tree_defer.log
timeout -k 10 400 python3 -u $R/tools/experiments/mode_fuzz.py $SEED $ROUNDS > $O/mode.log 2>&1 || { tail -15 $O/mode.log; exit 1; }
tail -1 $O/mode.log
timeout -k 10 400 python3 -u $R/tools/experiments/parquet_fuzz.py $SEED $((ROUNDS/2)) > $O/parquet.log 2>&1 || { tail -15 $O/parquet.log; exit 1; }
tail -1 $O/parquet.log
timeout -k 10 400 python3 -u $R/tools/experiments/dataframe_fuzz.py $SEED $ROUNDS > $O/dataframe.log 2>&1 || { tail -15 $O/dataframe.log; exit 1; }
tail -1 $O/dataframe.log
