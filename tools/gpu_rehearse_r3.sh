#!/bin/bash
# One-GPU rehearsal of the multi-rank bench paths: the library's exchange over real RCCL with one rank (few groups and a
# million), then two ranks sharing the device with the exchange through gloo (the launcher's children, their time limit
# and per-rank stderr files included), few groups and a million.  Every variant must find the same groups and rows.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/rehearse_r3
mkdir -p $O
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29544
show() { python3 - <<PY
import json
b=json.load(open("$1"))
print("$2", "n_gpus", b["n_gpus"], "ms/step", round(b["ms_per_step"],3), "merge ms", b["merge_ms_per_step"], "groups", b["result_groups"], "count", b["result_count_total"], b["config"]["merge"][:60])
PY
}
for variant in "abi 10 1000000000" "abi 1000000 500000000"; do
  set -- $variant
  QE_BENCH_FORCE_DIST=1 QE_BENCH_EXCHANGE=$1 timeout -k 10 300 python3 -u $R/bench.py --no-cpu --no-extra --groups $2 --rows $3 > $O/one_rank_$1_$2.json 2> $O/one_rank_$1_$2.err || { tail -20 $O/one_rank_$1_$2.err; exit 1; }
  show $O/one_rank_$1_$2.json "rccl, one rank, groups=$2:"
done
for variant in "10 400000000" "1000000 200000000"; do
  set -- $variant
  QE_BENCH_BACKEND=gloo QE_BENCH_LOG_DIR=$O/ranks_$1 QE_BENCH_TIMEOUT=240 timeout -k 10 300 python3 -u $R/bench.py --gpus 2 --no-cpu --no-extra --groups $1 --rows $2 > $O/two_ranks_$1.json 2> $O/two_ranks_$1.err || { tail -30 $O/two_ranks_$1.err; exit 1; }
  show $O/two_ranks_$1.json "gloo, two ranks on one device, groups=$1:"
done
# and one rank alone over the same rows, for the groups and counts to compare with
for variant in "10 400000000" "1000000 200000000"; do
  set -- $variant
  timeout -k 10 300 python3 -u $R/bench.py --no-cpu --no-extra --groups $1 --rows $2 > $O/single_$1.json 2> $O/single_$1.err || { tail -20 $O/single_$1.err; exit 1; }
  show $O/single_$1.json "single process, groups=$1:"
done
