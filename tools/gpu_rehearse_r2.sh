#!/bin/bash
# One-GPU rehearsal of the multi-rank bench path over real RCCL (world size 1): PARTIAL aggregate, exchange inside the
# library (qe_exchange_all_gather / qe_exchange_by_key), FINAL merge.  Then the same through torch.distributed.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r2
mkdir -p $O
export QE_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544
for variant in "abi 10 1000000000" "abi 1000000 500000000" "torch 10 1000000000"; do
  set -- $variant
  QE_BENCH_EXCHANGE=$1 timeout -k 10 300 python3 -u $R/bench.py --no-cpu --no-extra --groups $2 --rows $3 > $O/rehearse_$1_$2.json 2> $O/rehearse_$1_$2.err || { tail -20 $O/rehearse_$1_$2.err; exit 1; }
  python3 - <<PY
import json
b=json.load(open("$O/rehearse_$1_$2.json"))
print("$1 groups=$2", "ms/step", round(b["ms_per_step"],3), "merge ms", b["merge_ms_per_step"], "groups", b["result_groups"], "count", b["result_count_total"], b["config"]["merge"][:70], b["config"]["topology"].get("library_rccl_version"))
PY
done
