#!/bin/bash
# PMC passes on one probe configuration.  usage: gpu_pmc.sh <cfg> "<counters pass 1>" "<counters pass 2>" ...
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=$1; shift
O=$R/gpurun_out/pmc_$CFG
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "$@"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d $O/p$i -o pmc -- python3 -u $R/tools/probe.py $CFG 256000000 2 > $O/p$i.log 2>&1 || { tail -20 $O/p$i.log; exit 1; }
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("$O/p*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "qe_" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:60],r["Counter_Name"])].append(float(r["Counter_Value"]))
import json
out={}
for k,v in sorted(agg.items()):
    print(k[0],k[1],len(v),sum(v)/len(v))
    out.setdefault(k[0],{})[k[1]]={"launches":len(v),"mean_per_launch":sum(v)/len(v)}
json.dump({"config":"$CFG","rows_per_launch":256000000,"counters":out},open("$O/summary.json","w"),indent=1)
PY
