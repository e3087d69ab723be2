#!/bin/bash
# Kernel trace of one configuration of tools/probe.py: every launch with its duration, in order.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/trace_${1:-c4}
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- python3 -u $R/tools/probe.py ${1:-c4} ${2:-500000000} 3 ${3:-1} > $O/probe.log 2>&1 || { tail $O/probe.log; exit 1; }
python3 - <<PY
import csv,glob
f=glob.glob("$O/t/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=None
out=[]
for r in rows[-40:]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    if t0 is None: t0=s
    out.append(f"{(s-t0)/1e3:10.1f} us  +{(e-s)/1e3:8.1f} us  {r['Kernel_Name'][:90]}")
open("$O/summary.txt","w").write("\n".join(out)+"\n")
print("\n".join(out))
PY
find $O/t -name "*.csv" -delete
