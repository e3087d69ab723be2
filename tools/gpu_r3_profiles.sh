#!/bin/bash
# Evidence, one session: the default bench line; rocprofv3 kernel summaries of the headline, of C4, of the C5 plan
# over a million uniform keys and of the string shape, each next to the line the profiled process printed; PMC passes
# (one run each) on C4's kernels and on the string gather's.  Everything lands in gpurun_out/r3p.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3p
rm -rf $O && mkdir -p $O
rocm-smi --showclocks --showpower > $O/smi_before.txt 2>&1
timeout -k 10 500 python3 -u $R/bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
stats() {  # name, program arguments...
  local name=$1
  shift
  rm -rf $O/prof_$name
  timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o $name -- python3 -u "$@" > $O/$name.out 2> $O/$name.err || { tail -20 $O/$name.err; exit 1; }
  find $O/prof_$name -name "*kernel_trace*" -delete
  cp $(find $O/prof_$name -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv
}
stats headline $R/bench.py --no-cpu --no-extra
stats c4 $R/tools/probe.py c4 500000000 5 2
stats c5m $R/tools/probe.py c5m 500000000 5 2
stats parquet $R/tools/parquet_probe.py 32000000 3
stats utf8 $R/tools/probe.py utf8
pmc() {  # name, config, counters
  local name=$1 cfg=$2
  shift 2
  rm -rf $O/pmc_$name
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$name -o pmc -- python3 -u $R/tools/probe.py $cfg 500000000 2 2 > $O/pmc_$name.log 2>&1 || { tail -20 $O/pmc_$name.log; exit 1; }
}
pmc c4_sq c4 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pmc c4_tcc c4 TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum
pmc c4_fetch c4 FETCH_SIZE
pmc c4_write c4 WRITE_SIZE
pmc c5m_sq c5m SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pmc c5m_fetch c5m FETCH_SIZE
pmc c5m_write c5m WRITE_SIZE
pmc utf8_fetch utf8 FETCH_SIZE
pmc utf8_write utf8 WRITE_SIZE
rocm-smi --showclocks --showpower > $O/smi_after.txt 2>&1
python3 - <<PY
import csv, glob, collections, json, os
O = "$O"
def summarise(names, out, rows_note):
    agg = collections.defaultdict(list)
    for n in names:
        for f in glob.glob(os.path.join(O, "pmc_" + n, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "qe_" in r["Kernel_Name"] and "generate" not in r["Kernel_Name"]:
                    agg[(r["Kernel_Name"][:70] + " | grid " + r.get("Grid_Size", "?"), r["Counter_Name"])].append(float(r["Counter_Value"]))
    res = {}
    for (k, c), v in sorted(agg.items()):
        # launches of one kernel that did next to nothing next to its others (the 2^20-row trial that chooses the
        # many-groups first pass runs the first pass's kernel over the same grid) are kept out of the mean
        big = [x for x in v if x >= 0.1 * max(v)] if max(v) > 0 else v
        res.setdefault(k, {})[c] = {"launches": len(big), "mean_per_launch": sum(big) / len(big), "small_launches_left_out": len(v) - len(big)}
    json.dump({"what": rows_note, "counters": res}, open(os.path.join(O, out), "w"), indent=1)
    for k, cs in res.items():
        print(k, {c: round(x["mean_per_launch"] / 1e6, 2) for c, x in cs.items()})
summarise(["c5m_sq", "c5m_fetch", "c5m_write"], "pmc_c5m.json", "tools/probe.py c5m 500000000 2 2: the C5 plan over 1 M uniform keys, 500 M rows in two batches that share one launch of the first pass that partitions directly; counters per launch; FETCH_SIZE / WRITE_SIZE as rocprofv3 reports them (see profiles/README.md for the gfx950 reading)")
summarise(["c4_sq", "c4_tcc", "c4_fetch", "c4_write"], "pmc_c4.json", "tools/probe.py c4 500000000 2 2: 500 M rows, 1 M zipf(1.1) keys, two batches that share one launch of the first pass (the 2^20-row trial launches that choose the first pass are left out of the first-pass kernel's means); FETCH_SIZE / WRITE_SIZE in KiB-free raw units as rocprofv3 reports them (see profiles/README.md for the gfx950 reading)")
summarise(["utf8_fetch", "utf8_write"], "pmc_utf8.json", "tools/probe.py utf8: the reference's test_filter shape, 10 batches of 10 M rows a pass, 3 passes a run; per launch = per batch")
b = json.load(open(os.path.join(O, "bench.json")))
print("headline", b["value"], b["ms_per_step"], b["roofline"]["frac"], b["roofline"]["avg_launch_ms"])
for k, v in b.get("other_configs", {}).items():
    print(k, {x: (round(v[x], 4) if isinstance(v[x], float) else v[x]) for x in ("ms_per_step", "frac_hbm_peak_end_to_end", "frac_hbm_peak_main_kernel", "same_bits_on_a_second_run") if x in v})
PY
