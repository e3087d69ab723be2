#!/bin/bash
# Copies what tools/gpu_r3_profiles.sh left under gpurun_out/r3p into profiles/ under the names profiles/README.md lists.
O=gpurun_out/r3p; P=profiles
cp $O/bench.json $P/r03_bench.json
for n in headline c4 c5m utf8 parquet; do cp $O/${n}_kernel_stats.csv $P/r03_${n}_kernel_stats.csv; done
grep -v amdgpu.ids $O/headline.out | tail -1 > $P/r03_bench_under_rocprof.json
grep -v amdgpu.ids $O/c4.out | tail -3 > $P/r03_c4_probe_line.txt
grep -v amdgpu.ids $O/c5m.out | tail -3 > $P/r03_c5m_probe_line.txt
grep -v amdgpu.ids $O/utf8.out | tail -2 > $P/r03_utf8_probe_line.txt
grep -v amdgpu.ids $O/parquet.out > $P/r03_parquet_scan_probe_under_rocprof.txt
cp $O/pmc_c4.json $P/r03_pmc_c4.json; cp $O/pmc_c5m.json $P/r03_pmc_c5m.json; cp $O/pmc_utf8.json $P/r03_pmc_utf8.json
{ echo "== before"; cat $O/smi_before.txt; echo "== after"; cat $O/smi_after.txt; } > $P/r03_rocm_smi_before_after.txt
