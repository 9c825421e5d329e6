#!/bin/bash
# Copies what tools/refresh_evidence_r03.sh left under gpurun_out/rf3 into profiles/r03_* (run here, after the GPU call).
set -e
cd "$(dirname "$0")/.."
S=gpurun_out/rf3
cp $S/pmc_traffic.json profiles/pmc_traffic.json
cp $S/bench.json profiles/r03_bench.json
cp $S/kt/b_kernel_stats.csv profiles/r03_bench_kernel_stats.csv
cp $S/pmc_B64.txt profiles/r03_pmc_B64.txt
cp $S/pmc_rag4k_B4.txt profiles/r03_pmc_rag4k_B4.txt
cp $S/pmc_mosaic4096.txt profiles/r03_pmc_mosaic4096.txt
python tools/kstats.py $S/kt 3 > profiles/r03_bench_kernel_summary.txt
python tools/kstats.py $S/kt_rag 4 > profiles/r03_rag4k_kernel_summary.txt
python tools/kstats.py $S/kt_mosaic 4 > profiles/r03_mosaic_kernel_summary.txt
head -14 profiles/r03_bench_kernel_summary.txt
python - <<PY
import json
d = json.load(open("profiles/r03_bench.json"))
r = d["roofline"]
print(d["value"], d["ms_per_step"], r["launch_ms"], r["frac"], r["traffic"] and r["traffic"] / r["algorithmic_bytes_per_launch"])
for k in d["roofline_kernels"]: print(k["kernel"], k["launch_ms"], k["achieved"], k["frac"])
print(d["path"]); print(d["cpu_baseline"])
for c in d["configs"]: print(str(c.get("config"))[:70], c.get("ms_per_step"), c.get("value"))
PY
