#!/bin/bash
# Copies what tools/refresh_evidence.sh left under gpurun_out/rf into profiles/ (run here, after the GPU call).
# usage: bash tools/publish_evidence.sh <tag>        e.g. r01_g
set -e
cd "$(dirname "$0")/.."
T=$1
python tools/pmc_traffic.py gpurun_out/rf/pmc_fetch gpurun_out/rf/pmc_write $((64*1920*1080)) 1920x1080_K400 > /dev/null
cp gpurun_out/rf/bench.json profiles/${T}_bench.json
cp gpurun_out/rf/kt/b_kernel_stats.csv profiles/${T}_bench_kernel_stats.csv
cp gpurun_out/rf/pmc_B64.txt profiles/${T}_pmc_B64.txt
python tools/kstats.py gpurun_out/rf/kt 3 > profiles/${T}_bench_kernel_summary.txt
head -12 profiles/${T}_bench_kernel_summary.txt
python - <<PY
import json
d = json.load(open("profiles/${T}_bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["launch_ms"], d["roofline"]["frac"], d["roofline"]["traffic"] / d["roofline"]["algorithmic_bytes_per_launch"])
for k in d["roofline_kernels"]: print(k["kernel"], k["launch_ms"], k["achieved"], k["frac"])
print(d["cpu_baseline"])
PY
