# Round-3 evidence, run on the GPU box: bash tools/refresh_evidence_r03.sh ; results under gpurun_out/rf3
set -e
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/rf3
timeout -k 10 500 python bench.py --steps 5 --warmup 2 > gpurun_out/rf3/bench.json 2> gpurun_out/rf3/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rf3/kt -o b -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs > $R/gpurun_out/rf3/kt.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rf3/kt_rag -o r -- python3 $R/tools/time_rag4k.py 16 > $R/gpurun_out/rf3/kt_rag.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rf3/kt_mosaic -o m -- python3 $R/tools/time_mosaic_full.py 3 > $R/gpurun_out/rf3/kt_mosaic.log 2>&1
cd $R
bash tools/pmc_passes.sh all 64 rf3/pmc fetch write sq1 lds > gpurun_out/rf3/pmc_B64.txt 2>&1
bash tools/pmc_passes.sh rag4k 4 rf3/pmcrag fetch write sq1 lds > gpurun_out/rf3/pmc_rag4k_B4.txt 2>&1
bash tools/pmc_passes.sh mosaic 1 rf3/pmcmos fetch write sq1 lds > gpurun_out/rf3/pmc_mosaic4096.txt 2>&1
tail -n 5 gpurun_out/rf3/pmc_B64.txt; tail -n 5 gpurun_out/rf3/pmc_rag4k_B4.txt
