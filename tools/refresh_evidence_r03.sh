# Round-3 evidence, run on the GPU box: bash tools/refresh_evidence_r03.sh ; results under gpurun_out/rf3
# (the raw counter CSVs are summarised on the box and deleted: gpurun copies back at most 64 MiB)
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
cp profiles/pmc_traffic.json gpurun_out/rf3/pmc_traffic.json
bash tools/pmc_passes.sh all 64 rf3/pmc fetch write sq1 sq2 sq3 lds > gpurun_out/rf3/pmc_B64.txt 2>&1
python tools/pmc_traffic.py gpurun_out/rf3/pmc_fetch gpurun_out/rf3/pmc_write $((64*1920*1080)) 1920x1080_K400 gpurun_out/rf3/pmc_traffic.json > /dev/null
rm -rf gpurun_out/rf3/pmc_fetch gpurun_out/rf3/pmc_write gpurun_out/rf3/pmc_sq1 gpurun_out/rf3/pmc_sq2 gpurun_out/rf3/pmc_sq3 gpurun_out/rf3/pmc_lds
bash tools/pmc_passes.sh rag4k 4 rf3/pmcrag fetch write sq1 sq2 sq3 lds > gpurun_out/rf3/pmc_rag4k_B4.txt 2>&1
python tools/pmc_traffic.py gpurun_out/rf3/pmcrag_fetch gpurun_out/rf3/pmcrag_write $((4*3840*2160)) 3840x2160_K2000 gpurun_out/rf3/pmc_traffic.json > /dev/null
rm -rf gpurun_out/rf3/pmcrag_*
bash tools/pmc_passes.sh mosaic 1 rf3/pmcmos fetch write sq1 sq2 sq3 lds > gpurun_out/rf3/pmc_mosaic4096.txt 2>&1
rm -rf gpurun_out/rf3/pmcmos_*
for d in kt kt_rag kt_mosaic; do rm -f gpurun_out/rf3/$d/*kernel_trace.csv; done
tail -n 3 gpurun_out/rf3/pmc_B64.txt; du -sh gpurun_out
