set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/rf
timeout -k 10 400 python bench.py --steps 5 --warmup 2 > gpurun_out/rf/bench.json 2> gpurun_out/rf/bench.err
tail -c 3000 gpurun_out/rf/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/rf/kt -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rf/kt.log 2>&1
cd $GRAFT_REPO_ROOT
bash tools/pmc_passes.sh all 64 rf/pmc fetch write sq1 lds > gpurun_out/rf/pmc_B64.txt 2>&1
tail -30 gpurun_out/rf/pmc_B64.txt
