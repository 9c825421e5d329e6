# kernel tables of the dominant-colour stage, pixel lists against the box gather (run on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
K=${1:-400}
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_lists -o l -- python3 tools/time_dcd.py 64 lists $K > gpurun_out/kt_lists.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_gather -o g -- python3 tools/time_dcd.py 64 gather $K nolists > gpurun_out/kt_gather.log 2>&1 || exit 1
python3 tools/kstats.py gpurun_out/kt_lists 6 && python3 tools/kstats.py gpurun_out/kt_gather 6
