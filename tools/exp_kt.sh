#!/bin/bash
# usage (GPU box): bash tools/exp_kt.sh <stage> <B> <kernel-regex> tag1 tag2 ...  -- rocprof kernel times per experiment build
cd $GRAFT_REPO_ROOT
ST=$1; B=$2; RE=$3; shift 3
for t in "$@"; do
  if [ $t = base ]; then unset IMGAN_LIB; else export IMGAN_LIB=$GRAFT_REPO_ROOT/image-analysis_amd/imgan/exp/$t.so; fi
  (cd /tmp && TMPDIR=/tmp timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kt_$t -o b -- python3 $GRAFT_REPO_ROOT/tools/prof_stage.py $ST $B > /dev/null 2>&1)
  echo "#### $t"; python tools/kstats.py gpurun_out/kt_$t 2 | grep -E "$RE"
  rm -rf gpurun_out/kt_$t
done
