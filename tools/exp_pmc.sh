#!/bin/bash
# usage (GPU box): bash tools/exp_pmc.sh <stage> <B> <pass> tag1 tag2 ...   (counter pass over experiment builds)
cd $GRAFT_REPO_ROOT
ST=$1; B=$2; P=$3; shift 3
for t in "$@"; do
  if [ $t = base ]; then unset IMGAN_LIB; else export IMGAN_LIB=$PWD/image-analysis_amd/imgan/exp/$t.so; fi
  echo "#### $t"
  bash tools/pmc_passes.sh $ST $B xp_$t $P 2>&1 | grep -E "${FILTER:-.}"
done
