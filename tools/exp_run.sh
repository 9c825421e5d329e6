#!/bin/bash
# usage (GPU box): bash tools/exp_run.sh <script.py> <B> tag1 tag2 ...   (tags: files in imgan/exp/, or "base")
set -e
cd $GRAFT_REPO_ROOT
S=$1; B=$2; shift 2
cp image-analysis_amd/imgan/libimgan_hip.so /tmp/base.so
for t in "$@"; do
  if [ $t = base ]; then cp /tmp/base.so image-analysis_amd/imgan/libimgan_hip.so; else cp image-analysis_amd/imgan/exp/$t.so image-analysis_amd/imgan/libimgan_hip.so; fi
  timeout -k 10 120 python $S $B $t
done
