#!/bin/bash
# usage (GPU box): bash tools/exp_run.sh <script.py> <B> tag1 tag2 ...
# tags: experiment builds image-analysis_amd/imgan/exp/<tag>.so (made with IMGAN_OUT=...), or "base".
# The product library is never overwritten: imgan/_lib.py loads $IMGAN_LIB when it is set.
set -e
cd $GRAFT_REPO_ROOT
S=$1; B=$2; shift 2
for t in "$@"; do
  if [ $t = base ]; then unset IMGAN_LIB; else export IMGAN_LIB=$PWD/image-analysis_amd/imgan/exp/$t.so; fi
  timeout -k 10 120 python $S $B $t
done
