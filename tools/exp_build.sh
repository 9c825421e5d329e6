#!/bin/bash
# Experiment build: the product objects with some sources recompiled under extra flags.
# usage: bash tools/exp_build.sh <tag> "<extra hipcc flags>" file.hip [file.hip ...]
#   -> image-analysis_amd/imgan/exp/<tag>.so   (run with IMGAN_LIB=..., tools/exp_run.sh)
set -e
cd "$(dirname "$0")/../image-analysis_amd/csrc"
TAG=$1; EXTRA=$2; shift 2
OBJ=../../build/obj_$TAG
mkdir -p $OBJ ../imgan/exp
cp ../../build/obj/*.o $OBJ/
for f in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result $EXTRA -c $f -o $OBJ/${f%.hip}.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC $OBJ/*.o -o ../imgan/exp/$TAG.so
echo "built exp/$TAG.so"
