#!/bin/bash
# Builds libimgan_hip.so for gfx950 (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
OUT=${IMGAN_OUT:-../imgan/libimgan_hip.so}   # IMGAN_OUT: experiment builds (tools/)
SRCS=$(ls *.hip)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off \
      -Wall -Wno-unused-function -Wno-unused-result \
      $EXTRA_HIPCC_FLAGS $SRCS -o $OUT
echo "built $OUT"
