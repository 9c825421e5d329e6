#!/bin/bash
# Builds libimgan_hip.so for gfx950 (cross-compiles without a GPU).  One object per kernel file,
# compiled side by side (no relocatable device code: no kernel calls across files), then linked.
set -e
cd "$(dirname "$0")"
OUT=${IMGAN_OUT:-../imgan/libimgan_hip.so}   # IMGAN_OUT: experiment builds (tools/)
OBJ=${IMGAN_OBJ:-../../build/obj}            # IMGAN_OBJ: their objects (other flags, other directory)
JOBS=${IMGAN_JOBS:-8}
mkdir -p "$OBJ"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-result $EXTRA_HIPCC_FLAGS"
echo "$FLAGS" > "$OBJ/flags.new"
cmp -s "$OBJ/flags.new" "$OBJ/flags" || { rm -f "$OBJ"/*.o; mv "$OBJ/flags.new" "$OBJ/flags"; }
compile() {
    src=$1; obj="$OBJ/${src%.hip}.o"
    # rebuilt when the source, any header here or the public header is newer than the object
    if [ ! -f "$obj" ] || [ -n "$(find "$src" *.h *.inc ../../include/imgan.h -newer "$obj" 2>/dev/null)" ]; then
        hipcc $FLAGS -c "$src" -o "$obj" || exit 1
    fi
}
export -f compile; export OBJ FLAGS
ls *.hip | xargs -P "$JOBS" -I{} bash -c 'compile {}'
hipcc --offload-arch=gfx950 -shared -fPIC "$OBJ"/*.o -o "$OUT"
echo "built $OUT"
