#!/bin/bash
# Builds the C restatement (test infrastructure) into oracle/_build/liboracle_c.so
set -e
cd "$(dirname "$0")"
mkdir -p ../_build
gcc -O2 -fopenmp -ffp-contract=off -fPIC -shared -Wall -o ../_build/liboracle_c.so imgan_oracle.c -lm
echo "built oracle/_build/liboracle_c.so"
