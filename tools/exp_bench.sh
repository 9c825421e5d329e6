#!/bin/bash
# usage (GPU box): bash tools/exp_bench.sh tag1 tag2 ...   -- the bench step (256 x 1080p) per experiment build
cd $GRAFT_REPO_ROOT
for t in "$@"; do
  if [ $t = base ]; then unset IMGAN_LIB; else export IMGAN_LIB=$GRAFT_REPO_ROOT/image-analysis_amd/imgan/exp/$t.so; fi
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', d['ms_per_step'], 'dcd', d['roofline']['launch_ms'], 'slic', d['roofline_kernels'][0]['launch_ms'])"
done
