# ablation timing of the per-region descriptor kernel on the mosaic step
cd $GRAFT_REPO_ROOT
unset IMGAN_LIB
timeout -k 10 120 python tools/time_mosaic_full.py 3 2>&1 | grep -E "step|region|desc" || exit 1
for v in $VARIANTS; do
  echo "== $v"
  IMGAN_LIB=$PWD/image-analysis_amd/imgan/exp/$v.so timeout -k 10 120 python tools/time_mosaic_full.py 3 2>&1 | grep -E "step|region|desc" || exit 1
done
