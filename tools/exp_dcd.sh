# A/B on the dominant-colour stage: plain listed tiers against the dedup tiers, plus experiment builds
cd $GRAFT_REPO_ROOT
for K in ${KS:-400 800}; do
unset IMGAN_LIB
IMGAN_DCD_DEDUP=0 timeout -k 10 120 python tools/time_dcd.py 64 plain $K 2>&1 | grep -v amdgpu || exit 1
timeout -k 10 120 python tools/time_dcd.py 64 product $K 2>&1 | grep -v amdgpu || exit 1
for v in $VARIANTS; do
  export IMGAN_LIB=$PWD/image-analysis_amd/imgan/exp/$v.so
  timeout -k 10 120 python tools/time_dcd.py 64 $v $K 2>&1 | grep -v amdgpu || exit 1
done
done
