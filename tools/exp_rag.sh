# A/B of experiment builds on the configs[2] step (16 x 4K, SLIC + RAG merge)
cd $GRAFT_REPO_ROOT
unset IMGAN_LIB
timeout -k 10 120 python tools/time_rag4k.py 16 2>&1 | grep -v amdgpu || exit 1
for v in $VARIANTS; do
  echo "== $v"
  IMGAN_LIB=$PWD/image-analysis_amd/imgan/exp/$v.so timeout -k 10 120 python tools/time_rag4k.py 16 2>&1 | grep -v amdgpu || exit 1
done
