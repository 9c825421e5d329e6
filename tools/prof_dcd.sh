# kernel table of the dominant-colour stage for the product library and experiment builds (GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
K=${K:-400}
for v in product $VARIANTS; do
  if [ $v = product ]; then unset IMGAN_LIB; else export IMGAN_LIB=$PWD/image-analysis_amd/imgan/exp/$v.so; fi
  rm -rf gpurun_out/kt_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_$v -o k -- python3 tools/time_dcd.py 64 $v $K > gpurun_out/kt_$v.log 2>&1 || exit 1
  echo "== $v"; python3 tools/kstats.py gpurun_out/kt_$v 6 | grep "seg_\|dcd_segment"
done
