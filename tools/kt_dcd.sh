# per-kernel durations of the dominant-colour stage: the product library ("new") or experiment builds (GPU box)
# usage: bash tools/kt_dcd.sh <B> <outdir under gpurun_out> [variants: new, or tags of imgan/exp/<tag>.so]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B=${1:-256}; OUT=gpurun_out/${2:-ktd}; shift 2
mkdir -p $OUT
for v in ${*:-new}; do
  unset IMGAN_LIB
  if [ $v != new ]; then export IMGAN_LIB=$PWD/image-analysis_amd/imgan/exp/$v.so; fi
  rm -rf $OUT/$v
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/$v -o k -- python3 tools/time_dcd.py $B $v > $OUT/$v.log 2>&1 || { tail -5 $OUT/$v.log; exit 1; }
  echo "== $v"; grep "dcd stage" $OUT/$v.log
  python3 - $OUT/$v <<'PY'
import collections, csv, glob, sys
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if "dcd_" in k or "seg_" in k:
        print("%-100s n=%4d avg=%9.1f us" % (k[:100], len(v), sum(v) / len(v) / 1e3))
PY
done
