"""Which epilogue the bands of slic_assign_kernel take, per round (needs a build with
EXTRA_HIPCC_FLAGS=-DIMGAN_ASSIGN_COUNTS IMGAN_OUT=../imgan/exp/counts.so bash build.sh, copied
over libimgan_hip.so on the box: tools/exp_run.sh tools/assign_counts.py 64 counts)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-analysis_amd")); sys.path.insert(0, ROOT)
import torch
from bench import synth_batch
from imgan import ops, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
bgr = synth_batch(B, 1080, 1920, 400, 1234, "cuda")
lab = ops.bgr2lab(bgr)
plan = ops.SlicPlan(B, 1080, 1920, 400)
raw = ctypes.CDLL(_lib.lib._name) if hasattr(_lib.lib, "_name") else _lib.lib
fn = raw.imgan_debug_assign_counts
buf = (ctypes.c_ulonglong * 8)()
plan.seed(lab)
fn(buf, 1)
names = ["fast", "rows not full", "m > maxc", "uncovered", "tie replay"]
for it in range(9):
    plan.round(lab, 15, first=(it == 0))
    fn(buf, 1)
    tot = sum(buf[:4])
    print("round %d: " % (it + 1) + ", ".join("%s %.2f%%" % (n, 100.0 * buf[i] / max(tot, 1)) for i, n in enumerate(names)), flush=True)
