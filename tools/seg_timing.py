"""Phase clocks of the per-segment dominant-colour kernel (needs a build with
EXTRA_HIPCC_FLAGS=-DIMGAN_SEG_TIMING into a separate library; see the command in DESIGN.md)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-analysis_amd")); sys.path.insert(0, ROOT)
import torch
from bench import synth_batch
from imgan.pipeline import FullPipeline
from imgan import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
bgr = synth_batch(B, 1080, 1920, 400, 1234, "cuda")
pipe = FullPipeline(B, 1080, 1920, 400)
pipe.run_slic(bgr)
lib = _lib.lib
buf = (ctypes.c_ulonglong * 16)()
pipe.run_dcd()
lib.imgan_debug_seg_timing(buf, 1)
torch.cuda.synchronize()
import time
t0 = time.time()
pipe.run_dcd()
torch.cuda.synchronize()
print('run_dcd %.2f ms' % ((time.time() - t0) * 1e3))
lib.imgan_debug_seg_timing(buf, 1)
names = ["startup", "gather-bar", "assign", "reduce", "update", "stats", "coh-zero", "gather", "coh-largest", "coh-keep", "coh-clear", "coh-scatter", "coh-link", "coh-unite", "coh-flatten", "coh-count"]
tot = sum(buf)
for n, v in zip(names, buf):
    print("%-8s %12d ticks  %5.1f%%" % (n, v, 100.0 * v / max(tot, 1)))
print("per segment: %.1f us (100 MHz clock)" % (tot / (B * pipe.n) / 100.0))
