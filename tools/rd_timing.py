"""Phase clocks of the per-region descriptor kernel (needs a build with
EXTRA_HIPCC_FLAGS=-DIMGAN_RD_TIMING into a separate library, loaded through IMGAN_LIB)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-analysis_amd")); sys.path.insert(0, ROOT)
import torch
from bench import synth_batch
from imgan.pipeline import FullPipeline
from imgan import _lib
side = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
K = side * side // 5369
bgr = synth_batch(1, side, side, K, 99, "cuda")
pipe = FullPipeline(1, side, side, K, per_region=True)
pipe.step(bgr)
lib = _lib.lib
buf = (ctypes.c_ulonglong * 16)()
lib.imgan_debug_rd_timing(buf, 1)
pipe.step(bgr, timed=True)
torch.cuda.synchronize()
lib.imgan_debug_rd_timing(buf, 1)
print({k: round(v, 3) for k, v in pipe.stage_times_ms().items()})
names = ["stage", "sums+ehd", "slots", "clear", "build", "dilate-x", "dilate-y", "store", "finish"]
tot = sum(buf[:9])
for n, v in zip(names, buf):
    print("%-10s %12d ticks  %5.1f%%" % (n, v, 100.0 * v / max(tot, 1)))
print("regions %d, colour-structure passes %d, per region %.1f us (100 MHz clock)" % (pipe.n, buf[15], tot / pipe.n / 100.0))
