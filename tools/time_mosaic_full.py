"""configs[4] step (one 16384^2 mosaic, K=50,000, descriptors per superpixel) -- for rocprofv3 runs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-analysis_amd")); sys.path.insert(0, ROOT)
import torch
from bench import synth_batch
from imgan.pipeline import FullPipeline
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
bgr = synth_batch(1, 16384, 16384, 50000, 99, "cuda")
pipe = FullPipeline(1, 16384, 16384, 50000, per_region=True)
pipe.step(bgr)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    pipe.step(bgr, timed=True)
torch.cuda.synchronize()
print("step %.2f ms" % ((time.perf_counter() - t0) / steps * 1e3))
for k, v in pipe.stage_times_ms().items():
    print("%-20s %8.3f ms" % (k, v))
