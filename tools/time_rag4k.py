"""configs[2] timing: 3840x2160, K=2000, median 7 + Lab + SLIC + RAG merge (imgan.pipeline
.RagPipeline).  Development aid; the numbers quoted in DESIGN.md come from this script."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-analysis_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch
from bench import synth_batch
from imgan.pipeline import RagPipeline

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H, W, K = 2160, 3840, 2000
bgr = synth_batch(B, H, W, K, 4321, "cuda")
pipe = RagPipeline(B, H, W, K)
pipe.step(bgr)
torch.cuda.synchronize()
t0 = time.perf_counter()
steps = 3
for _ in range(steps):
    merged, luts = pipe.step(bgr, timed=True, check=False)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
for k, v in pipe.stage_times_ms().items():
    print("%-30s %8.3f ms (device, per call)" % (k, v))
for k, v in pipe.host_ms.items():
    print("%-30s %8.3f ms (wall)" % (k, float(np.mean(v))))
print("step %.2f ms for %d x 4K -> %.1f Mpx/s; regions in image 0: %d" %
      (dt * 1e3, B, B * H * W / dt / 1e6, int(merged[0].max())))
