"""Times imgan_region_descriptors on superpixels: python tools/time_region.py [B] [tag]
B images of 1920x1080 K=400 (or 'mosaic' for one 16384^2 K=50,000)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-analysis_amd")); sys.path.insert(0, ROOT)
import torch
from bench import synth_batch
from imgan.pipeline import FullPipeline
arg = sys.argv[1] if len(sys.argv) > 1 else "64"
tag = sys.argv[2] if len(sys.argv) > 2 else "base"
B, H, W, K = (1, 16384, 16384, 50000) if arg == "mosaic" else (int(arg), 1080, 1920, 400)
bgr = synth_batch(B, H, W, K, 99, "cuda")
pipe = FullPipeline(B, H, W, K, per_region=True)
pipe.run_slic(bgr)
for _ in range(2):
    pipe.region_desc.run(bgr, pipe.slic.labels, bbox=pipe.dcd.bbox)
e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
e[0].record()
for _ in range(3):
    pipe.region_desc.run(bgr, pipe.slic.labels, bbox=pipe.dcd.bbox)
e[1].record(); torch.cuda.synchronize()
ms = e[0].elapsed_time(e[1]) / 3
csd = pipe.region_desc.csd
ncol = (csd[..., :256] > 0).sum(-1).float()
print("%s region_desc %s: %.3f ms  %.1f GB/s (21 B/px); colours per region mean %.1f max %d; too big %d" %
      (tag, arg, ms, 21 * B * H * W / ms / 1e6, float(ncol.mean()), int(ncol.max()), int(pipe.region_desc.too_big.item())))
