"""Dominant-colour stage alone with k colours (argv[1]) on 64 x 1080p, K = 400 (IMGAN_DCD_DEDUP=0 for the plain tiers)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-analysis_amd")); sys.path.insert(0, ROOT)
import torch
from bench import synth_batch
from imgan.pipeline import FullPipeline
k = int(sys.argv[1])
bgr = synth_batch(64, 1080, 1920, 400, 1234, "cuda")
pipe = FullPipeline(64, 1080, 1920, 400, dominant=k)
pipe.run_slic(bgr); pipe.run_dcd(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(5):
    e0.record(); pipe.run_dcd(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print("k=%d dedup=%s: min %.3f ms, rounds %.1f" % (k, os.environ.get("IMGAN_DCD_DEDUP", "1"), min(ts), pipe.mean_lloyd_rounds()))
