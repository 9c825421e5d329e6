"""Per-round time of slic_assign inside FullPipeline.step (bench images).  usage: python tools/time_rounds.py [B]"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "image-analysis_amd"))
import torch
import bench
from imgan.pipeline import FullPipeline

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
bgr = bench.synth_batch(B, 1080, 1920, 400, 1234, torch.device("cuda", 0))
pipe = FullPipeline(B, 1080, 1920, 400)
for _ in range(2):
    pipe.step(bgr)
pipe.stage_events = {}
for _ in range(4):
    pipe.step(bgr, timed=True)
torch.cuda.synchronize()
ev = pipe.stage_events["slic_assign"]
n = pipe.num_iter - 1
for r in range(n):
    ts = [ev[s * n + r][0].elapsed_time(ev[s * n + r][1]) for s in range(4)]
    print("round %d: %.3f ms" % (r + 1, sum(ts) / len(ts)))
