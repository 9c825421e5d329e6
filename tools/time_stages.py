"""Stage times of FullPipeline.step on the bench batch.  usage: python tools/time_stages.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-analysis_amd")); sys.path.insert(0, ROOT)
import torch
from bench import synth_batch
from imgan.pipeline import FullPipeline
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
bgr = synth_batch(B, 1080, 1920, 400, 1234, "cuda")
pipe = FullPipeline(B, 1080, 1920, 400)
for _ in range(2):
    pipe.step(bgr)
pipe.stage_events = {}
for _ in range(4):
    pipe.step(bgr, timed=True)
torch.cuda.synchronize()
for k, v in pipe.stage_times_ms().items():
    print("%-20s %8.3f ms" % (k, v))
