"""Mean launch time of the SLIC assignment kernel at the bench shape (experiment builds: the
caller copies the variant library over imgan/libimgan_hip.so on the GPU box first)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-analysis_amd")); sys.path.insert(0, ROOT)
import torch
from bench import synth_batch
from imgan.pipeline import FullPipeline
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
bgr = synth_batch(B, 1080, 1920, 400, 1234, "cuda")
pipe = FullPipeline(B, 1080, 1920, 400)
pipe.run_slic(bgr); torch.cuda.synchronize()
best = []
for _ in range(3):
    pipe.stage_events = {}
    pipe.run_slic(bgr, timed=True); torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in pipe.stage_events["slic_assign"]]
    best.append(sum(ms) / len(ms))
print("%s assign B=%d: mean launch %.4f ms (best of 3 passes)" % (sys.argv[2] if len(sys.argv) > 2 else "", B, min(best)))
