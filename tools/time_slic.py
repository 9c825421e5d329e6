import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-analysis_amd")); sys.path.insert(0, ROOT)
import torch
from bench import synth_batch
from imgan.pipeline import FullPipeline
B = 64
bgr = synth_batch(B, 1080, 1920, 400, 1234, "cuda")
pipe = FullPipeline(B, 1080, 1920, 400)
for cfg in sys.argv[1:]:
    dbg, nsub = cfg.split(":")
    os.environ["IMGAN_SLIC_DEBUG"] = dbg
    if nsub != "0": os.environ["IMGAN_SLIC_NSUB"] = nsub
    else: os.environ.pop("IMGAN_SLIC_NSUB", None)
    pipe.run_slic(bgr); torch.cuda.synchronize()
    pipe.assign_events = []
    pipe.run_slic(bgr, timed=True); torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in pipe.assign_events]
    print("dbg=%s nsub=%s assign ms: first %.3f mean-rest %.3f" % (dbg, nsub, ms[0], sum(ms[1:]) / len(ms[1:])))
