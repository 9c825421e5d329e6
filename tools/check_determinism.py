"""Runs the bench step several times on the same batch and compares every output bit for bit
(region rows hold the coherence figures, the one place where lock-free unions could show a race).
usage: python tools/check_determinism.py [B] [repeats]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-analysis_amd")); sys.path.insert(0, ROOT)
import torch
from bench import synth_batch
from imgan.pipeline import FullPipeline
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
bgr = synth_batch(B, 1080, 1920, 400, 1234, "cuda")
pipe = FullPipeline(B, 1080, 1920, 400, want_assign=True)
ref = None
bad = 0
for r in range(reps):
    rows, feats = pipe.step(bgr)
    got = (rows.clone(), feats.clone().view(torch.int32), pipe.dcd.assign.clone(), pipe.slic.labels.clone())
    torch.cuda.synchronize()
    if ref is None:
        ref = got
    else:
        for name, a, b in zip(("image rows", "region rows", "assignment", "labels"), ref, got):
            if not torch.equal(a, b):
                bad += 1
                print("run %d: %s differ in %d places" % (r, name, int((a != b).sum())))
print("%d runs of %d images: %s" % (reps, B, "identical" if bad == 0 else "%d differences" % bad))
sys.exit(1 if bad else 0)
