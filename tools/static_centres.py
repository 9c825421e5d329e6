"""How many SLIC centres stand still from one round to the next, and how many 64 x 64 tiles see only
such centres (their labels and partial sums cannot change).  usage: python tools/static_centres.py [B H W K]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-analysis_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from bench import synth_batch
from imgan import ops
B, H, W, K = (int(v) for v in (sys.argv[1:5] + ["8", "1080", "1920", "400"][len(sys.argv) - 1:]))
bgr = synth_batch(B, H, W, K, 1234, "cuda")
lab = ops.bgr2lab(bgr)
plan = ops.SlicPlan(B, H, W, K)
S = plan.S
plan.seed(lab)
prev = plan.centres.cpu().numpy().copy()
TW = 64
ntx, nty = (W + TW - 1) // TW, (H + TW - 1) // TW
tx0 = np.arange(ntx) * TW
ty0 = np.arange(nty) * TW
for it in range(1, 10):
    plan.round(lab, 15, first=(it == 1))
    torch.cuda.synchronize()
    cur = plan.centres.cpu().numpy().copy()
    changed = (cur != prev).any(axis=2)                      # [B, n]: what the NEXT round sees differently
    frac = changed.mean()
    # tiles of the next round that see a changed centre under its old or its new position
    dirty = np.zeros((B, nty, ntx), dtype=bool)
    for c in (prev, cur):
        cx, cy = c[:, :, 3], c[:, :, 4]
        for b in range(B):
            for i in np.nonzero(changed[b])[0]:
                xs = (cx[b, i] - S < tx0 + TW) & (cx[b, i] + S > tx0)
                ys = (cy[b, i] - S < ty0 + TW) & (cy[b, i] + S > ty0)
                dirty[b] |= ys[:, None] & xs[None, :]
    print("after round %d: %.1f %% of the centres moved; round %d could skip %.1f %% of its tiles"
          % (it, 100 * frac, it + 1, 100 * (1 - dirty.mean())))
    prev = cur
