"""Stress of the expanded Lloyd distances of the dedup tiers (dcd_tiers.hip) against the subtract form of
the gather kernels: random label maps over images whose segments hold wide, bimodal, quantised or
nearly flat colour sets, hash-sampled and given centres, k = 2..16.  Run on the GPU box:
    python tools/stress_expanded.py [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-analysis_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch
from imgan import ops

def run(lab, dl, nseg, k, lists, seed, cen0):
    B, H, W = dl.shape
    plan = ops.DcdPlan(B, H, W, nseg, k, lists=lists)
    cen = torch.zeros((B, nseg, k, 3), dtype=torch.float32, device="cuda") if cen0 is None else torch.from_numpy(cen0.copy()).cuda()
    plan.kmeans(lab, dl, cen, seed=seed)
    stats = plan.statistics(lab, dl)
    rounds, done = plan.rounds()
    torch.cuda.synchronize()
    return [cen.cpu().numpy(), plan.assign.cpu().numpy().copy(), stats.cpu().numpy().copy(), rounds.cpu().numpy(), done.cpu().numpy()]

def image(rng, H, W, kind):
    yy, xx = np.mgrid[0:H, 0:W]
    if kind == 0:                                   # full-range noise
        return rng.randint(0, 256, size=(H, W, 3)).astype(np.uint8)
    if kind == 1:                                   # two far modes + noise
        m = ((xx // 7 + yy // 5) & 1).astype(np.float32)[..., None]
        a, b = rng.randint(0, 60, 3), rng.randint(190, 256, 3)
        return np.clip(a + m * (b - a) + rng.normal(0, 6, (H, W, 3)), 0, 255).astype(np.uint8)
    if kind == 2:                                   # few levels: many exact ties
        return (rng.randint(0, 5, size=(H, W, 3)) * 60).astype(np.uint8)
    if kind == 3:                                   # nearly flat
        return np.clip(128 + rng.normal(0, 1.2, (H, W, 3)), 0, 255).astype(np.uint8)
    g = (xx * 255 // W).astype(np.float32)          # gradient + noise
    return np.clip(np.stack([g, 255 - g, (yy * 255 // H)], -1) + rng.normal(0, 3, (H, W, 3)), 0, 255).astype(np.uint8)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = 0
for it in range(n):
    rng = np.random.RandomState(1000 + it)
    H, W = 192, 256
    img = image(rng, H, W, it % 5)
    cell = [24, 48, 64, 96][it % 4]                 # block segments of 576 .. 9216 px
    labels = ((np.arange(H)[:, None] // cell) * ((W + cell - 1) // cell) + (np.arange(W)[None, :] // cell)).astype(np.int32)
    nseg = int(labels.max()) + 1
    k = [2, 3, 5, 8, 8, 12, 16][it % 7]
    lab = torch.from_numpy(img).cuda().unsqueeze(0)
    dl = torch.from_numpy(labels).cuda().unsqueeze(0)
    cen0 = None
    if it % 3 == 2:                                 # a caller's fractional centres
        cen0 = rng.uniform(0, 255, size=(1, nseg, k, 3)).astype(np.float32)
    a = run(lab, dl, nseg, k, False, 7 + it, cen0)
    b = run(lab, dl, nseg, k, True, 7 + it, cen0)
    same = all(np.array_equal(x, y) for x, y in zip(a, b))
    print("case %2d kind %d cell %2d k %2d %s rounds %.1f %s" % (it, it % 5, cell, k, "given" if cen0 is not None else "hash ",
                                                          float(a[3].mean()), "ok" if same else "DIFFERENT"), flush=True)
    bad += not same
print("different: %d of %d" % (bad, n))
sys.exit(1 if bad else 0)
