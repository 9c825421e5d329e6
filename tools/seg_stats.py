import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-analysis_amd")); sys.path.insert(0, ROOT)
import torch, numpy as np
from bench import synth_batch
from imgan.pipeline import FullPipeline
B = 16
bgr = synth_batch(B, 1080, 1920, 400, 1234, "cuda")
pipe = FullPipeline(B, 1080, 1920, 400)
pipe.run_slic(bgr); pipe.run_dcd()
print("big segments", pipe.dcd.big_segments, "streaming rounds", pipe.dcd.rounds_run)
bb = pipe.dcd.bbox.view(torch.int32)[:B * pipe.n * 4].view(B, pipe.n, 4).cpu().numpy()
cnt = pipe.dcd.stats[..., 0].sum(-1).cpu().numpy(); area = (bb[..., 2] - bb[..., 0] + 1) * (bb[..., 3] - bb[..., 1] + 1)
print("count: mean %.0f max %d  >11776: %d" % (cnt.mean(), cnt.max(), (cnt > 11776).sum()))
print("area: mean %.0f max %d  >65535: %d" % (area.mean(), area.max(), (area > 65535).sum()))
print("area/count mean %.2f" % (area.sum() / cnt.sum()))
r, d = pipe.dcd.rounds()
r = r.cpu().numpy()
print("rounds: mean %.2f max %d ; hist" % (r.mean(), r.max()), np.bincount(r.ravel())[:40])
for cap in (8192, 9216, 10240, 10752, 11264, 11776):
    fit = (cnt <= cap) & (area * 3 + 8 <= cap * 6)
    print("cap %5d: %.1f %% of segments, %.1f %% of pixels fit" % (cap, 100.0 * fit.mean(), 100.0 * cnt[fit].sum() / cnt.sum()))
