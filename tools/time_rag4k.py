"""configs[2] timing: 3840x2160, K=2000, median 7 + Lab + SLIC + RAG merge (device sums and
adjacency, host Kruskal / leave-one-out cut per image, device relabel).  Development aid; the
numbers quoted in DESIGN.md come from this script."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-analysis_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch
from bench import synth_batch
from imgan import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H, W, K = 2160, 3840, 2000
bgr = synth_batch(B, H, W, K, 4321, "cuda")
plan = ops.SlicPlan(B, H, W, K)
blur = torch.empty_like(bgr); lab = torch.empty_like(bgr)

def step(timing=None):
    t = {}
    def tick(name, t0):
        torch.cuda.synchronize(); now = time.perf_counter(); t[name] = t.get(name, 0.0) + now - t0; return now
    t0 = time.perf_counter()
    ops.median7(bgr, out=blur); t0 = tick("median7", t0)
    ops.bgr2lab(blur, out=lab); t0 = tick("lab", t0)
    labels, _ = plan.run(lab, 15, 10); t0 = tick("slic", t0)
    sums, edges = ops.rag_accumulate(lab, labels, K); t0 = tick("rag accumulate + edge lists", t0)
    s = sums.cpu().numpy()
    luts = []
    for b in range(B):
        means = np.where(s[b, :, 3:4] > 0, s[b, :, :3] // np.maximum(s[b, :, 3:4], 1), 0)
        luts.append(ops.rag_merge_host(K, edges[b], 0.65, means=means)["lut"])
    t0 = tick("host merge", t0)
    merged = ops.relabel(labels, torch.from_numpy(np.stack(luts)).cuda()); t0 = tick("relabel", t0)
    return t, merged

step()
t, merged = step()
tot = sum(t.values())
for k, v in t.items():
    print("%-30s %8.2f ms" % (k, v * 1e3))
print("total %.2f ms for %d x 4K -> %.1f Mpx/s; regions in image 0: %d" %
      (tot * 1e3, B, B * H * W / tot / 1e6, int(merged[0].max())))
