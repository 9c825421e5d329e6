"""Debug: gather vs lists vs lists on the k = 3 superpixel case of tests/test_gpu_dcd_lists.py."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-analysis_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import voronoi_image
from oracle.color import bgr2lab_u8
from imgan import ops
k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
H, W = 360, 500
imgs = np.stack([voronoi_image(H, W, 50, seed=7 + i, noise=14.0) for i in range(2)])
lab = torch.from_numpy(np.stack([bgr2lab_u8(im) for im in imgs])).cuda()
slic = ops.SlicPlan(2, H, W, 30)
slic.run(lab, 20.0, 4)
dl = slic.labels
def run(lists):
    plan = ops.DcdPlan(2, H, W, slic.n, k, lists=lists)
    cen = torch.zeros((2, slic.n, k, 3), dtype=torch.float32, device="cuda")
    plan.kmeans(lab, dl, cen, seed=99)
    stats = plan.statistics(lab, dl)
    rounds, done = plan.rounds()
    torch.cuda.synchronize()
    return cen.cpu().numpy(), stats.cpu().numpy().copy(), rounds.cpu().numpy(), plan.assign.cpu().numpy().copy()
a = run(False); b = run(True); c = run(True)
lbl = dl.cpu().numpy()
for name, x, y in (("gather vs lists", a, b), ("lists vs lists", b, c)):
    bad = [(i, s) for i in range(2) for s in range(slic.n) if not (np.array_equal(x[0][i, s].view(np.uint32), y[0][i, s].view(np.uint32)) and np.array_equal(x[1][i, s], y[1][i, s]) and x[2][i, s] == y[2][i, s])]
    print(name, "segments that differ:", len(bad))
    for i, s in bad[:6]:
        n = int((lbl[i] == s).sum())
        print("  image %d segment %d px %d rounds %d / %d" % (i, s, n, x[2][i, s], y[2][i, s]))
        print("   centres", x[0][i, s].ravel()[:9], "|", y[0][i, s].ravel()[:9])
        print("   stats0", x[1][i, s][0], "|", y[1][i, s][0])
