"""Phase timing of the 16384^2 / K=50,000 mosaic (configs[4]) -- development aid."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-analysis_amd")); sys.path.insert(0, ROOT)
import torch
from bench import synth_batch
from imgan import ops

def tick(name, t0):
    torch.cuda.synchronize(); t = time.perf_counter(); print("%-28s %8.3f s" % (name, t - t0), flush=True); return t

H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
K = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
t = time.perf_counter()
bgr = synth_batch(1, H, W, K, 99, "cuda"); t = tick("synth", t)
lab = ops.bgr2lab(bgr); t = tick("lab", t)
plan = ops.SlicPlan(1, H, W, K); t = tick("plan", t)
labels, centres = plan.run(lab, 15, 10); t = tick("slic x9", t)
n = plan.n
dcd = ops.DcdPlan(1, H, W, n, 8); t = tick("dcd plan", t)
cen = torch.empty((1, n, 8, 3), dtype=torch.float32, device="cuda")
dcd.kmeans(lab, labels, cen, seed=4321, max_rounds=200, check_every=4); t = tick("dcd kmeans", t)
print("big segments", dcd.big_segments, "streaming rounds", dcd.rounds_run)
stats = dcd.statistics(lab, labels); t = tick("dcd stats", t)
sums, edges = ops.rag_accumulate(lab, labels, K); t = tick("rag accumulate+edges", t)
print("edges", len(edges[0]))
# where the adjacency time goes
t = time.perf_counter()
ops.rag_accumulate(lab, labels, K, want_edges=False); t = tick("rag sums only", t)
for _ in range(2):
    ops.rag_accumulate(lab, labels, K); t = tick("rag accumulate+edges again", t)
cap = 1 << 20
table = torch.zeros((1, cap), dtype=torch.int64, device="cuda"); ovf = torch.zeros(1, dtype=torch.int32, device="cuda")
s2 = torch.zeros((1, K, 4), dtype=torch.int64, device="cuda"); t = tick("alloc", t)
from imgan._lib import lib, check
check(lib.imgan_rag_accumulate(ops._ptr(lab), ops._ptr(labels), 1, H, W, K, ops._ptr(s2), ops._ptr(table), cap, ops._ptr(ovf), ops._stream())); t = tick("kernel with edge set", t)
o, _ = torch.sort(table, dim=1); t = tick("sort", t)
h = o.cpu().numpy(); t = tick("transfer", t)
