"""rag_accumulate_kernel with and without the edge set, 16 x 4K K=2000: python tools/time_ragacc.py [B] [tag]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-analysis_amd")); sys.path.insert(0, ROOT)
import torch
from bench import synth_batch
from imgan import ops
from imgan._lib import lib, check
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
tag = sys.argv[2] if len(sys.argv) > 2 else "base"
H, W, K = 2160, 3840, 2000
bgr = synth_batch(B, H, W, K, 4321, "cuda")
lab = ops.bgr2lab(ops.median7(bgr))
plan = ops.SlicPlan(B, H, W, K)
labels, _ = plan.run(lab, 15, 10)
rp = ops.RagPlan(B, K)
def run(edges):
    rp.sums.zero_(); rp.table.zero_(); rp.overflow.zero_()
    check(lib.imgan_rag_accumulate(ops._ptr(lab), ops._ptr(labels), B, H, W, K, ops._ptr(rp.sums),
                                   ops._ptr(rp.table) if edges else None, rp.cap if edges else 0,
                                   ops._ptr(rp.overflow) if edges else None, ops._stream()))
for edges in (False, True):
    for _ in range(2): run(edges)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    for _ in range(5): run(edges)
    e[1].record(); torch.cuda.synchronize()
    print("%s rag_accumulate edges=%s: %.3f ms (incl. 3 memsets)" % (tag, edges, e[0].elapsed_time(e[1]) / 5))
print("edges per image:", int((rp.table[0] != 0).sum()))
