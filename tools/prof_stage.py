"""Runs one stage of the path on a synthetic batch (for rocprofv3 counter passes).
usage: python3 tools/prof_stage.py {slic|dcd|desc|all|rag4k|mosaic} [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-analysis_amd"))
sys.path.insert(0, ROOT)
import torch
from bench import synth_batch
from imgan.pipeline import FullPipeline

stage = sys.argv[1] if len(sys.argv) > 1 else "slic"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
if stage == "rag4k":                       # configs[2]: 4 x 3840x2160, K = 2000, two steps
    from imgan.pipeline import RagPipeline
    B = min(B, 4)
    bgr = synth_batch(B, 2160, 3840, 2000, 4321, "cuda")
    rp = RagPipeline(B, 2160, 3840, 2000)
    for rep in range(2):
        rp.step(bgr, check=False)
    torch.cuda.synchronize()
    print("done", stage, B)
    sys.exit(0)
if stage == "mosaic":                      # configs[4] geometry on a 4096^2 crop: S = 73, per-region descriptors
    bgr = synth_batch(1, 4096, 4096, 3125, 99, "cuda")
    mp = FullPipeline(1, 4096, 4096, 3125, per_region=True)
    for rep in range(2):
        mp.step(bgr)
    torch.cuda.synchronize()
    print("done", stage)
    sys.exit(0)
bgr = synth_batch(B, 1080, 1920, 400, 1234, "cuda")
pipe = FullPipeline(B, 1080, 1920, 400)
for rep in range(2):
    pipe.run_slic(bgr)
    if stage in ("dcd", "all"):
        pipe.run_dcd()
    if stage in ("desc", "all"):
        pipe.desc.colour_layout_and_edge_histogram(bgr)
        pipe.desc.colour_structure(bgr)
torch.cuda.synchronize()
print("done", stage, B)
