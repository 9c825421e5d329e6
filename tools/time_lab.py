"""Times imgan_bgr2lab_u8 on a 256 x 1080p batch: python tools/time_lab.py [B] [tag]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-analysis_amd")); sys.path.insert(0, ROOT)
import torch
from imgan import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tag = sys.argv[2] if len(sys.argv) > 2 else "base"
g = torch.Generator(device="cuda").manual_seed(3)
from bench import synth_batch
for name, img in (("random", torch.randint(0, 256, (B, 1080, 1920, 3), dtype=torch.uint8, device="cuda", generator=g)),
                  ("mosaic", synth_batch(B, 1080, 1920, 400, 1234, "cuda"))):
  out = torch.empty_like(img)
  for _ in range(3):
    ops.bgr2lab(img, out=out)
  e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
  e[0].record()
  for _ in range(10):
    ops.bgr2lab(img, out=out)
  e[1].record(); torch.cuda.synchronize()
  ms = e[0].elapsed_time(e[1]) / 10
  print("%s bgr2lab %s B=%d 1080p: %.3f ms  %.1f GB/s (6 B/px)" % (tag, name, B, ms, 6 * img.numel() / 3 / ms / 1e6))
