"""Times imgan_median_u8c3 on 16 x 3840x2160 (configs[2]'s batch): python tools/time_median.py [B] [tag]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-analysis_amd")); sys.path.insert(0, ROOT)
import torch
from imgan import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
tag = sys.argv[2] if len(sys.argv) > 2 else "base"
g = torch.Generator(device="cuda").manual_seed(3)
img = torch.randint(0, 256, (B, 2160, 3840, 3), dtype=torch.uint8, device="cuda", generator=g)
out = torch.empty_like(img)
for ks in (7, 5, 3):
    for _ in range(2):
        ops.median_blur(img, ks, out=out)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    for _ in range(5):
        ops.median_blur(img, ks, out=out)
    e[1].record(); torch.cuda.synchronize()
    ms = e[0].elapsed_time(e[1]) / 5
    print("%s median%d B=%d 4K: %.3f ms  %.1f GB/s (6 B/px)" % (tag, ks, B, ms, 6 * img.numel() / 3 / ms / 1e6))
