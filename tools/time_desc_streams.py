"""Whole-image descriptors of the bench batch: colour structure and colour layout + edge histogram one
after the other on one stream against side by side on two (GPU box): python tools/time_desc_streams.py [B]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-analysis_amd")); sys.path.insert(0, ROOT)
import torch
from bench import synth_batch
from imgan.pipeline import FullPipeline
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
bgr = synth_batch(B, 1080, 1920, 400, 1234, "cuda")
pipe = FullPipeline(B, 1080, 1920, 400)
d = pipe.desc
s2 = torch.cuda.Stream()
def seq():
    d.colour_layout_and_edge_histogram(bgr); d.colour_structure(bgr)
def par():
    ev = torch.cuda.Event(); ev.record()
    with torch.cuda.stream(s2):
        s2.wait_event(ev)
        d.colour_structure(bgr)
        done = torch.cuda.Event(); done.record()
    d.colour_layout_and_edge_histogram(bgr)
    torch.cuda.current_stream().wait_event(done)
for name, fn in (("one stream", seq), ("two streams", par), ("one stream", seq), ("two streams", par)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print("%-12s %.3f ms" % (name, e0.elapsed_time(e1) / 20))
