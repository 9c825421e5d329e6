"""Wall time of the dominant-colour stage alone (experiment builds: the caller copies the
variant library over imgan/libimgan_hip.so on the GPU box before starting this)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-analysis_amd")); sys.path.insert(0, ROOT)
import torch
from bench import synth_batch
from imgan.pipeline import FullPipeline
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[3]) if len(sys.argv) > 3 else 400
bgr = synth_batch(B, 1080, 1920, K, 1234, "cuda")
pipe = FullPipeline(B, 1080, 1920, K, dcd_max_rounds=int(os.environ.get('DCD_MAX_ROUNDS', 200)))
if len(sys.argv) > 4 and sys.argv[4] == "nolists":
    from imgan import ops
    pipe.dcd = ops.DcdPlan(B, 1080, 1920, pipe.n, pipe.k, lists=False)
pipe.run_slic(bgr)
pipe.run_dcd()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(5):
    e0.record(); pipe.run_dcd(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print("K=%d" % K, "%s dcd stage B=%d: min %.3f ms  median %.3f ms" % (sys.argv[2] if len(sys.argv) > 2 else "", B, min(ts), sorted(ts)[2]))
