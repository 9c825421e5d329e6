"""Dumps Lab, SLIC labels and per-segment Lloyd round counts of the first image of the bench batch
(experiments on the CPU with the real segment statistics)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-analysis_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from bench import synth_batch
from imgan.pipeline import FullPipeline
bgr = synth_batch(2, 1080, 1920, 400, 1234, "cuda")
pipe = FullPipeline(2, 1080, 1920, 400)
pipe.run_slic(bgr); pipe.run_dcd(); torch.cuda.synchronize()
rounds, done = pipe.dcd.rounds()
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "seg_dump.npz"), lab=pipe.lab[0].cpu().numpy(),
                    labels=pipe.slic.labels[0].cpu().numpy(), rounds=rounds[0].cpu().numpy(),
                    centres=pipe.centres[0].cpu().numpy(), seed=np.int64(pipe.seed))
print("rounds mean %.2f max %d" % (rounds.float().mean().item(), rounds.max().item()))
