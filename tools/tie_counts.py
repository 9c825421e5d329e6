"""How often the expanded Lloyd distances of the dedup tiers hand an entry to the exact rule.
Needs an experiment build: bash tools/exp_build.sh ties -DIMGAN_TIE_COUNTS dcd_tiers.hip
then on the GPU box: IMGAN_LIB=.../imgan/exp/ties.so python tools/tie_counts.py [B]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-analysis_amd")); sys.path.insert(0, ROOT)
import torch
from bench import synth_batch
from imgan.pipeline import FullPipeline
from imgan import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
bgr = synth_batch(B, 1080, 1920, 400, 1234, "cuda")
pipe = FullPipeline(B, 1080, 1920, 400)
pipe.run_slic(bgr)
pipe.run_dcd()
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 8)()
lib.imgan_debug_tie_counts(buf)
trips, hit_trips, entries, hits, zeros, first = buf[0], buf[1], buf[2], buf[3], buf[4], buf[5]
print("wave trips %d, with a hand-over %d (%.2f%%)" % (trips, hit_trips, 100.0 * hit_trips / max(trips, 1)))
print("entries %d, handed over %d (%.4f%%): %d in first rounds, %d in second rounds, %d with two candidates at one literal distance" %
      (entries, hits, 100.0 * hits / max(entries, 1), first, buf[7], buf[6]))
