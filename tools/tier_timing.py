"""Phase clocks of the dedup tier kernels (dcd_tiers.hip), thread 0 of every workgroup, summed over
the segments.  Needs an experiment build: bash tools/exp_build.sh timing -DIMGAN_TIER_TIMING dcd_tiers.hip
then on the GPU box: IMGAN_LIB=.../imgan/exp/timing.so python tools/tier_timing.py [B]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-analysis_amd")); sys.path.insert(0, ROOT)
import torch
from bench import synth_batch
from imgan.pipeline import FullPipeline
from imgan import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
bgr = synth_batch(B, 1080, 1920, 400, 1234, "cuda")
pipe = FullPipeline(B, 1080, 1920, 400)
pipe.run_slic(bgr)
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 32)()
pipe.run_dcd()
lib.imgan_debug_tier_timing(buf, 1)
pipe.run_dcd()
torch.cuda.synchronize()
lib.imgan_debug_tier_timing(buf, 1)
names = ["init + clear", "load + table", "close-up", "assign", "barrier", "fold + update", "statistics",
         "positions + assign out", "-", "cc: clear grid", "cc: cluster bytes", "cc: links + join list",
         "cc: parents", "cc: unions", "cc: roots", "cc: clear counters", "cc: count", "cc: largest"]
nseg = buf[31]
tot = sum(buf[:len(names)])
for n, v in zip(names, buf):
    print("%-24s %12d ticks  %5.1f%%  %6.2f us / segment" % (n, v, 100.0 * v / max(tot, 1), v / 100.0 / max(nseg, 1)))
print("segments %d, per segment %.1f us of its workgroup's time (100 MHz clock)" % (nseg, tot / 100.0 / max(nseg, 1)))
for t in range(3):
    first, last, ends, wgs = ~buf[18 + 4 * t] & (2 ** 64 - 1), buf[19 + 4 * t], buf[20 + 4 * t], buf[21 + 4 * t]
    if wgs:
        print("tier %d: %4d workgroups, launch %.1f us, mean end %.1f us before the last one" %
              (t, wgs, (last - first) / 100.0, (last - ends / wgs) / 100.0))
rounds, _ = pipe.dcd.rounds()
print("mean rounds per segment %.2f" % float(rounds.float().mean()))
