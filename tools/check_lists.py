"""The position lists of the dominant colours' sort pass against the label map: every segment's run must
hold exactly the positions of its pixels.  usage: IMGAN_DCD_STOP_AFTER_SORT=1 python tools/check_lists.py [H W K reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-analysis_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from bench import synth_batch
from imgan.pipeline import FullPipeline
assert os.environ.get("IMGAN_DCD_STOP_AFTER_SORT"), "set IMGAN_DCD_STOP_AFTER_SORT=1"
H, W, K, reps = (int(v) for v in (sys.argv[1:5] + ["1050", "1573", "200", "5"][len(sys.argv) - 1:]))
B = 1
bgr = synth_batch(B, H, W, K, 7, "cuda")
pipe = FullPipeline(B, H, W, K, colour_weight=15, num_iter=10, dominant=8)
pipe.run_slic(bgr); torch.cuda.synchronize()
labels = pipe.slic.labels.cpu().numpy()[0]
nseg = pipe.n
al = lambda n: (n + 255) // 256 * 256
bad = 0
for rep in range(reps):
    pipe.dcd.list_scratch.fill_(0xAB)
    pipe.run_dcd(); torch.cuda.synchronize()
    raw = pipe.dcd.list_scratch.cpu().numpy()
    o = 256
    offs = raw[o:o + nseg * 4].view(np.uint32); o += al(nseg * 4)
    ends = raw[o:o + nseg * 4].view(np.uint32); o += al(nseg * 4)
    cur = raw[o:o + nseg * 4].view(np.uint32); o += al(nseg * 4)
    o += 4 * al(nseg * 4)
    sxy = raw[o:o + H * W * 4].view(np.uint32)
    hdr = raw[0:256].view(np.int32)
    o2 = 256 + 3 * al(nseg * 4)
    tiers = [raw[o2 + t * al(nseg * 4): o2 + t * al(nseg * 4) + nseg * 4].view(np.int32) for t in range(4)]
    listed = []
    for t in range(3):
        ids = tiers[t][:hdr[t]]
        if hdr[t] < 0 or hdr[t] > nseg or ((ids < 0) | (ids >= nseg)).any():
            bad += 1; print("rep %d tier %d: hdr %d ids %s" % (rep, t, hdr[t], ids[:8]), flush=True)
        if (tiers[t][max(hdr[t], 0):] != np.int32(-1414812757)).any():          # 0xABABABAB
            bad += 1; print("rep %d tier %d: entries beyond the list were written" % (rep, t), flush=True)
        listed += list(ids)
    if len(set(listed)) != len(listed):
        bad += 1; print("rep %d: a segment is on two tier lists" % rep, flush=True)
    covered = np.zeros(H * W, dtype=bool)
    for s in range(nseg):
        covered[offs[s]:min(int(offs[s]) + int(cur[s]), int(ends[s]))] = True
    stray = np.nonzero(~covered & (sxy != np.uint32(0xABABABAB)))[0]
    if len(stray):
        bad += 1; print("rep %d: %d entries outside every run were written, first at %d" % (rep, len(stray), stray[0]), flush=True)
    print("rep %d hdr %s tickets %s" % (rep, hdr[:4], hdr[8:12]), flush=True)
    for s in range(nseg):
        ys, xs = np.nonzero(labels == s)
        want = np.sort((ys.astype(np.uint32) << 16) | xs.astype(np.uint32))
        n = int(cur[s])
        got = np.sort(sxy[offs[s]:offs[s] + min(n, int(ends[s] - offs[s]))])
        if n != len(want) or not np.array_equal(got, want):
            bad += 1
            extra = np.setdiff1d(got, want); missing = np.setdiff1d(want, got)
            print("rep %d segment %d: cursor %d, pixels %d, run %d..%d, %d entries not of the segment %s, %d missing %s" %
                  (rep, s, n, len(want), offs[s], ends[s], len(extra), [hex(v) for v in extra[:4]], len(missing), [hex(v) for v in missing[:4]]), flush=True)
print("checked %d x %d segments: %d bad" % (reps, nseg, bad))
