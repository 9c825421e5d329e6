"""Batch pipeline for device-resident image batches (BASELINE.json configs[1]):

    BGR -> Lab -> SLIC (slic.py) -> dominant colours per superpixel (mpeg7Descriptors.py:76-130)
        + whole-image colour layout / edge histogram / colour structure

Everything runs in libimgan_hip.so on the current stream; no host round trip except the
per-round "segments still running" counter of the dominant-colour loop.  The outputs are the
fixed-width descriptor rows that the multi-GPU layout all-gathers (imgan.shard)."""
import ctypes

import torch

from . import ops
from ._lib import lib, check


class FullPipeline(object):
    def __init__(self, B, H, W, num_clusters, colour_weight=15, num_iter=10, dominant=8,
                 csd_bins=256, dcd_seed=12345, dcd_check_every=4, dcd_max_rounds=200):
        self.B, self.H, self.W = int(B), int(H), int(W)
        self.cw, self.num_iter, self.k = colour_weight, int(num_iter), int(dominant)
        self.seed, self.check_every, self.max_rounds = dcd_seed, dcd_check_every, dcd_max_rounds
        self.slic = ops.SlicPlan(B, H, W, num_clusters)
        self.n = self.slic.n
        dev = self.slic.device
        self.lab = torch.empty((self.B, self.H, self.W, 3), dtype=torch.uint8, device=dev)
        self.desc = ops.DescriptorPlan(B, H, W, csd_bins)
        self.dcd = ops.DcdPlan(B, H, W, self.n, self.k)
        self.centres = torch.empty((self.B, self.n, self.k, 3), dtype=torch.float32, device=dev)
        self.features = torch.empty((self.B, self.n, self.k, 8), dtype=torch.float32, device=dev)
        self.assign_events = []
        self.dcd_rounds_run = 0

    def run_slic(self, bgr, timed=False):
        ops.bgr2lab(bgr, out=self.lab)
        self.slic.init(self.lab)
        for _ in range(1, self.num_iter):
            if timed:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                self.slic.round(self.lab, self.cw, events=ev)
                self.assign_events.append(ev)
            else:
                self.slic.round(self.lab, self.cw)
        return self.slic.labels

    def run_dcd(self):
        labels = self.slic.labels
        p = ops._ptr
        self.dcd.kmeans(self.lab, labels, self.centres, seed=self.seed, max_rounds=self.max_rounds,
                        check_every=self.check_every)
        self.dcd_rounds_run = self.dcd.rounds_run
        stats = self.dcd.statistics(self.lab, labels)
        check(lib.imgan_dcd_features(p(stats), p(self.centres), self.B, self.n, self.k,
                                     p(self.features), ops._stream()))
        return self.features

    def step(self, bgr, timed=False):
        """One pass of the whole path over the resident batch.
        Returns (image_rows int32 [B, 12+80+16+csd_bins+1], region_rows float32 [B, n*k*8])."""
        self.run_slic(bgr, timed)
        cld = self.desc.colour_layout(bgr)
        ehd = self.desc.edge_histogram(bgr)
        csd = self.desc.colour_structure(bgr)
        feats = self.run_dcd()
        image_rows = torch.cat([cld[:, 0, :6], cld[:, 1, :3], cld[:, 2, :3], ehd, csd], dim=1)
        return image_rows, feats.view(self.B, -1)
