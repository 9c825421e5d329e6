"""Batch pipeline for device-resident image batches (BASELINE.json configs[1]):

    BGR -> Lab -> SLIC (slic.py) -> dominant colours per superpixel (mpeg7Descriptors.py:76-130)
        + whole-image colour layout / edge histogram / colour structure

Everything runs in libimgan_hip.so on the current stream, with no host round trip.  The outputs are the
fixed-width descriptor rows that the multi-GPU layout all-gathers (imgan.shard)."""
import ctypes
import os

import torch

from . import ops
from ._lib import lib, check


class FullPipeline(object):
    def __init__(self, B, H, W, num_clusters, colour_weight=15, num_iter=10, dominant=8,
                 csd_bins=256, dcd_seed=12345, dcd_check_every=4, dcd_max_rounds=200,
                 per_region=False, want_assign=False):
        """per_region: colour layout / edge histogram / colour structure of every superpixel
        (configs[4]) instead of the whole image.  want_assign: also write the per-pixel
        dominant-colour index map (self.dcd.assign), which the descriptor rows do not need."""
        self.want_assign = bool(want_assign)
        # IMGAN_REGION_BY_BOX=1: per-region descriptors by the scan of the boxes (for comparisons)
        self.region_by_box = os.environ.get("IMGAN_REGION_BY_BOX") == "1"
        self.B, self.H, self.W = int(B), int(H), int(W)
        self.cw, self.num_iter, self.k = colour_weight, int(num_iter), int(dominant)
        self.seed, self.check_every, self.max_rounds = dcd_seed, dcd_check_every, dcd_max_rounds
        self.slic = ops.SlicPlan(B, H, W, num_clusters)
        self.n = self.slic.n
        dev = self.slic.device
        self.lab = torch.empty((self.B, self.H, self.W, 3), dtype=torch.uint8, device=dev)
        self.per_region = bool(per_region)
        self.desc = ops.DescriptorPlan(B, H, W, csd_bins) if not per_region else None
        self.region_desc = ops.RegionDescriptorPlan(B, H, W, self.slic.n, csd_bins) if per_region else None
        self.dcd = ops.DcdPlan(B, H, W, self.n, self.k)
        self.centres = torch.empty((self.B, self.n, self.k, 3), dtype=torch.float32, device=dev)
        self.features = torch.empty((self.B, self.n, self.k, 8), dtype=torch.float32, device=dev)
        self.assign_events = []
        self.stage_events = {}
        self.dcd_rounds_run = 0

    def _timed(self, name, timed, fn):
        """Runs fn(); with timed, brackets it with events on the current (launch) stream."""
        if not timed:
            return fn()
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
        out = fn()
        ev[1].record()
        self.stage_events.setdefault(name, []).append(ev)
        return out

    def run_slic(self, bgr, timed=False):
        self._timed("bgr2lab", timed, lambda: ops.bgr2lab(bgr, out=self.lab))
        # with at least one round the first one writes the whole label map: no fill before it
        if self.num_iter > 1:
            self.slic.seed(self.lab)
        else:
            self.slic.init(self.lab)
        for it in range(1, self.num_iter):
            # the last round also leaves the bounding boxes of the superpixels for run_dcd
            box = self.dcd.bbox if (it == self.num_iter - 1 and self.dcd is not None) else None
            if timed:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                self.slic.round(self.lab, self.cw, events=ev, bbox=box, first=(it == 1))
                self.stage_events.setdefault("slic_assign", []).append(ev)
            else:
                self.slic.round(self.lab, self.cw, bbox=box, first=(it == 1))
        self.boxed = self.dcd is not None and self.num_iter > 1
        if self.boxed and self.dcd.lists:
            # ... and the pixel count of every superpixel (the divisor of its centre update)
            self.slic.label_counts(out=self.dcd.counts)
        return self.slic.labels

    def run_dcd(self, timed=False):
        labels = self.slic.labels
        p = ops._ptr

        def go():
            # a superpixel normally lies inside its centre's window, at most (2S)^2 pixels: the
            # count of segments beyond the on-chip capacity is not read here (no synchronisation
            # in a step); finish() settles it
            self._timed("dcd_segments", timed, lambda: self.dcd.kmeans(
                self.lab, labels, self.centres, seed=self.seed, max_rounds=self.max_rounds,
                check_every=self.check_every, boxed=getattr(self, "boxed", False),
                counted=getattr(self, "boxed", False) and self.dcd.lists,
                max_segment_px=(2 * self.slic.S) ** 2, write_assign=self.want_assign))
            self.dcd_rounds_run = self.dcd.rounds_run
            stats = self.dcd.statistics(self.lab, labels)
            check(lib.imgan_dcd_features(p(stats), p(self.centres), self.B, self.n, self.k,
                                         p(self.features), ops._stream()))
        self._timed("dcd", timed, go)
        return self.features

    def step(self, bgr, timed=False):
        """One pass of the whole path over the resident batch.
        Returns (image_rows int32 [B, 12+80+16+csd_bins+1], region_rows float32 [B, n*k*8])."""
        self.run_slic(bgr, timed)
        if self.per_region:
            # rows per superpixel [B, n, 12+96+csd_bins+1]; the boxes are the last SLIC round's.  The
            # dominant colours go first: their sort pass leaves every superpixel's pixels as one run
            # of positions, which the descriptor kernel walks instead of scanning the boxes
            listed = bool(self.dcd.lists) and not self.region_by_box
            if listed:
                feats = self.run_dcd(timed)
            cld, ehd, csd = self._timed("region_desc", timed, lambda: self.region_desc.run(
                bgr, self.slic.labels, bbox=self.dcd.bbox if (self.boxed or listed) else None,
                lists=self.dcd.list_scratch if listed else None))
            if not listed:
                feats = self.run_dcd(timed)
            rows = torch.cat([cld[:, :, 0, :6], cld[:, :, 1, :3], cld[:, :, 2, :3], ehd, csd], dim=2)
            return rows.view(self.B, -1), feats.view(self.B, -1)
        cld, ehd = self._timed("cld+ehd", timed, lambda: self.desc.colour_layout_and_edge_histogram(bgr))
        csd = self._timed("csd", timed, lambda: self.desc.colour_structure(bgr))
        feats = self.run_dcd(timed)
        image_rows = torch.cat([cld[:, 0, :6], cld[:, 1, :3], cld[:, 2, :3], ehd, csd], dim=1)
        return image_rows, feats.view(self.B, -1)

    def finish(self):
        """Settles the last step (one synchronisation): superpixels that did not fit the on-chip
        dominant-colour pass (a stale label stretched their box or count, see DcdPlan.kmeans) go
        through the streaming rounds, and their statistics and features are redone.  Returns the
        number of such superpixels (0 on every batch measured so far); the features returned
        by step() are final when it is 0."""
        left = self.dcd.finish_large(self.lab, self.slic.labels, self.centres,
                                     max_rounds=self.max_rounds, check_every=self.check_every)
        if left:
            stats = self.dcd.statistics(self.lab, self.slic.labels)
            check(lib.imgan_dcd_features(ops._ptr(stats), ops._ptr(self.centres), self.B, self.n,
                                         self.k, ops._ptr(self.features), ops._stream()))
        return left

    def stage_times_ms(self):
        """Mean device time per C call of every stage recorded with timed=True."""
        torch.cuda.synchronize()
        return {k: sum(a.elapsed_time(b) for a, b in v) / len(v)
                for k, v in self.stage_events.items()}

    def descriptor_bytes_per_px(self):
        """Algorithmic bytes per pixel of the CLD / EHD / CSD stage as this pipeline runs it
        (SURVEY 8d): whole-image descriptors read the image only, 3 B/px each; the per-region
        form reads image + labels, 7 B/px each (one fused pass here, priced as the survey does)."""
        return 21 if self.per_region else 9

    def segments_at_round_cap(self):
        """Segments of the last dominant-colour pass whose Lloyd loop was cut at dcd_max_rounds
        (the reference's loop is unbounded, mpeg7Descriptors.py:85-107)."""
        rounds, _ = self.dcd.rounds()
        return int((rounds >= self.max_rounds).sum().item())

    def mean_lloyd_rounds(self):
        """Pixel-weighted mean number of Lloyd rounds of the last dominant-colour pass."""
        rounds, _ = self.dcd.rounds()
        npx = self.dcd.stats[..., 0].sum(dim=2).to(torch.float64)
        tot = float(npx.sum().item())
        r = float((rounds.to(torch.float64) * npx).sum().item()) / max(tot, 1.0)
        return max(r, float(self.dcd_rounds_run))


class RagPipeline(object):
    """Batch form of RAGSegmentation's sequence (rag_segmentation.py:10-25, 38-184) for
    device-resident batches (BASELINE.json configs[2]): median 7x7 -> Lab -> SLIC -> per-label
    means + adjacency -> MST merge with the leave-one-out cut -> relabel.  Every stage is a kernel
    of libimgan_hip.so on the current stream; with at most 4096 labels per image the merge runs
    on the device too (ops.RagPlan) and a step enqueues without a host round trip.  Returns the
    merged label maps (int32, 0 = the unlabelled border) and the relabel tables."""

    def __init__(self, B, H, W, num_clusters, colour_weight=15, num_iter=10, c_factor=0.65,
                 median_blur=7):
        self.B, self.H, self.W = int(B), int(H), int(W)
        self.cw, self.num_iter, self.c_factor = colour_weight, int(num_iter), float(c_factor)
        self.ksize = int(median_blur)
        self.slic = ops.SlicPlan(B, H, W, num_clusters)
        # SLIC hands out n = nx * ny labels, which can exceed the requested count at small steps
        # (39 x 39, K = 96 -> S = 3, n = 144): size the tables for both
        self.K = max(int(num_clusters), self.slic.n)
        dev = self.slic.device
        self.blur = torch.empty((self.B, self.H, self.W, 3), dtype=torch.uint8, device=dev)
        self.lab = torch.empty_like(self.blur)
        self.merged = torch.empty((self.B, self.H, self.W), dtype=torch.int32, device=dev)
        self.rag = ops.RagPlan(B, self.K) if self.K <= ops.RagPlan.MAX_K else None
        self.stage_events = {}
        self.host_ms = {}
        self.redone = 0

    _timed = FullPipeline._timed

    def step(self, bgr, timed=False, check=True):
        """check=False leaves the status words unread (no synchronisation at all): the caller
        looks at self.rag.status later, as bench.py does after its timed loop."""
        import time
        self._timed("median", timed, lambda: ops.median_blur(bgr, self.ksize, out=self.blur))
        self._timed("bgr2lab", timed, lambda: ops.bgr2lab(self.blur, out=self.lab))
        if self.num_iter > 1:
            self.slic.seed(self.lab)
        else:
            self.slic.init(self.lab)
        for it in range(1, self.num_iter):
            if timed:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                self.slic.round(self.lab, self.cw, events=ev, first=(it == 1))
                self.stage_events.setdefault("slic_assign", []).append(ev)
            else:
                self.slic.round(self.lab, self.cw, first=(it == 1))
        labels = self.slic.labels
        if self.rag is not None:
            luts, _ = self._timed("rag_accumulate+merge", timed,
                                  lambda: self.rag.run(self.lab, labels, self.c_factor, slic=self.slic))
            if check:
                self.redone += self.rag.finish(self.lab, labels, self.c_factor)
        else:                                   # more labels than the device merge holds in LDS
            t0 = time.perf_counter()
            sums, edges = ops.rag_accumulate(self.lab, labels, self.K)      # synchronises
            t1 = time.perf_counter()
            luts = torch.from_numpy(ops.rag_merge_batch(self.K, edges, sums, self.c_factor)).to(labels.device)
            if timed:
                self.host_ms.setdefault("rag_accumulate+edge lists", []).append((t1 - t0) * 1e3)
                self.host_ms.setdefault("host merge", []).append((time.perf_counter() - t1) * 1e3)
        self._timed("relabel", timed, lambda: check_call_relabel(labels, luts, self.merged))
        return self.merged, luts

    stage_times_ms = FullPipeline.stage_times_ms


def check_call_relabel(labels, luts, out):
    check(lib.imgan_relabel_lut(ops._ptr(labels), ops._ptr(luts), ops._ptr(out), labels.shape[0],
                                labels[0].numel(), luts.shape[1], ops._stream()))
    return out
