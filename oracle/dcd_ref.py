"""Oracle: MPEG-7 dominant colour as scripts/mpeg7Descriptors.py computes it
(TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

Follows /root/reference/scripts/mpeg7Descriptors.py:20-26 (constructor),
:76-130 (dominant_colour) and :132-141 (mpeg7_dominant_colours).

Stated deviations:
  * the reference draws the initial centres with an unseeded random.sample
    (:79); here the caller injects the initial point indices.
  * an empty cluster is re-sampled with an unseeded random.sample in the
    reference (:101); here the caller injects, per cluster, the index of the
    point that cluster takes whenever it is empty (resample_idx; without it
    the cluster keeps its centre).  A cluster that is empty at the end
    reports var = NaN, percentage 0, coherence 0 (the reference would raise
    at :128 on max([])).
  * spatial coherence is the true 8-connected component size; the reference's
    flood fill clamps its window at 255 (:52,:54) and is only correct for
    images up to 255 x 255.
Parity unpinned by the reference (no fixture for this path).
"""
import numpy as np
from scipy import ndimage

from .color import bgr2lab_u8, bgr2gray_u8

_EIGHT = np.ones((3, 3), dtype=bool)


def mix32(h):
    """murmur3 finaliser on uint32 values held in uint64 arrays."""
    h = np.asarray(h).astype(np.uint64)
    m = np.uint64(0xffffffff)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85ebca6b)) & m
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xc2b2ae35)) & m
    h ^= h >> np.uint64(16)
    return h


def hash_resample_idx(flat_idx, seed, k):
    """The draws this repository puts in the place of random.sample(segment, 1) at :101 (the
    reference's are unseeded, so any draw is a run the reference could have made): cluster j of a
    segment takes the pixel p = y*W + x with the smallest
        (mix32((p * 0x9e3779b1 + seed) ^ ((j + 1) * 0x85ebca6b)), p)      (uint32 arithmetic).
    flat_idx: the segment's pixels as flat indices in raster order.  Returns, per cluster, the
    index into that list (the resample_idx of lloyd / dominant_colour)."""
    m = np.uint64(0xffffffff)
    p = np.asarray(flat_idx).astype(np.uint64)
    out = []
    for j in range(k):
        salt = np.uint64(((j + 1) * 0x85ebca6b) & 0xffffffff)
        h = mix32(((p * np.uint64(0x9e3779b1) + np.uint64(int(seed) & 0xffffffff)) & m) ^ salt)
        out.append(int(np.lexsort((p, h))[0]))
    return out


def lloyd(points_u8, init_idx, max_rounds=1000, resample_idx=None):
    """mpeg7Descriptors.py:78-107.  points_u8: N x 3 uint8 Lab, init_idx: k indices,
    resample_idx: k indices (the point cluster j takes when it is empty, :101) or None.
    Returns (assignment int32[N], centres float32[k,3], rounds)."""
    segment = np.asarray(points_u8, dtype=np.float32)
    k = len(init_idx)
    centers = [segment[i].copy() for i in init_idx]
    old_cen = [np.zeros(3, dtype=np.float32) for _ in range(k)]
    old_cen_sum = 0.0
    clusters = -np.ones(segment.shape[0])
    rounds = 0
    while True:
        rounds += 1
        distance = np.full(segment.shape[0], np.inf)
        for idc, cntr in enumerate(centers):
            col_difs = segment - cntr
            dist = np.sqrt(np.sum(np.square(col_difs), axis=1))      # float32
            idx = distance > dist
            distance[idx] = dist[idx]
            clusters[idx] = idc
        for c in range(k):
            cl_idx = (clusters == c)
            n = int(np.sum(cl_idx))
            if n:
                s = np.sum(segment[cl_idx], axis=0, dtype=np.float32)
                centers[c] = (s / np.float32(n)).astype(np.float32)
            elif resample_idx is not None:
                centers[c] = segment[resample_idx[c]].copy()         # :101, injected draw
            # else: keeps its centre
        shift = np.zeros(3, dtype=np.float32)
        for cnt, old in zip(centers, old_cen):
            shift = shift + np.abs(cnt - old)                         # float32 adds, cluster order
        cent_sum = float(shift[0]) + float(shift[1]) + float(shift[2])   # float64 adds (:102)
        if abs(old_cen_sum - cent_sum) < 0.1 or rounds >= max_rounds:
            break
        old_cen = [c.copy() for c in centers]
        old_cen_sum = cent_sum
    return clusters.astype(np.int32), np.stack(centers).astype(np.float32), rounds


def dominant_colour(points_u8, seg_idxs, init_idx, shape, resample_idx=None):
    """mpeg7Descriptors.py:76-130 for one segment.
    seg_idxs: N x 2 (row, col) of the segment's pixels; shape: (H, W)."""
    assign, centers, rounds = lloyd(points_u8, init_idx, resample_idx=resample_idx)
    pts = np.asarray(points_u8, dtype=np.float64)
    N = pts.shape[0]
    feats = {}
    for c in range(len(init_idx)):
        m = assign == c
        n = int(m.sum())
        mean = [float(v) for v in centers[c]]
        if n == 0:
            feats[c] = [mean, [float("nan")] * 3, 0.0, 0.0]
            continue
        var = np.var(pts[m], axis=0).tolist()
        mask = np.zeros(shape, dtype=bool)
        mask[seg_idxs[m, 0], seg_idxs[m, 1]] = True
        lab, ncomp = ndimage.label(mask, structure=_EIGHT)
        largest = int(np.bincount(lab.ravel())[1:].max())
        feats[c] = [mean, var, n / float(N), largest / float(n)]
    return feats, rounds


class DCDOracle(object):
    """MPEG7Descriptors restricted to the dominant-colour path."""

    def __init__(self, clusters, image):
        self.image_rgb = image
        self.image_lab = bgr2lab_u8(image)                    # un-blurred image (:22)
        self.image_gray = bgr2gray_u8(image)
        self.clusters = clusters
        self.segments = sorted(set(np.unique(clusters).tolist()) - {-1.0})

    def mpeg7_dominant_colours(self, max_cols, init_fn, resample_fn=None):
        """init_fn(segment_id, N, k) -> k point indices (raster order within the segment);
        resample_fn likewise: the point each cluster takes when it is empty (:101)."""
        out = {}
        self.rounds = {}
        for seg in self.segments:
            idx = (self.clusters == seg)
            seg_idxs = np.transpose(np.nonzero(idx))
            pts = self.image_lab[idx]
            init = init_fn(int(seg), pts.shape[0], max_cols)
            res = resample_fn(int(seg), pts.shape[0], max_cols) if resample_fn else None
            out[int(seg)], self.rounds[int(seg)] = dominant_colour(pts, seg_idxs, init,
                                                                   self.clusters.shape, res)
        return out
