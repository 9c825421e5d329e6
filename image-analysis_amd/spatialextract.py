"""Drop-in replacement for scripts/spatialextract.py (class SpatialExtractor, :5-173).

Same constructor, methods and return value ({segment: {'NW'|'N'|...: {other: ratio}}}); the
per-pixel work (bounding boxes, sector label counts) runs in libimgan_hip.so.

How the reference's geometry is kept:
  * the bounding box is cv2.boundingRect of the segment mask (:9-11,47: findContours(...)[0]
    is the image under the OpenCV 3 API the reference is written for, and boundingRect of an
    8-bit image is the box of its non-zero pixels);
  * the enlarged box, the five points per edge and the line coefficients (:48-69, :19-23) are
    computed on the host in float64 exactly as written (Python 2 integer division where both
    operands are ints);
  * each sector is the set of rows [lo(x), hi(x)) per column x that the loops at :71-171
    enumerate; the kernel evaluates int(a*x + b) in float64 in the same operation order.
"""
import ctypes

import numpy as np
import torch

from imgan import ops
from imgan._lib import lib, check

SECTORS = ['NW', 'N', 'NE', 'E', 'SE', 'S', 'SW', 'W']
CONST, LINE, LINE_MAX, LINE_MIN = 0, 1, 2, 3


def sector_pieces(x, y, w, h, im_h, im_w, c_param):
    """spatialextract.py:48-171 for one bounding box: list of
    (sector, xa, xb, (lo_kind, lo_a, lo_b, lo_c), (hi_kind, hi_a, hi_b, hi_c))."""
    x_cntr = x + w / 2.0
    y_cntr = y + h / 2.0
    new_h = np.sqrt(1 + c_param) * h
    new_w = np.sqrt(1 + c_param) * w
    new_x = int(x_cntr - new_w / 2)
    new_y = int(y_cntr - new_h / 2)
    new_h = int(new_h)
    new_w = int(new_w)
    new_x = new_x if new_x > 0 else 0
    new_y = new_y if new_y > 0 else 0
    new_w = new_w if (new_x + new_w) < im_w else (im_w - new_x - 1)
    new_h = new_h if (new_y + new_h) < im_h else (im_h - new_y - 1)
    new_x_cntr = new_x + new_w / 2.0
    new_y_cntr = new_y + new_h / 2.0
    ys = np.linspace(new_y, new_y + new_h, num=5)
    xs = np.linspace(new_x, new_x + new_w, num=5)
    point_l_y = [(y_p, new_x) for y_p in ys]
    point_r_y = [(y_p, new_x + new_w) for y_p in ys]
    point_u_x = [(new_y, x_p) for x_p in xs]
    point_d_x = [(new_y + new_h, x_p) for x_p in xs]
    centre = [new_y_cntr, new_x_cntr]

    def coeffs(pt):
        with np.errstate(divide="ignore", invalid="ignore"):
            a = (np.float64(centre[0]) - pt[0]) / (np.float64(centre[1]) - pt[1])
            b = pt[0] - a * pt[1]
        return float(a), float(b)

    half = new_x + new_w // 2
    bottom = new_y + new_h
    pieces = []

    def add(sector, xa, xb, lo, hi):
        pieces.append((SECTORS.index(sector), int(xa), int(xb), lo, hi))

    def line(pt, kind=LINE, c=0):
        a, b = coeffs(pt)
        return (kind, a, b, int(c))

    def const(c):
        return (CONST, 0.0, 0.0, int(c))

    add('NW', new_x, half, line(point_u_x[1], LINE_MAX, new_y), line(point_l_y[1]))        # :72-79
    add('N', int(point_u_x[1][1]), int(point_u_x[2][1]), const(new_y),
        line(point_u_x[1], LINE_MAX, new_y))                                               # :87-90
    add('N', int(point_u_x[2][1]), int(point_u_x[3][1]), const(new_y),
        line(point_u_x[3], LINE_MAX, new_y))                                               # :91-94
    add('NE', half, new_x + new_w, line(point_u_x[3], LINE_MAX, new_y), line(point_r_y[1]))  # :99-106
    add('E', half, new_x + new_w, line(point_r_y[1]), line(point_r_y[3]))                  # :111-118
    add('SE', half, new_x + new_w, line(point_r_y[3]), line(point_d_x[3], LINE_MIN, bottom))  # :123-130
    add('S', int(point_d_x[1][1]), int(point_d_x[2][1]), line(point_d_x[1]), const(bottom))  # :138-141
    add('S', int(point_d_x[2][1]), int(point_d_x[3][1]), line(point_d_x[3]), const(bottom))  # :142-145
    add('SW', new_x, half, line(point_l_y[3]), line(point_d_x[1], LINE_MIN, bottom))       # :150-157
    add('W', new_x, half, line(point_l_y[1]), line(point_l_y[3]))                          # :162-169
    return pieces


class SpatialExtractor(object):

    def __init__(self, clusters):
        # spatialextract.py:14-17
        if not isinstance(clusters, np.ndarray) or clusters.ndim != 2:
            raise ValueError("clusters must be an H x W label map")
        self.clusters = clusters
        self.segments = set(clusters.flatten())
        self.segments.discard(-1)
        # dense ids for every value in the map (-1 included: ratio_indexing counts it, :33-37)
        self._values, inverse = np.unique(clusters, return_inverse=True)
        self._dense = torch.from_numpy(inverse.reshape(clusters.shape).astype(np.int32)).cuda()
        self._bbox = None

    def contours_generator(self):
        """spatialextract.py:7-12.  Yields (mask image, mask, segment): the first element is
        what findContours(...)[0] is under the OpenCV 3 API, the 8-bit mask itself."""
        for segment_idx in self.segments:
            idx = 1 * (self.clusters == segment_idx)
            yield idx.astype(np.uint8), idx, segment_idx

    @staticmethod
    def linear_coeffs(point_o, point_t):
        """spatialextract.py:19-23."""
        a = (point_o[0] - point_t[0]) / (point_o[1] - point_t[1])
        b = point_t[0] - a * point_t[1]
        return a, b

    def ratio_indexing(self, nw, curr_idx):
        """spatialextract.py:25-38 for an explicit point list (host numpy; find_spatial does
        the same counting on the GPU without materialising the lists)."""
        np_idx = np.zeros(self.clusters.shape, dtype=bool)
        for point in nw:
            np_idx[point] = True
        segs_in_reg = self.clusters[np_idx]
        vals, cnts = np.unique(segs_in_reg, return_counts=True)
        ratio_rel = {}
        for sngl_seg, c in zip(vals, cnts):
            if sngl_seg != curr_idx:
                ratio_rel[int(sngl_seg)] = int(c) / float((self.clusters == sngl_seg).sum())
        return ratio_rel

    def bounding_boxes(self):
        """{segment: (x, y, w, h)} as cv2.boundingRect gives them (:47)."""
        nlab = len(self._values)
        H, W = self.clusters.shape
        bbox = torch.empty((nlab, 4), dtype=torch.int32, device=self._dense.device)
        check(lib.imgan_label_bbox(ops._ptr(self._dense), 1, H, W, nlab, ops._ptr(bbox),
                                   ops._stream()))
        bb = bbox.cpu().numpy()
        out = {}
        for i, v in enumerate(self._values):
            if v != -1:
                xmin, ymin, xmax, ymax = (int(t) for t in bb[i])
                out[v] = (xmin, ymin, xmax - xmin + 1, ymax - ymin + 1)
        return out

    def find_spatial(self, c_param):
        """spatialextract.py:40-173."""
        im_h, im_w = self.clusters.shape
        nlab = len(self._values)
        boxes = self.bounding_boxes()
        segs = list(self.segments)
        pi, pd = [], []
        for slot, idx in enumerate(segs):
            for sector, xa, xb, lo, hi in sector_pieces(*boxes[idx], im_h, im_w, c_param):
                pi.append([slot, sector, xa, xb, lo[0], hi[0], lo[3], hi[3]])
                pd.append([lo[1], lo[2], hi[1], hi[2]])
        dev = self._dense.device
        counts = torch.zeros((max(len(segs), 1), 8, nlab), dtype=torch.int32, device=dev)
        oob = torch.zeros(1, dtype=torch.int32, device=dev)
        if pi:
            pi_np = np.asarray(pi, dtype=np.int32)
            pd_np = np.nan_to_num(np.asarray(pd, dtype=np.float64), nan=0.0, posinf=0.0, neginf=0.0)
            max_cols = int(max(0, (pi_np[:, 3] - pi_np[:, 2]).max()))
            dpi, dpd = torch.from_numpy(pi_np).to(dev), torch.from_numpy(pd_np).to(dev)
            check(lib.imgan_spatial_sectors(ops._ptr(self._dense), im_h, im_w, nlab, ops._ptr(dpi),
                                            ops._ptr(dpd), len(pi), max_cols, ops._ptr(counts),
                                            ops._ptr(oob), ops._stream()))
        if int(oob.item()):
            raise IndexError("sector reaches outside the image")
        cnt = counts.cpu().numpy()
        sizes = np.bincount(self._dense.cpu().numpy().ravel(), minlength=nlab)
        relations = {}
        for slot, idx in enumerate(segs):
            local_rels = {}
            for d, name in enumerate(SECTORS):
                rel = {}
                for li in np.nonzero(cnt[slot, d])[0]:
                    v = self._values[li]
                    if v != idx:
                        rel[int(v)] = int(cnt[slot, d, li]) / float(sizes[li])
                local_rels[name] = rel
            relations[int(idx)] = local_rels
        return relations
