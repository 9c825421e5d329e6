"""TEST INFRASTRUCTURE -- CPU restatement of scripts/spatialextract.py:5-173 (SpatialExtractor).

Literal: the same point lists, built by the same loops (Python 2 semantics restated: `/` between
ints is floor division, xrange -> range, print dropped).  Pure-Python pixel loops, so only for
small label maps.  Parity unpinned by the reference (no fixtures; needs OpenCV): the bounding
box is restated as the box of the mask's non-zero pixels, which is what
cv2.boundingRect(cv2.findContours(mask, ...)[0]) returns under the OpenCV 3 API
(spatialextract.py:10-11,47), where element 0 is the (unmodified) image.
"""
import numpy as np


class SpatialOracle(object):

    def __init__(self, clusters):
        self.clusters = clusters                                  # :15
        self.segments = set(clusters.flatten())                   # :16
        self.segments.discard(-1)                                 # :17

    @staticmethod
    def linear_coeffs(point_o, point_t):                          # :19-23
        with np.errstate(divide="ignore", invalid="ignore"):
            a = (point_o[0] - point_t[0]) / (point_o[1] - point_t[1])
            b = point_t[0] - a * point_t[1]
        return a, b

    def ratio_indexing(self, nw, curr_idx):                       # :25-38
        np_idx = np.zeros(self.clusters.shape, dtype=bool)
        for point in nw:
            np_idx[point] = True
        segs_in_reg = self.clusters[np_idx]
        segs_set = set(segs_in_reg)
        ratio_rel = {}
        for sngl_seg in segs_set:
            if sngl_seg != curr_idx:
                s_idx = np.where(self.clusters == sngl_seg)
                ratio = segs_in_reg.tolist().count(sngl_seg) / float(s_idx[0].size)
                ratio_rel[int(sngl_seg)] = ratio
        return ratio_rel

    @staticmethod
    def bounding_rect(mask):
        ys, xs = np.nonzero(mask)
        x, y = int(xs.min()), int(ys.min())
        return x, y, int(xs.max()) - x + 1, int(ys.max()) - y + 1

    def find_spatial(self, c_param):                              # :40-173
        """Every sector is rebuilt as the explicit point list of the reference's loops."""
        im_h, im_w = self.clusters.shape
        root = np.sqrt(1 + c_param)
        relations = {}
        for idx in self.segments:
            x, y, w, h = self.bounding_rect(self.clusters == idx)               # :47
            bw, bh = root * w, root * h                                          # :52-53
            bx, by = int(x + w / 2.0 - bw / 2), int(y + h / 2.0 - bh / 2)        # :49-56
            bw, bh = int(bw), int(bh)                                            # :57-58
            bx, by = max(bx, 0), max(by, 0)                                      # :59-60
            if bx + bw >= im_w:                                                  # :61
                bw = im_w - bx - 1
            if by + bh >= im_h:                                                  # :62
                bh = im_h - by - 1
            centre = [by + bh / 2.0, bx + bw / 2.0]                              # :63-64
            rows5 = np.linspace(by, by + bh, num=5)                              # :66-69
            cols5 = np.linspace(bx, bx + bw, num=5)
            left = [(r, bx) for r in rows5]
            right = [(r, bx + bw) for r in rows5]
            top = [(by, c) for c in cols5]
            bottom = [(by + bh, c) for c in cols5]
            mid, end, floor = bx + bw // 2, bx + bw, by + bh                     # Py2 int division

            def ray(pt):
                a, b = self.linear_coeffs(centre, [pt[0], pt[1]])
                return lambda col: int(a * col + b)

            def not_above(f):                                                    # "... > new_y else new_y"
                return lambda col: f(col) if f(col) > by else by

            def not_below(f):                                                    # "... < new_y+new_h else ..."
                return lambda col: f(col) if f(col) < floor else floor

            def fixed(v):
                return lambda col: v

            c1, c2, c3 = int(top[1][1]), int(top[2][1]), int(top[3][1])
            d1, d2, d3 = int(bottom[1][1]), int(bottom[2][1]), int(bottom[3][1])
            # sector -> [(columns, first row, one-past-last row)]
            table = [
                ('NW', [(range(bx, mid), not_above(ray(top[1])), ray(left[1]))]),            # :72-79
                ('N', [(range(c1, c2), fixed(by), not_above(ray(top[1]))),                   # :87-90
                       (range(c2, c3), fixed(by), not_above(ray(top[3])))]),                 # :91-94
                ('NE', [(range(mid, end), not_above(ray(top[3])), ray(right[1]))]),          # :99-106
                ('E', [(range(mid, end), ray(right[1]), ray(right[3]))]),                    # :111-118
                ('SE', [(range(mid, end), ray(right[3]), not_below(ray(bottom[3])))]),       # :123-130
                ('S', [(range(d1, d2), ray(bottom[1]), fixed(floor)),                        # :138-141
                       (range(d2, d3), ray(bottom[3]), fixed(floor))]),                      # :142-145
                ('SW', [(range(bx, mid), ray(left[3]), not_below(ray(bottom[1])))]),         # :150-157
                ('W', [(range(bx, mid), ray(left[1]), ray(left[3]))]),                       # :162-169
            ]
            local_rels = {}
            for name, parts in table:
                pts = [(row, col) for cols, first, last in parts for col in cols
                       for row in range(first(col), last(col))]
                local_rels[name] = self.ratio_indexing(pts, idx)
            relations[int(idx)] = local_rels
        return relations
