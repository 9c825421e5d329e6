"""Oracle: SLIC superpixels exactly as scripts/slic.py computes them
(TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

Follows /root/reference/scripts/slic.py:7-127 with the Python-2 semantics
restated for Python 3 (integer `/` -> `//`, xrange -> range).  Parity
unpinned by the reference (it holds no fixture for this path).

Two distance modes:
  * "float64"  -- literal: sqrt((dc/cw)^2 + (ds/S)^2) in float64, slic.py:101-109.
  * "intkey"   -- the same ordering evaluated exactly over the integers,
                  key = dc^2 * S^2 + ds^2 * cw^2 (needs an integral colour weight).
                  Differs from "float64" only where two candidates have the
                  same real distance but float rounding breaks the tie.
"""
import numpy as np

from .color import bgr2lab_u8


def slic_step(height, width, num_clusters):
    """slic.py:29,33 -- int(sqrt((H*W) / K)) with Python-2 integer division."""
    return int(np.sqrt((height * width) // num_clusters))


def slic_grid(height, width, step):
    """slic.py:42-43 -- seed columns (outer loop) and rows (inner loop)."""
    xs = list(range(step, width - step // 2, step))
    ys = list(range(step, height - step // 2, step))
    return xs, ys


def find_local_minimum(image, center):
    """slic.py:57-76 -- 3x3 search for the smallest forward L-gradient,
    x outer / y inner, strict '<' so the first minimum wins."""
    min_grad = float("inf")
    loc_min = center
    for i in range(center[0] - 1, center[0] + 2):
        for j in range(center[1] - 1, center[1] + 2):
            l_1 = int(image[j + 1, i][0])
            l_2 = int(image[j, i + 1][0])
            l_3 = int(image[j, i][0])
            grad = abs(l_1 - l_3) + abs(l_2 - l_3)
            if grad < min_grad:
                min_grad = grad
                loc_min = (i, j)
    return loc_min


class SLICOracle(object):
    """Restatement of SLICSuperPixel (slic.py:5-160).

    `lab` may be given to skip the BGR->Lab step (all downstream parity is
    stated given identical Lab input).
    """

    def __init__(self, image=None, num_clusters=200, colour_weight=10, num_iter=10,
                 step=None, lab=None, mode="float64", update="bincount"):
        if lab is None:
            if not isinstance(image, np.ndarray):
                raise ValueError("Image value is None. Probably not loaded correctly!")
            lab = bgr2lab_u8(image)
        height, width = lab.shape[0], lab.shape[1]
        self.centers = []
        self.clusters = None
        self.image = np.asarray(lab, dtype="int32")            # slic.py:22
        self.num_iter = num_iter
        self.num_clusters = num_clusters
        self.colour_weight = colour_weight
        self.step = slic_step(height, width, num_clusters) if step is None else int(step)
        self.mode = mode
        self.update = update
        self.history = []          # labels after each round (for fixtures)

    def init_clusters(self):
        """slic.py:35-54 -- column-major seed order, centre = [L, a, b, x, y]."""
        height, width = self.image.shape[0], self.image.shape[1]
        self.centers = []
        xs, ys = slic_grid(height, width, self.step)
        for i in xs:
            for j in ys:
                new_pos = find_local_minimum(self.image, (i, j))
                colour = self.image[new_pos[1], new_pos[0]]
                self.centers.append([int(c) for c in colour] + [new_pos[0], new_pos[1]])

    def _assign(self, clusters, distance):
        """slic.py:88-116 -- one sweep over the centres in index order."""
        H, W = self.image.shape[0], self.image.shape[1]
        S = self.step
        cw = self.colour_weight
        for cntr in range(len(self.centers)):
            cx, cy = int(self.centers[cntr][3]), int(self.centers[cntr][4])
            x_range = [cx - S, cx + S]
            y_range = [cy - S, cy + S]
            x_l = x_range[0] if x_range[0] >= 0 else 0
            x_r = x_range[1] if x_range[1] < W else W - 1      # exclusive bound
            y_l = y_range[0] if y_range[0] >= 0 else 0
            y_r = y_range[1] if y_range[1] < H else H - 1
            if x_r <= x_l or y_r <= y_l:
                continue
            image_piece = self.image[y_l:y_r, x_l:x_r]
            colour_dif = image_piece - self.image[cy, cx]       # colour AT the centre position
            x_coords = np.arange(x_l, x_r)
            y_coords = np.arange(y_l, y_r).reshape(-1, 1)
            if self.mode == "float64":
                dc = np.sqrt(np.sum(np.square(colour_dif), axis=2))
                ds = np.sqrt((x_coords - cx) ** 2 + (y_coords - cy) ** 2)
                eucl = np.sqrt((dc / cw) ** 2 + (ds / S) ** 2)
            else:
                icw = int(cw)
                assert icw == cw, "intkey mode needs an integral colour weight"
                dc2 = np.sum(np.square(colour_dif.astype(np.int64)), axis=2)
                ds2 = ((x_coords - cx) ** 2 + (y_coords - cy) ** 2).astype(np.int64)
                eucl = dc2 * (S * S) + ds2 * (icw * icw)
            dist_pr = distance[y_l:y_r, x_l:x_r]
            idx = dist_pr > eucl
            dist_pr[idx] = eucl[idx]
            clusters[y_l:y_r, x_l:x_r][idx] = cntr

    def _update(self, clusters):
        """slic.py:118-126 -- per-label sums, Python-2 floor division, empty -> 0."""
        n = len(self.centers)
        H, W = clusters.shape
        if self.update == "scan":                               # literal cost structure
            for c in range(n):
                idx = (clusters == c)
                cols = self.image[idx]
                ys, xs = np.where(idx)
                cnt = int(np.sum(idx))
                sums = [int(v) for v in np.sum(cols, axis=0)] + [int(np.sum(xs)), int(np.sum(ys))]
                self.centers[c] = [(s // cnt) if cnt else 0 for s in sums]
            return
        lab = clusters.astype(np.int64).ravel()
        valid = lab >= 0
        lv = lab[valid]
        yy, xx = np.divmod(np.nonzero(valid)[0], W)
        img = self.image.reshape(-1, 3)[valid].astype(np.int64)
        cnt = np.bincount(lv, minlength=n)
        sums = [np.bincount(lv, weights=img[:, k], minlength=n).astype(np.int64)
                for k in range(3)]
        sums.append(np.bincount(lv, weights=xx, minlength=n).astype(np.int64))
        sums.append(np.bincount(lv, weights=yy, minlength=n).astype(np.int64))
        for c in range(n):
            self.centers[c] = [int(s[c]) // int(cnt[c]) if cnt[c] else 0 for s in sums]

    def run(self, keep_history=False):
        """slic.py:78-127."""
        self.init_clusters()
        clusters = -np.ones(self.image.shape[:-1])
        for _ in range(1, self.num_iter):                       # num_iter - 1 rounds
            if self.mode == "float64":
                distance = np.full(self.image.shape[:-1], np.inf)
            else:
                distance = np.full(self.image.shape[:-1], np.iinfo(np.int64).max, dtype=np.int64)
            self._assign(clusters, distance)
            self._update(clusters)
            if keep_history:
                self.history.append(clusters.astype(np.int32))
        self.clusters = clusters
        return clusters

    def get_clusters(self):
        return self.clusters


def enforce_connectivity(labels):
    """The build's own connectivity rule (scripts/slic.py has none; parity unpinned): per label
    the largest 8-connected component stays (ties: the one whose first raster pixel comes first),
    every other component takes the label of the pixel left of its first raster pixel (above it
    in column 0) as read from the input map, or keeps its label when that pixel is unlabelled or
    missing.  One pass.  Components via scipy.ndimage.label."""
    from scipy import ndimage
    labels = np.asarray(labels)
    H, W = labels.shape
    out = labels.copy()
    flat = labels.reshape(-1)
    eight = np.ones((3, 3), dtype=bool)
    for l in np.unique(labels):
        if l < 0:
            continue
        comp, ncomp = ndimage.label(labels == l, structure=eight)
        if ncomp < 2:
            continue
        cf = comp.reshape(-1)
        firsts = np.array([np.flatnonzero(cf == c)[0] for c in range(1, ncomp + 1)])
        sizes = np.bincount(cf, minlength=ncomp + 1)[1:]
        keep = min(range(ncomp), key=lambda c: (-sizes[c], firsts[c]))
        for c in range(ncomp):
            if c == keep:
                continue
            p = firsts[c]
            nb = p - 1 if p % W > 0 else p - W
            if nb >= 0 and flat[nb] >= 0:
                out[comp == c + 1] = flat[nb]
    return out

