"""Drop-in replacement for scripts/slic.py (class SLICSuperPixel, slic.py:5-160).

Same constructor, attributes, methods and return conventions as the reference;
the pixel work runs in hand-written HIP kernels through libimgan_hip.so.
Import it the way the reference's callers do (`from slic import SLICSuperPixel`,
rag_segmentation.py:4) with this directory on sys.path.
"""
import numpy as np
import torch

from imgan import ops


class SLICSuperPixel(object):

    def __init__(self, image, num_clusters, colour_weight=10, num_iter=10, step=None):
        # slic.py:10-11
        if not isinstance(image, np.ndarray):
            raise ValueError("Image value is None. Probably not loaded correctly!")
        if image.ndim != 3 or image.shape[2] != 3 or image.dtype != np.uint8:
            raise ValueError("Expected an H x W x 3 uint8 BGR image")

        self._dev_bgr = torch.from_numpy(np.ascontiguousarray(image)).cuda()
        self._dev_lab = ops.bgr2lab(self._dev_bgr)                    # slic.py:13
        height, width = image.shape[0], image.shape[1]

        self.centers = []
        self.clusters = None
        self.center_counts = []
        self.image_rgb = image
        self.image = self._dev_lab.cpu().numpy().astype("int32")     # slic.py:22
        self.num_iter = num_iter
        self.num_clusters = num_clusters
        self.colour_weight = colour_weight
        self.step, nx, ny = ops.slic_grid(height, width, num_clusters, step)   # slic.py:29-33
        # an empty seed grid (step larger than the image) is not an error in the reference:
        # its loops simply do nothing and every pixel stays -1
        self._plan = ops.SlicPlan(1, height, width, num_clusters, step) if nx * ny else None
        self._dev_labels = None

    def _pull_centres(self):
        if self._plan is None:
            self.centers, self.center_counts = [], []
            return
        c = self._plan.centres[0].cpu().numpy()
        self.centers = [[int(v) for v in row] for row in c]
        self.center_counts = [0.0] * len(self.centers)

    def init_clusters(self):
        """slic.py:35-54."""
        if self._plan is not None:
            self._plan.init(self._dev_lab)
        self._pull_centres()

    @staticmethod
    def find_local_minimum(image, center):
        """slic.py:57-76 (host helper kept for API parity; the kernel does the same search)."""
        min_grad = float("inf")
        loc_min = center
        for i in range(center[0] - 1, center[0] + 2):
            for j in range(center[1] - 1, center[1] + 2):
                l_1 = int(image[j + 1, i][0])
                l_2 = int(image[j, i + 1][0])
                l_3 = int(image[j, i][0])
                if abs(l_1 - l_3) + abs(l_2 - l_3) < min_grad:
                    min_grad = abs(l_1 - l_3) + abs(l_2 - l_3)
                    loc_min = (i, j)
        return loc_min

    def run(self):
        """slic.py:78-127."""
        if self._plan is None:
            self._pull_centres()
            self.clusters = np.full(self.image.shape[:2], -1.0)
            return
        labels, _ = self._plan.run(self._dev_lab, self.colour_weight, self.num_iter)
        self._dev_labels = labels[0]
        self._pull_centres()
        self.clusters = self._dev_labels.cpu().numpy().astype(np.float64)

    def find_contours(self):
        """slic.py:129-148 (visualisation helper, host numpy; same marking order)."""
        H, W = self.image.shape[0], self.image.shape[1]
        dx8 = [-1, -1, 0, 1, 1, 1, 0, -1]
        dy8 = [0, -1, -1, -1, 0, 1, 1, 1]
        istaken = np.zeros((H, W))
        contours = []
        for j in range(H):
            for i in range(W):
                nr_p = 0
                for k in range(8):
                    x, y = i + dx8[k], j + dy8[k]
                    if 0 <= x < W and 0 <= y < H:
                        if istaken[y, x] == 0 and self.clusters[j, i] != self.clusters[y, x]:
                            nr_p += 1
                if nr_p >= 2:
                    contours.append([j, i])
                    istaken[j, i] = 1
        return contours

    def plot(self):
        """slic.py:150-157 -- paints the contours; display needs OpenCV's GUI."""
        for cont in self.find_contours():
            self.image_rgb[cont[0], cont[1]] = [0, 0, 255]
        try:
            import cv2
        except ImportError:
            return self.image_rgb
        cv2.imshow('superPixel', self.image_rgb)
        cv2.waitKey(0)
        cv2.destroyAllWindows()

    def enforce_connectivity(self):
        """Additive (not in scripts/slic.py, named by the task): every label keeps its largest
        8-connected component, the others join the label to their left (imgan.ops.enforce_connectivity:
        label-propagation union-find on the device).  Updates and returns `clusters`."""
        if getattr(self, "_dev_labels", None) is None:
            return self.clusters
        from imgan import ops
        self._dev_labels = ops.enforce_connectivity(self._dev_labels)
        self.clusters = self._dev_labels.cpu().numpy().astype(np.float64)
        return self.clusters

    def get_clusters(self):
        return self.clusters
