"""Drop-in replacement for scripts/rag_segmentation.py (class RAGSegmentation, :8-203).

Same constructor, attributes, method names, argument meaning and return structure as the
reference.  Pixel passes (median blur, Lab, SLIC, per-label sums, adjacency, relabel, paint)
run in hand-written HIP kernels through libimgan_hip.so; the tiny graph part (<= a few
thousand edges: sort, Kruskal, leave-one-out cut) runs in the library's host C++.

Where the reference's result depends on Python-2 dict / set iteration order
(rag_segmentation.py:79-81,93) this module uses the canonical order: vertices ascending,
neighbours ascending.
"""
import numpy as np
import torch

from imgan import ops
from slic import SLICSuperPixel


def euclidean_dist(a, b):
    # the one helper find_edges takes from scripts/metric_methods.py (:5-6); the rest of that
    # module serves the GA and stays with the reference
    return np.sqrt(np.sum((a - b) ** 2))


def _labels_to_device(clusters):
    return torch.from_numpy(np.ascontiguousarray(clusters).astype(np.int32)).cuda()


class RAGSegmentation(object):

    def __init__(self, image, slic_clust_num=200, slic_cw=15, slic_steps=10, median_blur=7):
        # rag_segmentation.py:10-19
        if not isinstance(image, np.ndarray):
            raise ValueError("Image value is None. Probably not loaded correctly!")
        if not (isinstance(median_blur, (int, np.integer)) and median_blur > 1 and median_blur % 2 == 1):
            raise ValueError("median_blur must be an odd size greater than 1 (cv2.medianBlur)")
        dev = torch.from_numpy(np.ascontiguousarray(image)).cuda()
        self._dev_image = ops.median_blur(dev, median_blur)       # cv2.medianBlur(image, median_blur)
        self._dev_lab = ops.bgr2lab(self._dev_image)
        image = self._dev_image.cpu().numpy()
        self.image_lab = self._dev_lab.cpu().numpy()
        self.image_mean = image.copy()
        self.image = image
        self.slic_sp = None
        self.slic_clusters = None
        self.slic_clust_num = slic_clust_num
        self.slic_cw = slic_cw
        self.slic_steps = slic_steps

    def run_slic(self):
        """rag_segmentation.py:21-25."""
        self.slic_sp = SLICSuperPixel(self.image, self.slic_clust_num, colour_weight=self.slic_cw,
                                      num_iter=self.slic_steps)
        self.slic_sp.run()
        return self.slic_sp.get_clusters()

    def _means(self, clusters, host_image):
        """floor(sum / count) per label for ids 0..K-1, empty -> 0, and the
        painted image (rag_segmentation.py:27-48)."""
        K = self.slic_clust_num
        dev_img = torch.from_numpy(np.ascontiguousarray(host_image)).cuda()
        dev_lab = _labels_to_device(clusters)
        sums, _ = ops.rag_accumulate(dev_img, dev_lab, K, want_edges=False)
        s = sums[0].cpu().numpy()
        cnt = s[:, 3]
        means = np.where(cnt[:, None] > 0, s[:, :3] // np.maximum(cnt[:, None], 1), 0)
        return means.astype(np.uint64), cnt, dev_img, dev_lab

    def slic_mean_rgb(self, clusters):
        """rag_segmentation.py:27-36 (always paints image_mean)."""
        means, cnt, dev_img, dev_lab = self._means(clusters, self.image_mean)
        ops.paint_means(dev_img, dev_lab, torch.from_numpy(means.astype(np.uint8)).cuda())
        self.image_mean[...] = dev_img[0].cpu().numpy() if dev_img.dim() == 4 else dev_img.cpu().numpy()
        return {c: means[c] for c in range(self.slic_clust_num)}

    def slic_mean_lab(self, clusters, nchange=False):
        """rag_segmentation.py:38-48 (paints image_lab unless nchange)."""
        means, cnt, dev_img, dev_lab = self._means(clusters, self.image_lab)
        if not nchange:
            ops.paint_means(dev_img, dev_lab, torch.from_numpy(means.astype(np.uint8)).cuda())
            self.image_lab[...] = dev_img.cpu().numpy().reshape(self.image_lab.shape)
        return {c: means[c] for c in range(self.slic_clust_num)}

    def neighbours_regions(self, clusters):
        """rag_segmentation.py:50-74: {label: set(float neighbour labels)} for labels 0..K-1."""
        K = self.slic_clust_num
        dev_lab = _labels_to_device(clusters)
        kmax = max(K, int(dev_lab.max().item()) + 1)
        _, edges = ops.rag_accumulate(self._dev_image, dev_lab, kmax, want_edges=True)
        clust_ng = {c: set() for c in range(K)}
        for u, v in edges[0]:
            if u < K:
                clust_ng[int(u)].add(float(v))
            if v < K:
                clust_ng[int(v)].add(float(u))
        return clust_ng

    @staticmethod
    def find_edges(ng_dict, clust_col):
        """rag_segmentation.py:77-87."""
        edges_dict = {}
        for curr_region, ng_set in ng_dict.items():
            ng_edge = []
            for ng_reg in sorted(ng_set):
                curr_region_col = np.asarray(clust_col[curr_region], dtype="int32")
                ng_reg_col = np.asarray(clust_col[ng_reg], dtype="int32")
                ng_edge.append((ng_reg, euclidean_dist(curr_region_col, ng_reg_col)))
            edges_dict[curr_region] = ng_edge
        return edges_dict

    @staticmethod
    def _convert2arrays(edges_dict):
        """rag_segmentation.py:90-100: first orientation seen wins."""
        seen = set()
        edges_list, weight_list = [], []
        for vert_1, ng_list in edges_dict.items():
            for vert_2, weight in ng_list:
                new_edge = (vert_1, int(vert_2))
                if (new_edge[1], new_edge[0]) not in seen:
                    seen.add(new_edge)
                    edges_list.append([new_edge[0], new_edge[1]])
                    weight_list.append(weight)
        return edges_list, weight_list

    def kruskal_alg(self, edges_list, weight_list, vertex_list):
        """rag_segmentation.py:132-145 (edges must already be sorted by weight).
        Same list-of-sets semantics; sets found through an index instead of linear scans."""
        vertex_list = [list(v) for v in vertex_list]
        where = {}
        for i, s in enumerate(vertex_list):
            for e in s:
                where[e] = i
        n_alive = len(vertex_list)
        edges_mst, weight_mst = [], []
        for edge, weight in zip(edges_list, weight_list):
            if n_alive <= 1:
                break
            i1, i2 = where[edge[0]], where[edge[1]]
            if i1 != i2:
                edges_mst.append(edge)
                weight_mst.append(weight)
                for e in vertex_list[i2]:
                    where[e] = i1
                vertex_list[i1].extend(vertex_list[i2])
                vertex_list[i2] = None
                n_alive -= 1
        return edges_mst, weight_mst, [s for s in vertex_list if s is not None]

    def concat_similar_regs(self, edges_dict, clusters, c_factor=2.0):
        """rag_segmentation.py:147-184."""
        edges_list, weight_list = self._convert2arrays(edges_dict)
        vertices = list(edges_dict.keys())
        index = {v: i for i, v in enumerate(vertices)}
        K = len(vertices)
        uv = np.array([[index[e[0]], index[e[1]]] for e in edges_list], dtype=np.int32).reshape(-1, 2)
        res = ops.rag_merge_host(K, uv, c_factor, weights=np.asarray(weight_list, dtype=np.float64))
        edge_mst = [[vertices[u], vertices[v]] for u, v in res["mst"]]
        weight_mst = [np.float64(w) for w in res["mst_w"]]
        vertex_mst = [[vertices[i] for i in tree] for tree in res["sets"]]

        # relabel on the device: new id of a pixel = id of the set holding its label (:175-182)
        kmax = int(max(vertices)) + 1 if vertices else 1
        lut = np.zeros(kmax, dtype=np.int32)
        for i, v in enumerate(vertices):
            if isinstance(v, (int, np.integer)) and 0 <= v < kmax:
                lut[int(v)] = res["lut"][i]
        dev = ops.relabel(_labels_to_device(clusters), torch.from_numpy(lut).cuda())
        new_clust_array = dev[0].cpu().numpy().astype(np.float64)
        return new_clust_array, edge_mst, weight_mst, vertex_mst

    def plot_regions(self, clusters, edge_mst):
        """rag_segmentation.py:186-203 -- visualisation; needs OpenCV's drawing functions."""
        try:
            import cv2
        except ImportError:
            raise RuntimeError("plot_regions draws with OpenCV, which is not installed")
        reg_points = {}
        for clust_idx in range(self.slic_clust_num):
            ng_mask = clusters == clust_idx
            ys, xs = np.nonzero(ng_mask)
            if ys.size != 0:
                c_point = (int(xs.mean()), int(ys.mean()))
                reg_points[clust_idx] = c_point
                cv2.circle(self.image_mean, c_point, 5, (0, 0, 255), -1)
                cv2.putText(self.image_mean, str(clust_idx), c_point, cv2.FONT_HERSHEY_SIMPLEX,
                            0.5, (0, 0, 255), 1)
        for edge in edge_mst:
            cv2.line(self.image_mean, reg_points[edge[0]], reg_points[edge[1]], (255, 0, 0), 1)
