"""Oracle: region-adjacency-graph segmentation as scripts/rag_segmentation.py
computes it (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

Follows /root/reference/scripts/rag_segmentation.py:10-184 and
scripts/metric_methods.py:5-6.  Python-2 / legacy-numpy semantics restated:
  * colour means are floor(sum / count), empty -> 0       (:38-48)
  * dict / set iteration order is made canonical: vertices ascending,
    neighbours ascending (the reference depends on Py2 hash order, :79-81,:93)
Parity unpinned by the reference (no fixture for this path).
"""
import numpy as np

from .color import bgr2lab_u8, median_blur
from .slic_ref import SLICOracle


def euclidean_dist(col_one, col_two):
    """metric_methods.py:5-6."""
    return np.sqrt((col_one[0] - col_two[0]) ** 2 + (col_one[1] - col_two[1]) ** 2 +
                   (col_one[2] - col_two[2]) ** 2)


class RAGOracle(object):
    def __init__(self, image, slic_clust_num=200, slic_cw=15, slic_steps=10, median_blur_k=7,
                 slic_mode="float64"):
        """rag_segmentation.py:10-19."""
        image = median_blur(image, median_blur_k)
        self.image_lab = bgr2lab_u8(image)
        self.image_mean = image.copy()
        self.image = image
        self.slic_sp = None
        self.slic_clust_num = slic_clust_num
        self.slic_cw = slic_cw
        self.slic_steps = slic_steps
        self.slic_mode = slic_mode

    def run_slic(self):
        """rag_segmentation.py:21-25 (SLIC on the blurred image)."""
        self.slic_sp = SLICOracle(self.image, self.slic_clust_num, colour_weight=self.slic_cw,
                                  num_iter=self.slic_steps, mode=self.slic_mode)
        self.slic_sp.run()
        return self.slic_sp.get_clusters()

    def _mean(self, clusters, image, paint):
        mean_dict = {}
        flat = clusters.astype(np.int64).ravel()
        valid = flat >= 0
        K = self.slic_clust_num
        px = image.reshape(-1, 3)
        cnt = np.bincount(flat[valid], minlength=K)
        sums = np.stack([np.bincount(flat[valid], weights=px[valid, k].astype(np.int64),
                                     minlength=K).astype(np.int64) for k in range(3)], axis=1)
        for c in range(K):
            if cnt[c]:
                m = (sums[c] // cnt[c]).astype(np.uint64)
            else:
                m = np.zeros(3, dtype=np.uint64)                # int 0/0 -> 0 in legacy numpy
            mean_dict[c] = m
            if paint and cnt[c]:
                image[clusters == c] = m
        return mean_dict

    def slic_mean_lab(self, clusters, nchange=False):
        """rag_segmentation.py:38-48 -- floor means over the u8 Lab of the blurred image;
        paints them back unless nchange."""
        return self._mean(clusters, self.image_lab, paint=not nchange)

    def slic_mean_rgb(self, clusters):
        """rag_segmentation.py:27-36."""
        return self._mean(clusters, self.image_mean, paint=True)

    def neighbours_regions(self, clusters):
        """rag_segmentation.py:50-74 -- symmetric 8-connected adjacency between
        labelled pixels; -1 discarded; keys 0..K-1; members are floats."""
        K = self.slic_clust_num
        H, W = clusters.shape
        lab = clusters.astype(np.int64)
        pairs = set()
        # the clamped 3x3 window (:60-68) never reaches row H-1 / col W-1 for a
        # labelled pixel's lower/right side, and those hold -1 anyway.
        for dy, dx in ((0, 1), (1, 0), (1, 1), (1, -1)):
            a = lab[0:H - dy, max(0, -dx):W - max(0, dx)]
            b = lab[dy:H, max(0, dx):W + min(0, dx)]
            m = (a != b) & (a >= 0) & (b >= 0)
            if m.any():
                u = np.unique(np.stack([a[m], b[m]], axis=1), axis=0)
                pairs.update((int(p), int(q)) for p, q in u)
        ng = {c: set() for c in range(K)}
        for p, q in pairs:
            if p < K:
                ng[p].add(float(q))
            if q < K:
                ng[q].add(float(p))
        return ng

    @staticmethod
    def find_edges(ng_dict, clust_col):
        """rag_segmentation.py:77-87 -- weights = euclidean distance of int32-truncated means."""
        edges_dict = {}
        for curr_region in sorted(ng_dict.keys()):
            ng_edge = []
            for ng_reg in sorted(ng_dict[curr_region]):
                a = np.asarray(clust_col[curr_region], dtype="int32")
                b = np.asarray(clust_col[ng_reg], dtype="int32")
                ng_edge.append((ng_reg, euclidean_dist(a, b)))
            edges_dict[curr_region] = ng_edge
        return edges_dict

    @staticmethod
    def _convert2arrays(edges_dict):
        """rag_segmentation.py:90-100 -- first orientation seen wins."""
        seen = set()
        edges_list, weight_list = [], []
        for vert_1 in sorted(edges_dict.keys()):
            for vert_2, weight in edges_dict[vert_1]:
                new_edge = (vert_1, int(vert_2))
                if (new_edge[1], new_edge[0]) not in seen:
                    seen.add(new_edge)
                    edges_list.append(list(new_edge))
                    weight_list.append(weight)
        return edges_list, weight_list

    @staticmethod
    def _sort_edges(edges_list, weight_list):
        """rag_segmentation.py:103-114 -- stable insertion sort by weight."""
        order = sorted(range(len(edges_list)), key=lambda i: weight_list[i])   # stable
        return [edges_list[i] for i in order], [weight_list[i] for i in order]

    @staticmethod
    def kruskal_alg(edges_list, weight_list, vertex_list):
        """rag_segmentation.py:117-145 -- list-of-sets Kruskal; union appends the second
        endpoint's set to the first endpoint's set and pops the second."""
        vertex_list = [list(v) for v in vertex_list]
        where = {}
        for i, s in enumerate(vertex_list):
            for e in s:
                where[e] = i
        alive = [True] * len(vertex_list)
        n_alive = len(vertex_list)
        edges_mst, weight_mst = [], []
        edge_idx = 0
        while n_alive > 1 and edge_idx < len(edges_list):
            v1, v2 = edges_list[edge_idx]
            i1, i2 = where[v1], where[v2]
            if i1 != i2:
                edges_mst.append(edges_list[edge_idx])
                weight_mst.append(weight_list[edge_idx])
                for e in vertex_list[i2]:
                    where[e] = i1
                vertex_list[i1].extend(vertex_list[i2])
                vertex_list[i2] = []
                alive[i2] = False
                n_alive -= 1
            edge_idx += 1
        return edges_mst, weight_mst, [s for s, a in zip(vertex_list, alive) if a]

    def concat_similar_regs(self, edges_dict, clusters, c_factor=2.0):
        """rag_segmentation.py:147-184."""
        edges_list, weight_list = self._convert2arrays(edges_dict)
        vertex_list = [[v] for v in sorted(edges_dict.keys())]
        edges_list, weight_list = self._sort_edges(edges_list, weight_list)
        edge_mst, weight_mst, vertex_mst = self.kruskal_alg(edges_list, weight_list, vertex_list)

        new_edges, new_weight = [], []
        with np.errstate(all="ignore"):
            for edge_idx in range(len(edge_mst)):
                w_copy = list(weight_mst)
                w_copy.remove(w_copy[edge_idx])
                mst_mean = np.mean(w_copy) if w_copy else np.nan
                mst_std = np.std(w_copy) if w_copy else np.nan
                if weight_mst[edge_idx] < mst_mean + c_factor * mst_std:
                    new_edges.append(edge_mst[edge_idx])
                    new_weight.append(weight_mst[edge_idx])

        edge_mst, weight_mst, vertex_mst = self.kruskal_alg(new_edges, new_weight, vertex_list)

        lut = {}
        for new_cluster, tree in enumerate(vertex_mst, start=1):
            for vertex in tree:
                lut[vertex] = new_cluster
        new_clust_array = np.zeros(clusters.shape)                # -1 border -> 0 (:176)
        flat = clusters.astype(np.int64)
        K = max(lut.keys()) + 1 if lut else 0
        table = np.zeros(K + 1, dtype=np.float64)
        for vertex, new_cluster in lut.items():
            table[vertex] = new_cluster
        inside = (flat >= 0) & (flat < K)
        new_clust_array[inside] = table[flat[inside]]
        return new_clust_array, edge_mst, weight_mst, vertex_mst
