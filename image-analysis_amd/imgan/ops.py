"""Batch-first device operators: thin torch-tensor plumbing over the C ABI.

Every function takes / returns torch tensors that live on the current ROCm
device; torch is used only to own memory and to supply the HIP stream.  All
compute happens in libimgan_hip.so (hand-written HIP kernels).  There is no
CPU fallback: without a GPU these functions raise.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import lib, check


def _require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("imgan: no ROCm device available -- the HIP path is the only path")


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _dev(t, dtype, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise ValueError("%s must be a contiguous %s tensor on the GPU" % (name, dtype))
    return t


def as_batch(img):
    """[H,W,3] or [B,H,W,3] uint8 -> [B,H,W,3]."""
    if img.dim() == 3:
        img = img.unsqueeze(0)
    if img.dim() != 4 or img.shape[-1] != 3:
        raise ValueError("expected a [B,H,W,3] or [H,W,3] uint8 image tensor")
    return img


def lab_tables():
    gamma = np.zeros(256, dtype=np.uint16)
    cbrt = np.zeros(3072, dtype=np.uint16)
    coef = np.zeros(9, dtype=np.int32)
    check(lib.imgan_lab_tables_host(gamma.ctypes.data, cbrt.ctypes.data, coef.ctypes.data))
    return gamma, cbrt, coef.reshape(3, 3)


def bgr2lab(bgr, out=None):
    """cv2.cvtColor(bgr, COLOR_BGR2Lab) for a u8 [...,3] tensor (slic.py:13)."""
    _require_gpu()
    _dev(bgr, torch.uint8, "bgr")
    if out is None:
        out = torch.empty_like(bgr)
    check(lib.imgan_bgr2lab_u8(_ptr(bgr), _ptr(out), bgr.numel() // 3, _stream()))
    return out


def bgr2gray(bgr, out=None):
    """cv2.cvtColor(bgr, COLOR_BGR2GRAY) (mpeg7Descriptors.py:23)."""
    _require_gpu()
    _dev(bgr, torch.uint8, "bgr")
    if out is None:
        out = torch.empty(bgr.shape[:-1], dtype=torch.uint8, device=bgr.device)
    check(lib.imgan_bgr2gray_u8(_ptr(bgr), _ptr(out), bgr.numel() // 3, _stream()))
    return out


def median_blur(bgr, ksize=7, out=None):
    """cv2.medianBlur(bgr, ksize): any odd size; 3, 5 and 7 (rag_segmentation.py:11 uses 7) take
    the bit-sliced kernels, the others a plain search per pixel."""
    _require_gpu()
    img = as_batch(_dev(bgr, torch.uint8, "bgr"))
    if out is None:
        out = torch.empty_like(bgr)
    B, H, W, _ = img.shape
    check(lib.imgan_median_u8c3(_ptr(img), _ptr(out), B, H, W, int(ksize), _stream()))
    return out


def median7(bgr, out=None):
    return median_blur(bgr, 7, out)


def resize_linear(bgr, dsize, out=None):
    """cv2.resize(bgr, dsize) with the default INTER_LINEAR (optigen_utils.py:23); dsize is
    (width, height) as in OpenCV."""
    _require_gpu()
    img = as_batch(_dev(bgr, torch.uint8, "bgr"))
    B, Hs, Ws, _ = img.shape
    Wd, Hd = int(dsize[0]), int(dsize[1])
    if Wd < 1 or Hd < 1:
        raise ValueError("resize: empty destination size")
    if out is None:
        out = torch.empty((B, Hd, Wd, 3) if bgr.dim() == 4 else (Hd, Wd, 3), dtype=torch.uint8,
                          device=bgr.device)
    check(lib.imgan_resize_linear_u8c3(_ptr(img), _ptr(out), B, Hs, Ws, Hd, Wd, _stream()))
    return out


def slic_grid(H, W, num_clusters, step=None):
    """(step, nx, ny) exactly as slic.py:29-33,42-43 derive them."""
    s, nx, ny = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    check(lib.imgan_slic_grid_host(int(H), int(W), int(num_clusters), int(step) if step else 0,
                                   ctypes.byref(s), ctypes.byref(nx), ctypes.byref(ny)))
    return s.value, nx.value, ny.value


class SlicPlan(object):
    """Shapes + workspace for SLIC on a batch of equally sized images (reusable)."""

    def __init__(self, B, H, W, num_clusters, step=None, device=None):
        _require_gpu()
        self.B, self.H, self.W = int(B), int(H), int(W)
        self.S, self.nx, self.ny = slic_grid(H, W, num_clusters, step)
        self.n = self.nx * self.ny
        if self.n < 1:
            raise ValueError("SLIC seed grid is empty for this image size / cluster count")
        nbytes = lib.imgan_slic_workspace_bytes(self.B, self.H, self.W, self.S, self.n)
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self.centres = torch.empty((self.B, self.n, 5), dtype=torch.int32, device=self.device)
        self.labels = torch.empty((self.B, self.H, self.W), dtype=torch.int32, device=self.device)

    def _check(self, lab):
        lab = as_batch(_dev(lab, torch.uint8, "lab"))
        if tuple(lab.shape) != (self.B, self.H, self.W, 3):
            raise ValueError("lab has shape %s, plan is for %s" %
                             (tuple(lab.shape), (self.B, self.H, self.W, 3)))
        return lab

    def init(self, lab):
        lab = self._check(lab)
        check(lib.imgan_slic_init(_ptr(lab), self.B, self.H, self.W, self.S, self.nx, self.ny,
                                  _ptr(self.centres), _ptr(self.labels), _ptr(self.workspace),
                                  _stream()))

    def seed(self, lab):
        """init() without the pass that fills the label map with -1: the map is undefined until
        the first round, which must then be run as round(..., first=True)."""
        lab = self._check(lab)
        check(lib.imgan_slic_seed_centres(_ptr(lab), self.B, self.H, self.W, self.S, self.nx,
                                          self.ny, _ptr(self.centres), _stream()))

    def round(self, lab, colour_weight, events=None, bbox=None, first=False):
        """One iteration of slic.py:84-126.  `events`: optional (start, end) torch events
        recorded around the assignment kernel on the current stream (bench.py roofline).
        `bbox`: optional device buffer of at least B*n*16 bytes that receives the bounding box
        int32 [B,n,4] of every label of the updated map (for DcdPlan.kmeans(boxed=True)).
        `first`: the round after seed(): the label map counts as all -1 and is written in full."""
        lab = self._check(lab)
        a = (self.B, self.H, self.W, self.S, self.n)
        check(lib.imgan_slic_bin_centres(_ptr(lab), _ptr(self.centres), *a, float(colour_weight),
                                         _ptr(self.workspace), _stream()))
        if events:
            events[0].record()
        if bbox is not None and bbox.numel() * bbox.element_size() < self.B * self.n * 16:
            raise ValueError("bbox buffer too small")
        if first:
            check(lib.imgan_slic_assign_accumulate_first(
                _ptr(lab), _ptr(self.labels), *a, float(colour_weight), _ptr(self.workspace),
                _ptr(bbox) if bbox is not None else None, _stream()))
        elif bbox is None:
            check(lib.imgan_slic_assign_accumulate(_ptr(lab), _ptr(self.labels), *a,
                                                   float(colour_weight), _ptr(self.workspace),
                                                   _stream()))
        else:
            check(lib.imgan_slic_assign_accumulate_bbox(_ptr(lab), _ptr(self.labels), *a,
                                                        float(colour_weight), _ptr(self.workspace),
                                                        _ptr(bbox), _stream()))
        if events:
            events[1].record()
        check(lib.imgan_slic_finalise_centres(_ptr(self.centres), *a, _ptr(self.workspace),
                                              _stream()))

    def label_counts(self, out=None):
        """int32 [B,n]: pixels per label after the last round (valid until the next round())."""
        if out is None:
            out = torch.empty((self.B, self.n), dtype=torch.int32, device=self.device)
        check(lib.imgan_slic_label_counts(self.B, self.H, self.W, self.S, self.n,
                                          _ptr(self.workspace), _ptr(out), _stream()))
        return out

    def label_sums(self, out):
        """out int64 [B,K,4] = sum L, sum a, sum b, count per label after the last round (rows
        n..K-1 zero): rag_accumulate's sums for the image SLIC ran on, without another pass."""
        _dev(out, torch.int64, "out")
        if out.dim() != 3 or out.shape[0] != self.B or out.shape[2] != 4 or not out.is_contiguous():
            raise ValueError("out must be a contiguous [B, K, 4] tensor")
        check(lib.imgan_slic_label_sums(self.B, self.H, self.W, self.S, self.n, _ptr(self.workspace),
                                        int(out.shape[1]), _ptr(out), _stream()))
        return out

    def run(self, lab, colour_weight, num_iter):
        """SLICSuperPixel.run() (slic.py:78-127); returns (labels, centres) device tensors."""
        lab = self._check(lab)
        check(lib.imgan_slic_run(_ptr(lab), self.B, self.H, self.W, self.S, self.nx, self.ny,
                                 float(colour_weight), int(num_iter), _ptr(self.centres),
                                 _ptr(self.labels), _ptr(self.workspace), _stream()))
        return self.labels, self.centres


# ------------------------------------------------------------------------ RAG

def _pow2_at_least(x):
    p = 2
    while p < x:
        p <<= 1
    return p


def rag_accumulate(img, labels, K, want_edges=True, edge_cap=None):
    """Per-label colour sums + 8-connected adjacency (rag_segmentation.py:38-74).

    img: u8 [B,H,W,3] (any 3-channel image), labels: int32 [B,H,W].
    Returns (sums uint64-as-int64 [B,K,4] = c0,c1,c2,count, edges [list per image of int32 [E,2]
    arrays with u < v, lexicographically sorted] or None)."""
    _require_gpu()
    img = as_batch(_dev(img, torch.uint8, "img"))
    if labels.dim() == 2:
        labels = labels.unsqueeze(0)
    _dev(labels, torch.int32, "labels")
    B, H, W, _ = img.shape
    if tuple(labels.shape) != (B, H, W):
        raise ValueError("labels shape %s does not match image %s" % (tuple(labels.shape), (B, H, W)))
    sums = torch.zeros((B, K, 4), dtype=torch.int64, device=img.device)
    if not want_edges:
        check(lib.imgan_rag_accumulate(_ptr(img), _ptr(labels), B, H, W, int(K), _ptr(sums), None,
                                       0, None, _stream()))
        return sums, None
    cap = edge_cap or _pow2_at_least(16 * max(int(K), 64))
    while True:
        table = torch.zeros((B, cap), dtype=torch.int64, device=img.device)
        overflow = torch.zeros(B, dtype=torch.int32, device=img.device)
        sums.zero_()
        check(lib.imgan_rag_accumulate(_ptr(img), _ptr(labels), B, H, W, int(K), _ptr(sums),
                                       _ptr(table), cap, _ptr(overflow), _stream()))
        if not overflow.cpu().numpy().any():
            break
        cap *= 4                                     # pathological adjacency: retry larger
    # this form hands host lists to the class API: the hash sets cross to the host as they are and
    # are ordered there ((u+1)<<32 | (v+1) orders lexicographically by (u, v)); the batch path
    # (RagPlan) keeps them on the device
    host = table.cpu().numpy()
    edges = []
    for b in range(B):
        keys = np.sort(host[b][host[b] != 0])
        uv = np.stack([(keys >> 32) - 1, (keys & 0xffffffff) - 1], axis=1).astype(np.int32)
        edges.append(uv.reshape(-1, 2))
    return sums, edges


class RagPlan(object):
    """The RAG step of a device-resident batch without leaving the GPU: per-label sums + adjacency
    (rag_segmentation.py:38-74), then the merge of :77-184 by one workgroup per image
    (imgan_rag_merge_device), which writes the relabel tables in place.  Nothing synchronises;
    `status` (int32 [B], device) is nonzero for an image the device merge had to leave to the host
    (edge set overflowed, more than 16,384 edges): finish() deals with those."""

    MAX_K = 4096

    def __init__(self, B, K, edge_cap=None, device=None):
        _require_gpu()
        self.B, self.K = int(B), int(K)
        if self.K > self.MAX_K:
            raise OverflowError("RagPlan: more than %d labels per image; use rag_accumulate + "
                                "rag_merge_batch" % self.MAX_K)
        dev = device or torch.device("cuda", torch.cuda.current_device())
        self.cap = int(edge_cap or _pow2_at_least(16 * max(self.K, 64)))
        self.sums = torch.empty((self.B, self.K, 4), dtype=torch.int64, device=dev)
        self.table = torch.empty((self.B, self.cap), dtype=torch.int64, device=dev)
        self.overflow = torch.empty(self.B, dtype=torch.int32, device=dev)
        self.lut = torch.zeros((self.B, self.K), dtype=torch.int32, device=dev)
        self.status = torch.zeros(self.B, dtype=torch.int32, device=dev)

    def run(self, img, labels, c_factor, slic=None):
        """img u8 [B,H,W,3], labels int32 [B,H,W] -> (lut int32 [B,K], status int32 [B]), both on
        the device, enqueued on the current stream.  slic: the SlicPlan whose last round produced
        `labels` from `img` -- its sums are the per-label sums (rag_segmentation.py:38-48 on the
        image of run_slic :21-25), and only the adjacency pass over the labels is left."""
        img = as_batch(_dev(img, torch.uint8, "img"))
        _dev(labels, torch.int32, "labels")
        B, H, W, _ = img.shape
        if B != self.B or tuple(labels.shape) != (B, H, W):
            raise ValueError("batch does not match the plan")
        self.table.zero_()
        self.overflow.zero_()
        if slic is not None:
            if (slic.B, slic.H, slic.W) != (B, H, W) or slic.n > self.K:
                raise ValueError("SLIC plan does not match the batch")
            slic.label_sums(self.sums)
            check(lib.imgan_rag_accumulate(None, _ptr(labels), B, H, W, self.K, None,
                                           _ptr(self.table), self.cap, _ptr(self.overflow), _stream()))
        else:
            self.sums.zero_()
            check(lib.imgan_rag_accumulate(_ptr(img), _ptr(labels), B, H, W, self.K, _ptr(self.sums),
                                           _ptr(self.table), self.cap, _ptr(self.overflow), _stream()))
        check(lib.imgan_rag_merge_device(_ptr(self.sums), _ptr(self.table), self.cap,
                                         _ptr(self.overflow), B, self.K, float(c_factor),
                                         _ptr(self.lut), _ptr(self.status), _stream()))
        return self.lut, self.status

    def finish(self, img, labels, c_factor):
        """Synchronises once and redoes, through the host merge, the images whose status is
        nonzero (none in practice).  Returns the number of images redone."""
        bad = torch.nonzero(self.status).flatten().tolist()
        for b in bad:
            sums, edges = rag_accumulate(img[b:b + 1], labels[b:b + 1], self.K)
            self.lut[b] = torch.from_numpy(rag_merge_batch(self.K, edges, sums, c_factor)[0]).to(self.lut.device)
        return len(bad)


def rag_merge_batch(K, edges, sums, c_factor, threads=16):
    """The merge of rag_segmentation.py:77-184 for every image of a batch (means mode; host C++
    in the library, the images spread over `threads` host threads).  sums: the [B,K,4] tensor of
    rag_accumulate; edges: one int [E_b, 2] array per image.  Returns int32 [B,K]."""
    s = sums.cpu().numpy()
    means = np.ascontiguousarray(
        np.where(s[:, :, 3:4] > 0, s[:, :, :3] // np.maximum(s[:, :, 3:4], 1), 0).astype(np.int32))
    B = len(edges)
    offsets = np.zeros(B + 1, dtype=np.int64)
    offsets[1:] = np.cumsum([len(e) for e in edges])
    flat = np.ascontiguousarray(np.concatenate([np.asarray(e, dtype=np.int32).reshape(-1, 2)
                                                for e in edges]) if offsets[B] else
                                np.zeros((1, 2), np.int32))
    luts = np.zeros((B, K), dtype=np.int32)
    check(lib.imgan_rag_merge_host_batch(means.ctypes.data, B, int(K), flat.ctypes.data,
                                         offsets.ctypes.data, float(c_factor), luts.ctypes.data,
                                         int(threads)))
    return luts


def rag_merge_host(K, edges_uv, c_factor, means=None, weights=None):
    """rag_segmentation.py:77-184 on the host (C++ in the library).
    Either means (int [K,3]; canonical edge order) or weights (float [E]; edges in discovery
    order).  Returns dict(lut, mst, mst_w, sets)."""
    edges_uv = np.ascontiguousarray(edges_uv, dtype=np.int32).reshape(-1, 2)
    E = edges_uv.shape[0]
    if means is not None:
        means = np.ascontiguousarray(means, dtype=np.int32)
    if weights is not None:
        weights = np.ascontiguousarray(weights, dtype=np.float64)
    lut = np.zeros(K, dtype=np.int32)
    mst = np.zeros((max(K, 1), 2), dtype=np.int32)
    w = np.zeros(max(K, 1), dtype=np.float64)
    members = np.zeros(K, dtype=np.int32)
    offsets = np.zeros(K + 1, dtype=np.int32)
    n_mst, n_sets = ctypes.c_int32(), ctypes.c_int32()
    if (means is None) == (weights is None):
        raise ValueError("give either means or weights")
    if weights is not None and weights.shape[0] != E:
        raise ValueError("one weight per edge expected")
    if weights is not None and E == 0:
        weights = np.zeros(1, dtype=np.float64)       # keeps the pointer non-NULL (mode switch)
    check(lib.imgan_rag_merge_host(means.ctypes.data if means is not None else None,
                                   weights.ctypes.data if weights is not None else None,
                                   int(K), edges_uv.ctypes.data if E else None, E, float(c_factor),
                                   lut.ctypes.data, mst.ctypes.data, w.ctypes.data,
                                   ctypes.byref(n_mst), members.ctypes.data, offsets.ctypes.data,
                                   ctypes.byref(n_sets)))
    ns = n_sets.value
    sets = [members[offsets[i]:offsets[i + 1]].tolist() for i in range(ns)]
    return {"lut": lut, "mst": mst[:n_mst.value].copy(), "mst_w": w[:n_mst.value].copy(),
            "sets": sets}


def rag_merge_lut(K, edges_uv, c_factor, means):
    """The relabel table only (int32 [K]) of rag_merge_host in means mode: the batch path needs
    nothing else, and skipping the optional outputs keeps the per-image host cost small."""
    edges_uv = np.ascontiguousarray(edges_uv, dtype=np.int32).reshape(-1, 2)
    means = np.ascontiguousarray(means, dtype=np.int32)
    E = edges_uv.shape[0]
    lut = np.zeros(K, dtype=np.int32)
    n_mst, n_sets = ctypes.c_int32(), ctypes.c_int32()
    check(lib.imgan_rag_merge_host(means.ctypes.data, None, int(K),
                                   edges_uv.ctypes.data if E else None, E, float(c_factor),
                                   lut.ctypes.data, None, None, ctypes.byref(n_mst), None, None,
                                   ctypes.byref(n_sets)))
    return lut


def relabel(labels, lut):
    """out[p] = lut[labels[p]], labels < 0 -> 0 (rag_segmentation.py:175-182)."""
    _require_gpu()
    if labels.dim() == 2:
        labels = labels.unsqueeze(0)
    _dev(labels, torch.int32, "labels")
    if lut.dim() == 1:
        lut = lut.unsqueeze(0)
    _dev(lut, torch.int32, "lut")
    B = labels.shape[0]
    out = torch.empty_like(labels)
    check(lib.imgan_relabel_lut(_ptr(labels), _ptr(lut), _ptr(out), B, labels[0].numel(),
                                lut.shape[1], _stream()))
    return out


def paint_means(img, labels, means_u8):
    """img[labels == c] = means[c] in place (rag_segmentation.py:34,46)."""
    _require_gpu()
    imgb = as_batch(_dev(img, torch.uint8, "img"))
    if labels.dim() == 2:
        labels = labels.unsqueeze(0)
    if means_u8.dim() == 2:
        means_u8 = means_u8.unsqueeze(0)
    _dev(labels, torch.int32, "labels")
    _dev(means_u8, torch.uint8, "means")
    check(lib.imgan_paint_means(_ptr(imgb), _ptr(labels), _ptr(means_u8), imgb.shape[0],
                                labels[0].numel(), means_u8.shape[1], _stream()))
    return img


# ------------------------------------------------------------------------ CC / DCD

def cc_label(values):
    """8-connected components of equal non-negative values; returns root index per pixel."""
    _require_gpu()
    if values.dim() == 2:
        values = values.unsqueeze(0)
    _dev(values, torch.int32, "values")
    B, H, W = values.shape
    roots = torch.empty_like(values)
    check(lib.imgan_cc_label(_ptr(values), _ptr(roots), B, H, W, _stream()))
    return roots


def enforce_connectivity(labels, nlabels=None):
    """Connectivity enforcement for a superpixel label map (named by the task; scripts/slic.py has
    no such step, so the rule is this package's own and additive: SLICSuperPixel never applies it
    unless asked).  Components = 8-connected sets of equal labels, found by the label-propagation
    union-find kernel.  For every label its largest component stays (ties: the one that starts
    first in raster order); every other component takes the label of the pixel left of its first
    raster pixel (above it in column 0), as read from the input map; a component with no labelled
    pixel there keeps its label.  One pass; unlabelled pixels (-1) stay unlabelled.  All of it
    runs in imgan_enforce_connectivity.  labels: int32 [B,H,W] or [H,W] device tensor; nlabels: an
    upper bound on the label values + 1 (read from the map, with one synchronisation, when not
    given).  Returns a new tensor."""
    _require_gpu()
    squeeze = labels.dim() == 2
    if squeeze:
        labels = labels.unsqueeze(0)
    _dev(labels, torch.int32, "labels")
    B, H, W = labels.shape
    if nlabels is None:
        nlabels = max(int(labels.max().item()) + 1, 1)
    out = torch.empty_like(labels)
    scratch = torch.empty(lib.imgan_enforce_connectivity_scratch_bytes(B, H, W, int(nlabels)),
                          dtype=torch.uint8, device=labels.device)
    check(lib.imgan_enforce_connectivity(_ptr(labels), _ptr(out), B, H, W, int(nlabels), _ptr(scratch),
                                         _stream()))
    return out[0] if squeeze else out


class DcdPlan(object):
    """Dominant colours of every segment of a label map (mpeg7Descriptors.py:76-130).

    Segments that fit on chip are finished by the LDS-resident kernel in one launch (Lloyd
    rounds, statistics and the connected components behind the coherence value); larger ones
    (merged regions) continue through the streaming rounds and the whole-image component
    labelling.  Both give the same numbers."""

    def __init__(self, B, H, W, nseg, k, lists=True):
        """lists: sort the pixels by segment first (imgan_dcd_segments_listed; 8 bytes of
        scratch per pixel) instead of letting every segment scan its bounding box."""
        _require_gpu()
        self.B, self.H, self.W, self.nseg, self.k = int(B), int(H), int(W), int(nseg), int(k)
        dev = torch.device("cuda", torch.cuda.current_device())
        self.lists = bool(lists) and self.W <= 65536 and self.B * self.H * self.W < 2 ** 32
        self.list_scratch = self.counts = None
        if self.lists:
            self.list_scratch = torch.empty(
                lib.imgan_dcd_list_bytes(self.B, self.H, self.W, self.nseg), dtype=torch.uint8,
                device=dev)
            self.counts = torch.empty((self.B, self.nseg), dtype=torch.int32, device=dev)
        self.state = torch.zeros(lib.imgan_dcd_state_bytes(self.B, self.nseg, self.k),
                                 dtype=torch.uint8, device=dev)
        self.assign = torch.empty((self.B, self.H, self.W), dtype=torch.uint8, device=dev)
        self.active = torch.zeros(1, dtype=torch.int32, device=dev)
        self.n_big = torch.zeros(1, dtype=torch.int32, device=dev)
        self.stats = torch.empty((self.B, self.nseg, self.k, 8), dtype=torch.int64, device=dev)
        self.bbox = torch.empty(lib.imgan_dcd_bbox_bytes(self.B, self.nseg), dtype=torch.uint8,
                                device=dev)
        self.slots = None
        self.scratch = None
        self.rounds_run = 0          # streaming rounds issued for the large segments
        self.big_segments = 0

    # pixels the LDS-resident kernel holds per segment (its large tier, dcd_seg.hip)
    LDS_CAPACITY = 22528

    def kmeans(self, lab, labels, centres, seed=None, max_rounds=1000, check_every=1,
               streaming_only=False, boxed=False, max_segment_px=None, counted=False,
               resample_seed=0, write_assign=True):
        """Runs every segment's Lloyd loop to the reference's stop rule.
        centres: float32 [B,nseg,k,3], updated in place; with seed is not None they are
        sampled on the device (hash rule of imgan_dcd_init_hash), otherwise they must hold
        the initial centres.  boxed: self.bbox already holds the bounding boxes of `labels`
        (SlicPlan.round(..., bbox=self.bbox) on the round that made them).
        resample_seed: drives the re-seeding of clusters that run empty (mpeg7Descriptors.py:101;
        the rule is stated at imgan_dcd_kmeans_round in imgan.h) when `seed` is None; with a
        seed the same value serves both.
        counted: self.counts already holds the pixel count of every segment
        (SlicPlan.label_counts(out=self.counts)); counted otherwise (plans with lists only).
        write_assign: False leaves self.assign unwritten by the on-chip pass (the descriptor
        rows do not need the per-pixel map; pixel lists only).  Segments that end up in the
        streaming rounds write their part of it all the same.
        max_segment_px: the size the caller expects no segment to exceed (a SLIC superpixel
        normally lies inside its centre's 2S x 2S window, slic.py:94-100).  When it is within the
        on-chip capacity the count of segments left for the streaming rounds is not read back
        here, so the call does not synchronise.  That is an expectation, not a proof -- a pixel
        that no window reaches keeps a stale label (slic.py:81-116), which can stretch a label's
        box or count past the capacity -- so the counter stays on the device and the caller
        settles it later: unhandled_segments() reads it, finish_large() runs the streaming rounds
        for what was left (FullPipeline.finish(), bench.py after its timed loop)."""
        lab = as_batch(_dev(lab, torch.uint8, "lab"))
        if labels.dim() == 2:
            labels = labels.unsqueeze(0)
        _dev(labels, torch.int32, "labels")
        _dev(centres, torch.float32, "centres")
        if tuple(centres.shape) != (self.B, self.nseg, self.k, 3):
            raise ValueError("centres must be [B,nseg,k,3]")
        use_hash = seed is not None
        cseed = ctypes.c_uint32(int(seed if use_hash else resample_seed) & 0xffffffff)
        shape = (self.B, self.H, self.W, self.nseg, self.k)
        state_for_skip = _ptr(self.state)
        if streaming_only:
            self.state.zero_()
            self.big_segments = self.B * self.nseg
            state_for_skip = None
        else:
            if self.lists:
                if not boxed:
                    check(lib.imgan_label_bbox(_ptr(labels), self.B, self.H, self.W, self.nseg,
                                               _ptr(self.bbox), _stream()))
                if not counted:
                    check(lib.imgan_label_counts(_ptr(labels), self.B, self.H, self.W, self.nseg,
                                                 _ptr(self.counts), _stream()))
                check(lib.imgan_dcd_segments_listed(
                    _ptr(lab), _ptr(labels), *shape, 1 if use_hash else 0, cseed, _ptr(centres),
                    _ptr(self.assign) if write_assign else None, _ptr(self.state), _ptr(self.stats),
                    _ptr(self.bbox),
                    _ptr(self.counts), _ptr(self.list_scratch), _ptr(self.n_big), int(max_rounds),
                    _stream()))
            else:
                fn = lib.imgan_dcd_segments_boxed if boxed else lib.imgan_dcd_segments
                check(fn(_ptr(lab), _ptr(labels), *shape, 1 if use_hash else 0, cseed,
                         _ptr(centres), _ptr(self.assign), _ptr(self.state), _ptr(self.stats),
                         _ptr(self.bbox), _ptr(self.n_big), int(max_rounds), _stream()))
            self._pending = (use_hash, cseed, state_for_skip)
            if max_segment_px is not None and int(max_segment_px) <= self.LDS_CAPACITY:
                self.big_segments = 0               # deferred: see unhandled_segments()
                self.rounds_run = 0
                return centres, self.assign
            self.big_segments = int(self.n_big.item())
        self._pending = None
        return self._streaming_rounds(lab, labels, centres, use_hash, cseed, state_for_skip,
                                      max_rounds, check_every, streaming_only)

    def unhandled_segments(self):
        """Segments the last on-chip pass left for the streaming rounds (one synchronisation)."""
        return int(self.n_big.item())

    def finish_large(self, lab, labels, centres, max_rounds=1000, check_every=1):
        """Settles a kmeans() call that skipped the read-back of the counter: runs the streaming
        rounds for the segments the on-chip pass could not hold.  Returns their number."""
        if getattr(self, "_pending", None) is None:
            return 0
        use_hash, cseed, state_for_skip = self._pending
        self._pending = None
        self.big_segments = self.unhandled_segments()
        if self.big_segments:
            lab = as_batch(_dev(lab, torch.uint8, "lab"))
            if labels.dim() == 2:
                labels = labels.unsqueeze(0)
            self._streaming_rounds(lab, labels, centres, use_hash, cseed, state_for_skip,
                                   max_rounds, check_every, False)
        return self.big_segments

    def _streaming_rounds(self, lab, labels, centres, use_hash, cseed, state_for_skip, max_rounds,
                          check_every, streaming_only):
        shape = (self.B, self.H, self.W, self.nseg, self.k)
        self.rounds_run = 0
        if self.big_segments == 0:
            return centres, self.assign
        if use_hash:
            if self.slots is None:
                self.slots = torch.empty((self.B, self.nseg, self.k), dtype=torch.int64,
                                         device=self.state.device)
            check(lib.imgan_dcd_init_hash(_ptr(lab), _ptr(labels), *shape, cseed, _ptr(centres),
                                          _ptr(self.slots), state_for_skip, _stream()))
        r = 0
        while r < max_rounds:
            check(lib.imgan_dcd_kmeans_round(_ptr(lab), _ptr(labels), *shape, cseed, _ptr(centres),
                                             _ptr(self.assign), _ptr(self.state),
                                             1 if (r == 0 and streaming_only) else 0,
                                             _ptr(self.active), _stream()))
            r += 1
            if r % check_every == 0 and int(self.active.item()) == 0:
                break
        self.rounds_run = r
        return centres, self.assign

    def rounds(self):
        rounds = torch.empty((self.B, self.nseg), dtype=torch.int32, device=self.state.device)
        done = torch.empty_like(rounds)
        check(lib.imgan_dcd_rounds(_ptr(self.state), self.B, self.nseg, self.k, _ptr(rounds),
                                   _ptr(done), _stream()))
        return rounds, done

    def statistics(self, lab, labels, streaming_only=False):
        """uint64 [B,nseg,k,8]: count, sum c0..c2, sum of squares c0..c2, largest component."""
        lab = as_batch(_dev(lab, torch.uint8, "lab"))
        if labels.dim() == 2:
            labels = labels.unsqueeze(0)
        if not streaming_only and self.big_segments == 0:
            return self.stats            # every segment was finished on chip, coherence included
        if self.scratch is None:
            self.scratch = torch.empty((3, self.B, self.H, self.W), dtype=torch.int32,
                                       device=self.state.device)
        check(lib.imgan_dcd_stats(_ptr(lab), _ptr(labels), _ptr(self.assign), self.B, self.H,
                                  self.W, self.nseg, self.k, _ptr(self.stats), _ptr(self.scratch),
                                  None if streaming_only else _ptr(self.state), _stream()))
        return self.stats


# ------------------------------------------------------------------------ CLD / EHD / CSD

class DescriptorPlan(object):
    """Whole-image MPEG-7 colour layout, edge histogram and colour structure for a batch."""

    def __init__(self, B, H, W, nbins=256):
        _require_gpu()
        self.B, self.H, self.W, self.nbins = int(B), int(H), int(W), int(nbins)
        dev = torch.device("cuda", torch.cuda.current_device())
        self.cld = torch.empty((self.B, 3, 64), dtype=torch.int32, device=dev)
        self.ehd = torch.empty((self.B, 96), dtype=torch.int32, device=dev)
        self.csd = torch.empty((self.B, self.nbins + 1), dtype=torch.int32, device=dev)
        self.cld_scratch = torch.empty(max(lib.imgan_cld_scratch_bytes(self.B), 8),
                                       dtype=torch.uint8, device=dev)
        self.ehd_scratch = torch.empty(max(lib.imgan_ehd_scratch_bytes(self.B, self.H, self.W), 8),
                                       dtype=torch.uint8, device=dev)

    def _img(self, bgr):
        bgr = as_batch(_dev(bgr, torch.uint8, "bgr"))
        if tuple(bgr.shape) != (self.B, self.H, self.W, 3):
            raise ValueError("image batch does not match the plan")
        return bgr

    def colour_layout(self, bgr):
        bgr = self._img(bgr)
        check(lib.imgan_cld(_ptr(bgr), self.B, self.H, self.W, _ptr(self.cld),
                            _ptr(self.cld_scratch), _stream()))
        return self.cld

    def edge_histogram(self, bgr):
        bgr = self._img(bgr)
        check(lib.imgan_ehd(_ptr(bgr), self.B, self.H, self.W, _ptr(self.ehd),
                            _ptr(self.ehd_scratch), _stream()))
        return self.ehd

    def colour_layout_and_edge_histogram(self, bgr):
        """colour_layout() and edge_histogram() with one pass over the image for both
        (imgan_cld_ehd); returns (cld, ehd)."""
        bgr = self._img(bgr)
        check(lib.imgan_cld_ehd(_ptr(bgr), self.B, self.H, self.W, _ptr(self.cld), _ptr(self.ehd),
                                _ptr(self.cld_scratch), _ptr(self.ehd_scratch), _stream()))
        return self.cld, self.ehd

    def colour_structure(self, bgr):
        bgr = self._img(bgr)
        check(lib.imgan_csd(_ptr(bgr), self.B, self.H, self.W, self.nbins, _ptr(self.csd),
                            _stream()))
        return self.csd


def label_counts(labels, nseg):
    """int32 [B,nseg]: pixels per label (labels outside [0, nseg) are skipped)."""
    _require_gpu()
    if labels.dim() == 2:
        labels = labels.unsqueeze(0)
    _dev(labels, torch.int32, "labels")
    B, H, W = labels.shape
    counts = torch.empty((B, int(nseg)), dtype=torch.int32, device=labels.device)
    check(lib.imgan_label_counts(_ptr(labels), B, H, W, int(nseg), _ptr(counts), _stream()))
    return counts


def label_bbox(labels, nseg):
    """int32 [B,nseg,4] = xmin, ymin, xmax, ymax of every label (xmax = -1: no pixels)."""
    _require_gpu()
    if labels.dim() == 2:
        labels = labels.unsqueeze(0)
    _dev(labels, torch.int32, "labels")
    B, H, W = labels.shape
    bbox = torch.empty((B, int(nseg), 4), dtype=torch.int32, device=labels.device)
    check(lib.imgan_label_bbox(_ptr(labels), B, H, W, int(nseg), _ptr(bbox), _stream()))
    return bbox


class RegionDescriptorPlan(object):
    """Colour layout, edge histogram and colour structure of every region of a label map (one
    workgroup per region, bounding box staged in LDS; imgan_region_descriptors)."""

    MAX_BOX_CELLS = 24576

    def __init__(self, B, H, W, nseg, nbins=256):
        _require_gpu()
        self.B, self.H, self.W, self.nseg, self.nbins = int(B), int(H), int(W), int(nseg), int(nbins)
        dev = torch.device("cuda", torch.cuda.current_device())
        self.cld = torch.empty((self.B, self.nseg, 3, 64), dtype=torch.int32, device=dev)
        self.ehd = torch.empty((self.B, self.nseg, 96), dtype=torch.int32, device=dev)
        self.csd = torch.empty((self.B, self.nseg, self.nbins + 1), dtype=torch.int32, device=dev)
        self.too_big = torch.zeros(1, dtype=torch.int32, device=dev)
        self.large_scratch = None
        self._last = None
        self.scratch = torch.empty(max(lib.imgan_region_descriptors_scratch_bytes(self.B, self.nseg), 8),
                                   dtype=torch.uint8, device=dev)

    def run(self, bgr, labels, bbox=None, large=False, lists=None):
        """bbox: device buffer holding int32 [B,nseg,4] (e.g. the one the last SLIC round wrote);
        found with imgan_label_bbox when not given.  large: also finish the regions whose box is
        too large for the on-chip pass (finish_large(); synchronises).  lists: the list scratch of a
        DcdPlan whose listed dominant-colour pass has just run on the same labels and boxes
        (imgan_region_descriptors_listed: the regions' pixels come from their runs instead of a
        scan of their boxes).  Returns (cld, ehd, csd) device tensors."""
        bgr = as_batch(_dev(bgr, torch.uint8, "bgr"))
        if labels.dim() == 2:
            labels = labels.unsqueeze(0)
        _dev(labels, torch.int32, "labels")
        if tuple(bgr.shape) != (self.B, self.H, self.W, 3) or tuple(labels.shape) != (self.B, self.H, self.W):
            raise ValueError("batch does not match the plan")
        if bbox is not None and bbox.numel() * bbox.element_size() < self.B * self.nseg * 16:
            raise ValueError("bbox buffer too small")
        if lists is not None:
            if bbox is None:
                raise ValueError("the listed form needs the boxes the lists were made with")
            check(lib.imgan_region_descriptors_listed(
                _ptr(bgr), _ptr(bbox), _ptr(lists), self.B, self.H, self.W, self.nseg, self.nbins,
                _ptr(self.cld), _ptr(self.ehd), _ptr(self.csd), _ptr(self.too_big), _ptr(self.scratch),
                _stream()))
        else:
            if bbox is None:
                bbox = label_bbox(labels, self.nseg)
            check(lib.imgan_region_descriptors(_ptr(bgr), _ptr(labels), _ptr(bbox), self.B, self.H, self.W,
                                               self.nseg, self.nbins, _ptr(self.cld), _ptr(self.ehd),
                                               _ptr(self.csd), _ptr(self.too_big), _ptr(self.scratch),
                                               _stream()))
        self._last = (bgr, labels, bbox)
        if large:
            self.finish_large()
        return self.cld, self.ehd, self.csd

    def finish_large(self):
        """Regions whose box did not fit the on-chip pass (counted in self.too_big; their rows are
        zero until now) through imgan_region_descriptors_large, one region per call.  Synchronises
        once to read the counter, and once more for the boxes when it is not zero.  Returns the
        number of regions done this way."""
        n_big = int(self.too_big.item())
        if n_big == 0 or getattr(self, "_last", None) is None:
            return 0
        bgr, labels, bbox = self._last
        if isinstance(bbox, torch.Tensor) and bbox.dtype == torch.int32:
            boxes = bbox.reshape(-1)[:self.B * self.nseg * 4].view(self.B, self.nseg, 4).cpu().numpy()
        else:                                   # a byte buffer that starts with the boxes (DcdPlan.bbox)
            boxes = bbox.view(torch.int32)[:self.B * self.nseg * 4].view(self.B, self.nseg, 4).cpu().numpy()
        if self.large_scratch is None:
            self.large_scratch = torch.empty(lib.imgan_region_descriptors_large_scratch_bytes(self.H, self.W),
                                             dtype=torch.uint8, device=self.cld.device)
        done = 0
        for b in range(self.B):
            for r in range(self.nseg):
                x0, y0 = max(int(boxes[b, r, 0]), 0), max(int(boxes[b, r, 1]), 0)
                x1, y1 = min(int(boxes[b, r, 2]), self.W - 1), min(int(boxes[b, r, 3]), self.H - 1)
                if x1 < x0 or y1 < y0 or (x1 - x0 + 1) * (y1 - y0 + 1) <= self.MAX_BOX_CELLS:
                    continue
                check(lib.imgan_region_descriptors_large(
                    _ptr(bgr[b]), _ptr(labels[b]), self.H, self.W, r, x0, y0, x1, y1, self.nbins,
                    _ptr(self.cld[b, r]), _ptr(self.ehd[b, r]), _ptr(self.csd[b, r]),
                    _ptr(self.large_scratch), _stream()))
                done += 1
        self.too_big.zero_()
        return done


def hmmd_quantise(bgr, nbins=256):
    _require_gpu()
    _dev(bgr, torch.uint8, "bgr")
    out = torch.empty(bgr.shape[:-1], dtype=torch.uint8, device=bgr.device)
    check(lib.imgan_hmmd_quantise(_ptr(bgr), _ptr(out), bgr.numel() // 3, int(nbins), _stream()))
    return out
