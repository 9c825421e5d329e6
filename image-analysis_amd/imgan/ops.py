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


def median7(bgr, out=None):
    """cv2.medianBlur(bgr, 7) (rag_segmentation.py:11)."""
    _require_gpu()
    img = as_batch(_dev(bgr, torch.uint8, "bgr"))
    if out is None:
        out = torch.empty_like(bgr)
    B, H, W, _ = img.shape
    check(lib.imgan_median7_u8c3(_ptr(img), _ptr(out), B, H, W, _stream()))
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

    def round(self, lab, colour_weight, events=None):
        """One iteration of slic.py:84-126.  `events`: optional (start, end) torch events
        recorded around the assignment kernel on the current stream (bench.py roofline)."""
        lab = self._check(lab)
        a = (self.B, self.H, self.W, self.S, self.n)
        check(lib.imgan_slic_bin_centres(_ptr(lab), _ptr(self.centres), *a, _ptr(self.workspace),
                                         _stream()))
        if events:
            events[0].record()
        check(lib.imgan_slic_assign_accumulate(_ptr(lab), _ptr(self.labels), *a,
                                               float(colour_weight), _ptr(self.workspace),
                                               _stream()))
        if events:
            events[1].record()
        check(lib.imgan_slic_finalise_centres(_ptr(self.centres), *a, _ptr(self.workspace),
                                              _stream()))

    def run(self, lab, colour_weight, num_iter):
        """SLICSuperPixel.run() (slic.py:78-127); returns (labels, centres) device tensors."""
        lab = self._check(lab)
        check(lib.imgan_slic_run(_ptr(lab), self.B, self.H, self.W, self.S, self.nx, self.ny,
                                 float(colour_weight), int(num_iter), _ptr(self.centres),
                                 _ptr(self.labels), _ptr(self.workspace), _stream()))
        return self.labels, self.centres
