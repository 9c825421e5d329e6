"""Oracle: cv2.resize(img, (w, h)) with the default INTER_LINEAR for 8-bit images, as
optigen_utils.prepare_image (:19-24) and basegen.py:28-30 call it (TEST INFRASTRUCTURE ONLY).

The arithmetic lives in OpenCV (not vendored, version unpinned, absent from this image); restated
from the published algorithm of imgproc/resize.cpp (3.x), parity unpinned:
  * source coordinate of destination x: fx = float((x + 0.5) * scale - 0.5), scale = src / dst in
    double; sx = floor(fx); fx -= sx; sx < 0 -> (0, fx = 0); sx >= src - 1 -> (src - 1, fx = 0)
  * 11-bit weights: a0 = round(float(1 - fx) * 2048), a1 = round(fx * 2048) (round half to even,
    float32 products), the same along y
  * rows: h = S[sx] * a0 + S[sx + 1] * a1 (int); columns: ((b0 * (h0 >> 4)) >> 16) +
    ((b1 * (h1 >> 4)) >> 16) + 2) >> 2
  * an exact 2 x 2 reduction is routed to INTER_AREA: (a + b + c + d + 2) >> 2
OpenCV builds with IPP may take another path; no fixture of the reference pins either."""
import numpy as np


def _taps(src, dst):
    scale = np.float64(1.0) / (np.float64(dst) / np.float64(src))     # OpenCV: 1 / inv_scale
    x = np.arange(dst, dtype=np.float64)
    fx = ((x + 0.5) * scale - 0.5).astype(np.float32)
    sx = np.floor(fx).astype(np.int64)
    fx = (fx - sx.astype(np.float32)).astype(np.float32)
    low = sx < 0
    fx[low], sx[low] = 0.0, 0
    high = sx >= src - 1
    fx[high], sx[high] = 0.0, src - 1
    a0 = np.rint((np.float32(1.0) - fx).astype(np.float32) * np.float32(2048.0)).astype(np.int64)
    a1 = np.rint(fx * np.float32(2048.0)).astype(np.int64)
    return sx, np.minimum(sx + 1, src - 1), a0, a1


def resize_linear_u8(img, dsize):
    """img: H x W x C uint8; dsize = (width, height) as cv2.resize takes it."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    Hs, Ws = img.shape[:2]
    Wd, Hd = int(dsize[0]), int(dsize[1])
    assert Wd >= 1 and Hd >= 1
    if Ws == 2 * Wd and Hs == 2 * Hd:                          # the INTER_AREA fast path
        s = img.astype(np.int64)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    sx, sx1, a0, a1 = _taps(Ws, Wd)
    sy, sy1, b0, b1 = _taps(Hs, Hd)
    s = img.astype(np.int64)
    rows = s[:, sx] * a0[None, :, None] + s[:, sx1] * a1[None, :, None]      # Hs x Wd x C, x 2048
    h0, h1 = rows[sy], rows[sy1]
    out = (((b0[:, None, None] * (h0 >> 4)) >> 16) + ((b1[:, None, None] * (h1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def prepare_image(img, default_wres=320):
    """optigen_utils.py:19-24."""
    H, W = img.shape[:2]
    factor = H / float(W)
    return resize_linear_u8(img, (default_wres, int(default_wres * factor)))
