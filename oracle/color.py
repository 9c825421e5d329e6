"""Oracle: 8-bit BGR->CIELAB, BGR->gray and 7x7 median (TEST INFRASTRUCTURE ONLY).

Call sites in the reference: `cv2.cvtColor(.., COLOR_BGR2Lab)` slic.py:13,
rag_segmentation.py:12, mpeg7Descriptors.py:22; `COLOR_BGR2GRAY`
mpeg7Descriptors.py:23; `cv2.medianBlur(image, 7)` rag_segmentation.py:11.

The arithmetic lives in OpenCV (not vendored, version unpinned, absent from
this image).  It is restated from the published OpenCV 3.x algorithm
(imgproc/color.cpp `RGB2Lab_b`, `RGB2Gray<uchar>`) with one
stated substitution: OpenCV's approximate `cvCbrt` is replaced by the
correctly rounded cube root.  Parity unpinned: no cv2 here to check against.
"""
import numpy as np

LAB_SHIFT = 12
GAMMA_SHIFT = 3
LAB_SHIFT2 = 15
CBRT_TAB_SIZE = 256 * 3 // 2 * (1 << GAMMA_SHIFT)  # 3072

_M = np.array([[0.412453, 0.357580, 0.180423],
               [0.212671, 0.715160, 0.072169],
               [0.019334, 0.119193, 0.950227]], dtype=np.float64)
_WHITE = np.array([0.950456, 1.0, 1.088754], dtype=np.float64)


def _round_half_even(x):
    return np.rint(x).astype(np.int64)


def lab_tables():
    """(gammaTab u16[256], cbrtTab u16[3072], coeffs int32[3][3] for R,G,B order)."""
    f32 = np.float32
    i = np.arange(256, dtype=np.float32)
    x = i * f32(1.0 / 255.0)                                   # float32, as OpenCV
    lin = x * f32(1.0 / 12.92)
    pw = np.power((x.astype(np.float64) + 0.055) * (1.0 / 1.055), 2.4).astype(np.float32)
    g = np.where(x <= f32(0.04045), lin, pw).astype(np.float32)
    gamma = np.clip(_round_half_even(f32(255.0 * (1 << GAMMA_SHIFT)) * g), 0, 65535).astype(np.uint16)

    j = np.arange(CBRT_TAB_SIZE, dtype=np.float32)
    y = j * f32(1.0 / (255.0 * (1 << GAMMA_SHIFT)))
    lo = y * f32(7.787) + f32(0.13793103448275862)
    cb = np.cbrt(y.astype(np.float64)).astype(np.float32)       # correctly rounded cbrt (see header)
    c = np.where(y < f32(0.008856), lo, cb).astype(np.float32)
    cbrt = np.clip(_round_half_even(f32(1 << LAB_SHIFT2) * c), 0, 65535).astype(np.uint16)

    coeffs = _round_half_even((1 << LAB_SHIFT) * _M / _WHITE[:, None]).astype(np.int32)
    return gamma, cbrt, coeffs


_TABLES = None


def _tables():
    global _TABLES
    if _TABLES is None:
        _TABLES = lab_tables()
    return _TABLES


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def bgr2lab_u8(img):
    """H x W x 3 uint8 BGR -> H x W x 3 uint8 (L*255/100, a+128, b+128)."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.shape[-1] == 3
    gamma, cbrt, C = _tables()
    g = gamma.astype(np.int64)
    B = g[img[..., 0]]
    G = g[img[..., 1]]
    R = g[img[..., 2]]
    cb = cbrt.astype(np.int64)
    C = C.astype(np.int64)
    fX = cb[_descale(R * C[0, 0] + G * C[0, 1] + B * C[0, 2], LAB_SHIFT)]
    fY = cb[_descale(R * C[1, 0] + G * C[1, 1] + B * C[1, 2], LAB_SHIFT)]
    fZ = cb[_descale(R * C[2, 0] + G * C[2, 1] + B * C[2, 2], LAB_SHIFT)]
    Lscale = (116 * 255 + 50) // 100
    Lshift = -((16 * 255 * (1 << LAB_SHIFT2) + 50) // 100)
    L = _descale(Lscale * fY + Lshift, LAB_SHIFT2)
    a = _descale(500 * (fX - fY) + 128 * (1 << LAB_SHIFT2), LAB_SHIFT2)
    b = _descale(200 * (fY - fZ) + 128 * (1 << LAB_SHIFT2), LAB_SHIFT2)
    out = np.stack([L, a, b], axis=-1)
    return np.clip(out, 0, 255).astype(np.uint8)


def bgr2gray_u8(img):
    """H x W x 3 uint8 BGR -> H x W uint8; (4899 R + 9617 G + 1868 B + 8192) >> 14."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.shape[-1] == 3
    b = img[..., 0].astype(np.int64)
    g = img[..., 1].astype(np.int64)
    r = img[..., 2].astype(np.int64)
    return ((4899 * r + 9617 * g + 1868 * b + 8192) >> 14).astype(np.uint8)


def median_blur(img, ksize=7):
    """cv2.medianBlur(img, ksize) for uint8 H x W x C: per-channel ksize x ksize
    median with replicated border (rag_segmentation.py:11)."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and ksize % 2 == 1
    r = ksize // 2
    squeeze = img.ndim == 2
    if squeeze:
        img = img[..., None]
    H, W, C = img.shape
    pad = np.pad(img, ((r, r), (r, r), (0, 0)), mode="edge")
    stack = np.empty((ksize * ksize, H, W, C), dtype=np.uint8)
    k = 0
    for dy in range(ksize):
        for dx in range(ksize):
            stack[k] = pad[dy:dy + H, dx:dx + W]
            k += 1
    out = np.partition(stack, (ksize * ksize) // 2, axis=0)[(ksize * ksize) // 2]
    return out[..., 0] if squeeze else out
