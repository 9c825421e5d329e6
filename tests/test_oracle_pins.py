"""Pins of oracle pieces to third-party libraries importable here (scipy), and to the float64
definition of CIELAB: the reference holds no fixtures for this path and cannot run,
so these are the only checks of the oracle that do not come from the oracle itself."""
import numpy as np
import pytest
import scipy.fft
import scipy.ndimage

from oracle import color, mpeg7_std


@pytest.mark.parametrize("ksize", [3, 5, 7])
def test_median_blur_equals_scipy_median_filter(ksize):
    # cv2.medianBlur (rag_segmentation.py:11): per-channel k x k median, replicated border
    rng = np.random.RandomState(5 + ksize)
    img = rng.randint(0, 256, (37, 53, 3)).astype(np.uint8)
    img[10:20, 5:30] = rng.randint(0, 4, (10, 25, 3))          # many equal values
    want = scipy.ndimage.median_filter(img, size=(ksize, ksize, 1), mode="nearest")
    assert np.array_equal(color.median_blur(img, ksize), want)


def test_dct_table_is_the_orthonormal_dct2():
    c = mpeg7_std.dct_table()
    want = scipy.fft.dct(np.eye(8), type=2, norm="ortho", axis=0)
    assert np.abs(c - want).max() < 1e-15
    rng = np.random.RandomState(1)
    blk = rng.randint(0, 256, (8, 8)).astype(np.float64)
    # the serial statement of cld_from_sums is c @ blk @ c.T
    assert np.abs(c @ blk @ c.T - scipy.fft.dctn(blk, type=2, norm="ortho")).max() < 1e-10


def test_cld_coefficients_follow_scipy_dctn():
    """The unquantised DCT of cld_from_sums, recomputed with scipy.fft.dctn, quantises to the
    same descriptor (a coefficient within 1e-6 of a rounding boundary would be a coincidence)."""
    rng = np.random.RandomState(7)
    img = rng.randint(0, 256, (64, 96, 3)).astype(np.uint8)
    sums, cnt = mpeg7_std.cld_block_sums(img)
    got = mpeg7_std.cld_from_sums(sums, cnt)
    avg = sums // cnt[:, None]
    B, G, R = (avg[:, k].astype(np.float64) for k in range(3))
    yy = (0.299 * R + 0.587 * G + 0.114 * B) / 256.0
    planes = [np.floor(219.0 * yy + 16.5), np.floor(224.0 * 0.564 * (B / 256.0 - yy) + 128.5),
              np.floor(224.0 * 0.713 * (R / 256.0 - yy) + 128.5)]
    for ch in range(3):
        coef = np.floor(scipy.fft.dctn(planes[ch].reshape(8, 8), type=2, norm="ortho") + 0.499999)
        coef = coef.astype(np.int64).ravel()
        dc = mpeg7_std._cdiv(int(coef[0]), 8)
        want0 = (mpeg7_std._quant_ydc(dc) >> 1) if ch == 0 else mpeg7_std._quant_cdc(dc)
        assert got[ch, 0] == want0
        for i in range(1, 64):
            v = int(coef[mpeg7_std.ZIGZAG[i]])
            want = mpeg7_std._quant_ac(mpeg7_std._cdiv(v, 2) if ch == 0 else v) >> 3
            assert got[ch, i] == want


def _float_lab(b, g, r):
    """CIELAB in float64 from 8-bit sRGB with the matrix / white point the 8-bit path rounds
    (oracle.color._M, _WHITE), scaled as cv2 scales 8-bit output: L*255/100, a+128, b+128."""
    def lin(c):
        x = c / 255.0
        return np.where(x <= 0.04045, x / 12.92, ((x + 0.055) / 1.055) ** 2.4)
    R, G, B = lin(r), lin(g), lin(b)
    M, Wt = color._M, color._WHITE
    X = (M[0, 0] * R + M[0, 1] * G + M[0, 2] * B) / Wt[0]
    Y = (M[1, 0] * R + M[1, 1] * G + M[1, 2] * B) / Wt[1]
    Z = (M[2, 0] * R + M[2, 1] * G + M[2, 2] * B) / Wt[2]

    def f(t):
        return np.where(t > 0.008856, np.cbrt(t), 7.787 * t + 16.0 / 116.0)
    fx, fy, fz = f(X), f(Y), f(Z)
    L = np.where(Y > 0.008856, 116.0 * fy - 16.0, 903.3 * Y)
    return L * 255.0 / 100.0, 500.0 * (fx - fy) + 128.0, 200.0 * (fy - fz) + 128.0


# measured over all 2^24 colours (this test): the fixed-point path stays within these many 8-bit
# steps of the float64 definition; DESIGN.md §2 quotes them
LAB_MAX_ERR = (1.55, 2.67, 1.70)


def test_integer_lab_tracks_float_cielab_over_all_colours():
    worst = np.zeros(3)
    g8, r8 = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    for b in range(256):
        img = np.stack([np.full_like(g8, b), g8, r8], axis=-1)
        got = color.bgr2lab_u8(img).astype(np.float64)
        L, A, Bq = _float_lab(float(b), g8.astype(np.float64), r8.astype(np.float64))
        for k, want in enumerate((L, A, Bq)):
            worst[k] = max(worst[k], np.abs(got[..., k] - np.clip(want, 0, 255)).max())
    print("max |integer Lab - float64 CIELAB| in 8-bit steps (L, a, b):", worst)
    assert (worst <= np.array(LAB_MAX_ERR)).all(), worst
