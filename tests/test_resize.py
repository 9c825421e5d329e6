"""cv2.resize (INTER_LINEAR, 8 bit) as prepare_image uses it (optigen_utils.py:19-24): the numpy
restatement against the float64 definition of the filter, and the HIP kernel against the
restatement (exact).  Parity unpinned: no OpenCV here, no fixture in the reference."""
import numpy as np
import pytest

from oracle.resize_ref import resize_linear_u8, prepare_image


def _float_bilinear(img, Wd, Hd):
    Hs, Ws = img.shape[:2]

    def taps(src, dst):
        f = (np.arange(dst) + 0.5) * src / dst - 0.5
        s = np.floor(f).astype(int)
        f = f - s
        f[s < 0], s[s < 0] = 0, 0
        hi = s >= src - 1
        f[hi], s[hi] = 0, src - 1
        return s, np.minimum(s + 1, src - 1), f
    sx, sx1, fx = taps(Ws, Wd)
    sy, sy1, fy = taps(Hs, Hd)
    a = img.astype(np.float64)
    r = a[:, sx] * (1 - fx)[None, :, None] + a[:, sx1] * fx[None, :, None]
    return r[sy] * (1 - fy)[:, None, None] + r[sy1] * fy[:, None, None]


@pytest.mark.parametrize("src,dst", [((413, 795), (320, 166)), ((50, 60), (151, 97)), ((261, 320), (320, 261)),
                                     ((7, 5), (3, 11)), ((1, 9), (4, 1))])
def test_fixed_point_filter_tracks_float_bilinear(src, dst):
    rng = np.random.RandomState(src[0])
    img = rng.randint(0, 256, src + (3,)).astype(np.uint8)
    got = resize_linear_u8(img, dst)
    assert got.shape == (dst[1], dst[0], 3)
    assert np.abs(got.astype(np.float64) - _float_bilinear(img, dst[0], dst[1])).max() <= 1.0


def test_known_answers():
    const = np.full((30, 40, 3), 77, dtype=np.uint8)
    assert (resize_linear_u8(const, (17, 9)) == 77).all()             # weights sum to 2048
    img = np.arange(8 * 6 * 3, dtype=np.uint8).reshape(8, 6, 3)
    half = resize_linear_u8(img, (3, 4))                               # exact 2 x 2 -> INTER_AREA
    s = img.astype(int)
    assert np.array_equal(half, (s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2)
    assert np.array_equal(resize_linear_u8(img, (6, 8)), img)          # identity
    bear = np.zeros((1050, 1573, 3), dtype=np.uint8)
    assert prepare_image(bear).shape == (213, 320, 3)                  # SURVEY section 8 grid table


@pytest.mark.gpu
@pytest.mark.parametrize("src,dst", [((413, 795), (320, 166)), ((50, 60), (151, 97)), ((680, 1024), (320, 212)),
                                     ((64, 128), (64, 32)), ((7, 5), (3, 11)), ((1, 9), (4, 1))])
def test_gpu_resize_equals_restatement(src, dst):
    import torch
    from imgan import ops
    rng = np.random.RandomState(dst[0])
    imgs = rng.randint(0, 256, (2,) + src + (3,)).astype(np.uint8)
    got = ops.resize_linear(torch.from_numpy(imgs).cuda(), dst).cpu().numpy()
    for b in range(2):
        assert np.array_equal(got[b], resize_linear_u8(imgs[b], dst)), b


@pytest.mark.gpu
def test_gpu_prepare_image():
    from imgan.prepare import prepare_image as gpu_prepare
    rng = np.random.RandomState(3)
    img = rng.randint(0, 256, (478, 850, 3)).astype(np.uint8)          # data/bird.jpg's size
    assert np.array_equal(gpu_prepare(img), prepare_image(img))
    with pytest.raises(ValueError):
        gpu_prepare(None)
