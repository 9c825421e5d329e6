"""Homogeneous texture (scripts/mpeg7Descriptors.py:277-365): the oracle runs the reference's
own third-party calls (scipy.ndimage.rotate, scipy.fftpack); the HIP path restates scipy's
spline rotation and must match within 1e-6 relative (north_star: 1e-4 for float descriptors)."""
import numpy as np
import pytest

from oracle import texture_ref


def test_oracle_texture_known_answers():
    z = texture_ref.radon_feature_extract(np.zeros((12, 20), np.uint8))
    assert len(z) == 32 and z[30] == 0.0 and z[31] == 0.0
    assert all(f['e'] == 0.0 and f['d'] == 0.0 for f in z[:30])       # log10(1 + 0)
    img = np.full((16, 16), 255, np.uint8)
    f = texture_ref.radon_feature_extract(img)
    assert abs(f[30] - 1.0) < 1e-7 and f[31] < 1e-7                   # mean 1, std 0
    # padding to a square (:313-318): a 8 x 16 image is padded below, the mean halves
    f = texture_ref.radon_feature_extract(np.full((8, 16), 255, np.uint8))
    assert abs(f[30] - 0.5) < 1e-7 and abs(f[31] - 0.5) < 1e-7
    g = texture_ref.gabor_filters(5, 6, (329, 24))
    assert len(g) == 30 and g[0].shape == (329, 24) and all((m >= 0).all() and (m <= 1).all() for m in g)


def _close(a, b, rtol=1e-6, atol=1e-9):
    return abs(a - b) <= atol + rtol * abs(b)


def _compare(got, want):
    assert len(got) == len(want) == 32
    for c in range(30):
        assert _close(got[c]['e'], want[c]['e']) and _close(got[c]['d'], want[c]['d']), (c, got[c], want[c])
    assert _close(got[30], want[30], 1e-12) and _close(got[31], want[31], 1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,seed", [(24, 24, 1), (20, 36, 2), (45, 31, 3), (64, 64, 4)])
def test_gpu_radon_features_match_scipy_oracle(H, W, seed):
    import torch
    assert torch.cuda.is_available()
    from mpeg7Descriptors import MPEG7Descriptors
    from conftest import voronoi_image
    img = voronoi_image(H, W, 6, seed=seed)[:, :, 1].copy()
    got = MPEG7Descriptors.radon_feature_extract(img)
    _compare(got, texture_ref.radon_feature_extract(img))


@pytest.mark.gpu
def test_gpu_homogeneous_texture_per_segment():
    import torch
    assert torch.cuda.is_available()
    from mpeg7Descriptors import MPEG7Descriptors
    from conftest import voronoi_image
    H, W = 40, 56
    img = voronoi_image(H, W, 5, seed=9)
    rng = np.random.RandomState(3)
    sy, sx = rng.randint(0, H, 4), rng.randint(0, W, 4)
    yy, xx = np.mgrid[0:H, 0:W]
    cl = ((yy[..., None] - sy) ** 2 + (xx[..., None] - sx) ** 2).argmin(axis=2).astype(np.float64)
    cl[-1, :] = -1
    cl[:, -1] = -1
    mp = MPEG7Descriptors(cl, img)
    got = mp.mpeg7_homogeneus_texture()
    want = texture_ref.homogeneous_texture(cl, mp.image_gray)
    assert set(got) == set(want)
    for s in want:
        _compare(got[s], want[s])
