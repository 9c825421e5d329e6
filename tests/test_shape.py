"""Region shape / ART (scripts/mpeg7Descriptors.py:144-186, 233-275): oracle known answers on
CPU, HIP path against the oracle on the GPU (polar image bit exact, ART within 1e-9; the
north_star tolerance for float descriptors is 1e-4 relative)."""
import numpy as np
import pytest

from oracle import shape_ref


def test_oracle_cart2radial_known_answers():
    H, W = 12, 20
    flat = np.full((H, W), 77, dtype=np.uint8)
    pol = shape_ref.cart2radial(flat)
    assert set(np.unique(pol)) <= {0, 77}          # window means of a constant, 0 where empty
    assert pol[0, 0] == 77                         # radius 0 -> the image centre
    # Python slice quirk (:165-174): at angle pi the sample leaves the frame on the left; with
    # int(px) <= -3 the stop index is negative and the window spans most of the row
    img = np.zeros((H, W), dtype=np.uint8)
    img[H // 2:H // 2 + 2, :] = 200                 # two bright rows through the centre
    pol = shape_ref.cart2radial(img)
    max_rad = np.sqrt(W ** 2 + H ** 2) / 2.0
    xa = W - 1
    px = xa * (max_rad / W) * np.cos((H // 2) * 2 * np.pi / H) + W / 2.0
    assert int(px) == -1 and pol[H // 2, xa] == 200   # [0:1] -> one column of the bright rows
    # a tall frame so that the sample goes 3+ pixels left of it: stop = int(px) + 2 < 0
    H2, W2, xa = 40, 8, 4
    img2 = np.zeros((H2, W2), dtype=np.uint8)
    img2[H2 // 2:H2 // 2 + 2, 0:2] = 100
    pol2 = shape_ref.cart2radial(img2)
    px2 = xa * (np.sqrt(W2 ** 2 + H2 ** 2) / 2.0 / W2) * np.cos((H2 // 2) * 2 * np.pi / H2) + W2 / 2.0
    stop = int(px2) + 2
    assert -W2 < stop < 0
    ncols = W2 + stop                               # image[:, 0:stop] counts from the end
    assert pol2[H2 // 2, xa] == (100 * 2 * min(2, ncols)) // (2 * ncols) != 0


def test_oracle_art_known_answers():
    H, W = 16, 16
    assert np.array(shape_ref.art_transform(np.zeros((H, W), np.uint8))).max() == 0.0
    rng = np.random.RandomState(1)
    art = np.array(shape_ref.art_transform(rng.randint(0, 255, (H, W)).astype(np.uint8)))
    assert art.shape == (3, 12) and art[0, 0] == 1.0 and (art >= 0).all()
    # a constant polar image has no angular variation: every m > 0 term vanishes
    pol = np.full((H, W), 9, np.uint8)
    art = np.array(shape_ref.art_transform(pol, polar_image=pol))
    assert np.abs(art[:, 1:]).max() < 1e-12


def _label_map(H, W, nseg, seed):
    rng = np.random.RandomState(seed)
    sy, sx = rng.randint(0, H, nseg), rng.randint(0, W, nseg)
    yy, xx = np.mgrid[0:H, 0:W]
    cl = ((yy[..., None] - sy) ** 2 + (xx[..., None] - sx) ** 2).argmin(axis=2).astype(np.float64)
    cl[-1, :] = -1
    cl[:, -1] = -1
    return cl


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,seed", [(24, 40, 1), (37, 29, 2), (64, 48, 3), (40, 8, 4), (5, 70, 5)])
def test_gpu_cart2radial_and_art_match_oracle(H, W, seed):
    import torch
    assert torch.cuda.is_available()
    from mpeg7Descriptors import MPEG7Descriptors
    rng = np.random.RandomState(seed)
    img = rng.randint(0, 256, (H, W)).astype(np.uint8)
    img[rng.rand(H, W) < 0.5] = 0
    pol = MPEG7Descriptors.cart2radial(img, one_dim=True)
    want = shape_ref.cart2radial(img)
    assert pol.dtype == np.uint8 and np.array_equal(pol, want)        # integer path: bit exact
    bgr = np.stack([img, img[::-1].copy(), img[:, ::-1].copy()], axis=-1)
    pol3 = MPEG7Descriptors.cart2radial(bgr)
    for c in range(3):
        assert np.array_equal(pol3[:, :, c], shape_ref.cart2radial(np.ascontiguousarray(bgr[:, :, c])))
    mp = MPEG7Descriptors(np.zeros((H, W)), np.stack([img] * 3, axis=-1))
    got = np.array(mp.art_transform(img))
    ref = np.array(shape_ref.art_transform(img, polar_image=want))
    assert got.shape == (3, 12)
    assert np.allclose(got, ref, rtol=1e-9, atol=1e-12)                # tolerance: 1e-9 (spec 1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,nseg,seed", [(32, 44, 4, 11), (48, 36, 6, 12)])
def test_gpu_region_shape_matches_oracle(H, W, nseg, seed):
    import torch
    assert torch.cuda.is_available()
    from mpeg7Descriptors import MPEG7Descriptors
    from conftest import voronoi_image
    cl = _label_map(H, W, nseg, seed)
    img = voronoi_image(H, W, 5, seed=seed)
    mp = MPEG7Descriptors(cl, img)
    got = mp.mpeg7_region_shape()
    want = shape_ref.region_shape(cl, mp.image_gray)
    assert set(got) == set(want)
    for s in want:
        assert np.allclose(np.array(got[s]), np.array(want[s]), rtol=1e-9, atol=1e-12), s
