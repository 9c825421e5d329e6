"""Edge cases of the drop-in path on the GPU: degenerate sizes, empty / single segments,
error behaviour (the reference's only explicit error is ValueError, slic.py:10-11; the others
are what its numpy / random calls raise on the same input)."""
import numpy as np
import pytest

from conftest import voronoi_image

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def test_slic_degenerate_inputs():
    _torch()
    from slic import SLICSuperPixel
    from oracle.slic_ref import SLICOracle
    with pytest.raises(ValueError):
        SLICSuperPixel(None, 10)                                   # slic.py:10-11
    with pytest.raises(ValueError):
        SLICSuperPixel([[1, 2, 3]], 10)
    img = voronoi_image(12, 16, 2, seed=1)
    with pytest.raises((ValueError, ZeroDivisionError)):
        SLICSuperPixel(img, 12 * 16 + 1).run()                     # step = int(sqrt(0)) = 0
    # step larger than the image: the seed grid is empty, the loops do nothing (no error)
    sp = SLICSuperPixel(img, 1, num_iter=3)
    sp.run()
    ora = SLICOracle(img, 1, num_iter=3, mode="float64")
    ora.run()
    assert sp.centers == [] and len(ora.centers) == 0 and sp.step == ora.step
    assert np.array_equal(sp.get_clusters(), ora.get_clusters()) and (sp.get_clusters() == -1).all()
    # one centre only: every covered pixel gets label 0, the rest stays -1
    sp = SLICSuperPixel(img, 4, num_iter=3, step=7)
    sp.run()
    ora = SLICOracle(img, 4, num_iter=3, step=7, mode="float64")
    ora.run()
    assert len(sp.centers) == 1
    assert np.array_equal(sp.get_clusters(), ora.get_clusters())
    assert set(np.unique(sp.get_clusters())) <= {-1.0, 0.0}
    # num_iter = 1: zero rounds (range(1, 1)), nothing is labelled
    sp = SLICSuperPixel(img, 4, num_iter=1)
    sp.run()
    assert (sp.get_clusters() == -1).all()
    # flat image: pure spatial Voronoi with lowest-index ties (all keys tie in colour)
    flat = np.full((40, 52, 3), 90, np.uint8)
    sp = SLICSuperPixel(flat, 12, colour_weight=15, num_iter=4)
    sp.run()
    ora = SLICOracle(flat, 12, colour_weight=15, num_iter=4, mode="float64")
    ora.run()
    assert np.array_equal(sp.get_clusters(), ora.get_clusters())


@pytest.mark.parametrize("shape", [(9, 13), (33, 35), (64, 61)])
def test_slic_ragged_widths_match_oracle(shape):
    _torch()
    from slic import SLICSuperPixel
    from oracle.slic_ref import SLICOracle
    H, W = shape
    img = voronoi_image(H, W, 4, seed=H)
    sp = SLICSuperPixel(img, 6, colour_weight=15, num_iter=5)
    sp.run()
    ora = SLICOracle(img, 6, colour_weight=15, num_iter=5, mode="float64")
    ora.run()
    assert np.array_equal(sp.get_clusters(), ora.get_clusters())


def test_rag_on_degenerate_label_maps():
    torch = _torch()
    from imgan import ops
    H, W, K = 24, 32, 8
    img = torch.from_numpy(voronoi_image(H, W, 3, seed=2)).cuda()
    none = torch.full((1, H, W), -1, dtype=torch.int32, device="cuda")
    sums, edges = ops.rag_accumulate(img, none, K)
    assert int(sums.abs().sum()) == 0 and len(edges[0]) == 0       # nothing labelled
    one = torch.zeros((1, H, W), dtype=torch.int32, device="cuda")
    sums, edges = ops.rag_accumulate(img, one, K)
    assert int(sums[0, 0, 3]) == H * W and len(edges[0]) == 0      # a single region, no edges
    res = ops.rag_merge_host(K, edges[0], 0.65, means=np.zeros((K, 3), np.int32))
    assert res["lut"].tolist() == list(range(1, K + 1))            # no edges: every id its own set
    merged = ops.relabel(none, torch.from_numpy(res["lut"]).cuda())
    assert int(merged.abs().sum()) == 0                            # -1 -> region 0


def test_dominant_colour_small_segments():
    _torch()
    from mpeg7Descriptors import MPEG7Descriptors
    img = voronoi_image(20, 28, 3, seed=3)
    cl = np.zeros((20, 28))
    cl[0, 0:3] = 1                                                 # a 3-pixel segment
    mp = MPEG7Descriptors(cl, img, seed=1)
    with pytest.raises(ValueError):                                # random.sample(3 points, 8) (:79)
        mp.mpeg7_dominant_colours(8)
    out = mp.mpeg7_dominant_colours(2)
    assert set(out) == {0, 1} and len(out[1]) == 2
    pct = sum(v[2] for v in out[1].values())
    assert abs(pct - 1.0) < 1e-12


def test_whole_image_descriptors_tiny_images():
    _torch()
    from mpeg7Descriptors import MPEG7Descriptors
    from oracle import mpeg7_std as M
    for H, W in ((8, 8), (9, 17), (16, 8)):
        img = voronoi_image(H, W, 2, seed=H + W)
        mp = MPEG7Descriptors(np.zeros((H, W)), img)
        assert mp.mpeg7_colour_layout() == M.colour_layout(img).tolist()
        c, n = M.edge_histogram_counts(img)
        assert mp.mpeg7_edge_histogram(quantised=False) == (c.tolist(), n.tolist())
        h, nw = M.colour_structure_counts(img)
        got_h, got_nw = mp.mpeg7_colour_structure(quantised=False)
        assert got_h == h.tolist() and got_nw == nw


def test_spatial_and_shape_single_segment():
    _torch()
    from spatialextract import SpatialExtractor
    from mpeg7Descriptors import MPEG7Descriptors
    from oracle.spatial_ref import SpatialOracle
    from oracle import shape_ref
    cl = np.zeros((18, 26))
    rel = SpatialExtractor(cl).find_spatial(0.85)
    assert rel == SpatialOracle(cl).find_spatial(0.85)
    assert all(v == {} for v in rel[0].values())                   # no other segment to relate to
    img = voronoi_image(18, 26, 2, seed=5)
    mp = MPEG7Descriptors(cl, img)
    got = mp.mpeg7_region_shape()
    want = shape_ref.region_shape(cl, mp.image_gray)
    assert np.allclose(np.array(got[0]), np.array(want[0]), rtol=1e-9, atol=1e-12)
    line = np.arange(40, dtype=np.uint8).reshape(1, 40) * 5        # a 1 x N image
    assert np.array_equal(MPEG7Descriptors.cart2radial(line, one_dim=True), shape_ref.cart2radial(line))


def test_connectivity_enforcement_matches_own_oracle():
    """Additive step named by the task (no counterpart in scripts/slic.py, parity unpinned): the
    device path (union-find components + per-component relabel) against the scipy restatement."""
    torch = _torch()
    from imgan import ops
    from oracle.slic_ref import enforce_connectivity
    rng = np.random.RandomState(12)
    cases = []
    a = np.repeat(np.repeat(rng.randint(0, 6, (9, 12)), 8, axis=0), 8, axis=1).astype(np.int32)
    a[rng.rand(*a.shape) < 0.04] = 3                       # speckle: many one-pixel orphans
    a[-1, :] = -1
    a[:, -1] = -1                                          # the border SLIC leaves unlabelled
    cases.append(a)
    b = np.zeros((20, 30), np.int32)
    b[:, 10:20] = 1
    b[5:8, 12:15] = 0                                      # an island of 0 inside 1
    b[0, 0] = 1                                            # orphan in the corner: no left / upper pixel
    b[10, 0] = 1                                           # orphan in column 0: takes the label above
    cases.append(b)
    cases.append(np.full((7, 9), -1, np.int32))            # nothing labelled
    cases.append(np.arange(6 * 5, dtype=np.int32).reshape(6, 5))   # every pixel its own label
    for c in cases:
        got = ops.enforce_connectivity(torch.from_numpy(c).cuda()).cpu().numpy()
        assert np.array_equal(got, enforce_connectivity(c))
    # on the class: labels after run() stay a valid superpixel map, orphans only shrink in number
    from slic import SLICSuperPixel
    from scipy import ndimage
    sp = SLICSuperPixel(voronoi_image(120, 160, 10, seed=8, noise=25.0), 40, colour_weight=4, num_iter=5)
    sp.run()
    before = sp.get_clusters().copy()
    after = sp.enforce_connectivity()

    def ncomp(m):
        return sum(ndimage.label(m == l, structure=np.ones((3, 3)))[1] for l in np.unique(m) if l >= 0)
    assert np.array_equal(after, enforce_connectivity(before.astype(np.int32)).astype(np.float64))
    assert ncomp(after) <= ncomp(before) and (after[before < 0] == -1).all()
