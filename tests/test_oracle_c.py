"""The C restatement (oracle/c/imgan_oracle.c) against the numpy oracle: two independent
statements of slic.py:35-127 and mpeg7Descriptors.py:76-130 must agree bit for bit."""
import numpy as np
import pytest

from conftest import voronoi_image
from oracle import c_oracle, dcd_ref
from oracle.color import bgr2lab_u8
from oracle.slic_ref import SLICOracle


@pytest.mark.parametrize("H,W,K,cw,iters", [(64, 96, 24, 10, 10), (101, 157, 60, 15, 6),
                                            (120, 90, 3, 40, 4), (50, 70, 100, 15, 3)])
def test_c_slic_equals_numpy_oracle(H, W, K, cw, iters):
    lab = bgr2lab_u8(voronoi_image(H, W, max(K, 4), seed=H + W))
    o = SLICOracle(lab=lab, num_clusters=K, colour_weight=cw, num_iter=iters)
    want = o.run()
    labels, centres = c_oracle.slic_run(lab, K, cw, iters)
    assert np.array_equal(labels.astype(np.float64), want)
    assert np.array_equal(centres, np.asarray(o.centers, dtype=np.int32).reshape(-1, 5))


def test_c_dominant_colour_equals_numpy_oracle():
    rng = np.random.RandomState(3)
    H, W = 40, 52
    lab = bgr2lab_u8(voronoi_image(H, W, 6, seed=11))
    mask = np.zeros((H, W), bool)
    mask[5:33, 7:45] = rng.rand(28, 38) > 0.15            # ragged segment with holes
    idx = np.transpose(np.nonzero(mask))
    pts = lab[mask]
    for k in (1, 3, 8):
        init = rng.choice(pts.shape[0], k, replace=False)
        a_np, c_np, r_np = dcd_ref.lloyd(pts, init)
        assign, centres, stats, rounds = c_oracle.dcd_segment(pts, idx, init)
        assert rounds == r_np and np.array_equal(assign, a_np)
        assert np.array_equal(centres.view(np.uint32), c_np.view(np.uint32))
        feats, _ = dcd_ref.dominant_colour(pts, idx, init, (H, W))
        for j in range(k):
            n = int(stats[j, 0])
            assert n == int((a_np == j).sum())
            if n == 0:
                continue
            mean = stats[j, 1:4].astype(np.float64) / n
            var = stats[j, 4:7].astype(np.float64) / n - mean * mean
            assert np.allclose(var, feats[j][1], rtol=1e-9, atol=1e-9)
            assert feats[j][2] == n / float(pts.shape[0])
            assert feats[j][3] == int(stats[j, 7]) / float(n)


def test_c_path_batch_runs_threads():
    imgs = np.stack([bgr2lab_u8(voronoi_image(48, 64, 6, seed=s)) for s in (1, 2, 3)])
    labels, used, pixel_rounds = c_oracle.path_batch(imgs, 12, 15, 4, 3, 2)
    assert used == 2 and pixel_rounds > 0
    for b in range(3):
        want, _ = c_oracle.slic_run(imgs[b], 12, 15, 4)
        assert np.array_equal(labels[b], want)
