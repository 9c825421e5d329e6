"""CPU checks of the oracle itself: known-answer tests that follow from reading the reference
source (last column) and the committed golden vectors."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, voronoi_image
from oracle.color import bgr2lab_u8, bgr2gray_u8, median_blur
from oracle.slic_ref import SLICOracle
from oracle.rag_ref import RAGOracle
from oracle.dcd_ref import lloyd, dominant_colour
from oracle import mpeg7_std as M


def test_lab_known_answers():
    px = np.array([[[0, 0, 0], [255, 255, 255], [128, 128, 128], [7, 7, 7]]], dtype=np.uint8)
    lab = bgr2lab_u8(px)
    assert tuple(lab[0, 0]) == (0, 128, 128) and tuple(lab[0, 1]) == (255, 128, 128)
    assert lab[0, 2, 1] == 128 and lab[0, 2, 2] == 128 and lab[0, 3, 1] == 128     # greys: a=b=128
    assert bgr2gray_u8(px)[0].tolist() == [0, 255, 128, 7]


def test_median_known_answers():
    const = np.full((9, 11, 3), 41, dtype=np.uint8)
    assert np.array_equal(median_blur(const, 7), const)
    img = np.zeros((7, 7, 1), dtype=np.uint8)
    img[:, :4] = 200                                  # 28 of 49 -> median 200 at the centre
    assert median_blur(img, 7)[3, 3, 0] == 200


def test_slic_uniform_image_quirks():
    flat = np.full((60, 90, 3), 120, dtype=np.uint8)
    o = SLICOracle(flat, 12, colour_weight=10, num_iter=3)
    o.init_clusters()
    S = o.step
    assert o.centers[0][3:] == [S - 1, S - 1]         # first scanned neighbour wins (slic.py:73)
    ny = len(range(S, 60 - S // 2, S))
    assert o.centers[1][3:] == [S - 1, 2 * S - 1] if ny > 1 else True     # column-major order
    cl = o.run()
    assert (cl[-1, :] == -1).all() and (cl[:, -1] == -1).all()            # slic.py:94-100
    # flat colour -> pure spatial Voronoi of the first round; centres are floor centroids
    for c, cen in enumerate(o.centers):
        ys, xs = np.nonzero(cl == c)
        if len(xs):
            assert cen[3] == int(xs.sum()) // len(xs) and cen[4] == int(ys.sum()) // len(ys)


def test_slic_runs_num_iter_minus_one_rounds():
    img = voronoi_image(48, 64, 8, seed=2)
    a = SLICOracle(img, 8, num_iter=1)
    a.run()
    assert (a.get_clusters() == -1).all()             # range(1, 1) is empty (slic.py:84)
    b = SLICOracle(img, 8, num_iter=4)
    b.run(keep_history=True)
    assert len(b.history) == 3


def test_slic_modes_agree_on_update_paths():
    img = voronoi_image(70, 90, 10, seed=3)
    a = SLICOracle(img, 20, colour_weight=15, num_iter=5, update="scan")
    b = SLICOracle(img, 20, colour_weight=15, num_iter=5, update="bincount")
    assert np.array_equal(a.run(), b.run()) and a.centers == b.centers


def test_adjacency_and_merge_rules():
    rag = RAGOracle(np.zeros((8, 8, 3), np.uint8), slic_clust_num=4)
    cl = np.array([[0, 1, -1], [2, 3, -1], [-1, -1, -1]], dtype=np.float64)
    ng = rag.neighbours_regions(cl)
    assert ng == {0: {1.0, 2.0, 3.0}, 1: {0.0, 2.0, 3.0}, 2: {0.0, 1.0, 3.0}, 3: {0.0, 1.0, 2.0}}
    col = {0: [0, 0, 0], 1: [0, 0, 1], 2: [0, 0, 2], 3: [90, 0, 0]}
    new, e, w, v = rag.concat_similar_regs(rag.find_edges(ng, col), cl, c_factor=0.0)
    assert new[2, 2] == 0 and new[0, 0] >= 1          # -1 border -> id 0, ids start at 1 (:175-176)


def test_dcd_stop_rule_and_ties():
    pts = np.array([[10, 10, 10]] * 5 + [[200, 50, 50]] * 5, dtype=np.uint8)
    assign, cen, rounds = lloyd(pts, [0, 5])
    assert assign.tolist() == [0] * 5 + [1] * 5 and rounds == 3
    # duplicate initial colours: the lower index keeps every point (strict '>' at :90)
    assign, cen, rounds = lloyd(pts, [0, 1], max_rounds=1)
    assert set(assign.tolist()) == {0}
    assert cen[1].tolist() == [10.0, 10.0, 10.0]      # empty cluster keeps its centre
    # with an injected re-sample draw (:101) the empty cluster moves to that point, takes its
    # five pixels in the next round and the loop ends on the two colours
    assign, cen, rounds = lloyd(pts, [0, 1], resample_idx=[3, 7])
    assert assign.tolist() == [0] * 5 + [1] * 5
    assert cen[0].tolist() == [10.0, 10.0, 10.0] and cen[1].tolist() == [200.0, 50.0, 50.0]


@pytest.mark.parametrize("name", ["road_320", "bear_320", "synth_96x64"])
def test_oracle_reproduces_golden(name):
    f = np.load(os.path.join(GOLDEN, name + ".npz"))
    img = f["bgr"]
    K = f["means"].shape[0]
    assert np.array_equal(bgr2lab_u8(img), f["lab"])
    assert np.array_equal(bgr2gray_u8(img), f["gray"])
    assert np.array_equal(median_blur(img, 7), f["blurred"])
    s = SLICOracle(lab=f["lab_blurred"], num_clusters=K, colour_weight=15, num_iter=10)
    assert np.array_equal(s.run().astype(np.int32), f["slic_labels"])
    assert np.array_equal(np.array(s.centers), f["slic_centres"])
    assert np.array_equal(M.cld_from_sums(*M.cld_block_sums(img)), f["cld"])
    assert np.array_equal(np.concatenate(M.edge_histogram_counts(img)), f["ehd"])
    h, n = M.colour_structure_counts(img)
    assert np.array_equal(h, f["csd"][:-1]) and n == f["csd"][-1]
    # dominant colours of the merged regions: injected initial points and re-sample draws
    from oracle.dcd_ref import DCDOracle, hash_resample_idx
    init = {int(s): f["dcd_init"][i].tolist() for i, s in enumerate(f["dcd_segments"])}
    flat = f["merged"].ravel()
    d = DCDOracle(f["merged"].astype(np.float64), img)
    k = f["dcd_init"].shape[1]
    feats = d.mpeg7_dominant_colours(
        k, lambda seg, n_, kk: init[seg],
        lambda seg, n_, kk: hash_resample_idx(np.nonzero(flat == seg)[0], int(f["dcd_resample_seed"]), kk))
    for i, s in enumerate(f["dcd_segments"]):
        assert d.rounds[int(s)] == f["dcd_rounds"][i]
        for j in range(k):
            assert np.array_equal(feats[int(s)][j][0], f["dcd_mean"][i, j])
            assert feats[int(s)][j][2] == f["dcd_pct"][i, j] and feats[int(s)][j][3] == f["dcd_coh"][i, j]


def test_mpeg7_std_sanity():
    grey = np.full((64, 64, 3), 128, dtype=np.uint8)
    counts, nblk = M.edge_histogram_counts(grey)
    assert counts.sum() == 0 and nblk.sum() > 0               # no edges in a flat image
    hist, nwin = M.colour_structure_counts(grey)
    assert hist.sum() == nwin and (hist > 0).sum() == 1       # one colour, present in every window
    q = M.hmmd_quantise(np.array([[[0, 0, 0], [255, 255, 255], [0, 0, 255]]], np.uint8))
    assert q[0, 0] == 0 and q[0, 1] == 31                     # achromatic sub-space, sum extremes
    step = np.zeros((64, 64, 3), np.uint8)
    step[:, 32:] = 255
    c, n = M.edge_histogram_counts(step)                      # block size 2: no block straddles x=32
    assert c.sum() == 0
    cld = M.colour_layout(grey)
    assert cld.shape == (12,) and (cld[1:6] == 16).all()      # AC terms of a flat image quantise to 16


def test_float32_centre_sums_part_from_integer_sums_beyond_2_24():
    """mpeg7Descriptors.py:96-99 adds a cluster's points up in float32 (np.sum(..., dtype=float32)).
    The product and the C restatement add them as integers, which is the same number while the
    sum stays below 2^24 (a cluster of at most 65,793 pixels can never exceed it: 65,793 x 255 <
    2^24) and a different one beyond that.  This pins where the two part and by how much (the
    deviation row of DESIGN.md section 2); the gap depends on numpy's summation order, so the
    bounds are loose."""
    rng = np.random.RandomState(3)
    exact_n = (1 << 24) // 255                                   # 65,793
    worst = np.full((exact_n, 3), 255, dtype=np.float32)
    assert np.array_equal(np.sum(worst, axis=0, dtype=np.float32), worst.astype(np.int64).sum(axis=0))
    for n in (1000, 20000, exact_n):
        pts = rng.randint(0, 256, (n, 3)).astype(np.float32)
        assert np.array_equal(np.sum(pts, axis=0, dtype=np.float32), pts.astype(np.int64).sum(axis=0)), n
    gaps = {}
    for n in (300000, 2000000):
        pts = rng.randint(128, 256, (n, 3)).astype(np.float32)   # bright cluster: sums well past 2^24
        f32 = np.sum(pts, axis=0, dtype=np.float32).astype(np.float64)
        exact = pts.astype(np.int64).sum(axis=0).astype(np.float64)
        assert (exact > (1 << 24)).all()
        gaps[n] = float(np.abs(f32 - exact).max() / exact.max())
    # the float32 running sum is off, but by little: a relative gap that grows with the cluster
    assert 0.0 < gaps[300000] < 2e-4 and gaps[300000] <= gaps[2000000] < 2e-3, gaps
