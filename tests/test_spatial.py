"""Spatial relations (scripts/spatialextract.py): oracle known answers and the host-side sector
geometry on CPU; the HIP path against the oracle on the GPU."""
import numpy as np
import pytest

from oracle.spatial_ref import SpatialOracle


def _blocks(H=48, W=64):
    cl = np.zeros((H, W), dtype=np.float64)
    cl[16:32, 24:40] = 1          # centre square
    cl[0:10, 24:40] = 2           # north of it
    cl[38:48, 24:40] = 3          # south
    cl[16:32, 0:12] = 4           # west
    cl[16:32, 52:64] = 5          # east
    return cl


def _random_map(H, W, nseg, seed, border=True):
    rng = np.random.RandomState(seed)
    sy, sx = rng.randint(0, H, nseg), rng.randint(0, W, nseg)
    yy, xx = np.mgrid[0:H, 0:W]
    d = (yy[..., None] - sy) ** 2 + (xx[..., None] - sx) ** 2
    cl = d.argmin(axis=2).astype(np.float64)
    if border:                                    # raw SLIC maps keep -1 on the last row / column
        cl[-1, :] = -1
        cl[:, -1] = -1
    return cl


def test_oracle_compass_known_answers():
    """A square with one block on each side: each block shows up in its own sector only
    (background 0 is everywhere); ratios are pixel counts over the block size."""
    rel = SpatialOracle(_blocks()).find_spatial(0.85)[1]
    assert set(rel) == {'NW', 'N', 'NE', 'E', 'SE', 'S', 'SW', 'W'}
    # enlarged box of the 16x16 square: sqrt(1.85)*16 = 21.76 -> 21 wide from (21, 13)
    for name, other in (('N', 2), ('S', 3), ('W', 4), ('E', 5)):
        assert other not in rel[name]             # blocks lie outside the 21x21 enlarged box
    wide = SpatialOracle(_blocks()).find_spatial(8.0)[1]    # 3x box: reaches all four blocks
    assert 2 in wide['N'] and 3 in wide['S'] and 4 in wide['W'] and 5 in wide['E']
    assert 2 not in wide['S'] and 4 not in wide['E'] and 5 not in wide['W']
    for v in wide.values():
        assert 1 not in v                          # the segment itself is never listed (:34)
        assert all(0.0 < r <= 1.0 for r in v.values())


def test_oracle_counts_unlabelled_pixels_like_a_segment():
    cl = _random_map(40, 56, 6, 3)
    rel = SpatialOracle(cl).find_spatial(0.85)
    assert -1 not in rel                           # :17
    # the clamps at :61-62 keep every sector off the last row / column, where SLIC leaves -1
    assert not any(-1 in sec for r in rel.values() for sec in r.values())
    cl[18:22, 26:30] = -1                          # an unlabelled patch inside the image
    rel = SpatialOracle(cl).find_spatial(0.85)
    assert -1 not in rel
    assert any(-1 in sec for r in rel.values() for sec in r.values())   # :33-37 counts it


@pytest.mark.parametrize("H,W,nseg,c,seed", [(40, 56, 6, 0.85, 3), (64, 48, 9, 0.85, 4),
                                              (33, 70, 4, 2.5, 5), (25, 25, 3, 0.1, 6)])
def test_sector_pieces_match_oracle_point_lists(H, W, nseg, c, seed):
    """The product's column rules [lo(x), hi(x)) evaluated on the host give the same counts as
    the oracle's explicit point lists (no GPU: checks the host geometry only)."""
    from spatialextract import sector_pieces, SECTORS
    cl = _random_map(H, W, nseg, seed)
    ora = SpatialOracle(cl)
    want = ora.find_spatial(c)
    for idx in ora.segments:
        box = ora.bounding_rect(cl == idx)
        got = {name: {} for name in SECTORS}
        for sector, xa, xb, lo, hi in sector_pieces(*box, H, W, c):
            def bound(rule, x):
                kind, a, b, k = rule
                if kind == 0:
                    return k
                t = int(np.float64(a) * x + np.float64(b))
                return max(t, k) if kind == 2 else min(t, k) if kind == 3 else t
            for x in range(xa, xb):
                for y in range(bound(lo, x), bound(hi, x)):
                    v = cl[y, x]
                    got[SECTORS[sector]][v] = got[SECTORS[sector]].get(v, 0) + 1
        for name in SECTORS:
            exp = want[int(idx)][name]
            mine = {int(v): n / float((cl == v).sum()) for v, n in got[name].items() if v != idx}
            assert mine == exp, (idx, name)


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,nseg,c,seed,border", [(40, 56, 6, 0.85, 3, True), (96, 128, 14, 0.85, 7, False),
                                                     (213, 320, 25, 0.85, 8, True), (64, 64, 5, 3.0, 9, False)])
def test_gpu_find_spatial_matches_oracle(H, W, nseg, c, seed, border):
    import torch
    assert torch.cuda.is_available()
    from spatialextract import SpatialExtractor
    cl = _random_map(H, W, nseg, seed, border)
    if border:
        cl[H // 2:H // 2 + 3, W // 3:W // 3 + 5] = -1            # unlabelled pixels inside
    got = SpatialExtractor(cl).find_spatial(c)
    want = SpatialOracle(cl).find_spatial(c)
    assert got == want                            # integer counts / integer sizes: exact floats


@pytest.mark.gpu
def test_gpu_spatial_after_rag_merge():
    """The caller's sequence (optigen_utils.py:41-56): merged RAG map -> find_spatial(0.85)."""
    import torch
    assert torch.cuda.is_available()
    from conftest import voronoi_image
    from rag_segmentation import RAGSegmentation
    from spatialextract import SpatialExtractor
    img = voronoi_image(120, 160, 10, seed=21)
    rag = RAGSegmentation(img.copy(), slic_clust_num=60, slic_cw=15)
    cl = rag.run_slic()
    new = rag.concat_similar_regs(rag.find_edges(rag.neighbours_regions(cl), rag.slic_mean_lab(cl)),
                                  cl, c_factor=0.65)[0]
    ext = SpatialExtractor(new)
    got = ext.find_spatial(0.85)
    assert got == SpatialOracle(new).find_spatial(0.85)
    boxes = ext.bounding_boxes()
    for seg in ext.segments:
        assert boxes[seg] == SpatialOracle.bounding_rect(new == seg)
