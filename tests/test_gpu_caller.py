"""The callers' sequence end to end on the drop-in modules (what scripts/optigen_utils.py:41-56 and
basegen.py:32-70 call, in their order) on a 320-wide image (the size prepare_image makes,
optigen_utils.py:19-24).  The test checks what the callers rely on afterwards -- the shapes and key
sets of the four descriptor dictionaries, that every (segment, dominant colour) yields the 106
numbers the SVM record is made of (30 x (e, d) + mean + std, 3 x 12 shape
coefficients, 8 colour values), that the spatial relations cover the 8 directions, and that
everything is JSON-serialisable (basegen.py:79-80) -- without restating the callers' own loops."""
import json

import numpy as np
import pytest

from conftest import voronoi_image

pytestmark = pytest.mark.gpu

DIRECTIONS = {"NW", "N", "NE", "E", "SE", "S", "SW", "W"}


def _flat_len(value):
    """Number of scalars in a nested list / tuple / scalar."""
    if isinstance(value, (list, tuple)):
        return sum(_flat_len(v) for v in value)
    return 1


def test_extract_descriptors_sequence():
    import torch
    assert torch.cuda.is_available()
    from rag_segmentation import RAGSegmentation
    from mpeg7Descriptors import MPEG7Descriptors
    from spatialextract import SpatialExtractor
    img = voronoi_image(213, 320, 9, seed=77)     # large blobs: every region has >= 8 pixels
    rag = RAGSegmentation(img, slic_clust_num=200, slic_cw=15, median_blur=7)
    clusters = rag.run_slic()
    means = rag.slic_mean_lab(clusters)
    edges = rag.find_edges(rag.neighbours_regions(clusters), means)
    merged = rag.concat_similar_regs(edges, clusters, c_factor=0.65)[0]
    mpeg = MPEG7Descriptors(merged, img, seed=5)
    colour = mpeg.mpeg7_dominant_colours(8)
    texture = mpeg.mpeg7_homogeneus_texture()
    shape = mpeg.mpeg7_region_shape()
    spatial = SpatialExtractor(merged).find_spatial(0.85)

    segs = sorted(int(v) for v in np.unique(merged))
    assert segs[0] == 0 and -1 not in segs                  # the border strip is region 0 (a12)
    for d in (colour, texture, shape, spatial):
        assert sorted(int(key) for key in d) == segs
    for seg in segs:
        # 30 Gabor channels {e, d} + mean + std
        assert len(texture[seg]) == 32
        assert all(set(ch) == {"e", "d"} for ch in texture[seg][:30])
        tex_numbers = 2 * 30 + 2
        assert np.array(shape[seg]).shape == (3, 12)
        assert 1 <= len(colour[seg]) <= 8
        # the callers drop a record with a NaN in it (a cluster left without pixels); one stays
        assert any(np.isfinite(np.array(list(e[0]) + list(e[1]) + list(e[2:]), dtype=np.float64)).all()
                   for e in colour[seg].values())
        for entry in colour[seg].values():                  # [mean(3), var(3), share, coherence]
            assert _flat_len(entry) == 8
            assert tex_numbers + 36 + _flat_len(entry) == 106
        assert np.isfinite([[ch["e"], ch["d"]] for ch in texture[seg][:30]]).all()
        assert np.isfinite(texture[seg][30:]).all() and np.isfinite(shape[seg]).all()
        # relations: per direction a dict {other segment: share}, never negative
        assert set(spatial[seg]) <= DIRECTIONS
        for side, others in spatial[seg].items():
            assert seg not in others and set(int(o) for o in others) <= set(segs)
            assert all(v >= 0.0 for v in others.values())
    json.dumps({"color": colour, "tex": texture, "shape": shape, "spatial": spatial})
