"""The caller's sequence, end to end on the drop-in modules: optigen_utils.extract_descriptors
(scripts/optigen_utils.py:41-100, same steps in basegen.py:32-70) replayed on a 320-wide image
(prepare_image size, optigen_utils.py:19-24), including the 106-float record per (segment,
dominant colour) and the 8-direction relation tensor."""
import copy
import json

import numpy as np
import pytest

from conftest import voronoi_image

pytestmark = pytest.mark.gpu

side2id = {'NW': 0, 'N': 1, 'NE': 2, 'E': 3, 'SE': 4, 'S': 5, 'SW': 6, 'W': 7}


def extract_descriptors(n_image, c_param=0.85):
    from rag_segmentation import RAGSegmentation
    from mpeg7Descriptors import MPEG7Descriptors
    from spatialextract import SpatialExtractor
    rag = RAGSegmentation(n_image, slic_clust_num=200, slic_cw=15, median_blur=7)   # :41
    clusters = rag.run_slic()
    clust_col = rag.slic_mean_lab(clusters)
    clust_ng = rag.neighbours_regions(clusters)
    edges = rag.find_edges(clust_ng, clust_col)
    new_clusters = rag.concat_similar_regs(edges, clusters, c_factor=0.65)[0]         # :46
    mpeg = MPEG7Descriptors(new_clusters, n_image, seed=5)
    color_desc = mpeg.mpeg7_dominant_colours(8)
    tex_desc = mpeg.mpeg7_homogeneus_texture()
    shape_desc = mpeg.mpeg7_region_shape()
    spatial_rels = SpatialExtractor(new_clusters).find_spatial(c_param)               # :55-56
    seg_desc = {}
    for seg_idx in set(color_desc.keys()):                                            # :59-85
        record = []
        for tex_val in tex_desc[seg_idx][:-2]:
            record.append(tex_val['e'])
            record.append(tex_val['d'])
        record += tex_desc[seg_idx][-2:]
        for shape_val in shape_desc[seg_idx]:
            record += shape_val
        ex_dataset = []
        for clust_key, clust_val in color_desc[seg_idx].items():
            new_rec = copy.deepcopy(record)
            for cl_elem in clust_val:
                if isinstance(cl_elem, list):
                    new_rec += cl_elem
                else:
                    new_rec.append(cl_elem)
            if np.isnan(new_rec).any():
                continue
            ex_dataset.append(new_rec)
        seg_desc[seg_idx] = ex_dataset
    nseg = len(seg_desc)
    rel_matrix = np.zeros((nseg, nseg, 8))                                             # :88-100
    for ref_seg_id, ref_seg_val in spatial_rels.items():
        for rkey, rval in ref_seg_val.items():
            for p_seg_k, p_seg_val in rval.items():
                rel_matrix[ref_seg_id, p_seg_k, side2id[rkey]] += p_seg_val
    return new_clusters, seg_desc, rel_matrix, (color_desc, tex_desc, shape_desc, spatial_rels)


def test_extract_descriptors_sequence():
    import torch
    assert torch.cuda.is_available()
    img = voronoi_image(213, 320, 9, seed=77)
    # large blobs so that every merged region has more pixels than dominant colours
    new_clusters, seg_desc, rel_matrix, parts = extract_descriptors(img)
    segs = sorted(set(new_clusters.flatten()))
    assert segs[0] == 0.0 and -1 not in segs                # border strip is region 0 (a12)
    assert set(seg_desc) == set(int(s) for s in segs)
    for seg, rows in seg_desc.items():
        assert 1 <= len(rows) <= 8
        for r in rows:
            assert len(r) == 106                            # 60 + 2 + 36 + 8
            assert np.isfinite(r).all()
    assert rel_matrix.shape == (len(segs), len(segs), 8) and (rel_matrix >= 0).all()
    # basegen.py:79-80 dumps the descriptors as JSON
    json.dumps({"color": parts[0], "tex": parts[1], "shape": parts[2], "spatial": parts[3]})
