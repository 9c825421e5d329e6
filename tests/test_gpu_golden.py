"""GPU parity against the committed golden vectors (tests/golden/, tools/make_golden.py):
the whole caller sequence of optigen_utils.extract_descriptors :41-56 through the drop-in
classes, on inputs derived from the reference's own sample images."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,k", [("road_320", 8), ("bear_320", 8), ("synth_96x64", 4)])
def test_pipeline_reproduces_golden(name, k):
    import torch
    assert torch.cuda.is_available()
    from rag_segmentation import RAGSegmentation
    from mpeg7Descriptors import MPEG7Descriptors
    from imgan import ops
    f = np.load(os.path.join(GOLDEN, name + ".npz"))
    img = f["bgr"]
    K = f["means"].shape[0]
    dev = torch.from_numpy(img).cuda()
    assert np.array_equal(ops.bgr2lab(dev).cpu().numpy(), f["lab"])
    assert np.array_equal(ops.bgr2gray(dev).cpu().numpy(), f["gray"])

    rag = RAGSegmentation(img.copy(), slic_clust_num=K, slic_cw=15, median_blur=7)
    assert np.array_equal(rag.image, f["blurred"])
    assert np.array_equal(rag.image_lab, f["lab_blurred"])
    cl = rag.run_slic()
    assert np.array_equal(cl.astype(np.int32), f["slic_labels"])
    assert np.array_equal(np.array(rag.slic_sp.centers), f["slic_centres"])
    col = rag.slic_mean_lab(cl)
    assert np.array_equal(np.array([col[c] for c in range(K)], dtype=np.int32), f["means"])
    cn = rag.neighbours_regions(cl)
    edges = sorted((u, int(v)) for u in cn for v in cn[u] if u < v)
    assert np.array_equal(np.array(edges, dtype=np.int32), f["edges"])
    new, emst, wmst, vmst = rag.concat_similar_regs(rag.find_edges(cn, col), cl, c_factor=0.65)
    assert np.array_equal(new.astype(np.int32), f["merged"])
    assert np.array_equal(np.array(emst, dtype=np.int32).reshape(-1, 2), f["mst"])
    assert np.array_equal(np.array(wmst), f["mst_w"])

    # SLIC state after intermediate rounds, through the stepwise C-ABI calls
    plan = ops.SlicPlan(1, img.shape[0], img.shape[1], K)
    lab_blur = torch.from_numpy(f["lab_blurred"]).cuda()
    plan.init(lab_blur)
    for r in range(1, 6):
        plan.round(lab_blur, 15)
        if r == 1:
            assert np.array_equal(plan.labels[0].cpu().numpy(), f["slic_labels_round1"])
    assert np.array_equal(plan.labels[0].cpu().numpy(), f["slic_labels_round5"])

    mp = MPEG7Descriptors(new, img)
    init = {int(s): f["dcd_init"][i].tolist() for i, s in enumerate(f["dcd_segments"])}
    dcd = mp.mpeg7_dominant_colours(k, init=init, resample_seed=int(f["dcd_resample_seed"]))
    for i, s in enumerate(f["dcd_segments"]):
        s = int(s)
        assert mp.dcd_rounds[s] == f["dcd_rounds"][i]
        for j in range(k):
            mean, var, pct, coh = dcd[s][j]
            assert np.allclose(mean, f["dcd_mean"][i, j], rtol=1e-4, atol=0)
            assert np.allclose(var, f["dcd_var"][i, j], rtol=1e-4, atol=1e-9, equal_nan=True)
            assert pct == f["dcd_pct"][i, j] and coh == f["dcd_coh"][i, j]
    plan_d = ops.DescriptorPlan(1, img.shape[0], img.shape[1])
    assert np.array_equal(plan_d.colour_layout(dev)[0].cpu().numpy(), f["cld"])
    assert np.array_equal(plan_d.edge_histogram(dev)[0].cpu().numpy(), f["ehd"])
    assert np.array_equal(plan_d.colour_structure(dev)[0].cpu().numpy(), f["csd"])
