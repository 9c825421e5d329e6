"""GPU parity: median blur, RAG (means, adjacency, merge, relabel) vs the CPU oracle."""
import numpy as np
import pytest

from conftest import voronoi_image

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


@pytest.mark.parametrize("shape", [(40, 56), (213, 320), (9, 5), (67, 131)])
def test_median7_matches_oracle(shape):
    torch = _torch()
    from imgan import ops
    from oracle.color import median_blur
    rng = np.random.RandomState(11)
    img = rng.randint(0, 256, shape + (3,)).astype(np.uint8)
    got = ops.median7(torch.from_numpy(img).cuda()).cpu().numpy()
    assert np.array_equal(got, median_blur(img, 7))
    const = np.full(shape + (3,), 93, dtype=np.uint8)          # KAT: median of a constant
    assert np.array_equal(ops.median7(torch.from_numpy(const).cuda()).cpu().numpy(), const)


@pytest.mark.parametrize("ksize", [3, 5, 7])
@pytest.mark.parametrize("shape", [(150, 300), (131, 257), (64, 128), (1, 1), (3, 200), (70, 33)])
def test_median_sizes_tiles_and_borders(ksize, shape):
    """Several 128 x 64 tiles, widths with and without the dword-aligned fast path, images
    smaller than the window, a batch, and value patterns with many ties (0 / 255 / plateaus)."""
    torch = _torch()
    from imgan import ops
    from oracle.color import median_blur
    rng = np.random.RandomState(ksize * 100 + shape[1])
    imgs = np.stack([rng.randint(0, 256, shape + (3,)), rng.randint(0, 2, shape + (3,)) * 255,
                     (rng.randint(0, 256, shape + (3,)) // 64) * 64]).astype(np.uint8)
    got = ops.median_blur(torch.from_numpy(imgs).cuda(), ksize).cpu().numpy()
    for b in range(3):
        assert np.array_equal(got[b], median_blur(imgs[b], ksize)), (b, ksize, shape)


def test_median_other_odd_sizes_and_rejections():
    """cv2.medianBlur takes any odd size: 9, 11 and 15 go through the plain per-pixel search (1 is
    the identity for the C entry point); even sizes are refused, and RAGSegmentation refuses what
    cv2 refuses (even, or not greater than 1)."""
    torch = _torch()
    from imgan import ops
    from oracle.color import median_blur
    from rag_segmentation import RAGSegmentation
    rng = np.random.RandomState(11)
    img = np.stack([rng.randint(0, 256, (37, 53, 3)), (rng.randint(0, 256, (37, 53, 3)) // 64) * 64]).astype(np.uint8)
    dev = torch.from_numpy(img).cuda()
    for ksize in (9, 11, 15):
        got = ops.median_blur(dev, ksize).cpu().numpy()
        for b in range(2):
            assert np.array_equal(got[b], median_blur(img[b], ksize)), (b, ksize)
    assert np.array_equal(ops.median_blur(dev, 1).cpu().numpy(), img)
    for ksize in (0, 4, 256, -3):
        with pytest.raises(ValueError):
            ops.median_blur(dev, ksize)
    for ksize in (1, 4):
        with pytest.raises(ValueError):
            RAGSegmentation(img[0], median_blur=ksize)
    rag = RAGSegmentation(img[0].copy(), slic_clust_num=12, median_blur=9)
    assert np.array_equal(rag._dev_image.cpu().numpy(), median_blur(img[0], 9))


@pytest.mark.parametrize("H,W,K,c", [(213, 320, 200, 0.65), (120, 160, 60, 0.65),
                                     (101, 157, 40, 2.0), (96, 128, 300, 0.6)])
def test_rag_pipeline_matches_oracle(H, W, K, c):
    """The caller's sequence (optigen_utils.py:41-47) step by step against the oracle."""
    torch = _torch()
    from rag_segmentation import RAGSegmentation
    from oracle.rag_ref import RAGOracle
    img = voronoi_image(H, W, max(K // 4, 6), seed=77 + H)
    rag = RAGSegmentation(img.copy(), slic_clust_num=K, slic_cw=15, median_blur=7)
    ora = RAGOracle(img.copy(), slic_clust_num=K, slic_cw=15, median_blur_k=7)
    assert np.array_equal(rag.image, ora.image)
    assert np.array_equal(rag.image_lab, ora.image_lab)

    t_clusters = rag.run_slic()
    o_clusters = ora.run_slic()
    assert np.array_equal(t_clusters, o_clusters)

    col = rag.slic_mean_lab(t_clusters)
    ocol = ora.slic_mean_lab(o_clusters)
    assert sorted(col.keys()) == list(range(K))
    for k in range(K):
        assert np.array_equal(np.asarray(col[k], dtype=np.int64), np.asarray(ocol[k], dtype=np.int64))
    assert np.array_equal(rag.image_lab, ora.image_lab)         # painted means (:45-46)

    cn = rag.neighbours_regions(t_clusters)
    ocn = ora.neighbours_regions(o_clusters)
    assert cn == ocn
    for k, s in cn.items():                                      # symmetric, -1 never appears
        assert -1.0 not in s
        for v in s:
            assert float(k) in cn[int(v)]

    ed = rag.find_edges(cn, col)
    oed = ora.find_edges(ocn, ocol)
    assert ed.keys() == oed.keys()
    for k in ed:
        assert [(a, float(b)) for a, b in ed[k]] == [(a, float(b)) for a, b in oed[k]]

    new, emst, wmst, vmst = rag.concat_similar_regs(ed, t_clusters, c_factor=c)
    onew, oemst, owmst, ovmst = ora.concat_similar_regs(oed, o_clusters, c_factor=c)
    assert new.dtype == np.float64
    assert np.array_equal(new, onew)                             # ids included (canonical order)
    assert emst == oemst and vmst == ovmst
    assert np.allclose(wmst, owmst, rtol=0, atol=0)
    assert (new[-1, :] == 0).all() and (new[:, -1] == 0).all()   # -1 border becomes id 0 (:176)

    rgb = rag.slic_mean_rgb(new)
    orgb = ora.slic_mean_rgb(onew)
    for k in range(K):
        assert np.array_equal(np.asarray(rgb[k], dtype=np.int64), np.asarray(orgb[k], dtype=np.int64))
    assert np.array_equal(rag.image_mean, ora.image_mean)


def test_rag_batch_merge_means_mode():
    """Device sums + hash-set adjacency + C++ merge (means mode) + LUT relabel, batch of 2."""
    torch = _torch()
    from imgan import ops
    from oracle.rag_ref import RAGOracle
    K = 80
    imgs = [voronoi_image(150, 200, 30, seed=s) for s in (5, 6)]
    oras = [RAGOracle(im, slic_clust_num=K) for im in imgs]
    labels = np.stack([o.run_slic() for o in oras]).astype(np.int32)
    blur = np.stack([o.image for o in oras])
    lab = np.stack([o.image_lab for o in oras])
    dlab = torch.from_numpy(lab).cuda()
    dlabels = torch.from_numpy(labels).cuda()
    sums, edges = ops.rag_accumulate(dlab, dlabels, K)
    s = sums.cpu().numpy()
    luts = []
    for b, o in enumerate(oras):
        col = o.slic_mean_lab(labels[b].astype(np.float64), nchange=True)
        means = np.where(s[b, :, 3:4] > 0, s[b, :, :3] // np.maximum(s[b, :, 3:4], 1), 0)
        assert np.array_equal(means, np.array([col[c] for c in range(K)], dtype=np.int64))
        ng = o.neighbours_regions(labels[b].astype(np.float64))
        ref_edges = sorted((u, int(v)) for u in ng for v in ng[u] if u < v)
        assert [tuple(e) for e in edges[b].tolist()] == ref_edges
        res = ops.rag_merge_host(K, edges[b], 0.65, means=means)
        onew, oe, ow, ov = o.concat_similar_regs(o.find_edges(ng, col), labels[b].astype(np.float64), 0.65)
        assert res["mst"].tolist() == oe and res["sets"] == ov
        luts.append(res["lut"])
        out = ops.relabel(dlabels[b:b + 1], torch.from_numpy(res["lut"]).cuda())[0].cpu().numpy()
        assert np.array_equal(out.astype(np.float64), onew)
    both = ops.relabel(dlabels, torch.from_numpy(np.stack(luts)).cuda()).cpu().numpy()
    assert both.shape == labels.shape


def test_adjacency_checkerboard_kat():
    """2x2 checkerboard of labels: diagonal neighbours count (8-connectivity)."""
    torch = _torch()
    from imgan import ops
    lab = np.array([[0, 1, -1], [2, 3, -1], [-1, -1, -1]], dtype=np.int32)
    img = torch.zeros((1, 3, 3, 3), dtype=torch.uint8, device="cuda")
    _, edges = ops.rag_accumulate(img, torch.from_numpy(lab).cuda().unsqueeze(0), 4)
    assert [tuple(e) for e in edges[0].tolist()] == [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]


def test_rag_pipeline_batch_class_matches_oracle():
    """imgan.pipeline.RagPipeline (median -> Lab -> SLIC -> sums/adjacency -> merge -> relabel on
    a resident batch) against the oracle's per-image sequence."""
    torch = _torch()
    from imgan.pipeline import RagPipeline
    from oracle.rag_ref import RAGOracle
    K, H, W = 60, 120, 160
    imgs = [voronoi_image(H, W, 8, seed=s) for s in (31, 32, 33)]
    pipe = RagPipeline(len(imgs), H, W, K, colour_weight=15, num_iter=10, c_factor=0.65)
    merged, luts = pipe.step(torch.from_numpy(np.stack(imgs)).cuda(), timed=True)
    merged = merged.cpu().numpy()
    for b, im in enumerate(imgs):
        o = RAGOracle(im.copy(), slic_clust_num=K, slic_cw=15)
        cl = o.run_slic()
        want = o.concat_similar_regs(o.find_edges(o.neighbours_regions(cl), o.slic_mean_lab(cl)),
                                     cl, 0.65)[0]
        assert np.array_equal(merged[b].astype(np.float64), want), b
    assert set(pipe.stage_times_ms()) >= {"median", "bgr2lab", "slic_assign", "rag_accumulate+merge", "relabel"}
    assert pipe.redone == 0 and int(pipe.rag.status.abs().sum()) == 0      # merged on the device


def _device_merge(torch, K, means, edge_lists, c, cap=None):
    """imgan_rag_merge_device on hand-made sums / edge hash sets (the layout imgan_rag_accumulate
    leaves): returns (luts, status)."""
    from imgan import ops
    from imgan._lib import lib, check
    B = len(edge_lists)
    plan = ops.RagPlan(B, K, edge_cap=cap)
    sums = np.zeros((B, K, 4), dtype=np.int64)
    sums[:, :, :3] = means * 7                      # floor(7 m / 7) = m
    sums[:, :, 3] = 7
    table = np.zeros((B, plan.cap), dtype=np.uint64)
    rng = np.random.RandomState(1)
    for b, uv in enumerate(edge_lists):
        keys = ((uv[:, 0].astype(np.uint64) + 1) << np.uint64(32)) | (uv[:, 1].astype(np.uint64) + 1)
        slots = rng.choice(plan.cap, len(keys), replace=False)      # any slot: the merge only scans
        table[b, slots] = keys
    plan.sums.copy_(torch.from_numpy(sums))
    plan.table.copy_(torch.from_numpy(table.view(np.int64)))
    plan.overflow.zero_()
    check(lib.imgan_rag_merge_device(ops._ptr(plan.sums), ops._ptr(plan.table), plan.cap,
                                     ops._ptr(plan.overflow), B, K, float(c), ops._ptr(plan.lut),
                                     ops._ptr(plan.status), ops._stream()))
    return plan.lut.cpu().numpy(), plan.status.cpu().numpy()


@pytest.mark.parametrize("K,levels,c", [(50, 256, 0.65), (300, 8, 0.65), (2000, 256, 0.65), (4096, 32, 2.0),
                                        (700, 3, 0.1), (5, 256, 0.65), (1, 256, 0.65)])
def test_device_merge_equals_host_merge(K, levels, c):
    """Random planar-ish graphs (grid neighbours + diagonals, some vertices isolated), colours
    quantised to `levels` values so that equal weights are common: relabel tables of the device
    merge against imgan_rag_merge_host (itself checked against the oracle and the golden files)."""
    torch = _torch()
    from imgan import ops
    rng = np.random.RandomState(K + levels)
    B = 3
    side = max(1, int(np.sqrt(K)))
    means = (rng.randint(0, levels, (B, K, 3)) * (256 // levels)).astype(np.int64)
    edge_lists = []
    for b in range(B):
        uv = set()
        for v in range(K):
            if rng.rand() < 0.05:
                continue                                     # an isolated vertex (empty label id)
            for d in (1, side, side + 1, side - 1):
                if v + d < K and d > 0 and rng.rand() < 0.8:
                    uv.add((v, v + d))
        edge_lists.append(np.array(sorted(uv), dtype=np.int64).reshape(-1, 2))
    luts, status = _device_merge(torch, K, means, edge_lists, c)
    assert (status == 0).all()
    for b in range(B):
        want = ops.rag_merge_host(K, edge_lists[b], c, means=means[b])["lut"]
        assert np.array_equal(luts[b], want), b


def test_device_merge_hands_over_what_it_cannot_hold():
    """More than 16,384 edges, or an edge set that overflowed: status says so, and
    RagPlan.finish() redoes the image through the host merge."""
    torch = _torch()
    from imgan import ops
    K = 4096
    rng = np.random.RandomState(3)
    means = rng.randint(0, 256, (1, K, 3)).astype(np.int64)
    u = rng.randint(0, K - 1, 40000)
    v = np.minimum(u + 1 + rng.randint(0, 50, 40000), K - 1)
    uv = np.unique(np.stack([u, v], 1)[u < v], axis=0)
    assert len(uv) > 16384
    luts, status = _device_merge(torch, K, means, [uv], 0.65, cap=1 << 17)
    assert status[0] == 1


def test_rag_pipeline_small_step_more_labels_than_requested():
    """39 x 39 with K = 96 gives S = 3 and n = 144 labels > K (the reference dies with a KeyError
    in find_edges there): the batch path sizes its tables for n and merges all of them."""
    torch = _torch()
    from imgan.pipeline import RagPipeline
    img = voronoi_image(39, 39, 6, seed=9)
    pipe = RagPipeline(1, 39, 39, 96)
    assert pipe.slic.n == 144 and pipe.K == 144
    merged, luts = pipe.step(torch.from_numpy(img[None]).cuda())
    assert int(merged.max()) >= 1 and pipe.redone == 0


def test_slic_sums_and_adjacency_only_pass_equal_the_full_pass():
    """The per-label sums the last SLIC round leaves (imgan_slic_label_sums) are the sums of
    imgan_rag_accumulate over the same image and labels (rag_segmentation.py:38-48 on the image of
    run_slic :21-25), and the pass without image and sums fills the same edge set."""
    torch = _torch()
    from imgan import ops
    from oracle.color import bgr2lab_u8
    B, H, W, K = 2, 180, 244, 70
    lab = torch.from_numpy(np.stack([bgr2lab_u8(voronoi_image(H, W, 25, seed=s, noise=6.0))
                                     for s in (3, 4)])).cuda()
    plan = ops.SlicPlan(B, H, W, K)
    labels, _ = plan.run(lab, 15, 10)
    Kt = max(K, plan.n) + 5                                   # rows beyond n must come back zero
    want_sums, want_edges = ops.rag_accumulate(lab, labels, Kt)
    got = torch.full((B, Kt, 4), -1, dtype=torch.int64, device="cuda")
    plan.label_sums(got)
    assert torch.equal(got, want_sums)
    assert int(got[:, plan.n:].abs().sum()) == 0
    rp_full, rp_adj = ops.RagPlan(B, Kt), ops.RagPlan(B, Kt)
    lut_full, st_full = rp_full.run(lab, labels, 0.65)
    lut_adj, st_adj = rp_adj.run(lab, labels, 0.65, slic=plan)
    assert torch.equal(rp_full.sums, rp_adj.sums)
    assert torch.equal(torch.sort(rp_full.table, dim=1).values, torch.sort(rp_adj.table, dim=1).values)
    assert torch.equal(lut_full, lut_adj) and int(st_full.abs().sum()) == 0 == int(st_adj.abs().sum())
    for b in range(B):
        keys = rp_adj.table[b][rp_adj.table[b] != 0].cpu().numpy().astype(np.uint64)
        pairs = sorted((int(k >> np.uint64(32)) - 1, int(k & np.uint64(0xffffffff)) - 1) for k in keys)
        assert pairs == [tuple(e) for e in want_edges[b].tolist()]
