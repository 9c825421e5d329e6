"""GPU parity at BASELINE.json's sizes against the CPU oracles, exact (not sampled properties):

  * 1920x1080, K = 400 (configs[1]): Lab, SLIC labels + centres, the dominant colours of every
    superpixel with injected initial points, CLD / EHD / CSD (CSD sub-sampling factor 4);
  * 3840x2160, K = 2000 (configs[2]): median, SLIC, per-label means, adjacency, merged map;
  * 1050x1573, K = 200 (configs[0]'s geometry: bear.jpg at full size, S = 90) on a synthetic image;
  * a 4096x4096 crop for the colour structure with sub-sampling factor 16.

The C restatement (oracle/c, bit identical to the numpy oracle: tests/test_oracle_c.py) makes the
SLIC and Lloyd loops affordable at these sizes; the rest is the numpy oracle itself."""
import numpy as np
import pytest

from conftest import voronoi_image, resample_rule

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def _gpu_slic(torch, lab_np, K, cw=15, num_iter=10):
    from imgan import ops
    H, W = lab_np.shape[:2]
    lab = torch.from_numpy(lab_np).cuda().unsqueeze(0)
    plan = ops.SlicPlan(1, H, W, K)
    labels, centres = plan.run(lab, cw, num_iter)
    return lab, plan, labels, centres


def _check_slic_vs_c(torch, lab_np, K, grid):
    from oracle import c_oracle
    lab, plan, labels, centres = _gpu_slic(torch, lab_np, K)
    assert (plan.S, plan.nx, plan.ny) == grid
    want_labels, want_centres = c_oracle.slic_run(lab_np, K, 15, 10)
    got = labels[0].cpu().numpy()
    # north_star allows 0.1 % of the pixels to differ (float ties); the integer key with the
    # float64 replay of equal keys gives none
    assert np.array_equal(got, want_labels), int((got != want_labels).sum())
    assert np.array_equal(centres[0].cpu().numpy(), want_centres)
    return lab, plan, labels


def test_1080p_k400_whole_path_against_oracles():
    torch = _torch()
    from imgan import ops
    from imgan._lib import lib, check
    from oracle import c_oracle, mpeg7_std as M
    from oracle.color import bgr2lab_u8, bgr2gray_u8
    H, W, K, k = 1080, 1920, 400, 8
    img = voronoi_image(H, W, K, seed=1080)
    dev = torch.from_numpy(img).cuda()
    lab_np = bgr2lab_u8(img)
    assert np.array_equal(ops.bgr2lab(dev).cpu().numpy(), lab_np)
    assert np.array_equal(ops.bgr2gray(dev).cpu().numpy(), bgr2gray_u8(img))
    lab, plan, labels = _check_slic_vs_c(torch, lab_np, K, (72, 26, 14))

    # whole-image descriptors (the colour structure runs with sub-sampling factor 4 here)
    assert M.csd_subsample(H, W) == 4
    dp = ops.DescriptorPlan(1, H, W)
    batch = dev.unsqueeze(0)
    sums, cnt = M.cld_block_sums(img)
    assert np.array_equal(dp.colour_layout(batch)[0].cpu().numpy(), M.cld_from_sums(sums, cnt))
    counts, nblk = M.edge_histogram_counts(img)
    ehd = dp.edge_histogram(batch)[0].cpu().numpy()
    assert np.array_equal(ehd[:80], counts) and np.array_equal(ehd[80:], nblk)
    hist, nwin = M.colour_structure_counts(img)
    csd = dp.colour_structure(batch)[0].cpu().numpy()
    assert np.array_equal(csd[:256], hist) and csd[256] == nwin

    # dominant colours of every superpixel: injected initial points, C oracle per segment
    n = plan.n
    flat = labels[0].cpu().numpy().ravel()
    labf = lab_np.reshape(-1, 3)
    order = np.argsort(flat, kind="stable")                      # raster order inside a label
    bounds = np.searchsorted(flat[order], np.arange(n + 1))
    rng = np.random.RandomState(17)
    cen0 = np.zeros((1, n, k, 3), dtype=np.float32)
    want = {}
    for s in range(n):
        idx = order[bounds[s]:bounds[s + 1]]
        assert len(idx) >= k
        init = np.sort(rng.choice(len(idx), k, replace=False)).astype(np.int32)
        cen0[0, s] = labf[idx[init]]
        rc = np.stack(np.divmod(idx, W), axis=1).astype(np.int32)
        want[s] = (idx,) + c_oracle.dcd_segment(labf[idx], rc, init,
                                                resample_idx=resample_rule(idx, 2468, k))
    dcd = ops.DcdPlan(1, H, W, n, k)
    cen = torch.from_numpy(cen0).cuda()
    dcd.kmeans(lab, labels, cen, max_rounds=1000, resample_seed=2468)
    stats = dcd.statistics(lab, labels).cpu().numpy()[0]
    rounds = dcd.rounds()[0].cpu().numpy()[0]
    assign = dcd.assign[0].cpu().numpy().ravel()
    cen = cen.cpu().numpy()[0]
    for s in range(n):
        idx, w_assign, w_cen, w_stats, w_rounds = want[s]
        assert rounds[s] == w_rounds, (s, rounds[s], w_rounds)
        assert np.array_equal(assign[idx], w_assign.astype(np.uint8)), s
        assert np.array_equal(cen[s].view(np.uint32), w_cen.view(np.uint32)), s   # same float bits
        assert np.array_equal(stats[s].astype(np.uint64), w_stats), s
    assert (assign[flat < 0] == 0xff).all()
    # float features from the integer statistics: mean / variance / share / coherence (1e-4 rel.)
    feats = torch.empty((1, n, k, 8), dtype=torch.float32, device="cuda")
    check(lib.imgan_dcd_features(ops._ptr(dcd.stats), ops._ptr(torch.from_numpy(cen).cuda().unsqueeze(0).contiguous()),
                                 1, n, k, ops._ptr(feats), ops._stream()))
    f = feats.cpu().numpy()[0]
    for s in range(0, n, 7):
        idx, w_assign, w_cen, w_stats, _ = want[s]
        st = w_stats.astype(np.float64)
        for j in range(k):
            if st[j, 0] == 0:
                continue
            # population variance of the cluster's own pixels (mpeg7Descriptors.py:114-116)
            var = st[j, 4:7] / st[j, 0] - (st[j, 1:4] / st[j, 0]) ** 2
            assert np.allclose(f[s, j, 0:3], w_cen[j], rtol=1e-4, atol=0)
            assert np.allclose(f[s, j, 3:6], var, rtol=1e-4, atol=1e-3)
            assert np.isclose(f[s, j, 6], st[j, 0] / len(idx), rtol=1e-6)
            assert np.isclose(f[s, j, 7], st[j, 7] / st[j, 0], rtol=1e-6)


def test_4k_k2000_slic_rag_against_oracles():
    torch = _torch()
    from imgan import ops
    from oracle.color import bgr2lab_u8, median_blur
    from oracle.rag_ref import RAGOracle
    H, W, K = 2160, 3840, 2000
    img = voronoi_image(H, W, K, seed=2160)
    dev = torch.from_numpy(img).cuda()
    blur = ops.median7(dev).cpu().numpy()
    # the 7x7 median on three bands (top and bottom borders, middle); rows 3 from a band's cut
    # edge see pixels outside the band
    for r0, r1 in ((0, 200), (1000, 1200), (H - 200, H)):
        want = median_blur(img[r0:r1], 7)
        a, b = (0 if r0 == 0 else 3), (r1 - r0 if r1 == H else r1 - r0 - 3)
        assert np.array_equal(blur[r0 + a:r0 + b], want[a:b]), (r0, r1)
    lab_np = bgr2lab_u8(blur)
    assert np.array_equal(ops.bgr2lab(torch.from_numpy(blur).cuda()).cpu().numpy(), lab_np)
    lab, plan, labels = _check_slic_vs_c(torch, lab_np, K, (64, 59, 33))

    # rag_segmentation.py:38-184 on the oracle side, from the same blurred image and label map
    ora = RAGOracle.__new__(RAGOracle)
    ora.image, ora.image_lab, ora.image_mean = blur, lab_np.copy(), blur.copy()
    ora.slic_clust_num = K
    clusters = labels[0].cpu().numpy().astype(np.float64)
    col = ora.slic_mean_lab(clusters, nchange=True)
    ng = ora.neighbours_regions(clusters)
    merged_want = ora.concat_similar_regs(ora.find_edges(ng, col), clusters, c_factor=0.65)[0]

    sums, edges = ops.rag_accumulate(lab, labels, K)
    s = sums[0].cpu().numpy()
    means = np.where(s[:, 3:4] > 0, s[:, :3] // np.maximum(s[:, 3:4], 1), 0)
    assert np.array_equal(means, np.stack([col[c] for c in range(K)]).astype(np.int64))
    pairs = set((int(u), int(v)) for u, v in edges[0])
    want_pairs = set((c, int(v)) for c in ng for v in ng[c] if c < v)
    assert pairs == want_pairs
    luts = ops.rag_merge_batch(K, edges, sums, 0.65)
    merged = ops.relabel(labels, torch.from_numpy(luts).cuda())[0].cpu().numpy()
    assert np.array_equal(merged.astype(np.float64), merged_want)
    assert merged.max() > 1

    # the path bench.py times for configs[2]: RagPipeline.step = median + Lab + SLIC + the
    # adjacency-only pass + the per-label sums of the last SLIC round + the merge on the device
    # (rag_merge_kernel: LDS sort of the edge keys, Boruvka forest, leave-one-out cut, replay),
    # at this geometry (n = 1947 labels, about 5.8 k edges) against the same oracle result
    from imgan.pipeline import RagPipeline
    pipe = RagPipeline(1, H, W, K)
    assert pipe.rag is not None, "K = 2000 must take the device merge"
    got, luts_dev = pipe.step(dev.unsqueeze(0))
    assert int(pipe.rag.status.cpu().numpy().max()) == 0 and pipe.redone == 0
    assert np.array_equal(pipe.slic.labels[0].cpu().numpy(), labels[0].cpu().numpy())
    assert np.array_equal(got[0].cpu().numpy().astype(np.float64), merged_want)
    assert np.array_equal(luts_dev[0].cpu().numpy()[:K], luts[0][:K])


def test_bear_geometry_1050x1573_k200():
    """configs[0] as written: the reference's own CPU case runs SLIC on bear.jpg at full size
    (1573 x 1050, K = 200 -> S = 90, 16 x 11 = 176 centres).  The photograph does not travel
    (unknown licence), a synthetic image of that shape does."""
    torch = _torch()
    from oracle.color import bgr2lab_u8
    img = voronoi_image(1050, 1573, 120, seed=1573, noise=10.0)
    _check_slic_vs_c(torch, bgr2lab_u8(img), 200, (90, 16, 11))


def test_colour_structure_subsampling_16():
    torch = _torch()
    from imgan import ops
    from oracle import mpeg7_std as M
    H = W = 4096
    assert M.csd_subsample(H, W) == 16
    img = voronoi_image(H, W, 600, seed=4096, noise=12.0)
    dp = ops.DescriptorPlan(1, H, W)
    csd = dp.colour_structure(torch.from_numpy(img).cuda().unsqueeze(0))[0].cpu().numpy()
    hist, nwin = M.colour_structure_counts(img)
    assert np.array_equal(csd[:256], hist) and csd[256] == nwin
    counts, nblk = M.edge_histogram_counts(img)                 # 122-pixel blocks
    ehd = dp.edge_histogram(torch.from_numpy(img).cuda().unsqueeze(0))[0].cpu().numpy()
    assert np.array_equal(ehd[:80], counts) and np.array_equal(ehd[80:], nblk)


def test_1080p_full_pipeline_hash_init_against_c_oracle():
    """configs[1] as bench.py runs it: FullPipeline.step with the device-side hash rule for the
    initial centres (imgan_dcd_init_hash) and the pixel lists; every superpixel's centres,
    statistics and round count against the C oracle given the restated hash picks."""
    torch = _torch()
    from imgan.pipeline import FullPipeline
    from oracle import c_oracle
    from oracle.color import bgr2lab_u8
    from oracle.dcd_ref import mix32
    H, W, K, k, seed = 1080, 1920, 400, 8, 12345
    img = voronoi_image(H, W, K, seed=1081)
    dev_img = torch.from_numpy(img).cuda().unsqueeze(0)
    # bench.py's form writes no per-pixel assignment map; the same step with the map asked for must
    # give the same rows, and it is that map the per-pixel comparison below reads
    bare = FullPipeline(1, H, W, K, dominant=k, dcd_seed=seed)
    rows0, feats0 = bare.step(dev_img)
    rows0, feats0 = rows0.clone(), feats0.clone()
    del bare
    pipe = FullPipeline(1, H, W, K, dominant=k, dcd_seed=seed, want_assign=True)
    rows, feats = pipe.step(dev_img)
    assert torch.equal(rows, rows0) and torch.equal(feats.view(torch.int32), feats0.view(torch.int32))
    assert pipe.dcd.lists and int(pipe.dcd.n_big.item()) == 0
    lab_np = bgr2lab_u8(img)
    want_labels, _ = c_oracle.slic_run(lab_np, K, 15, 10)
    labels = pipe.slic.labels[0].cpu().numpy()
    assert np.array_equal(labels, want_labels)
    n = pipe.n
    flat = labels.ravel()
    labf = lab_np.reshape(-1, 3)
    order = np.argsort(flat, kind="stable")
    bounds = np.searchsorted(flat[order], np.arange(n + 1))
    # the device rule, restated: slot j of a segment takes the pixel with the smallest
    # (h, p), h = mix32(p * 0x9e3779b1 + seed), among those with h % k == j; an empty slot takes
    # the smallest key of the segment
    p_all = np.arange(H * W, dtype=np.uint64)
    h_all = mix32((p_all * np.uint64(0x9e3779b1) + np.uint64(seed)) & np.uint64(0xffffffff))
    stats = pipe.dcd.stats[0].cpu().numpy()
    cen = pipe.centres[0].cpu().numpy()
    rounds = pipe.dcd.rounds()[0].cpu().numpy()[0]
    f = feats.view(1, n, k, 8)[0].cpu().numpy()
    for s in range(n):
        idx = order[bounds[s]:bounds[s + 1]]
        assert len(idx) >= k
        hs = h_all[idx]
        keys = (hs << np.uint64(32)) | idx.astype(np.uint64)
        init = []
        for j in range(k):
            cand = np.nonzero(hs % np.uint64(k) == j)[0]
            pick = cand[np.argmin(keys[cand])] if len(cand) else int(np.argmin(keys))
            init.append(int(pick))
        rc = np.stack(np.divmod(idx, W), axis=1).astype(np.int32)
        # the injected form takes sorted distinct indices or not: the oracle takes them as given
        w_assign, w_cen, w_stats, w_rounds = c_oracle.dcd_segment(
            labf[idx], rc, np.array(init, np.int32), max_rounds=200,
            resample_idx=resample_rule(idx, seed, k))
        assert rounds[s] == w_rounds, (s, rounds[s], w_rounds)
        assert np.array_equal(cen[s].view(np.uint32), w_cen.view(np.uint32)), s
        assert np.array_equal(stats[s].astype(np.uint64), w_stats), s
        assert np.array_equal(pipe.dcd.assign[0].view(-1)[torch.from_numpy(idx).cuda()].cpu().numpy(),
                              w_assign.astype(np.uint8)), s
        st = w_stats.astype(np.float64)
        live = st[:, 0] > 0
        assert np.allclose(f[s, live, 6], st[live, 0] / len(idx), rtol=1e-6)
        assert np.allclose(f[s, live, 7], st[live, 7] / st[live, 0], rtol=1e-6)


def test_region_descriptors_at_mosaic_geometry():
    """configs[4]'s per-superpixel CLD / EHD / CSD at the mosaic's step: a 1168 x 1168 image with
    K = 256 has S = 73 like the 16384^2 / K = 50,000 mosaic (boxes up to 146^2 cells, 4-pixel edge
    blocks, all colours of a region pooled in one pass of region_desc_kernel), through
    FullPipeline(per_region=True) as bench.py runs it; every region against oracle.mpeg7_std."""
    torch = _torch()
    from imgan.pipeline import FullPipeline
    from oracle import mpeg7_std as M
    H = W = 1168
    K = 256
    img = voronoi_image(H, W, K // 2, seed=1168, noise=8.0)
    pipe = FullPipeline(1, H, W, K, dominant=8, per_region=True)
    assert pipe.slic.S == 73 and pipe.n == 15 * 15
    rows, _ = pipe.step(torch.from_numpy(img).cuda().unsqueeze(0))
    assert int(pipe.region_desc.too_big.item()) == 0
    labels = pipe.slic.labels[0].cpu().numpy()
    cld = pipe.region_desc.cld[0].cpu().numpy()
    ehd = pipe.region_desc.ehd[0].cpu().numpy()
    csd = pipe.region_desc.csd[0].cpu().numpy()
    assert rows.shape == (1, pipe.n * (12 + 96 + 257))
    checked = 0
    for r in range(pipe.n):
        ys, xs = np.nonzero(labels == r)
        if len(ys) == 0:
            assert not cld[r].any() and not ehd[r].any() and not csd[r].any()
            continue
        # the oracle on a crop that holds the whole region (its forms are relative to the box)
        y0, y1, x0, x1 = ys.min(), ys.max() + 1, xs.min(), xs.max() + 1
        ci, cl = img[y0:y1, x0:x1], labels[y0:y1, x0:x1]
        assert np.array_equal(cld[r], M.region_colour_layout_full(ci, cl, r)), r
        counts, nblk = M.region_edge_histogram_counts(ci, cl, r)
        assert np.array_equal(ehd[r], np.concatenate([counts, nblk])), r
        hist, nwin = M.region_colour_structure_counts(ci, cl, r)
        assert np.array_equal(csd[r], np.concatenate([hist, [nwin]])), r
        checked += 1
    assert checked >= pipe.n - 2
    assert ehd[:, 80:].sum() > 0 and csd[:, 256].sum() > 0
    # the step above took the regions' pixels from the dominant colours' pixel lists; the scan of
    # the boxes against the label map (imgan_region_descriptors) gives the same rows
    assert pipe.dcd.lists
    pipe.region_desc.run(torch.from_numpy(img).cuda().unsqueeze(0), pipe.slic.labels, bbox=pipe.dcd.bbox)
    assert np.array_equal(pipe.region_desc.cld[0].cpu().numpy(), cld)
    assert np.array_equal(pipe.region_desc.ehd[0].cpu().numpy(), ehd)
    assert np.array_equal(pipe.region_desc.csd[0].cpu().numpy(), csd)
