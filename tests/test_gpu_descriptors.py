"""GPU parity: connected components, dominant colour, CLD / EHD / CSD vs the CPU oracle."""
import numpy as np
import pytest

from conftest import voronoi_image, resample_rule, mix32 as _mix32

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


@pytest.mark.parametrize("shape,nval", [((37, 53), 3), ((128, 200), 5), ((1, 9), 2), ((300, 300), 2)])
def test_cc_label_matches_scipy(shape, nval):
    torch = _torch()
    from scipy import ndimage
    from imgan import ops
    rng = np.random.RandomState(3)
    vals = rng.randint(-1, nval, shape).astype(np.int32)
    roots = ops.cc_label(torch.from_numpy(vals).cuda())[0].cpu().numpy()
    assert ((roots < 0) == (vals < 0)).all()
    flat = np.arange(vals.size).reshape(shape)
    for v in range(nval):
        lab, n = ndimage.label(vals == v, structure=np.ones((3, 3)))
        if n == 0:
            continue
        mins = ndimage.minimum(flat, lab, index=np.arange(1, n + 1))   # smallest raster index
        want = np.asarray(mins)[lab[vals == v] - 1]
        assert np.array_equal(roots[vals == v], want)


def test_cc_spiral_long_chain():
    """One long snake component: stresses the union-find path lengths."""
    torch = _torch()
    from imgan import ops
    H, W = 65, 129
    vals = np.full((H, W), -1, dtype=np.int32)
    for y in range(0, H, 2):
        vals[y, :] = 0
        if y + 1 < H:
            vals[y + 1, W - 1 if (y // 2) % 2 == 0 else 0] = 0
    roots = ops.cc_label(torch.from_numpy(vals).cuda())[0].cpu().numpy()
    assert (roots[vals == 0] == 0).all()


def _merged_clusters(img, K):
    from oracle.rag_ref import RAGOracle
    rag = RAGOracle(img, slic_clust_num=K)
    cl = rag.run_slic()
    col = rag.slic_mean_lab(cl, nchange=True)
    ng = rag.neighbours_regions(cl)
    return rag.concat_similar_regs(rag.find_edges(ng, col), cl, c_factor=0.65)[0]


def _resample_fn(clusters, mp):
    """resample_fn for DCDOracle: the hash rule of the product with the seed the class drew."""
    flat = np.asarray(clusters).ravel()
    return lambda seg, n, kk: resample_rule(np.nonzero(flat == seg)[0], mp.resample_seed, kk)


@pytest.mark.parametrize("H,W,K,k", [(120, 160, 60, 8), (213, 320, 200, 8), (90, 110, 30, 3)])
def test_dominant_colour_matches_oracle(H, W, K, k):
    torch = _torch()
    from mpeg7Descriptors import MPEG7Descriptors
    from oracle.dcd_ref import DCDOracle
    img = voronoi_image(H, W, 12, seed=21 + H, noise=14.0)
    clusters = _merged_clusters(img, K)
    rng = np.random.RandomState(9)
    init = {}
    for seg in np.unique(clusters):
        n = int((clusters == seg).sum())
        if n < k:
            pytest.skip("segment smaller than k")
        init[int(seg)] = sorted(rng.choice(n, k, replace=False).tolist())
    mp = MPEG7Descriptors(clusters, img)
    got = mp.mpeg7_dominant_colours(k, init=init)
    ora = DCDOracle(clusters, img)
    want = ora.mpeg7_dominant_colours(k, lambda seg, n, kk: init[seg], _resample_fn(clusters, mp))
    assert got.keys() == want.keys()
    for seg in want:
        assert mp.dcd_rounds[seg] == ora.rounds[seg], seg            # same stop round
        for j in range(k):
            gm, gv, gp, gc = got[seg][j]
            wm, wv, wp, wc = want[seg][j]
            # float descriptors: 1e-4 relative (north_star); integers behind them are exact
            assert np.allclose(gm, wm, rtol=1e-4, atol=0), (seg, j, gm, wm)
            assert gp == wp and gc == wc, (seg, j)
            if np.isnan(wv).any():
                assert np.isnan(gv).all()
            else:
                assert np.allclose(gv, wv, rtol=1e-4, atol=1e-9), (seg, j, gv, wv)


@pytest.mark.parametrize("levels", [64, 128])
def test_dominant_colour_exact_ties(levels):
    """Few distinct colours per segment: initial centres coincide (equal distances for every
    pixel, the lower cluster index must win and the other stays empty) and pixels sit at equal
    distance from two centres, so the near-tie replay of the fast loop decides most pixels."""
    torch = _torch()
    from mpeg7Descriptors import MPEG7Descriptors
    from oracle.dcd_ref import DCDOracle
    H, W, k = 96, 160, 6
    img = (voronoi_image(H, W, 40, seed=levels, noise=0.0) // levels * levels).astype(np.uint8)
    clusters = (np.arange(W)[None, :] // 40 + 0 * np.arange(H)[:, None]).astype(np.float64)
    rng = np.random.RandomState(levels)
    init = {seg: sorted(rng.choice(H * 40, k, replace=False).tolist()) for seg in range(4)}
    mp = MPEG7Descriptors(clusters, img, seed=levels)
    got = mp.mpeg7_dominant_colours(k, init=init)
    ora = DCDOracle(clusters, img)
    want = ora.mpeg7_dominant_colours(k, lambda seg, n, kk: init[seg], _resample_fn(clusters, mp))
    empties = 0
    for seg in want:
        assert mp.dcd_rounds[seg] == ora.rounds[seg], seg
        for j in range(k):
            gm, gv, gp, gc = got[seg][j]
            wm, wv, wp, wc = want[seg][j]
            assert np.allclose(gm, wm, rtol=1e-6, atol=0), (seg, j, gm, wm)
            assert gp == wp and gc == wc, (seg, j, gp, wp, gc, wc)
            if np.isnan(wv).any():
                empties += 1
                assert np.isnan(gv).all()
            else:
                assert np.allclose(gv, wv, rtol=1e-6, atol=1e-9), (seg, j, gv, wv)
    assert empties > 0                                   # coinciding centres did occur


def test_dominant_colour_single_segment_api():
    torch = _torch()
    from mpeg7Descriptors import MPEG7Descriptors
    from oracle.dcd_ref import dominant_colour
    from oracle.color import bgr2lab_u8
    img = voronoi_image(60, 80, 5, seed=4, noise=20.0)
    clusters = np.zeros((60, 80))
    clusters[:, 40:] = 1
    mp = MPEG7Descriptors(clusters, img, seed=123)
    idx = (clusters == 1)
    seg_idxs = np.transpose(np.nonzero(idx))
    pts = bgr2lab_u8(img)[idx]
    init = [5, 100, 700, 1500]
    got = mp.dominant_colour(4, pts, seg_idxs, init=init)
    res = resample_rule(np.nonzero(idx.ravel())[0], mp.resample_seed, 4)
    want, _ = dominant_colour(pts, seg_idxs, init, clusters.shape, res)
    for j in range(4):
        assert np.allclose(got[j][0], want[j][0], rtol=1e-4)
        assert got[j][2] == want[j][2] and got[j][3] == want[j][3]
    with pytest.raises(ValueError):                                  # N < k (:79)
        mp.dominant_colour(8, pts[:3], seg_idxs[:3])


SHAPES = [(213, 320), (64, 96), (101, 157), (480, 640), (17, 23), (7, 5)]


@pytest.mark.parametrize("shape", SHAPES)
def test_cld_ehd_csd_match_oracle(shape):
    torch = _torch()
    from imgan import ops
    from oracle import mpeg7_std as M
    H, W = shape
    imgs = np.stack([voronoi_image(H, W, 10, seed=s, noise=10.0) for s in (1, 2, 3)])
    dev = torch.from_numpy(imgs).cuda()
    plan = ops.DescriptorPlan(3, H, W)
    cld = plan.colour_layout(dev).cpu().numpy()
    ehd = plan.edge_histogram(dev).cpu().numpy()
    csd = plan.colour_structure(dev).cpu().numpy()
    for b in range(3):
        sums, cnt = M.cld_block_sums(imgs[b])
        assert np.array_equal(cld[b], M.cld_from_sums(sums, cnt))
        counts, nblk = M.edge_histogram_counts(imgs[b])
        assert np.array_equal(ehd[b, :80], counts) and np.array_equal(ehd[b, 80:], nblk)
        hist, nwin = M.colour_structure_counts(imgs[b])
        assert np.array_equal(csd[b, :256], hist) and csd[b, 256] == nwin
    # the batch pipeline's form of the first two: one pass over the image for both (imgan_cld_ehd)
    plan2 = ops.DescriptorPlan(3, H, W)
    cld2, ehd2 = plan2.colour_layout_and_edge_histogram(dev)
    assert np.array_equal(cld2.cpu().numpy(), cld) and np.array_equal(ehd2.cpu().numpy(), ehd)


@pytest.mark.parametrize("shape", [(1080, 1920), (1050, 1573), (2160, 3840), (33, 4000), (600, 9)])
def test_cld_ehd_one_pass_equals_the_two_calls(shape):
    """imgan_cld_ehd against imgan_cld + imgan_ehd at the bench's sizes (edge blocks of 42 / 36 / 86
    pixels, strips that straddle a row of the colour grid, rows below the last edge block), an
    unaligned width, and two shapes so flat / so narrow that the call falls back to the two passes."""
    torch = _torch()
    from imgan import ops
    H, W = shape
    rng = np.random.RandomState(H + W)
    img = torch.from_numpy(rng.randint(0, 256, (2, H, W, 3), dtype=np.uint8)).cuda()
    plan = ops.DescriptorPlan(2, H, W)
    cld = plan.colour_layout(img).clone()
    ehd = plan.edge_histogram(img).clone()
    plan2 = ops.DescriptorPlan(2, H, W)
    cld2, ehd2 = plan2.colour_layout_and_edge_histogram(img)
    assert torch.equal(cld2, cld) and torch.equal(ehd2, ehd)
    assert int(ehd[:, 80:].sum()) > 0 or min(H, W) < 12


@pytest.mark.parametrize("nbins", [32, 64, 128, 256])
def test_hmmd_quantiser_exhaustive_slice(nbins):
    torch = _torch()
    from imgan import ops
    from oracle.mpeg7_std import hmmd_quantise
    # every (g, r) for 32 blue levels incl. extremes: 2^21 colours
    b = np.array(sorted(set(list(range(0, 256, 9)) + [1, 2, 127, 128, 254, 255])), dtype=np.uint8)
    g, r = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8))
    img = np.stack([np.broadcast_to(b[:, None, None], (len(b), 256, 256)),
                    np.broadcast_to(g, (len(b), 256, 256)),
                    np.broadcast_to(r, (len(b), 256, 256))], axis=-1).copy()
    got = ops.hmmd_quantise(torch.from_numpy(img).cuda(), nbins).cpu().numpy()
    want = hmmd_quantise(img, nbins)
    assert np.array_equal(got, want)
    assert want.max() == nbins - 1 and want.min() == 0


def test_descriptor_methods_on_class():
    torch = _torch()
    from mpeg7Descriptors import MPEG7Descriptors
    from oracle import mpeg7_std as M
    img = voronoi_image(213, 320, 10, seed=8)
    mp = MPEG7Descriptors(np.zeros((213, 320)), img)
    assert mp.mpeg7_colour_layout() == M.colour_layout(img).tolist()
    c, n = M.edge_histogram_counts(img)
    assert mp.mpeg7_edge_histogram() == M.ehd_quantise(c, n).tolist()
    h, nw = M.colour_structure_counts(img)
    assert mp.mpeg7_colour_structure() == M.csd_quantise(h, nw).tolist()
    shape = mp.mpeg7_region_shape()                 # one segment (label 0) covering the frame
    assert list(shape) == [0] and np.array(shape[0]).shape == (3, 12)


def test_full_pipeline_batch_matches_oracle():
    """configs[1] in miniature: Lab -> SLIC -> per-superpixel dominant colours (device-side
    hash init) + CLD/EHD/CSD for a batch of 2, each stage against the oracle."""
    torch = _torch()
    from imgan.pipeline import FullPipeline
    from oracle.color import bgr2lab_u8
    from oracle.slic_ref import SLICOracle
    from oracle.dcd_ref import dominant_colour
    from oracle import mpeg7_std as M
    H, W, K, k, seed = 120, 160, 30, 4, 777
    imgs = np.stack([voronoi_image(H, W, 10, seed=s, noise=12.0) for s in (31, 32)])
    pipe = FullPipeline(2, H, W, K, colour_weight=15, num_iter=10, dominant=k, dcd_seed=seed,
                        dcd_check_every=1)
    rows, feats = pipe.step(torch.from_numpy(imgs).cuda())
    rows = rows.cpu().numpy()
    feats = feats.cpu().numpy().reshape(2, pipe.n, k, 8)
    labels = pipe.slic.labels.cpu().numpy()
    for b in range(2):
        lab = bgr2lab_u8(imgs[b])
        o = SLICOracle(lab=lab, num_clusters=K, colour_weight=15, num_iter=10)
        assert np.array_equal(o.run().astype(np.int32), labels[b])
        cld = M.colour_layout(imgs[b])
        counts, nblk = M.edge_histogram_counts(imgs[b])
        hist, nwin = M.colour_structure_counts(imgs[b])
        want = np.concatenate([cld, counts, nblk, hist, [nwin]])
        assert np.array_equal(rows[b], want)
        # device init rule restated: slot j of a segment <- smallest (mix32(p*c + seed), p), h % k == j
        p = np.arange(H * W, dtype=np.uint64)
        h = _mix32((p * np.uint64(0x9e3779b1) + np.uint64(seed)) & np.uint64(0xffffffff))
        flat = labels[b].ravel()
        for s in range(pipe.n):
            idx = np.nonzero(flat == s)[0]
            if len(idx) == 0:
                continue
            hs = h[idx]
            init = []
            for j in range(k):
                cand = np.nonzero(hs % np.uint64(k) == j)[0]
                assert len(cand), "test segments are large enough for every slot"
                best = cand[np.lexsort((idx[cand], hs[cand]))[0]]
                init.append(int(best))
            pts = lab.reshape(-1, 3)[idx]
            seg_idxs = np.stack(np.divmod(idx, W), axis=1)
            want_f, _ = dominant_colour(pts, seg_idxs, init, (H, W), resample_rule(idx, seed, k))
            for j in range(k):
                wm, wv, wp, wc = want_f[j]
                g = feats[b, s, j]
                assert np.allclose(g[0:3], wm, rtol=1e-4, atol=0), (b, s, j)
                if np.isnan(wv).any():
                    assert np.isnan(g[3:6]).all()
                else:
                    assert np.allclose(g[3:6], wv, rtol=1e-4, atol=1e-6), (b, s, j)
                assert np.isclose(g[6], wp, rtol=1e-6) and np.isclose(g[7], wc, rtol=1e-6)


@pytest.mark.parametrize("k", [8, 12])
def test_dcd_lds_path_equals_streaming_path(k):
    """The LDS-resident segment kernel and the streaming rounds are two schedules of the same
    arithmetic: centres, assignment, statistics and round counts must be identical, also when
    some segments are too large for LDS and take the streaming rounds."""
    torch = _torch()
    from imgan import ops
    from oracle.color import bgr2lab_u8
    H, W = 200, 260
    img = voronoi_image(H, W, 14, seed=61, noise=15.0)
    lab = torch.from_numpy(bgr2lab_u8(img)).cuda().unsqueeze(0)
    # 30 small square segments + one large segment (> LDS capacity) + unlabelled border
    labels = np.full((H, W), -1, dtype=np.int32)
    labels[0:150, 0:240] = 30
    sid = 0
    for gy in range(5):
        for gx in range(6):
            labels[150 + gy * 9:150 + gy * 9 + 9, gx * 40:gx * 40 + 37] = sid
            sid += 1
    dl = torch.from_numpy(labels).cuda().unsqueeze(0)
    rng = np.random.RandomState(4)
    cen0 = np.zeros((1, 31, k, 3), dtype=np.float32)
    labf = bgr2lab_u8(img).reshape(-1, 3)
    for s in range(31):
        idx = np.nonzero(labels.ravel() == s)[0]
        cen0[0, s] = labf[rng.choice(idx, k, replace=False)]
    out = []
    for streaming in (False, True):
        plan = ops.DcdPlan(1, H, W, 31, k)
        cen = torch.from_numpy(cen0.copy()).cuda()
        plan.kmeans(lab, dl, cen, streaming_only=streaming)
        stats = plan.statistics(lab, dl, streaming_only=streaming)
        rounds, done = plan.rounds()
        out.append((cen.cpu().numpy(), plan.assign.cpu().numpy().copy(), stats.cpu().numpy().copy(),
                    rounds.cpu().numpy(), plan.big_segments))
    assert out[0][4] == 1 and out[1][4] == 31            # one segment exceeded the LDS capacity
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][1], out[1][1])
    assert np.array_equal(out[0][2], out[1][2])
    assert np.array_equal(out[0][3], out[1][3])


def _region_oracle_rows(img, labels, r):
    from oracle import mpeg7_std as M
    cld = M.region_colour_layout_full(img, labels, r)
    counts, nblk = M.region_edge_histogram_counts(img, labels, r)
    hist, nwin = M.region_colour_structure_counts(img, labels, r)
    return cld, np.concatenate([counts, nblk]), np.concatenate([hist, [nwin]])


def test_region_descriptors_match_oracle_on_superpixels():
    """Per-region colour layout / edge histogram / colour structure of every SLIC superpixel of two
    images (boxes handed over by the last SLIC round) against oracle.mpeg7_std's per-region forms."""
    torch = _torch()
    from imgan.pipeline import FullPipeline
    H, W, K = 150, 230, 24
    imgs = np.stack([voronoi_image(H, W, 9, seed=s, noise=12.0) for s in (71, 72)])
    pipe = FullPipeline(2, H, W, K, dominant=4, per_region=True)
    rows, _ = pipe.step(torch.from_numpy(imgs).cuda())
    labels = pipe.slic.labels.cpu().numpy()
    cld = pipe.region_desc.cld.cpu().numpy()
    ehd = pipe.region_desc.ehd.cpu().numpy()
    csd = pipe.region_desc.csd.cpu().numpy()
    assert int(pipe.region_desc.too_big.item()) == 0
    assert rows.shape == (2, pipe.n * (12 + 96 + 257))
    checked = 0
    for b in range(2):
        for r in range(pipe.n):
            if not (labels[b] == r).any():
                assert not cld[b, r].any() and not ehd[b, r].any() and not csd[b, r].any()
                continue
            wc, we, ws = _region_oracle_rows(imgs[b], labels[b], r)
            assert np.array_equal(cld[b, r], wc), (b, r)
            assert np.array_equal(ehd[b, r], we), (b, r)
            assert np.array_equal(csd[b, r], ws), (b, r)
            checked += 1
    assert checked >= 2 * pipe.n - 4


def test_region_descriptors_odd_shapes_and_limits():
    """Hand-made label maps: a ring (empty colour-layout blocks inside its box), thin and tiny
    regions (no window fits), a region reaching the image corner, an id without pixels, and a
    region too large for the chip (counted, rows zero)."""
    torch = _torch()
    from imgan import ops
    H, W = 200, 260
    img = voronoi_image(H, W, 30, seed=5, noise=20.0)
    labels = np.full((H, W), -1, dtype=np.int32)
    labels[:, :] = 6                                            # background region: box = whole frame
    yy, xx = np.mgrid[0:H, 0:W]
    d = np.hypot(yy - 60, xx - 70)
    labels[(d < 40) & (d >= 25)] = 0                            # ring
    labels[120:123, 10:150] = 1                                 # thin strip
    labels[150:155, 200:205] = 2                                # 5 x 5
    labels[H - 40:, W - 50:] = 3                                # touches the corner
    labels[10:12, 200:202] = 5                                  # 2 x 2 (id 4 has no pixels)
    nseg = 7
    plan = ops.RegionDescriptorPlan(1, H, W, nseg)
    dev = torch.from_numpy(img).cuda().unsqueeze(0)
    cld, ehd, csd = plan.run(dev, torch.from_numpy(labels).cuda().unsqueeze(0))
    cld, ehd, csd = cld.cpu().numpy()[0], ehd.cpu().numpy()[0], csd.cpu().numpy()[0]
    assert int(plan.too_big.item()) == 1                        # region 6: 52,000 cells
    assert not cld[6].any() and not csd[6].any() and not cld[4].any()
    for r in (0, 1, 2, 3, 5):
        wc, we, ws = _region_oracle_rows(img, labels, r)
        assert np.array_equal(cld[r], wc), r
        assert np.array_equal(ehd[r], we), r
        assert np.array_equal(csd[r], ws), r
    assert csd[0, 256] > 0 and csd[1, 256] == 0 and ehd[3, 80:].sum() > 0
    # the same rows from the pixel lists of a listed dominant-colour pass over the same map
    # (imgan_region_descriptors_listed): bit for bit, the oversized region counted again
    dcd = ops.DcdPlan(1, H, W, nseg, 2)
    assert dcd.lists
    ldev = torch.from_numpy(labels).cuda().unsqueeze(0)
    cen = torch.empty((1, nseg, 2, 3), dtype=torch.float32, device="cuda")
    dcd.kmeans(ops.bgr2lab(dev), ldev, cen, seed=3, max_rounds=2)
    plan2 = ops.RegionDescriptorPlan(1, H, W, nseg)
    cld2, ehd2, csd2 = plan2.run(dev, ldev, bbox=dcd.bbox, lists=dcd.list_scratch)
    assert int(plan2.too_big.item()) == 1
    assert np.array_equal(cld2.cpu().numpy()[0], cld) and np.array_equal(ehd2.cpu().numpy()[0], ehd)
    assert np.array_equal(csd2.cpu().numpy()[0], csd)


def test_large_cluster_centre_sums_integer_vs_float32():
    """A region whose clusters hold more than 65,793 pixels (a merged region of the class API on
    a large image): mpeg7Descriptors.py:96-99 adds the points up in float32, which rounds beyond
    2^24; the product adds integers (deviation table).  The product must
    equal the C restatement (integer sums) bit for bit, and stay within north_star's 1e-4 of the
    literal numpy rule (np.sum(dtype=float32), oracle/dcd_ref.py)."""
    torch = _torch()
    from imgan import ops
    from imgan._lib import lib, check
    from oracle import c_oracle
    from oracle.color import bgr2lab_u8
    from oracle.dcd_ref import dominant_colour
    H = W = 400
    k = 2
    rng = np.random.RandomState(9)
    img = np.zeros((H, W, 3), dtype=np.float64)
    img[:, :220] = (200, 210, 220)                               # two bright, well separated colours
    img[:, 220:] = (60, 240, 150)
    img = np.clip(img + rng.normal(0, 6, img.shape), 0, 255).astype(np.uint8)
    lab_np = bgr2lab_u8(img)
    labels = np.zeros((H, W), dtype=np.int32)
    idx = np.arange(H * W)
    init = np.array([5 * W + 7, 300 * W + 390], dtype=np.int32)
    cen0 = lab_np.reshape(-1, 3)[init].astype(np.float32).reshape(1, 1, k, 3)
    plan = ops.DcdPlan(1, H, W, 1, k)
    lab = torch.from_numpy(lab_np).cuda().unsqueeze(0)
    dl = torch.from_numpy(labels).cuda().unsqueeze(0)
    cen = torch.from_numpy(cen0.copy()).cuda()
    plan.kmeans(lab, dl, cen, max_rounds=1000)
    assert plan.big_segments == 1                                # streaming rounds (beyond LDS)
    stats = plan.statistics(lab, dl).cpu().numpy()[0, 0]
    rounds = int(plan.rounds()[0].cpu().numpy()[0, 0])
    rc = np.stack(np.divmod(idx, W), axis=1).astype(np.int32)
    w_assign, w_cen, w_stats, w_rounds = c_oracle.dcd_segment(lab_np.reshape(-1, 3), rc, init)
    assert (w_stats[:, 0] > 65793).all() and (w_stats[:, 1:4].max(axis=0) > (1 << 24)).any()
    assert rounds == w_rounds
    assert np.array_equal(cen.cpu().numpy()[0, 0].view(np.uint32), w_cen.view(np.uint32))
    assert np.array_equal(stats.astype(np.uint64), w_stats)
    assert np.array_equal(plan.assign[0].cpu().numpy().ravel(), w_assign.astype(np.uint8))
    feats = torch.empty((1, 1, k, 8), dtype=torch.float32, device="cuda")
    check(lib.imgan_dcd_features(ops._ptr(plan.stats), ops._ptr(cen), 1, 1, k, ops._ptr(feats),
                                 ops._stream()))
    f = feats.cpu().numpy()[0, 0]
    lit, lit_rounds = dominant_colour(lab_np.reshape(-1, 3), rc, list(init), (H, W))
    assert lit_rounds == rounds
    for j in range(k):
        mean, var, share, coh = lit[j]
        assert np.allclose(f[j, 0:3], mean, rtol=1e-4, atol=0), (j, f[j, 0:3], mean)
        assert np.allclose(f[j, 3:6], var, rtol=1e-4, atol=1e-3)
        assert np.isclose(f[j, 6], share, rtol=1e-6) and np.isclose(f[j, 7], coh, rtol=1e-6)


def _check_regions_against_oracle(torch, img, labels, nseg, expect_large):
    from imgan import ops
    plan = ops.RegionDescriptorPlan(1, img.shape[0], img.shape[1], nseg)
    cld, ehd, csd = plan.run(torch.from_numpy(img).cuda().unsqueeze(0),
                             torch.from_numpy(labels).cuda().unsqueeze(0))
    assert int(plan.too_big.item()) == expect_large
    assert plan.finish_large() == expect_large and int(plan.too_big.item()) == 0
    cld, ehd, csd = cld.cpu().numpy()[0], ehd.cpu().numpy()[0], csd.cpu().numpy()[0]
    for r in range(nseg):
        if not (labels == r).any():
            assert not cld[r].any() and not ehd[r].any() and not csd[r].any()
            continue
        wc, we, ws = _region_oracle_rows(img, labels, r)
        assert np.array_equal(cld[r], wc), r
        assert np.array_equal(ehd[r], we), r
        assert np.array_equal(csd[r], ws), r
    return cld, ehd, csd


def test_region_descriptors_of_merged_regions_golden(golden_dir):
    """The regions MPEG7Descriptors is actually built on (optigen_utils.py:46-49): the merged map of
    the golden road image, where two regions' boxes (53,120 and 28,710 cells) exceed the on-chip
    pass and take imgan_region_descriptors_large; every region against oracle.mpeg7_std."""
    import os
    torch = _torch()
    f = np.load(os.path.join(golden_dir, "road_320.npz"))
    img, merged = f["bgr"], f["merged"].astype(np.int32)
    _check_regions_against_oracle(torch, img, merged, int(merged.max()) + 1, expect_large=2)
    # the class methods on the same map (float64 labels as the reference hands them over)
    from mpeg7Descriptors import MPEG7Descriptors, ehd_quantise, csd_quantise
    from oracle import mpeg7_std as M
    mp = MPEG7Descriptors(merged.astype(np.float64), img)
    layout, edges, structure = (mp.mpeg7_region_colour_layout(), mp.mpeg7_region_edge_histogram(),
                                mp.mpeg7_region_colour_structure())
    assert sorted(layout) == sorted(int(v) for v in np.unique(merged))
    for r in layout:
        wc, we, ws = _region_oracle_rows(img, merged, r)
        assert layout[r] == [int(v) for v in np.concatenate([wc[0, :6], wc[1, :3], wc[2, :3]])]
        assert edges[r] == ehd_quantise(we[:80], we[80:])
        assert structure[r] == csd_quantise(ws[:256], int(ws[256]))


def test_region_descriptors_large_boxes_with_subsampling():
    """Boxes of 2^17 cells and more sub-sample the colour structure (factor 2 here), and their
    edge blocks grow past 4 pixels: a frame-filling background with holes, a ring and a half
    frame, all through the streaming path, against the oracle."""
    torch = _torch()
    H, W = 520, 760
    img = voronoi_image(H, W, 60, seed=12, noise=18.0)
    labels = np.zeros((H, W), dtype=np.int32)                   # 0: the whole frame minus the rest
    yy, xx = np.mgrid[0:H, 0:W]
    d = np.hypot(yy - 260, xx - 300)
    labels[(d < 230) & (d >= 120)] = 1                          # ring, box 460 x 460 -> factor 2
    labels[:, 600:] = 2                                         # strip, box 160 x 520 -> factor 1
    labels[400:, :200] = 3                                      # 120 x 200: on chip
    labels[5:9, 700:704] = -1                                   # unlabelled hole
    _check_regions_against_oracle(torch, img, labels, 5, expect_large=3)
