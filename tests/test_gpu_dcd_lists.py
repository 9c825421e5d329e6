"""The pixel-list front end of the dominant-colour kernel (imgan_dcd_segments_listed) against the
bounding-box gather it replaces (imgan_dcd_segments / _boxed): same arithmetic on the same pixels,
so centres, assignment, statistics and round counts must be identical (bit-exact), whichever
capacity tier a segment lands in.  The gather path itself is pinned to the oracle in
test_gpu_descriptors.py / test_gpu_fullsize_oracle.py."""
import numpy as np
import pytest

from conftest import voronoi_image, resample_rule

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch


def _run(ops, torch, lab, dl, nseg, k, lists, cen0=None, seed=None, **kw):
    B, H, W = dl.shape
    plan = ops.DcdPlan(B, H, W, nseg, k, lists=lists)
    assert plan.lists == lists
    if cen0 is None:
        cen = torch.zeros((B, nseg, k, 3), dtype=torch.float32, device="cuda")
    else:
        cen = torch.from_numpy(cen0.copy()).cuda()
    plan.kmeans(lab, dl, cen, seed=seed, **kw)
    stats = plan.statistics(lab, dl)
    rounds, done = plan.rounds()
    torch.cuda.synchronize()
    return (cen.cpu().numpy(), plan.assign.cpu().numpy().copy(), stats.cpu().numpy().copy(),
            rounds.cpu().numpy(), done.cpu().numpy(), plan.big_segments)


def _same(a, b):
    assert a[5] == b[5]
    for i in range(5):
        assert np.array_equal(a[i], b[i]), i


def test_label_counts_match_bincount():
    torch = _torch()
    from imgan import ops
    rng = np.random.RandomState(3)
    for (B, H, W, nseg) in [(1, 1, 1, 1), (2, 37, 301, 11), (1, 129, 517, 700), (3, 64, 64, 5)]:
        labels = rng.randint(-2, nseg + 3, size=(B, H, W)).astype(np.int32)
        if nseg == 700:                       # every pixel its own run: more labels per block
            labels = rng.randint(0, nseg, size=(B, H, W)).astype(np.int32)   # than table slots
        got = ops.label_counts(torch.from_numpy(labels).cuda(), nseg).cpu().numpy()
        for b in range(B):
            l = labels[b].ravel()
            want = np.bincount(l[(l >= 0) & (l < nseg)], minlength=nseg)
            assert np.array_equal(got[b], want), (B, H, W, nseg, b)


def test_slic_label_counts_are_the_counts_of_the_label_map():
    torch = _torch()
    from imgan import ops
    from oracle.color import bgr2lab_u8
    img = voronoi_image(150, 230, 40, seed=5, noise=12.0)
    lab = torch.from_numpy(bgr2lab_u8(img)).cuda().unsqueeze(0)
    plan = ops.SlicPlan(1, 150, 230, 60)
    plan.seed(lab)
    for it in range(3):
        plan.round(lab, 20.0, first=(it == 0))
        labels = plan.labels.cpu().numpy().ravel()
        want = np.bincount(labels[labels >= 0], minlength=plan.n)
        assert np.array_equal(plan.label_counts().cpu().numpy()[0], want), it


@pytest.mark.parametrize("k", [3, 5, 8, 12])
def test_listed_equals_gather_on_superpixels(k):
    """SLIC superpixels of two images (all three dedup tiers for k <= 8 -- 7,168 / 11,520 / 22,528
    px -- and the two for k = 12), hash-sampled initial centres."""
    torch = _torch()
    from imgan import ops
    from oracle.color import bgr2lab_u8
    H, W = 360, 500
    imgs = np.stack([voronoi_image(H, W, 50, seed=7 + i, noise=14.0) for i in range(2)])
    lab = torch.from_numpy(np.stack([bgr2lab_u8(im) for im in imgs])).cuda()
    # 30 clusters -> S = 77, superpixels of ~9000 px: every tier gets some
    slic = ops.SlicPlan(2, H, W, 30)
    slic.run(lab, 20.0, 4)
    dl = slic.labels
    a = _run(ops, torch, lab, dl, slic.n, k, False, seed=99)
    b = _run(ops, torch, lab, dl, slic.n, k, True, seed=99)
    _same(a, b)
    first = dl[0].cpu().numpy().ravel()
    sizes = np.bincount(first[first >= 0])
    assert (sizes <= 7168).any() and ((sizes > 11520) & (sizes <= 22528)).any(), sizes


def test_listed_equals_gather_edge_cases():
    """Empty segment ids, unlabelled pixels, a segment beyond the on-chip capacity (left to the
    streaming rounds by both), a one-pixel segment, a segment whose box is too large for its tier,
    given initial centres."""
    torch = _torch()
    from imgan import ops
    from oracle.color import bgr2lab_u8
    H, W = 220, 300
    k = 4
    img = voronoi_image(H, W, 20, seed=17, noise=20.0)
    labf = bgr2lab_u8(img)
    lab = torch.from_numpy(labf).cuda().unsqueeze(0)
    labels = np.full((H, W), -1, dtype=np.int32)
    labels[0:160, 0:200] = 0                       # 32000 px: beyond the largest tier
    labels[170:172, 0:50] = 1
    labels[180, 10] = 2                            # one pixel
    # id 3 stays empty
    labels[0:200:7, 210:290:5] = 4                 # few pixels scattered over a large box
    labels[190:220, 100:200] = 5
    labels[165, :] = 7                             # a thin line across the image; id 6 empty
    nseg = 9                                       # id 8 empty too
    dl = torch.from_numpy(labels).cuda().unsqueeze(0)
    rng = np.random.RandomState(1)
    cen0 = np.zeros((1, nseg, k, 3), dtype=np.float32)
    for s in range(nseg):
        idx = np.nonzero(labels.ravel() == s)[0]
        if len(idx):
            cen0[0, s] = labf.reshape(-1, 3)[rng.choice(idx, k, replace=len(idx) < k)]
    a = _run(ops, torch, lab, dl, nseg, k, False, cen0=cen0)
    b = _run(ops, torch, lab, dl, nseg, k, True, cen0=cen0)
    assert a[5] == 1
    _same(a, b)
    a = _run(ops, torch, lab, dl, nseg, k, False, seed=5)
    b = _run(ops, torch, lab, dl, nseg, k, True, seed=5)
    _same(a, b)


@pytest.mark.parametrize("k", [4, 8, 12])
def test_listed_equals_gather_with_a_callers_centres(k):
    """The tier kernels compute the rounds' distances in expanded form around a centring point and
    hand near ties to the literal rule by an error margin; the gather kernels subtract.  Initial
    centres a caller may pass -- fractions, values far outside 0..255, absurdly large ones (no margin
    holds: every entry takes the literal rule), copies of one another, all equal -- must give the
    same centres, assignment, statistics and round counts on both."""
    torch = _torch()
    from imgan import ops
    from oracle.color import bgr2lab_u8
    H, W = 240, 320
    img = voronoi_image(H, W, 40, seed=23, noise=16.0)
    labf = bgr2lab_u8(img)
    lab = torch.from_numpy(labf).cuda().unsqueeze(0)
    slic = ops.SlicPlan(1, H, W, 24)
    slic.run(lab, 20.0, 3)
    dl = slic.labels
    nseg = slic.n
    rng = np.random.RandomState(4)
    flat = labf.reshape(-1, 3).astype(np.float32)
    lbl = dl[0].cpu().numpy().ravel()
    cen0 = np.zeros((1, nseg, k, 3), dtype=np.float32)
    for s in range(nseg):
        idx = np.nonzero(lbl == s)[0]
        if not len(idx):
            continue
        c = flat[rng.choice(idx, k, replace=len(idx) < k)].copy()
        kind = s % 6
        if kind == 0:
            c += rng.uniform(-0.5, 0.5, size=c.shape).astype(np.float32)      # fractions
        elif kind == 1:
            c[0] = [300.5, -40.25, 128.0]                                        # outside the colour cube
            c[1] += 1000.0
        elif kind == 2:
            c[k - 1] = [1.0e20, -3.0e25, 7.0e18]                                 # beyond any margin
        elif kind == 3:
            c[1:] = c[0]                                                         # all one colour
        elif kind == 4:
            c[k // 2] = c[0]                                                     # one copy
            c[k - 1] += 0.25
        cen0[0, s] = c
    a = _run(ops, torch, lab, dl, nseg, k, False, cen0=cen0, seed=3)
    b = _run(ops, torch, lab, dl, nseg, k, True, cen0=cen0, seed=3)
    _same(a, b)
    assert (a[3] > 1).any()


@pytest.mark.parametrize("case", range(10))
def test_listed_equals_gather_on_hard_colour_sets(case):
    """Segments whose colours stress the error margin of the tiers' expanded distances: the full
    colour cube, two far modes, a few levels (exact ties between integer centres), nearly flat,
    gradients; block segments of 576 to 9216 pixels, k = 2..16, hash-sampled and fractional given
    centres (tools/stress_expanded.py runs more of the same)."""
    torch = _torch()
    from imgan import ops
    rng = np.random.RandomState(1000 + case)
    H, W = 192, 256
    yy, xx = np.mgrid[0:H, 0:W]
    kind = case % 5
    if kind == 0:
        img = rng.randint(0, 256, size=(H, W, 3)).astype(np.uint8)
    elif kind == 1:
        m = ((xx // 7 + yy // 5) & 1).astype(np.float32)[..., None]
        lo, hi = rng.randint(0, 60, 3), rng.randint(190, 256, 3)
        img = np.clip(lo + m * (hi - lo) + rng.normal(0, 6, (H, W, 3)), 0, 255).astype(np.uint8)
    elif kind == 2:
        img = (rng.randint(0, 5, size=(H, W, 3)) * 60).astype(np.uint8)
    elif kind == 3:
        img = np.clip(128 + rng.normal(0, 1.2, (H, W, 3)), 0, 255).astype(np.uint8)
    else:
        g = (xx * 255 // W).astype(np.float32)
        img = np.clip(np.stack([g, 255 - g, (yy * 255 // H)], -1) + rng.normal(0, 3, (H, W, 3)), 0, 255).astype(np.uint8)
    cell = [24, 48, 64, 96][case % 4]
    labels = ((np.arange(H)[:, None] // cell) * ((W + cell - 1) // cell) + (np.arange(W)[None, :] // cell)).astype(np.int32)
    nseg = int(labels.max()) + 1
    k = [2, 3, 5, 8, 8, 12, 16][case % 7]
    lab = torch.from_numpy(img).cuda().unsqueeze(0)
    dl = torch.from_numpy(labels).cuda().unsqueeze(0)
    cen0 = rng.uniform(0, 255, size=(1, nseg, k, 3)).astype(np.float32) if case % 3 == 2 else None
    a = _run(ops, torch, lab, dl, nseg, k, False, cen0=cen0, seed=7 + case)
    b = _run(ops, torch, lab, dl, nseg, k, True, cen0=cen0, seed=7 + case)
    _same(a, b)


def test_listed_with_wrong_counts_stays_in_bounds():
    """Counts that disagree with the label map (a caller's mistake) lose pixels, not memory
    safety: the call finishes and the segments whose counts were right are unaffected."""
    torch = _torch()
    from imgan import ops
    from oracle.color import bgr2lab_u8
    H, W = 120, 160
    img = voronoi_image(H, W, 12, seed=3, noise=10.0)
    lab = torch.from_numpy(bgr2lab_u8(img)).cuda().unsqueeze(0)
    labels = (np.arange(H)[:, None] // 30 * 4 + np.arange(W)[None, :] // 40).astype(np.int32)
    dl = torch.from_numpy(labels).cuda().unsqueeze(0)
    nseg, k = 16, 4
    good = _run(ops, torch, lab, dl, nseg, k, True, seed=8)
    plan = ops.DcdPlan(1, H, W, nseg, k)
    ops.label_bbox(dl, nseg)                                      # warm the helper as well
    plan.bbox[:nseg * 16].copy_(ops.label_bbox(dl, nseg).view(torch.uint8).flatten())
    counts = ops.label_counts(dl, nseg)
    counts[0, 3] = 10                      # too small
    counts[0, 5] = 2000                    # too large
    counts[0, 9] = -4                      # nonsense
    plan.counts.copy_(counts)
    cen = torch.zeros((1, nseg, k, 3), dtype=torch.float32, device="cuda")
    plan.kmeans(lab, dl, cen, seed=8, boxed=True, counted=True)
    torch.cuda.synchronize()
    stats = plan.stats.cpu().numpy()[0]
    assert stats[3, :, 0].sum() == 10 and stats[9, :, 0].sum() == 0
    assert stats[5, :, 0].sum() == 30 * 40
    for s in (0, 1, 2, 15):
        assert np.array_equal(stats[s], good[2][0, s])
    # a total beyond the pixel count: the runs behind it are cut short, nothing is written outside
    counts[0, 12] = 1 << 30
    plan.counts.copy_(counts)
    plan.kmeans(lab, dl, cen, seed=8, boxed=True, counted=True)
    torch.cuda.synchronize()
    stats = plan.stats.cpu().numpy()[0]
    for s in (0, 1, 2):
        assert np.array_equal(stats[s], good[2][0, s])
    assert stats[:, :, 0].sum() <= H * W


@pytest.mark.parametrize("levels", [32, 64, 128])
def test_empty_clusters_are_reseeded_the_same_way_everywhere(levels):
    """Quantised colours: sampled centres coincide, clusters run empty and are re-seeded
    (mpeg7Descriptors.py:101, hash rule in place of random.sample).  The pixel lists, the box
    gather and the streaming rounds must agree bit for bit, and with the oracle given the same
    draws."""
    torch = _torch()
    from imgan import ops
    from oracle.color import bgr2lab_u8
    from oracle import c_oracle
    H, W, k, seed = 120, 200, 6, 4242 + levels
    img = (voronoi_image(H, W, 60, seed=levels, noise=3.0) // levels * levels).astype(np.uint8)
    labf = bgr2lab_u8(img)
    lab = torch.from_numpy(labf).cuda().unsqueeze(0)
    labels = (np.arange(H)[:, None] // 40 * 5 + np.arange(W)[None, :] // 40).astype(np.int32)
    nseg = 15
    dl = torch.from_numpy(labels).cuda().unsqueeze(0)
    rng = np.random.RandomState(levels)
    flat = labels.ravel()
    cen0 = np.zeros((1, nseg, k, 3), dtype=np.float32)
    inits = {}
    for s in range(nseg):
        idx = np.nonzero(flat == s)[0]
        inits[s] = np.sort(rng.choice(len(idx), k, replace=False)).astype(np.int32)
        cen0[0, s] = labf.reshape(-1, 3)[idx[inits[s]]]
    a = _run(ops, torch, lab, dl, nseg, k, False, cen0=cen0, resample_seed=seed)
    b = _run(ops, torch, lab, dl, nseg, k, True, cen0=cen0, resample_seed=seed)
    c = _run(ops, torch, lab, dl, nseg, k, True, cen0=cen0, resample_seed=seed, streaming_only=True)
    _same(a, b)
    for i in range(5):
        assert np.array_equal(a[i], c[i]), i
    reseeded = 0
    for s in range(nseg):
        idx = np.nonzero(flat == s)[0]
        rc = np.stack(np.divmod(idx, W), axis=1).astype(np.int32)
        res = resample_rule(idx, seed, k)
        w_assign, w_cen, w_stats, w_rounds = c_oracle.dcd_segment(labf.reshape(-1, 3)[idx], rc,
                                                                  inits[s], resample_idx=res)
        k_assign, _, _, k_rounds = c_oracle.dcd_segment(labf.reshape(-1, 3)[idx], rc, inits[s])
        reseeded += int(k_rounds != w_rounds or not np.array_equal(k_assign, w_assign))
        assert b[3][0, s] == w_rounds, s
        assert np.array_equal(b[1][0].ravel()[idx], w_assign.astype(np.uint8)), s
        assert np.array_equal(b[0][0, s].view(np.uint32), w_cen.view(np.uint32)), s
        assert np.array_equal(b[2][0, s].astype(np.uint64), w_stats), s
    assert reseeded > 0                    # the draws did change the outcome of some segment


def test_dedup_table_overflow_falls_back_to_the_plain_body():
    """Uniform colour noise: nearly every pixel of a segment has a colour of its own, the colour
    table of the dedup tiers overflows and the workgroup runs the plain body on the segment; flat
    colours beside it: a handful of table entries.  Both must equal the box gather bit for bit."""
    torch = _torch()
    from imgan import ops
    rng = np.random.RandomState(11)
    H, W, k = 240, 400, 8
    labf = rng.randint(0, 256, size=(H, W, 3)).astype(np.uint8)        # taken as Lab directly
    labf[:, 300:] = (labf[:, 300:] // 128) * 128                       # 8 colours on the right
    lab = torch.from_numpy(labf).cuda().unsqueeze(0)
    labels = np.full((H, W), -1, dtype=np.int32)
    labels[0:70, 0:100] = 0            # 7,000 px, all different: overflows the smallest tier's table
    labels[70:180, 0:100] = 1          # 11,000 px: overflows the middle tier's
    labels[0:200, 100:200] = 2         # 20,000 px: the large tier's
    labels[200:240, 0:60] = 3          # 2,400 px: fits
    labels[0:70, 300:400] = 4          # flat colours, three sizes
    labels[70:180, 300:400] = 5
    labels[0:200, 200:300] = 6         # noise and, from column 300 on, nothing: plain noise
    dl = torch.from_numpy(labels).cuda().unsqueeze(0)
    a = _run(ops, torch, lab, dl, 7, k, False, seed=31)
    b = _run(ops, torch, lab, dl, 7, k, True, seed=31)
    assert a[5] == 0
    _same(a, b)


def test_checkerboard_overflows_the_join_list():
    """Two colours in a checkerboard: every pixel's north neighbour is of the other cluster while
    its north-east one is of its own, so every cell asks for an explicit union in the coherence
    pass and the list of such cells overflows in every tier (the rest take the flagged route).
    Each colour is one 8-connected component.  Gather, lists and the oracle must agree."""
    torch = _torch()
    from imgan import ops
    from oracle import c_oracle
    H, W, k = 150, 330, 2
    yy, xx = np.mgrid[0:H, 0:W]
    labf = np.where(((yy + xx) & 1)[..., None] == 0, np.array([60, 120, 140]),
                    np.array([180, 130, 100])).astype(np.uint8)
    lab = torch.from_numpy(labf).cuda().unsqueeze(0)
    labels = np.full((H, W), -1, dtype=np.int32)
    labels[0:150, 0:150] = 0           # 22,500 px: the large tier
    labels[0:100, 150:250] = 1         # 10,000 px: the middle one
    labels[0:80, 250:330] = 2          # 6,400 px: the smallest
    dl = torch.from_numpy(labels).cuda().unsqueeze(0)
    flat = labels.ravel()
    cen0 = np.zeros((1, 3, k, 3), dtype=np.float32)
    for s in range(3):
        idx = np.nonzero(flat == s)[0]
        cen0[0, s] = labf.reshape(-1, 3)[idx[:2]]
    a = _run(ops, torch, lab, dl, 3, k, False, cen0=cen0)
    b = _run(ops, torch, lab, dl, 3, k, True, cen0=cen0)
    assert a[5] == 0
    _same(a, b)
    for s in range(3):
        idx = np.nonzero(flat == s)[0]
        rc = np.stack(np.divmod(idx, W), axis=1).astype(np.int32)
        _, w_cen, w_stats, _ = c_oracle.dcd_segment(labf.reshape(-1, 3)[idx], rc,
                                                    np.array([0, 1], dtype=np.int32))
        assert np.array_equal(b[0][0, s].view(np.uint32), w_cen.view(np.uint32)), s
        assert np.array_equal(b[2][0, s].astype(np.uint64), w_stats), s
        assert sorted(int(v) for v in w_stats[:, 7]) == [len(idx) // 2, (len(idx) + 1) // 2], s


def test_cell_grid_up_to_the_tier_limit_keeps_clear_of_the_join_list():
    """A sparse segment of the smallest tier whose bounding box has 143 x 147 cells: with its border
    (one cell above, left and right) the coherence pass's 2-byte cell grid has 145 x 148 = 21,460
    cells and reaches within 80 bytes of the 6 x 7168 bytes the plan allows (tier_fits: 2 area + 8
    <= 6 CAP).  The colour table of that tier is 256 bytes shorter than
    that (its slot count is rounded down), and the accumulators -- later the list of cells that need
    an explicit union -- used to start right behind it: the last parents of such a box lay on top of
    the first 128 entries of the list (wrong components, or a union-find walk that never ends).
    Random membership gives ~800 list entries; lists, gather and the oracle must agree."""
    torch = _torch()
    from imgan import ops
    from oracle import c_oracle
    bw, bh, k = 143, 147, 2
    H, W = bh + 5, bw + 7
    rng = np.random.RandomState(11)
    for trial in range(3):
        member = rng.rand(bh, bw) < 0.32
        member[0, 0] = member[0, -1] = member[-1, 0] = member[-1, -1] = True   # the box is the whole 143 x 147
        member[-1, ::2] = True                                                 # members in the last row of cells
        labels = np.full((H, W), -1, dtype=np.int32)
        labels[3:3 + bh, 2:2 + bw][member] = 0
        n = int(member.sum())
        grid = (bw + 2) * (bh + 1)
        assert n <= 7168 and 2 * grid + 8 <= 6 * 7168 and 2 * grid > 6 * ((7168 * 2 // 3) // 64 * 64) + 2 * 7168
        labf = rng.randint(0, 2, size=(H, W, 1)) * np.array([120, 10, 60]) + np.array([60, 120, 100])
        labf = labf.astype(np.uint8)
        lab = torch.from_numpy(labf).cuda().unsqueeze(0)
        dl = torch.from_numpy(labels).cuda().unsqueeze(0)
        idx = np.nonzero(labels.ravel() == 0)[0]
        px = labf.reshape(-1, 3)[idx]
        first = [int(np.nonzero((px == c).all(axis=1))[0][0]) for c in np.unique(px, axis=0)]
        cen0 = px[first].astype(np.float32).reshape(1, 1, k, 3)
        a = _run(ops, torch, lab, dl, 1, k, False, cen0=cen0)
        for rep in range(4):                    # the list's order is a matter of timing
            b = _run(ops, torch, lab, dl, 1, k, True, cen0=cen0)
            _same(a, b)
        rc = np.stack(np.divmod(idx, W), axis=1).astype(np.int32)
        _, w_cen, w_stats, _ = c_oracle.dcd_segment(px, rc, np.array(first, dtype=np.int32))
        assert np.array_equal(b[0][0, 0].view(np.uint32), w_cen.view(np.uint32)), trial
        assert np.array_equal(b[2][0, 0].astype(np.uint64), w_stats), trial


def test_step_outputs_do_not_depend_on_the_run():
    """The pixel lists come out in the order the sort pass's atomics land and the coherence pass
    unites cells lock-free: the results must not depend on either.  Three passes of the bench step
    over the same four 1080p images, every output compared bit for bit."""
    torch = _torch()
    from bench import synth_batch
    from imgan.pipeline import FullPipeline
    bgr = synth_batch(4, 1080, 1920, 400, 77, "cuda")
    pipe = FullPipeline(4, 1080, 1920, 400, want_assign=True)
    ref = None
    for _ in range(3):
        rows, feats = pipe.step(bgr)
        got = (rows.clone(), feats.clone().view(torch.int32), pipe.dcd.assign.clone())
        torch.cuda.synchronize()
        if ref is None:
            ref = got
        else:
            for a, b in zip(ref, got):
                assert torch.equal(a, b)


@pytest.mark.parametrize("k", [3, 8])
def test_listed_equals_gather_on_a_bench_sized_batch(k):
    """The check that caught the round-2 builds that came out wrong (tools/check_lists_vs_gather.py),
    as a collected test: 8 bench images (1080p, K = 400: 2,912 superpixels, every tier), the dedup
    tiers against the box gather and against a second run of themselves, every segment's centres,
    statistics, round counts and assignment bit for bit."""
    torch = _torch()
    from bench import synth_batch
    from imgan import ops
    B, H, W, K = 8, 1080, 1920, 400
    lab = ops.bgr2lab(synth_batch(B, H, W, K, 4242, "cuda"))
    slic = ops.SlicPlan(B, H, W, K)
    slic.run(lab, 15, 10)
    dl = slic.labels

    def run(lists):
        plan = ops.DcdPlan(B, H, W, slic.n, k, lists=lists)
        cen = torch.zeros((B, slic.n, k, 3), dtype=torch.float32, device="cuda")
        plan.kmeans(lab, dl, cen, seed=99 + k)
        stats = plan.statistics(lab, dl)
        rounds, _ = plan.rounds()
        return cen.view(torch.int32).clone(), stats.clone(), rounds.clone(), plan.assign.clone(), plan.big_segments

    gather, lists, again = run(False), run(True), run(True)
    for x, y in ((gather, lists), (lists, again)):
        assert x[4] == y[4] == 0
        differ = ((x[0] != y[0]).flatten(2).any(2) | (x[1] != y[1]).flatten(2).any(2) | (x[2] != y[2]))
        assert int(differ.sum()) == 0, int(differ.sum())
        assert torch.equal(x[3], y[3])
