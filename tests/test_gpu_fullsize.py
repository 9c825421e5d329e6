"""GPU checks at BASELINE.json's full sizes (1080p K=400, 4K K=2000, 16384^2 K=50,000).

The numpy oracle needs minutes to hours at these sizes, so the checks are size-independent
properties, each evaluated by an independent torch restatement of the reference rule:
  * the label of a sampled pixel is the first minimum of slic.py:101-109 over ALL centres whose
    window (slic.py:94-100) holds it (brute force, exact int64 keys; the kernel only looks at
    the centres binned around the pixel's tile),
  * the final centres are floor(sum / count) of the final labels (slic.py:118-126),
  * last row / column stay -1, batch entries are independent, runs are reproducible,
  * RAG sums and the adjacency set equal a torch restatement of rag_segmentation.py:38-74,
  * dominant-colour statistics are consistent with the label map (counts, sums, coherence).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def _synth(B, H, W, K, seed):
    from bench import synth_batch
    return synth_batch(B, H, W, K, seed, "cuda")


def _segment_sums(torch, keys, values, n):
    """Counts and per-key integer sums of `values` (list of 1-D tensors) for keys in [0, n).
    Sort + prefix sums: torch.bincount with weights crawls on spatially coherent label maps
    (every lane of a wave adds to the same address)."""
    skeys, perm = torch.sort(keys.long())
    edges = torch.searchsorted(skeys, torch.arange(n + 1, device=keys.device))
    cnt = edges[1:] - edges[:-1]
    out = []
    for v in values:
        cs = torch.cat([torch.zeros(1, dtype=torch.long, device=keys.device),
                        torch.cumsum(v[perm].long(), 0)])
        out.append(cs[edges[1:]] - cs[edges[:-1]])
    return cnt, out


def _slic_rounds(plan, lab, cw, num_iter):
    """init + (num_iter-1) rounds; returns the centres that the LAST assignment used."""
    plan.init(lab)
    prev = None
    for it in range(1, num_iter):
        if it == num_iter - 1:
            prev = plan.centres.clone()
        plan.round(lab, cw)
    return prev


def _brute_force_labels(torch, lab_b, prev_b, S, cw, xs, ys, chunk=1024):
    """First minimum of the exact integer key over every centre whose window holds the pixel.
    Returns (label or -2 where no window covers the pixel, tie flag)."""
    H, W, _ = lab_b.shape
    n = prev_b.shape[0]
    cx = prev_b[:, 3].long().clamp(0, W - 1)
    cy = prev_b[:, 4].long().clamp(0, H - 1)
    ccol = lab_b[cy, cx].long()                                  # colour AT the centre position
    out = torch.empty(xs.numel(), dtype=torch.long, device=lab_b.device)
    tie = torch.zeros(xs.numel(), dtype=torch.bool, device=lab_b.device)
    idx = torch.arange(n, device=lab_b.device).view(1, n)
    big = torch.iinfo(torch.int64).max
    for i in range(0, xs.numel(), chunk):
        x = xs[i:i + chunk].view(-1, 1)
        y = ys[i:i + chunk].view(-1, 1)
        p = lab_b[y.view(-1), x.view(-1)].long()                 # [P, 3]
        inwin = ((x >= (cx - S).clamp(min=0).view(1, n)) & (x < (cx + S).clamp(max=W - 1).view(1, n)) &
                 (y >= (cy - S).clamp(min=0).view(1, n)) & (y < (cy + S).clamp(max=H - 1).view(1, n)))
        dc2 = ((p.view(-1, 1, 3) - ccol.view(1, n, 3)) ** 2).sum(-1)
        ds2 = (x - cx.view(1, n)) ** 2 + (y - cy.view(1, n)) ** 2
        key = dc2 * (S * S) + ds2 * (cw * cw)
        key = torch.where(inwin, key, torch.full_like(key, big))
        kmin = key.min(dim=1, keepdim=True).values
        first = torch.where(key == kmin, idx, torch.full_like(idx, n)).min(dim=1).values
        covered = inwin.any(dim=1)
        out[i:i + chunk] = torch.where(covered, first, torch.full_like(first, -2))
        tie[i:i + chunk] = ((key == kmin) & inwin).sum(dim=1) > 1
    return out, tie


def _check_slic(torch, B, H, W, K, cw=15, num_iter=10, nsample=2048, seed=99):
    from imgan import ops
    bgr = _synth(B, H, W, K, seed)
    lab = ops.bgr2lab(bgr)
    plan = ops.SlicPlan(B, H, W, K)
    prev = _slic_rounds(plan, lab, cw, num_iter)
    labels = plan.labels.clone()
    centres = plan.centres.clone()
    n, S = plan.n, plan.S

    # reproducible, and run() is the same loop
    l2, c2 = ops.SlicPlan(B, H, W, K).run(lab, cw, num_iter)
    assert torch.equal(l2, labels) and torch.equal(c2, centres)

    # reference quirks at full size
    assert int(labels.min()) >= -1 and int(labels.max()) < n
    assert bool((labels[:, -1, :] == -1).all()) and bool((labels[:, :, -1] == -1).all())
    # nearly every inner pixel is covered (a nudged first seed can leave column / row 0 out)
    assert float((labels[:, :-1, :-1] >= 0).float().mean()) > 0.98

    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    ties = 0
    for b in range(B):
        # centres = floor(sum / count) over the final labels (slic.py:118-126)
        lb = labels[b].view(-1)
        ok = lb >= 0
        yy, xx = torch.meshgrid(torch.arange(H, device="cuda"), torch.arange(W, device="cuda"),
                                indexing="ij")
        vals = [lab[b, :, :, c].reshape(-1)[ok] for c in range(3)] + [xx.reshape(-1)[ok], yy.reshape(-1)[ok]]
        cnt, sums = _segment_sums(torch, lb[ok], vals, n)
        for f, s in enumerate(sums):
            want = torch.where(cnt > 0, s // cnt.clamp(min=1), torch.zeros_like(s))
            assert torch.equal(want.int(), centres[b, :, f]), (b, f)

        # sampled pixels against the brute-force rule over all centres
        xs = torch.randint(0, W - 1, (nsample,), generator=g, device="cuda")
        ys = torch.randint(0, H - 1, (nsample,), generator=g, device="cuda")
        want, tie = _brute_force_labels(torch, lab[b], prev[b], S, cw, xs, ys,
                                        chunk=max(64, (1 << 24) // n))
        got = labels[b, ys, xs].long()
        sure = (want >= 0) & ~tie                                # equal keys go by float64 ties
        assert torch.equal(got[sure], want[sure]), (b, int((got[sure] != want[sure]).sum()))
        ties += int(tie.sum())
    assert ties < 0.01 * B * nsample
    return bgr, lab, plan, labels


def test_slic_1080p_batch_properties():
    torch = _torch()
    B, H, W, K = 4, 1080, 1920, 400
    bgr, lab, plan, labels = _check_slic(torch, B, H, W, K)
    assert (plan.S, plan.nx, plan.ny) == (72, 26, 14)
    # batch entries are independent: image 2 alone gives the same map
    from imgan import ops
    l1, _ = ops.SlicPlan(1, H, W, K).run(lab[2:3].contiguous(), 15, 10)
    assert torch.equal(l1[0], labels[2])


def _torch_rag(torch, img_b, labels_b, K):
    """rag_segmentation.py:38-74 restated with torch ops: sums per label and the sorted set of
    (u, v), u < v, of 8-connected differently labelled pixels (both labelled)."""
    H, W = labels_b.shape
    lb = labels_b.view(-1)
    ok = (lb >= 0) & (lb < K)
    cnt, sums = _segment_sums(torch, lb[ok], [img_b[:, :, c].reshape(-1)[ok] for c in range(3)], K)
    keys = []
    L = labels_b.long()
    for dy, dx in ((0, 1), (1, 0), (1, 1), (1, -1)):
        a = L[0:H - dy, max(0, -dx):W - max(0, dx)]
        c = L[dy:H, max(0, dx):W + min(0, dx)]
        m = (a != c) & (a >= 0) & (c >= 0)
        u, v = torch.minimum(a[m], c[m]), torch.maximum(a[m], c[m])
        keys.append(torch.unique(u * (1 << 32) + v))
    keys = torch.unique(torch.cat(keys))
    return torch.stack(sums + [cnt], dim=1), torch.stack([keys >> 32, keys & 0xffffffff], dim=1)


def test_slic_rag_4k_properties():
    """configs[2]: 3840x2160, K=2000, SLIC + RAG merge."""
    torch = _torch()
    from imgan import ops
    B, H, W, K = 2, 2160, 3840, 2000
    bgr, lab, plan, labels = _check_slic(torch, B, H, W, K, nsample=1024)
    assert (plan.S, plan.nx, plan.ny) == (64, 59, 33)
    blur = ops.median7(bgr)
    lab_blur = ops.bgr2lab(blur)
    sums, edges = ops.rag_accumulate(lab_blur, labels, K)
    for b in range(B):
        want_sums, want_edges = _torch_rag(torch, lab_blur[b], labels[b], K)
        assert torch.equal(sums[b], want_sums)
        assert np.array_equal(edges[b].astype(np.int64), want_edges.cpu().numpy())
        s = sums[b].cpu().numpy()
        means = np.where(s[:, 3:4] > 0, s[:, :3] // np.maximum(s[:, 3:4], 1), 0)
        res = ops.rag_merge_host(K, edges[b], 0.65, means=means)
        lut = res["lut"]
        # merged ids: 1-based, one per surviving set, every vertex in exactly one set
        assert sorted(v for st in res["sets"] for v in st) == list(range(K))
        assert lut.min() >= 1 and len(np.unique(lut)) == len(res["sets"])
        # kept edges form a forest over the vertices: #sets = K - #kept edges
        assert len(res["sets"]) == K - len(res["mst"])
        merged = ops.relabel(labels[b:b + 1], torch.from_numpy(lut).cuda())[0]
        inner = labels[b] >= 0
        assert torch.equal(merged[inner].long(), torch.from_numpy(lut).cuda().long()[labels[b][inner].long()])
        assert bool((merged[~inner] == 0).all())                 # the -1 border becomes region 0


def test_mosaic_16k_superpixel_descriptors():
    """configs[4]: one 16384x16384 mosaic, K=50,000, dominant colours per superpixel."""
    torch = _torch()
    from imgan import ops
    from imgan._lib import lib, check
    B, H, W, K, k = 1, 16384, 16384, 50000, 8
    bgr, lab, plan, labels = _check_slic(torch, B, H, W, K, nsample=512)
    assert (plan.S, plan.nx, plan.ny) == (73, 223, 223)
    n = plan.n
    dcd = ops.DcdPlan(B, H, W, n, k)
    centres = torch.empty((B, n, k, 3), dtype=torch.float32, device="cuda")
    dcd.kmeans(lab, labels, centres, seed=4321, max_rounds=200, check_every=4)
    stats = dcd.statistics(lab, labels)
    feats = torch.empty((B, n, k, 8), dtype=torch.float32, device="cuda")
    check(lib.imgan_dcd_features(ops._ptr(stats), ops._ptr(centres), B, n, k, ops._ptr(feats),
                                 ops._stream()))
    torch.cuda.synchronize()
    lb = labels[0].view(-1)
    ok = lb >= 0
    seg_px, seg_sums = _segment_sums(torch, lb[ok], [lab[0, :, :, c].reshape(-1)[ok] for c in range(3)], n)
    assert torch.equal(stats[0, :, :, 0].sum(dim=1), seg_px)     # every pixel in one cluster
    for c in range(3):
        assert torch.equal(stats[0, :, :, 1 + c].sum(dim=1), seg_sums[c])
    a = dcd.assign[0].view(-1)
    assert bool((a[ok] < k).all()) and bool((a[~ok] == 0xff).all())
    # per (segment, cluster) counts recomputed from the assignment map
    pair = lb[ok].long() * k + a[ok].long()
    assert torch.equal(_segment_sums(torch, pair, [], n * k)[0].view(n, k), stats[0, :, :, 0])
    # largest component is at least one pixel and at most the cluster size
    big = stats[0, :, :, 7]
    cnt = stats[0, :, :, 0]
    assert bool((big <= cnt).all()) and bool((big[cnt > 0] >= 1).all())
    pct = feats[0, :, :, 6].double().sum(dim=1)
    assert bool(((pct - 1.0).abs()[seg_px > 0] < 1e-4).all())
    # adjacency at K = 50,000 goes through the hash set (no K x K bitmap)
    sums, edges = ops.rag_accumulate(lab, labels, K)
    want_sums, want_edges = _torch_rag(torch, lab[0], labels[0], K)
    assert torch.equal(sums[0], want_sums)
    assert np.array_equal(edges[0].astype(np.int64), want_edges.cpu().numpy())

    # per-superpixel colour layout / edge histogram / colour structure at this size
    # (region_desc_kernel as bench.py's configs[4] step runs it): properties for all 49,729 rows,
    # and a sample of regions exactly against the oracle on a crop that holds the region
    from oracle import mpeg7_std as M
    rd = ops.RegionDescriptorPlan(B, H, W, n)
    cld, ehd, csd = rd.run(bgr, labels)
    torch.cuda.synchronize()
    assert int(rd.too_big.item()) == 0
    box = ops.label_bbox(labels, n)[0]
    empty = seg_px == 0
    assert not bool(cld[0][empty].any()) and not bool(ehd[0][empty].any()) and not bool(csd[0][empty].any())
    bw = (box[:, 2] - box[:, 0] + 1).clamp(min=0).long()
    bh = (box[:, 3] - box[:, 1] + 1).clamp(min=0).long()
    nwin = csd[0, :, 256].long()
    # boxes below 256 x 256: sub-sampling 1, so at most (bw - 7)(bh - 7) windows; a window with a
    # cell of the region counts once and holds at least one colour; a colour at most nwin times
    assert bool((nwin <= (bw - 7).clamp(min=0) * (bh - 7).clamp(min=0)).all())
    hist = csd[0, :, :256].long()
    assert bool((hist.max(dim=1).values <= nwin).all()) and bool((hist.sum(dim=1) >= nwin).all())
    # edge blocks: the five edge types of a sub-image never exceed its full blocks
    assert bool((ehd[0, :, :80].view(n, 16, 5).sum(dim=2) <= ehd[0, :, 80:]).all())
    assert int(ehd[0, :, 80:].sum()) > n and int(nwin.sum()) > n
    rng = np.random.RandomState(5)
    for r in rng.choice(n, 48, replace=False).tolist() + [0, n - 1, 222, 223]:
        x0, y0, x1, y1 = [int(v) for v in box[r].tolist()]
        if x1 < x0:
            continue
        ci = bgr[0, y0:y1 + 1, x0:x1 + 1].cpu().numpy()
        cl = labels[0, y0:y1 + 1, x0:x1 + 1].cpu().numpy()
        assert np.array_equal(cld[0, r].cpu().numpy(), M.region_colour_layout_full(ci, cl, r)), r
        counts, nblk = M.region_edge_histogram_counts(ci, cl, r)
        assert np.array_equal(ehd[0, r].cpu().numpy(), np.concatenate([counts, nblk])), r
        h_, nw_ = M.region_colour_structure_counts(ci, cl, r)
        assert np.array_equal(csd[0, r].cpu().numpy(), np.concatenate([h_, [nw_]])), r
    # ... and the listed form (the regions' pixels from the runs the dominant-colour pass left
    # behind, as bench.py's configs[4] step takes them): all 49,729 rows bit for bit
    rd2 = ops.RegionDescriptorPlan(B, H, W, n)
    cld2, ehd2, csd2 = rd2.run(bgr, labels, bbox=dcd.bbox, lists=dcd.list_scratch)
    torch.cuda.synchronize()
    assert int(rd2.too_big.item()) == 0
    assert torch.equal(cld2, cld) and torch.equal(ehd2, ehd) and torch.equal(csd2, csd)
