"""Bounding boxes of the superpixels as a by-product of the last SLIC round
(imgan_slic_assign_accumulate_bbox) and the dominant-colour kernel that takes them
(imgan_dcd_segments_boxed): both against the stand-alone pass over the label map."""
import numpy as np
import pytest

from conftest import voronoi_image

pytestmark = pytest.mark.gpu


def _boxes_numpy(labels, n):
    out = np.empty((n, 4), np.int32)
    out[:, :2] = 0x7fffffff
    out[:, 2:] = -1
    ys, xs = np.nonzero(labels >= 0)
    lb = labels[ys, xs]
    np.minimum.at(out[:, 0], lb, xs)
    np.minimum.at(out[:, 1], lb, ys)
    np.maximum.at(out[:, 2], lb, xs)
    np.maximum.at(out[:, 3], lb, ys)
    return out


@pytest.mark.parametrize("B,H,W,K,cw", [
    (2, 213, 320, 200, 15),      # S = 18: 4 rows per lane
    (1, 270, 480, 100, 10),      # S = 36: 8 rows per lane, one band per tile
    (2, 200, 260, 16, 15),       # S = 57: two bands per tile
    (1, 203, 259, 16, 15),       # unaligned rows
    (1, 96, 128, 12, 40.5),      # non-integral colour weight: literal float64 key path
    (1, 64, 64, 256, 15),        # S = 4: more centres than a tile record holds (slow tile path)
])
def test_slic_round_boxes_equal_label_map_boxes(B, H, W, K, cw):
    import ctypes
    import torch
    from imgan import ops
    from imgan._lib import lib, check
    img = np.stack([voronoi_image(H, W, max(4, min(K, 64)), seed=50 + b) for b in range(B)])
    lab = ops.bgr2lab(torch.from_numpy(img).cuda())
    plan = ops.SlicPlan(B, H, W, K)
    box = torch.full((B, plan.n, 4), 12345, dtype=torch.int32, device="cuda")
    plan.init(lab)
    for it in range(3):
        plan.round(lab, cw, bbox=box if it == 2 else None)
    labels = plan.labels.cpu().numpy()
    got = box.cpu().numpy()
    for b in range(B):
        assert np.array_equal(got[b], _boxes_numpy(labels[b], plan.n)), b
    # and the stand-alone kernel agrees
    ref = torch.empty_like(box)
    check(lib.imgan_label_bbox(ops._ptr(plan.labels), B, H, W, plan.n, ops._ptr(ref), ops._stream()))
    assert torch.equal(ref, box)


def test_boxes_follow_pixels_that_keep_their_label():
    """A pixel no window reaches in the last round keeps its label (slic.py:81,113) and still
    counts for the box of that label."""
    import torch
    from imgan import ops
    H, W, K = 120, 160, 12
    img = voronoi_image(H, W, 6, seed=9)
    lab = ops.bgr2lab(torch.from_numpy(img[None]).cuda())
    plan = ops.SlicPlan(1, H, W, K)
    plan.init(lab)
    plan.round(lab, 15)
    # move every centre into one corner: most of the image is out of reach of every window
    plan.centres[:, :, 3] = 3
    plan.centres[:, :, 4] = 3
    box = torch.empty((1, plan.n, 4), dtype=torch.int32, device="cuda")
    before = plan.labels.clone()
    plan.round(lab, 15, bbox=box)
    labels = plan.labels.cpu().numpy()[0]
    assert (labels == before.cpu().numpy()[0]).mean() > 0.5          # most pixels kept their label
    assert np.array_equal(box.cpu().numpy()[0], _boxes_numpy(labels, plan.n))


def test_dominant_colours_with_given_boxes_equal_the_plain_call():
    import torch
    from imgan.pipeline import FullPipeline
    B, H, W, K = 3, 216, 384, 60
    img = np.stack([voronoi_image(H, W, 20, seed=70 + b) for b in range(B)])
    bgr = torch.from_numpy(img).cuda()
    pipe = FullPipeline(B, H, W, K, want_assign=True)
    pipe.run_slic(bgr)
    assert pipe.boxed
    pipe.run_dcd()
    feats, assign = pipe.features.clone(), pipe.dcd.assign.clone()
    stats = pipe.dcd.stats.clone()
    pipe.boxed = False                                  # the kernel finds the boxes itself
    pipe.run_dcd()
    assert torch.equal(pipe.dcd.stats, stats) and torch.equal(pipe.dcd.assign, assign)
    assert torch.equal(pipe.features.view(torch.int32), feats.view(torch.int32))


def test_boxes_beyond_the_image_are_clamped():
    """Boxes handed to imgan_dcd_segments_boxed come from the caller: sides pushed past the image
    edge must be clamped by the kernel (same result as the exact boxes), not read out of bounds."""
    import torch
    from imgan.pipeline import FullPipeline
    B, H, W, K = 2, 216, 384, 60
    img = np.stack([voronoi_image(H, W, 20, seed=80 + b) for b in range(B)])
    bgr = torch.from_numpy(img).cuda()
    pipe = FullPipeline(B, H, W, K, want_assign=True)
    pipe.run_slic(bgr)
    pipe.run_dcd()
    feats, assign, stats = pipe.features.clone(), pipe.dcd.assign.clone(), pipe.dcd.stats.clone()
    box = pipe.dcd.bbox[:B * pipe.n * 16].view(torch.int32).view(B, pipe.n, 4)
    live = box[..., 2] >= 0
    box[..., 0] = torch.where(live & (box[..., 0] == 0), torch.full_like(box[..., 0], -3), box[..., 0])
    box[..., 1] = torch.where(live & (box[..., 1] == 0), torch.full_like(box[..., 1], -5), box[..., 1])
    box[..., 2] = torch.where(live & (box[..., 2] == W - 2), torch.full_like(box[..., 2], W + 7), box[..., 2])
    box[..., 3] = torch.where(live & (box[..., 3] == H - 2), torch.full_like(box[..., 3], H + 2), box[..., 3])
    assert int((box[..., 0] < 0).sum()) > 0 and int((box[..., 3] >= H).sum()) > 0
    pipe.run_dcd()
    assert torch.equal(pipe.dcd.stats, stats) and torch.equal(pipe.dcd.assign, assign)
    assert torch.equal(pipe.features.view(torch.int32), feats.view(torch.int32))


def test_crowded_tile_takes_the_fast_loop_with_global_overflow_slots():
    """More centres reach one tile than its LDS accumulators were sized for (but no more than a
    tile record holds): the pixels won by the slots beyond go to global memory one by one.
    Labels against a brute-force evaluation of the exact key, centres against the label map,
    boxes against the label map."""
    import torch
    from imgan import ops
    from test_gpu_fullsize import _brute_force_labels, _segment_sums
    H, W, K, cw = 240, 320, 48, 15
    img = voronoi_image(H, W, 30, seed=3)
    lab = ops.bgr2lab(torch.from_numpy(img[None]).cuda())
    plan = ops.SlicPlan(1, H, W, K)
    n, S = plan.n, plan.S
    assert n == 35 and S == 40
    plan.init(lab)
    plan.round(lab, cw)
    before = plan.labels.clone()
    # every centre into a 40 x 30 patch: the tiles around it see all 35 of them
    idx = torch.arange(n, device="cuda")
    plan.centres[0, :, 3] = (100 + (idx % 7) * 6).int()
    plan.centres[0, :, 4] = (80 + (idx // 7) * 6).int()
    prev = plan.centres[0].clone()
    box = torch.empty((1, n, 4), dtype=torch.int32, device="cuda")
    plan.round(lab, cw, bbox=box)
    labels = plan.labels[0]
    ys, xs = torch.meshgrid(torch.arange(H - 1, device="cuda"), torch.arange(W - 1, device="cuda"),
                            indexing="ij")
    want, tie = _brute_force_labels(torch, lab[0], prev, S, cw, xs.reshape(-1), ys.reshape(-1))
    got = labels[:H - 1, :W - 1].reshape(-1).long()
    kept = before[0, :H - 1, :W - 1].reshape(-1).long()
    expect = torch.where(want == -2, kept, want)
    assert int(((got != expect) & ~tie).sum()) == 0
    assert int((want != -2).sum()) > 5000                    # the crowded tiles were exercised
    lb = labels.reshape(-1)
    ok = lb >= 0
    yy, xx = torch.meshgrid(torch.arange(H, device="cuda"), torch.arange(W, device="cuda"), indexing="ij")
    vals = [lab[0, :, :, c].reshape(-1)[ok] for c in range(3)] + [xx.reshape(-1)[ok], yy.reshape(-1)[ok]]
    cnt, sums = _segment_sums(torch, lb[ok], vals, n)
    for f, s in enumerate(sums):
        wantc = torch.where(cnt > 0, s // cnt.clamp(min=1), torch.zeros_like(s))
        assert torch.equal(wantc.int(), plan.centres[0, :, f]), f
    assert np.array_equal(box.cpu().numpy()[0], _boxes_numpy(labels.cpu().numpy(), n))
