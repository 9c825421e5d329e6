"""GPU parity: BGR->Lab and SLIC (HIP path through the C ABI) vs the CPU oracle."""
import numpy as np
import pytest

from conftest import voronoi_image

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


@pytest.mark.parametrize("shape", [(1, 64, 96), (2, 213, 320), (1, 101, 157), (1, 1, 3)])
def test_bgr2lab_gray_match_oracle(shape):
    torch = _torch()
    from imgan import ops
    from oracle.color import bgr2lab_u8, bgr2gray_u8
    rng = np.random.RandomState(7)
    img = rng.randint(0, 256, shape + (3,)).astype(np.uint8)
    dev = torch.from_numpy(img).cuda()
    lab = ops.bgr2lab(dev).cpu().numpy()
    gray = ops.bgr2gray(dev).cpu().numpy()
    assert np.array_equal(lab, bgr2lab_u8(img))          # integer path: bit exact
    assert np.array_equal(gray, bgr2gray_u8(img))


def test_bgr2lab_exhaustive_2p24():
    """Every one of the 2^24 BGR triples (table driven, so this is cheap)."""
    torch = _torch()
    from imgan import ops
    from oracle.color import bgr2lab_u8
    v = np.arange(1 << 24, dtype=np.uint32)
    img = np.stack([v & 255, (v >> 8) & 255, (v >> 16) & 255], axis=-1).astype(np.uint8)
    img = img.reshape(4096, 4096, 3)
    lab = ops.bgr2lab(torch.from_numpy(img).cuda()).cpu().numpy()
    ref = bgr2lab_u8(img)
    assert np.array_equal(lab, ref)
    assert tuple(lab[0, 0]) == (0, 128, 128) and tuple(lab[-1, -1]) == (255, 128, 128)


CASES = [
    # H, W, K, cw, num_iter
    (64, 96, 24, 10, 10),
    (213, 320, 200, 15, 10),       # prepare_image size (optigen_utils.py:19-24), W % 4 == 0
    (101, 157, 60, 15, 6),         # unaligned rows
    (270, 480, 100, 10, 10),
    (96, 128, 12, 40, 4),          # large step -> u64 key path check via cw
    (200, 260, 16, 15, 10),        # S = 57: two bands per tile, 8 rows per lane, 32 x 16 wave patches
    (203, 259, 16, 15, 5),         # the same with unaligned rows
    (300, 420, 12, 15, 4),         # S = 102
]


@pytest.mark.parametrize("H,W,K,cw,iters", CASES)
def test_slic_matches_oracle(H, W, K, cw, iters):
    torch = _torch()
    from slic import SLICSuperPixel
    from oracle.slic_ref import SLICOracle
    img = voronoi_image(H, W, K, seed=1234 + H)
    sp = SLICSuperPixel(img, K, colour_weight=cw, num_iter=iters)
    sp.run()
    got = sp.get_clusters()
    assert got.dtype == np.float64 and got.shape == (H, W)

    # literal float64 oracle (slic.py:101-109).  north_star asks for >= 99.9 % agreement; the
    # kernel orders candidates by the exact integer key and falls back to the literal float64
    # expression only when two keys are equal, which reproduces the reference's float ties,
    # so the labels and the final centres are expected to be identical.
    lit = SLICOracle(img, K, colour_weight=cw, num_iter=iters, mode="float64")
    lit.run()
    assert sp.step == lit.step and len(sp.centers) == len(lit.centers)
    agree = float((got == lit.get_clusters()).mean())
    assert agree >= 0.999, agree
    assert np.array_equal(got, lit.get_clusters()), agree
    assert sp.centers == [list(map(int, c)) for c in lit.centers]
    # reference quirk: last row / column never labelled
    assert (got[-1, :] == -1).all() and (got[:, -1] == -1).all()


def test_slic_noninteger_colour_weight():
    torch = _torch()
    from slic import SLICSuperPixel
    from oracle.slic_ref import SLICOracle
    img = voronoi_image(120, 160, 40, seed=5)
    sp = SLICSuperPixel(img, 40, colour_weight=12.5, num_iter=5)
    sp.run()
    lit = SLICOracle(img, 40, colour_weight=12.5, num_iter=5, mode="float64")
    lit.run()
    assert np.array_equal(sp.get_clusters(), lit.get_clusters())


def test_slic_init_matches_oracle_and_kat():
    torch = _torch()
    from slic import SLICSuperPixel
    from oracle.slic_ref import SLICOracle
    img = voronoi_image(90, 130, 30, seed=3)
    sp = SLICSuperPixel(img, 30)
    sp.init_clusters()
    o = SLICOracle(img, 30)
    o.init_clusters()
    assert sp.centers == o.centers
    # KAT: uniform image -> every seed moves to (x-1, y-1) (first scanned, gradient 0)
    flat = np.full((90, 130, 3), 77, dtype=np.uint8)
    sp = SLICSuperPixel(flat, 30)
    sp.init_clusters()
    S = sp.step
    assert sp.centers[0][3:] == [S - 1, S - 1]


def test_slic_rejects_non_array():
    from slic import SLICSuperPixel
    with pytest.raises(ValueError):
        SLICSuperPixel(None, 10)


@pytest.mark.parametrize("kind", ["flat", "blocks", "quantised", "gradient"])
@pytest.mark.parametrize("K,cw", [(12, 15), (24, 1), (20, 40)])
def test_slic_exact_ties_large_step(kind, K, cw):
    """Images built to produce equal keys (flat areas, few colours, mirror symmetry) at steps
    that take the 8-rows-per-lane, two-band tiles: the second-lowest-key tracking and the
    float64 replay must leave exactly the reference's labels.  Oracle: the C restatement
    (bit-identical to the numpy oracle, tests/test_oracle_c.py; the numpy one needs minutes)."""
    torch = _torch()
    from imgan import ops
    from oracle import c_oracle
    from oracle.color import bgr2lab_u8
    H, W = 240, 320
    rng = np.random.RandomState(K + int(cw))
    if kind == "flat":
        img = np.full((H, W, 3), 77, np.uint8)
    elif kind == "blocks":
        img = np.zeros((H, W, 3), np.uint8)
        img[(np.arange(H)[:, None] // 40 + np.arange(W)[None, :] // 40) % 2 == 1] = (200, 120, 40)
    elif kind == "quantised":
        img = (voronoi_image(H, W, 30, seed=K) // 64 * 64).astype(np.uint8)
    else:
        img = np.empty((H, W, 3), np.uint8)
        img[...] = (np.arange(W)[None, :, None] // 16 * 12).astype(np.uint8)    # mirror-free ramps
        img[:, W // 2:] = img[:, :W // 2][:, ::-1]                                # and a mirror
    lab = bgr2lab_u8(img)
    want_labels, want_centres = c_oracle.slic_run(lab, K, cw, 6)
    plan = ops.SlicPlan(1, H, W, K)
    labels, centres = plan.run(torch.from_numpy(lab[None]).cuda(), cw, 6)
    assert plan.S >= 48
    assert np.array_equal(labels.cpu().numpy()[0], want_labels)
    assert np.array_equal(centres.cpu().numpy()[0], want_centres)


@pytest.mark.parametrize("H,W,K,B", [(512, 768, 6000, 1), (96, 160, 40, 3), (300, 200, 1500, 2),
                                      (65, 65, 1056, 1), (100, 130, 1400, 2)])    # the last two: S = 2 and 3
def test_slic_binning_paths(H, W, K, B):
    """The centres are binned by one workgroup per image while the cell grid fits LDS (4096 cells
    and centres) and by the count / scan / fill kernels beyond: both against the C oracle
    (bit-identical to the numpy oracle, tests/test_oracle_c.py), exactly."""
    torch = _torch()
    from imgan import ops
    from oracle import c_oracle
    from oracle.color import bgr2lab_u8
    labs = np.stack([bgr2lab_u8(voronoi_image(H, W, max(K // 8, 6), seed=K + b)) for b in range(B)])
    plan = ops.SlicPlan(B, H, W, K)
    labels, centres = plan.run(torch.from_numpy(labs).cuda(), 15, 5)
    if K == 6000:
        assert plan.n > 4096                              # the three-kernel path
    for b in range(B):
        want_labels, want_centres = c_oracle.slic_run(labs[b], K, 15, 5)
        assert np.array_equal(labels.cpu().numpy()[b], want_labels)
        assert np.array_equal(centres.cpu().numpy()[b], want_centres)


@pytest.mark.parametrize("H,W,K,cw", [(101, 157, 60, 15), (213, 320, 200, 15), (64, 64, 256, 10),
                                       (200, 260, 16, 15), (96, 128, 12, 0.5)])
def test_slic_first_round_writes_the_whole_map(H, W, K, cw):
    """seed() + round(first=True) never fills the label map with -1 (slic.py:81): the first round
    must write every pixel itself, the uncovered ones and the border included.  The map is
    poisoned beforehand; the result must equal the init() path and the oracle, exactly."""
    torch = _torch()
    from imgan import ops
    from oracle import c_oracle
    from oracle.color import bgr2lab_u8
    rng = np.random.RandomState(H + K)
    # a noisy image makes the centres drift, which leaves pixels out of every window
    img = (voronoi_image(H, W, max(K // 4, 5), seed=H) // 2 + rng.randint(0, 128, (H, W, 3))).astype(np.uint8)
    lab = bgr2lab_u8(img)
    dev = torch.from_numpy(lab[None]).cuda()
    for rounds in (1, 4):
        ref = ops.SlicPlan(1, H, W, K)
        ref.init(dev)
        for _ in range(rounds):
            ref.round(dev, cw)
        plan = ops.SlicPlan(1, H, W, K)
        plan.labels.fill_(0x5a5a5a5a)
        plan.seed(dev)
        for it in range(rounds):
            plan.round(dev, cw, first=(it == 0))
        assert torch.equal(plan.labels, ref.labels)
        assert torch.equal(plan.centres, ref.centres)
        fused = ops.SlicPlan(1, H, W, K)
        fused.labels.fill_(0x5a5a5a5a)
        fused.run(dev, cw, rounds + 1)
        assert torch.equal(fused.labels, ref.labels)
        got = ref.labels.cpu().numpy()[0]
        assert (got[-1] == -1).all() and (got[:, -1] == -1).all()
    if float(cw) == int(cw):
        want_labels, want_centres = c_oracle.slic_run(lab, K, cw, 5)
        assert np.array_equal(got, want_labels)
    # no round at all: the map is all -1
    none = ops.SlicPlan(1, H, W, K)
    none.labels.fill_(7)
    none.run(dev, cw, 1)
    assert int((none.labels != -1).sum()) == 0
