"""CPU checks of the C-ABI boundary: the library loads, exports exactly what include/imgan.h
declares, and its host-only entry points agree with the oracle.  No kernel is launched."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, GOLDEN


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "imgan.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(imgan_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from imgan import _lib
    declared = _header_symbols()
    assert len(declared) >= 25
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), "declared in imgan.h but not exported: " + name
    assert sorted(_lib.SIGNATURES.keys()) == declared      # ctypes table mirrors the header


def test_every_entry_point_cites_the_reference():
    text = open(os.path.join(ROOT, "include", "imgan.h")).read()
    for ref in ("slic.py:", "rag_segmentation.py:", "mpeg7Descriptors.py:"):
        assert ref in text


def test_version_and_error_plumbing():
    from imgan import _lib
    assert _lib.lib.imgan_version() >= 100
    rc = _lib.lib.imgan_bgr2lab_u8(None, None, 16, None)         # rejected before any HIP call
    assert rc == _lib.E_INVALID
    assert "null" in _lib.last_error()
    with pytest.raises(ValueError):
        _lib.check(rc)


@pytest.mark.parametrize("H,W,K,expect", [
    (1050, 1573, 200, (90, 16, 11)), (213, 320, 200, (18, 17, 11)), (1080, 1920, 400, (72, 26, 14)),
    (2160, 3840, 2000, (64, 59, 33)), (16384, 16384, 50000, (73, 223, 223))])
def test_slic_grid_table(H, W, K, expect):
    """SURVEY.md section 8 grid table (slic.py:29,33,42-43 with Python-2 floor division)."""
    from imgan import ops
    from oracle.slic_ref import slic_step, slic_grid
    assert ops.slic_grid(H, W, K) == expect
    S = slic_step(H, W, K)
    xs, ys = slic_grid(H, W, S)
    assert (S, len(xs), len(ys)) == expect


def test_slic_grid_rejects_bad_input():
    from imgan import ops
    with pytest.raises(ValueError):
        ops.slic_grid(10, 10, 1000)          # step 0: more clusters than pixels
    with pytest.raises(ValueError):
        ops.slic_grid(0, 10, 4)


def test_lab_tables_match_oracle_and_golden():
    from imgan import ops
    from oracle.color import lab_tables
    g, c, C = ops.lab_tables()
    og, oc, oC = lab_tables()
    assert np.array_equal(g, og) and np.array_equal(c, oc) and np.array_equal(C, oC)
    gold = np.load(os.path.join(GOLDEN, "lab_tables.npz"))
    assert np.array_equal(g, gold["gamma"]) and np.array_equal(c, gold["cbrt"])
    assert np.array_equal(C, gold["coeffs"])


@pytest.mark.parametrize("name", ["road_320", "bear_320", "synth_96x64"])
def test_rag_merge_host_matches_golden(name):
    from imgan import ops
    f = np.load(os.path.join(GOLDEN, name + ".npz"))
    K = f["means"].shape[0]
    res = ops.rag_merge_host(K, f["edges"], 0.65, means=f["means"])
    assert np.array_equal(res["mst"], f["mst"])
    assert np.array_equal(res["mst_w"], f["mst_w"])
    lut = res["lut"]
    lab = f["slic_labels"]
    merged = np.where(lab >= 0, lut[np.maximum(lab, 0)], 0)
    assert np.array_equal(merged, f["merged"])
    # weights mode (the drop-in class path): same result when fed the same edges in order
    w = np.sqrt(((f["means"][f["edges"][:, 0]] - f["means"][f["edges"][:, 1]]) ** 2).sum(1).astype(np.float64))
    res2 = ops.rag_merge_host(K, f["edges"], 0.65, weights=w)
    assert np.array_equal(res2["lut"], lut) and res2["sets"] == res["sets"]


def test_rag_merge_host_edge_cases():
    from imgan import ops
    means = np.array([[10, 10, 10], [10, 10, 12], [200, 0, 0]], dtype=np.int32)
    # no edges: every vertex its own set, ids 1..K
    r = ops.rag_merge_host(3, np.zeros((0, 2), np.int32), 0.65, means=means)
    assert r["lut"].tolist() == [1, 2, 3] and len(r["mst"]) == 0
    # one MST edge: mean of the empty remainder is NaN -> dropped (rag_segmentation.py:166-168)
    r = ops.rag_merge_host(2, np.array([[0, 1]], np.int32), 2.0, means=means[:2])
    assert r["lut"].tolist() == [1, 2]
    # union appends the second endpoint's set to the first's and keeps the first's slot
    m4 = np.array([[0, 0, 0], [0, 0, 1], [0, 0, 2], [90, 0, 0]], dtype=np.int32)
    r = ops.rag_merge_host(4, np.array([[0, 1], [1, 2], [2, 3]], np.int32), 0.0, means=m4)
    assert r["sets"] == [[0, 1, 2], [3]] and r["lut"].tolist() == [1, 1, 1, 2]
    with pytest.raises(ValueError):
        ops.rag_merge_host(2, np.array([[0, 5]], np.int32), 1.0, means=means[:2])


def test_product_path_needs_the_gpu():
    """No CPU fallback: without a device the operators raise instead of computing."""
    import torch
    from imgan import ops
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        ops.bgr2lab(torch.zeros((2, 2, 3), dtype=torch.uint8))
    from slic import SLICSuperPixel
    with pytest.raises(ValueError):
        SLICSuperPixel(None, 10)             # slic.py:10-11 check comes first


def test_rag_merge_host_batch_equals_per_image_merges():
    """imgan_rag_merge_host_batch (images spread over host threads) against one
    imgan_rag_merge_host call per image, ragged edge lists, an image without edges, and the
    error of a worker thread reaching the caller."""
    import ctypes
    from imgan import ops

    class FakeSums(object):                       # rag_merge_batch only calls .cpu().numpy()
        def __init__(self, a):
            self.a = a

        def cpu(self):
            return self

        def numpy(self):
            return self.a

    rng = np.random.RandomState(5)
    B, K = 9, 40
    sums = rng.randint(0, 4000, (B, K, 4)).astype(np.int64)
    sums[:, :, 3] = rng.randint(0, 30, (B, K))     # some empty labels (count 0 -> mean 0)
    edges = []
    for b in range(B):
        E = 0 if b == 3 else rng.randint(1, 120)
        uv = rng.randint(0, K, (E, 2))
        edges.append(uv[uv[:, 0] != uv[:, 1]].astype(np.int32))
    for threads in (1, 4, 64):
        luts = ops.rag_merge_batch(K, edges, FakeSums(sums), 0.65, threads=threads)
        for b in range(B):
            means = np.where(sums[b, :, 3:4] > 0, sums[b, :, :3] // np.maximum(sums[b, :, 3:4], 1), 0)
            assert np.array_equal(luts[b], ops.rag_merge_lut(K, edges[b], 0.65, means)), (threads, b)
    edges[5] = np.array([[0, K + 3]], np.int32)
    with pytest.raises(ValueError):
        ops.rag_merge_batch(K, edges, FakeSums(sums), 0.65, threads=4)


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under image-analysis_amd/ may import,
    include, load or execute it."""
    pkg = os.path.join(ROOT, "image-analysis_amd")
    bad = re.compile(r"^\s*(from|import)\s+oracle\b|#include\s*[<\"].*oracle|"
                     r"(import_module|__import__|CDLL|open)\(.*oracle", re.M)
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp", ".sh")):
                src = open(os.path.join(dirpath, fn)).read()
                assert not bad.search(src), "product file uses the oracle: " + fn


def test_leave_one_out_cut_with_equal_weights_follows_numpy():
    """rag_segmentation.py:163-170 on forests whose other weights are all equal: numpy's std is
    exactly 0 there (one other weight, or a uniform forest of a size whose sum is exact), so the
    strict '<' drops an edge that weighs the same as the others.  The merge must agree with the
    literal numpy rule (oracle/rag_ref.py), not keep the edge on rounding noise."""
    from imgan import ops
    from oracle.rag_ref import RAGOracle

    def literal(means, edges, c):
        ora = RAGOracle.__new__(RAGOracle)
        K = len(means)
        ng = {v: set() for v in range(K)}
        for u, v in edges:
            ng[u].add(float(v))
            ng[v].add(float(u))
        col = {v: np.array(means[v]) for v in range(K)}
        clusters = np.arange(K, dtype=np.float64).reshape(1, K)
        out = ora.concat_similar_regs(ora.find_edges(ng, col), clusters, c_factor=c)[0]
        return out.ravel().astype(np.int64)

    cases = [
        # two forest edges of the same weight sqrt(14): each sees one other of its own weight
        ([[0, 0, 0], [1, 2, 3], [2, 4, 6]], [[0, 1], [1, 2]]),
        # a uniform path of five vertices (four equal weights, each sees three equal others)
        ([[10 * i, 0, 3 * i] for i in range(5)], [[i, i + 1] for i in range(4)]),
        # uniform weights with a heavier one: the heavy edge goes, the others stay
        ([[0, 0, 0], [3, 0, 0], [6, 0, 0], [9, 0, 0], [9, 40, 0]], [[0, 1], [1, 2], [2, 3], [3, 4]]),
        # one other weight, smaller / larger
        ([[0, 0, 0], [5, 0, 0], [5, 7, 0]], [[0, 1], [1, 2]]),
    ]
    for means, edges in cases:
        for c in (0.0, 0.65, 2.0):
            m = np.array(means, np.int32)
            e = np.array(edges, np.int32)
            got = ops.rag_merge_host(len(means), e, c, means=m)["lut"]
            want = literal(means, edges, c)
            # same partition, canonical numbering on both sides for these path graphs
            assert np.array_equal(got, want), (means, edges, c, got, want)
