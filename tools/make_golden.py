"""Generates tests/golden/*.npz: small inputs derived from the reference's sample images
(decoded here with PIL, resized to 320 wide as optigen_utils.prepare_image does, plus one
synthetic image) together with the ORACLE's outputs for them.

The reference itself cannot run here (Python 2 + OpenCV), so these vectors pin
the oracle against itself over time and give the GPU tests fixed targets; they are data
(arrays), not reference code.  Run from the repository root:  python tools/make_golden.py
"""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.color import bgr2lab_u8, bgr2gray_u8, median_blur     # noqa: E402
from oracle.slic_ref import SLICOracle                            # noqa: E402
from oracle.rag_ref import RAGOracle                              # noqa: E402
from oracle.dcd_ref import DCDOracle, hash_resample_idx          # noqa: E402
from oracle import mpeg7_std as M                                 # noqa: E402
from oracle.synth import voronoi_image                            # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
RESAMPLE_SEED = 20261111          # seed of the re-sample draws (mpeg7Descriptors.py:101) in the fixtures
REF_DATA = "/root/reference/data"


def load_small(name, wres=320):
    im = Image.open(os.path.join(REF_DATA, name)).convert("RGB")
    w, h = im.size
    im = im.resize((wres, int(wres * (h / float(w)))), Image.BILINEAR)
    return np.ascontiguousarray(np.asarray(im)[..., ::-1])          # BGR, as cv2.imread gives


def init_indices(clusters, k, seed):
    rng = np.random.RandomState(seed)
    init = {}
    for seg in np.unique(clusters):
        n = int((clusters == seg).sum())
        init[int(seg)] = sorted(rng.choice(n, k, replace=False).tolist()) if n >= k else None
    return init


def make(name, img, K=200, k=8):
    rag = RAGOracle(img.copy(), slic_clust_num=K, slic_cw=15)
    slic = SLICOracle(rag.image, K, colour_weight=15, num_iter=10)
    slic.run(keep_history=True)
    cl = slic.get_clusters()
    col = rag.slic_mean_lab(cl, nchange=True)
    ng = rag.neighbours_regions(cl)
    ed = rag.find_edges(ng, col)
    merged, emst, wmst, vmst = rag.concat_similar_regs(ed, cl, c_factor=0.65)
    init = init_indices(merged, k, seed=7)
    ok = all(v is not None for v in init.values())
    out = dict(
        bgr=img, blurred=rag.image, lab_blurred=bgr2lab_u8(rag.image), lab=bgr2lab_u8(img),
        gray=bgr2gray_u8(img), slic_labels_round1=slic.history[0], slic_labels_round5=slic.history[4],
        slic_labels=cl.astype(np.int32), slic_centres=np.array(slic.centers, dtype=np.int32),
        means=np.array([col[c] for c in range(K)], dtype=np.int32),
        edges=np.array(sorted((u, int(v)) for u in ng for v in ng[u] if u < v), dtype=np.int32),
        merged=merged.astype(np.int32), mst=np.array(emst, dtype=np.int32).reshape(-1, 2),
        mst_w=np.array(wmst, dtype=np.float64),
        cld=M.cld_from_sums(*M.cld_block_sums(img)),
        ehd=np.concatenate(M.edge_histogram_counts(img)),
        csd=np.concatenate([M.colour_structure_counts(img)[0], [M.colour_structure_counts(img)[1]]]),
    )
    if ok:
        d = DCDOracle(merged, img)
        flat = merged.ravel()
        feats = d.mpeg7_dominant_colours(
            k, lambda seg, n, kk: init[seg],
            lambda seg, n, kk: hash_resample_idx(np.nonzero(flat == seg)[0], RESAMPLE_SEED, kk))
        out["dcd_resample_seed"] = np.int64(RESAMPLE_SEED)
        segs = sorted(feats.keys())
        out["dcd_segments"] = np.array(segs, dtype=np.int32)
        out["dcd_init"] = np.array([init[s] for s in segs], dtype=np.int32)
        out["dcd_mean"] = np.array([[feats[s][j][0] for j in range(k)] for s in segs], dtype=np.float64)
        out["dcd_var"] = np.array([[feats[s][j][1] for j in range(k)] for s in segs], dtype=np.float64)
        out["dcd_pct"] = np.array([[feats[s][j][2] for j in range(k)] for s in segs], dtype=np.float64)
        out["dcd_coh"] = np.array([[feats[s][j][3] for j in range(k)] for s in segs], dtype=np.float64)
        out["dcd_rounds"] = np.array([d.rounds[s] for s in segs], dtype=np.int32)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, img.shape, "superpixels", len(slic.centers), "regions", len(np.unique(merged)),
          "dcd" if ok else "no-dcd")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    make("road_320", load_small("road.jpg"))
    make("bear_320", load_small("bear.jpg"))
    make("synth_96x64", voronoi_image(64, 96, 16, seed=1), K=40, k=4)
    g, c, C = __import__("oracle.color", fromlist=["lab_tables"]).lab_tables()
    np.savez_compressed(os.path.join(OUT, "lab_tables.npz"), gamma=g, cbrt=c, coeffs=C)
