"""Synthetic test inputs (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py)."""
import numpy as np


def voronoi_image(H, W, K, seed, noise=8.0):
    """Synthetic input of SURVEY.md 8d: ceil(K/2) Voronoi sites with random BGR colours +
    N(0, noise^2), clipped to u8."""
    rng = np.random.RandomState(seed)
    ns = max(2, (K + 1) // 2)
    sx = rng.randint(0, W, ns)
    sy = rng.randint(0, H, ns)
    cols = rng.randint(0, 256, (ns, 3)).astype(np.float32)
    # nearest site on a coarse grid, then upsample (keeps this cheap for large images)
    g = 4
    yy, xx = np.mgrid[0:(H + g - 1) // g, 0:(W + g - 1) // g]
    best = np.full(yy.shape, np.inf, dtype=np.float32)
    idx = np.zeros(yy.shape, dtype=np.int32)
    for i in range(ns):
        d = (xx * g - sx[i]).astype(np.float32) ** 2 + (yy * g - sy[i]).astype(np.float32) ** 2
        m = d < best
        best[m] = d[m]
        idx[m] = i
    idx = np.repeat(np.repeat(idx, g, axis=0), g, axis=1)[:H, :W]
    img = cols[idx] + rng.normal(0.0, noise, (H, W, 3)).astype(np.float32)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)
