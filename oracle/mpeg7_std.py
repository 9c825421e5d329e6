"""Oracle: MPEG-7 colour layout (CLD), edge histogram (EHD) and colour structure
(CSD) descriptors (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

The reference does NOT contain these descriptors (SURVEY.md section 0 item 2, rows
a15-a17): scripts/mpeg7Descriptors.py implements dominant colour, homogeneous
texture and region shape only.  `north_star` names them, so they are defined
here from ISO/IEC 15938-3 and its XM reference software as recalled (no copy of
either exists in this image).  **Parity unpinned**: these functions are the
definition for this repository; the HIP kernels are checked bit-exactly
against them.

All decisions are taken in exact integer arithmetic (stated per function) so
CPU and GPU cannot disagree through floating-point rounding; the only double
arithmetic (CLD colour transform and DCT) uses a fixed operation order.
"""
import math

import numpy as np

from .color import bgr2gray_u8

# ----------------------------------------------------------------------------- CLD

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5,
                   12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                   58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63], dtype=np.int32)


def dct_table():
    """c[i][j] = s_i * cos(pi/8 * i * (j + 0.5)), s_0 = sqrt(1/8), s_i = 1/2."""
    c = np.zeros((8, 8), dtype=np.float64)
    for i in range(8):
        s = math.sqrt(0.125) if i == 0 else 0.5
        for j in range(8):
            c[i, j] = s * math.cos((math.pi / 8.0) * i * (j + 0.5))     # libm, as the C side
    return c


def cld_block_sums(img):
    """Per 8x8-grid block sums of B,G,R and pixel counts.
    block(x, y) = (x*8 // W, y*8 // H).  Returns (sums int64[64,3], counts int64[64])."""
    H, W = img.shape[:2]
    bx = (np.arange(W) * 8) // W
    by = (np.arange(H) * 8) // H
    blk = (by[:, None] * 8 + bx[None, :]).ravel()
    cnt = np.bincount(blk, minlength=64).astype(np.int64)
    px = img.reshape(-1, 3).astype(np.int64)
    sums = np.stack([np.bincount(blk, weights=px[:, k], minlength=64).astype(np.int64)
                     for k in range(3)], axis=1)
    return sums, cnt


def _quant_ydc(i):
    if i > 192:
        return 112 + (i - 192) // 4
    if i > 160:
        return 96 + (i - 160) // 2
    if i > 96:
        return 32 + (i - 96)
    if i > 64:
        return 16 + (i - 64) // 2
    return i // 4


def _quant_cdc(i):
    if i > 191:
        return 63
    if i > 160:
        return 56 + (i - 160) // 4
    if i > 144:
        return 48 + (i - 144) // 2
    if i > 112:
        return 16 + (i - 112)
    if i > 96:
        return 8 + (i - 96) // 2
    if i > 64:
        return (i - 64) // 4
    return 0


def _quant_ac(i):
    i = min(i, 239)
    i = max(i, -256)
    a = abs(i)
    if a > 127:
        j = 64 + a // 4
    elif a > 63:
        j = 32 + a // 2
    else:
        j = a
    j = -j if i < 0 else j
    return j + 128


def _cdiv(a, b):
    """C integer division (truncation toward zero)."""
    q = abs(a) // b
    return -q if a < 0 else q


def cld_from_sums(sums, cnt):
    """Block averages (integer division) -> YCbCr -> 8x8 DCT -> zig-zag -> quantise.
    Returns int32[3,64]: rows Y, Cb, Cr in zig-zag order."""
    c = dct_table()
    small = np.zeros((3, 64), dtype=np.float64)
    for k in range(64):
        if cnt[k]:
            B = float(int(sums[k, 0]) // int(cnt[k]))
            G = float(int(sums[k, 1]) // int(cnt[k]))
            R = float(int(sums[k, 2]) // int(cnt[k]))
        else:
            B = G = R = 0.0
        yy = (0.299 * R + 0.587 * G + 0.114 * B) / 256.0
        small[0, k] = float(int(219.0 * yy + 16.5))
        small[1, k] = float(int(224.0 * 0.564 * (B / 256.0 * 1.0 - yy) + 128.5))
        small[2, k] = float(int(224.0 * 0.713 * (R / 256.0 * 1.0 - yy) + 128.5))
    out = np.zeros((3, 64), dtype=np.int32)
    for ch in range(3):
        blk = small[ch]
        tmp = np.zeros(64, dtype=np.float64)
        for i in range(8):
            for j in range(8):
                s = 0.0
                for k in range(8):
                    s += c[j, k] * blk[8 * i + k]
                tmp[8 * i + j] = s
        coef = np.zeros(64, dtype=np.int64)
        for j in range(8):
            for i in range(8):
                s = 0.0
                for k in range(8):
                    s += c[i, k] * tmp[8 * k + j]
                coef[8 * i + j] = int(np.floor(s + 0.499999))
        if ch == 0:
            out[ch, 0] = _quant_ydc(_cdiv(int(coef[0]), 8)) >> 1
        else:
            out[ch, 0] = _quant_cdc(_cdiv(int(coef[0]), 8))
        for i in range(1, 64):
            v = int(coef[ZIGZAG[i]])
            if ch == 0:
                out[ch, i] = _quant_ac(_cdiv(v, 2)) >> 3
            else:
                out[ch, i] = _quant_ac(v) >> 3
    return out


def colour_layout(img, ny=6, nc=3):
    """Default MPEG-7 CLD: 6 Y + 3 Cb + 3 Cr coefficients."""
    sums, cnt = cld_block_sums(img)
    q = cld_from_sums(sums, cnt)
    return np.concatenate([q[0, :ny], q[1, :nc], q[2, :nc]]).astype(np.int32)


# ----------------------------------------------------------------------------- EHD

EHD_QUANT = np.array([
    [0.010867, 0.057915, 0.099526, 0.144849, 0.195573, 0.260504, 0.358031, 0.530128],
    [0.012266, 0.069934, 0.125879, 0.182307, 0.243396, 0.314563, 0.411728, 0.564319],
    [0.004193, 0.025852, 0.046860, 0.068519, 0.093286, 0.123490, 0.161505, 0.228960],
    [0.004174, 0.025924, 0.046232, 0.067163, 0.089655, 0.115391, 0.151904, 0.217745],
    [0.006778, 0.051667, 0.108650, 0.166257, 0.224226, 0.285691, 0.356375, 0.450972]])


def ehd_block_size(H, W, desired=1100):
    temp = np.sqrt(W * H / float(desired))
    bs = int(temp / 2) * 2
    return bs if bs else 2


def edge_histogram_counts(img, threshold=11):
    """Returns (counts int32[80], blocks_per_subimage int32[16]).

    Image blocks of bs x bs (bs even) tile the image from the top-left; each is
    split into 2x2 quadrants with gray sums s1..s4 (q = (bs/2)^2 pixels each).
    Edge strengths, scaled by q: v=|s1-s2+s3-s4| h=|s1+s2-s3-s4|
    d45=sqrt2|s1-s4| d135=sqrt2|s2-s3| nd=2|s1-s2-s3+s4|.  Maximum taken in
    the order v,h,45,135,nd with strict '>' on the exact squares; the block
    votes iff max >= threshold*q.  bin = subimage*5 + type."""
    gray = bgr2gray_u8(img).astype(np.int64)
    H, W = gray.shape
    bs = ehd_block_size(H, W)
    hb = bs // 2
    q = hb * hb
    counts = np.zeros(80, dtype=np.int32)
    nblk = np.zeros(16, dtype=np.int32)
    for j in range(0, H - bs + 1, bs):
        for i in range(0, W - bs + 1, bs):
            sub = (i * 4) // W + ((j * 4) // H) * 4
            nblk[sub] += 1
            s1 = int(gray[j:j + hb, i:i + hb].sum())
            s2 = int(gray[j:j + hb, i + hb:i + bs].sum())
            s3 = int(gray[j + hb:j + bs, i:i + hb].sum())
            s4 = int(gray[j + hb:j + bs, i + hb:i + bs].sum())
            sq = [(s1 - s2 + s3 - s4) ** 2, (s1 + s2 - s3 - s4) ** 2,
                  2 * (s1 - s4) ** 2, 2 * (s2 - s3) ** 2, 4 * (s1 - s2 - s3 + s4) ** 2]
            best, e = sq[0], 0
            for t in range(1, 5):
                if sq[t] > best:
                    best, e = sq[t], t
            if best >= (threshold * q) ** 2:
                counts[sub * 5 + e] += 1
    return counts, nblk


def ehd_quantise(counts, nblk):
    """80 x 3-bit bins: nearest reconstruction level of counts/blocks."""
    out = np.zeros(80, dtype=np.int32)
    for b in range(80):
        n = nblk[b // 5]
        v = counts[b] / float(n) if n else 0.0
        tab = EHD_QUANT[b % 5]
        j = 0
        while j < 7 and v >= (tab[j] + tab[j + 1]) / 2.0:
            j += 1
        out[b] = j
    return out


# ----------------------------------------------------------------------------- CSD

HMMD_THRESH = {256: [0, 6, 20, 60, 110, 256], 128: [0, 6, 20, 60, 110, 256],
               64: [0, 6, 20, 60, 110, 256], 32: [0, 6, 60, 110, 256]}
HMMD_NHUE = {256: [1, 4, 16, 16, 16], 128: [1, 4, 8, 8, 8], 64: [1, 4, 4, 8, 8], 32: [1, 4, 4, 4]}
HMMD_NSUM = {256: [32, 8, 4, 4, 4], 128: [16, 4, 4, 4, 4], 64: [8, 4, 4, 2, 1], 32: [8, 4, 1, 1]}


def hmmd_quantise(img, nbins=256):
    """H x W x 3 uint8 BGR -> H x W uint8 HMMD cell index.

    max/min/diff/sum2=max+min; hue in integer degrees, rounded half up, with
    priority R, G, B for the maximum; sub-space from diff thresholds; hue and
    sum uniformly quantised inside the sub-space:
      iHue = hue*nHue // 360,  iSum = (sum2 - lo)*nSum // (2*(256 - lo))."""
    b = img[..., 0].astype(np.int64)
    g = img[..., 1].astype(np.int64)
    r = img[..., 2].astype(np.int64)
    mx = np.maximum(np.maximum(r, g), b)
    mn = np.minimum(np.minimum(r, g), b)
    d = mx - mn
    s2 = mx + mn
    num = np.where(r == mx, 60 * (g - b),
                   np.where(g == mx, 120 * d + 60 * (b - r), 240 * d + 60 * (r - g)))
    num = np.where(num < 0, num + 360 * d, num)
    dd = np.maximum(d, 1)
    hue = (2 * num + dd) // (2 * dd)
    hue = np.where(hue >= 360, hue - 360, hue)
    hue = np.where(d == 0, 0, hue)
    thr = HMMD_THRESH[nbins]
    nh = np.array(HMMD_NHUE[nbins], dtype=np.int64)
    ns = np.array(HMMD_NSUM[nbins], dtype=np.int64)
    off = np.concatenate([[0], np.cumsum(nh * ns)[:-1]])
    sub = np.zeros_like(d)
    for t in thr[1:-1]:
        sub += (d >= t)
    lo = np.array(thr[:-1], dtype=np.int64)[sub]
    ihue = hue * nh[sub] // 360
    isum = (s2 - lo) * ns[sub] // (2 * (256 - lo))
    return (off[sub] + ihue * ns[sub] + isum).astype(np.uint8)


def csd_subsample(H, W):
    p = int(np.floor(0.5 * np.log2(W * H) - 8 + 0.5))
    return 1 << max(0, p)


def colour_structure_counts(img, nbins=256):
    """Returns (hist int32[nbins], n_windows).  8x8 structuring element with
    sub-sampling K = 2^max(0, round(0.5 log2(WH) - 8)); window origins step by K;
    each colour present in a window counts once."""
    H, W = img.shape[:2]
    K = csd_subsample(H, W)
    nwx = (W - 8 * K) // K + 1 if W >= 8 * K else 0
    nwy = (H - 8 * K) // K + 1 if H >= 8 * K else 0
    hist = np.zeros(nbins, dtype=np.int32)
    if nwx <= 0 or nwy <= 0:
        return hist, 0
    sub = img[0:(nwy + 7) * K:K, 0:(nwx + 7) * K:K]
    qimg = hmmd_quantise(sub, nbins)
    for c in np.unique(qimg):
        m = (qimg == c)
        # window (u,v) holds colour c iff any of m[v:v+8, u:u+8]
        acc = np.zeros((nwy, nwx), dtype=bool)
        for dy in range(8):
            for dx in range(8):
                acc |= m[dy:dy + nwy, dx:dx + nwx]
        hist[int(c)] = int(acc.sum())
    return hist, nwx * nwy


CSD_AMP_THRESH = [0.000000001, 0.037, 0.08, 0.195, 0.32]
CSD_AMP_LEVELS = [1, 25, 20, 35, 35, 140]


def csd_quantise(hist, nwin):
    """Non-uniform 8-bit amplitude quantisation of hist / n_windows."""
    out = np.zeros(len(hist), dtype=np.int32)
    if nwin == 0:
        return out
    edges = [0.0] + CSD_AMP_THRESH + [1.0]
    for i, h in enumerate(hist):
        v = h / float(nwin)
        base = 0
        val = 255
        for r in range(6):
            lo, hi = edges[r], edges[r + 1]
            if v < hi or r == 5:
                if r == 0:
                    val = 0
                else:
                    val = base + int((v - lo) / (hi - lo) * CSD_AMP_LEVELS[r])
                    val = min(val, base + CSD_AMP_LEVELS[r] - 1)
                break
            base += CSD_AMP_LEVELS[r]
        out[i] = min(val, 255)
    return out


# ----------------------------------------------------------------------------- per region
# The same three descriptors for one region of a label map (BASELINE.json configs[4]:
# "descriptors per superpixel region"; SURVEY.md 8d prices them at 3 + 4 B/px: image + labels).
# Own definition, after the standard's treatment of arbitrarily shaped regions; parity unpinned.
# Everything is taken relative to the region's bounding box (x0, y0, bw, bh):
#   CLD: the 8 x 8 grid spans the box; a block's mean runs over the region's pixels in it; a block
#        without such pixels takes the region's mean colour (integer floor, as the block means)
#   EHD: image blocks tile the box from its corner, size from the box's area; a block votes only
#        if all of its pixels belong to the region
#   CSD: the structuring element slides over the box (sub-sampling from the box's area); only the
#        region's pixels put their colour into a window; windows that hold none are not counted

def region_bbox(labels, r):
    ys, xs = np.nonzero(labels == r)
    if ys.size == 0:
        return None
    return int(xs.min()), int(ys.min()), int(xs.max() - xs.min() + 1), int(ys.max() - ys.min() + 1)


def region_cld_block_sums(img, labels, r):
    x0, y0, bw, bh = region_bbox(labels, r)
    crop = img[y0:y0 + bh, x0:x0 + bw].astype(np.int64)
    m = labels[y0:y0 + bh, x0:x0 + bw] == r
    gx = (np.arange(bw) * 8) // bw
    gy = (np.arange(bh) * 8) // bh
    blk = (gy[:, None] * 8 + gx[None, :])[m]
    cnt = np.bincount(blk, minlength=64).astype(np.int64)
    sums = np.stack([np.bincount(blk, weights=crop[..., k][m], minlength=64).astype(np.int64)
                     for k in range(3)], axis=1)
    return sums, cnt


def region_colour_layout_full(img, labels, r):
    """int32 [3, 64]; empty blocks take floor(region sum / region count)."""
    sums, cnt = region_cld_block_sums(img, labels, r)
    tot, n = sums.sum(axis=0), int(cnt.sum())
    fill = np.array([int(t) // n for t in tot], dtype=np.int64)
    sums2, cnt2 = sums.copy(), cnt.copy()
    for k in range(64):
        if cnt[k] == 0:
            sums2[k], cnt2[k] = fill, 1
    return cld_from_sums(sums2, cnt2)


def region_edge_histogram_counts(img, labels, r, threshold=11):
    x0, y0, bw, bh = region_bbox(labels, r)
    gray = bgr2gray_u8(img[y0:y0 + bh, x0:x0 + bw]).astype(np.int64)
    m = labels[y0:y0 + bh, x0:x0 + bw] == r
    bs = ehd_block_size(bh, bw)
    hb = bs // 2
    q = hb * hb
    counts = np.zeros(80, dtype=np.int32)
    nblk = np.zeros(16, dtype=np.int32)
    for j in range(0, bh - bs + 1, bs):
        for i in range(0, bw - bs + 1, bs):
            if not m[j:j + bs, i:i + bs].all():
                continue
            sub = (i * 4) // bw + ((j * 4) // bh) * 4
            nblk[sub] += 1
            s1 = int(gray[j:j + hb, i:i + hb].sum())
            s2 = int(gray[j:j + hb, i + hb:i + bs].sum())
            s3 = int(gray[j + hb:j + bs, i:i + hb].sum())
            s4 = int(gray[j + hb:j + bs, i + hb:i + bs].sum())
            sq = [(s1 - s2 + s3 - s4) ** 2, (s1 + s2 - s3 - s4) ** 2,
                  2 * (s1 - s4) ** 2, 2 * (s2 - s3) ** 2, 4 * (s1 - s2 - s3 + s4) ** 2]
            best, e = sq[0], 0
            for t in range(1, 5):
                if sq[t] > best:
                    best, e = sq[t], t
            if best >= (threshold * q) ** 2:
                counts[sub * 5 + e] += 1
    return counts, nblk


def region_colour_structure_counts(img, labels, r, nbins=256):
    x0, y0, bw, bh = region_bbox(labels, r)
    K = csd_subsample(bh, bw)
    nwx = (bw - 8 * K) // K + 1 if bw >= 8 * K else 0
    nwy = (bh - 8 * K) // K + 1 if bh >= 8 * K else 0
    hist = np.zeros(nbins, dtype=np.int32)
    if nwx <= 0 or nwy <= 0:
        return hist, 0
    crop = img[y0:y0 + bh, x0:x0 + bw][0:(nwy + 7) * K:K, 0:(nwx + 7) * K:K]
    m = (labels[y0:y0 + bh, x0:x0 + bw] == r)[0:(nwy + 7) * K:K, 0:(nwx + 7) * K:K]
    qimg = hmmd_quantise(crop, nbins).astype(np.int64)
    qimg[~m] = -1
    anyin = np.zeros((nwy, nwx), dtype=bool)
    for c in np.unique(qimg[qimg >= 0]):
        mc = qimg == c
        acc = np.zeros((nwy, nwx), dtype=bool)
        for dy in range(8):
            for dx in range(8):
                acc |= mc[dy:dy + nwy, dx:dx + nwx]
        hist[int(c)] = int(acc.sum())
        anyin |= acc
    return hist, int(anyin.sum())
