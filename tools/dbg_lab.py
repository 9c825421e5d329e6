import sys, numpy as np, torch
sys.path.insert(0, 'image-analysis_amd'); sys.path.insert(0, '.')
from imgan import ops
from oracle.color import bgr2lab_u8, bgr2gray_u8
rng = np.random.RandomState(7)
img = rng.randint(0, 256, (1, 64, 96, 3)).astype(np.uint8)
lab = ops.bgr2lab(torch.from_numpy(img).cuda()).cpu().numpy()
ref = bgr2lab_u8(img)
bad = np.argwhere(lab != ref)
print(len(bad), 'mismatches of', lab.size)
print('by channel', np.bincount(bad[:, 3], minlength=3), 'by x%4', np.bincount(bad[:, 2] % 4, minlength=4))
for b in bad[:10]:
    print(b, img[tuple(b[:3])], lab[tuple(b[:3])], ref[tuple(b[:3])])
g = ops.bgr2gray(torch.from_numpy(img).cuda()).cpu().numpy()
print('gray mismatches', (g != bgr2gray_u8(img)).sum())
