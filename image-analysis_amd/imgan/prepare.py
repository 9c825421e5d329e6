"""The image preparation the reference's callers run before the hot path
(optigen_utils.prepare_image :19-24, basegen.py:28-30): resize to 320 pixels wide with
cv2.resize's default bilinear filter, here on the GPU (imgan_resize_linear_u8c3)."""
import numpy as np
import torch

from . import ops


def resize(img, dsize):
    """cv2.resize(img, dsize) for an H x W x 3 uint8 ndarray; dsize = (width, height)."""
    if not isinstance(img, np.ndarray):
        raise ValueError("Image value is None. Probably not loaded correctly!")
    dev = torch.from_numpy(np.ascontiguousarray(img)).cuda()
    return ops.resize_linear(dev, dsize).cpu().numpy()


def prepare_image(img, default_wres=320):
    """optigen_utils.py:19-24."""
    if not isinstance(img, np.ndarray):
        raise ValueError("Image value is None. Probably not loaded correctly!")
    im_res = img.shape[:-1]
    factor = im_res[0] / float(im_res[1])
    return resize(img, (default_wres, int(default_wres * factor)))
