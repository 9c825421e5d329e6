"""imgan_cld_static (the entry point that allocates nothing, include/imgan.h) against imgan_cld,
through the C ABI: bit exact, on two calls in a row, the second with a smaller input (the static
scratch keeps the first call's contents and must be cleared where it starts from zero)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cld_pair(lib, check, bgr):
    B, H, W, _ = bgr.shape
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ref = torch.full((B, 3, 64), -7, dtype=torch.int32, device="cuda")
    out = torch.full((B, 3, 64), -9, dtype=torch.int32, device="cuda")
    ws = torch.empty(lib.imgan_cld_scratch_bytes(B), dtype=torch.uint8, device="cuda")
    check(lib.imgan_cld(bgr.data_ptr(), B, H, W, ref.data_ptr(), ws.data_ptr(), st))
    check(lib.imgan_cld_static(bgr.data_ptr(), B, H, W, out.data_ptr(), st))
    torch.cuda.synchronize()
    return ref.cpu().numpy(), out.cpu().numpy()


def test_cld_static_matches_cld_twice():
    from imgan._lib import lib, check
    g = torch.Generator(device="cuda").manual_seed(77)
    big = torch.randint(0, 256, (5, 270, 484, 3), dtype=torch.uint8, device="cuda", generator=g)
    small = torch.randint(0, 256, (2, 61, 93, 3), dtype=torch.uint8, device="cuda", generator=g)
    for bgr in (big, small):                     # second call: fewer images, smaller, W % 4 != 0
        ref, out = _cld_pair(lib, check, bgr)
        assert np.array_equal(ref, out)
        assert (ref[:, :, 0] > 0).all()          # a real descriptor, not an untouched buffer


def test_cld_static_rejects_oversize_batch():
    from imgan._lib import lib, last_error
    bgr = torch.zeros((1, 8, 8, 3), dtype=torch.uint8, device="cuda")
    out = torch.zeros((1, 3, 64), dtype=torch.int32, device="cuda")
    rc = lib.imgan_cld_static(bgr.data_ptr(), 1025, 8, 8, out.data_ptr(), None)
    assert rc == -3 and "1024" in last_error()


def test_cld_static_prepare_is_idempotent():
    from imgan._lib import lib, check
    check(lib.imgan_cld_static_prepare())
    check(lib.imgan_cld_static_prepare())
    bgr = torch.randint(0, 256, (1, 40, 56, 3), dtype=torch.uint8, device="cuda")
    ref, out = _cld_pair(lib, check, bgr)
    assert np.array_equal(ref, out)
