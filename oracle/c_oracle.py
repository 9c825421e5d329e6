"""TEST INFRASTRUCTURE -- ctypes access to the C restatement (oracle/c/imgan_oracle.c): a second,
independent statement of the SLIC and dominant-colour loops, and the multi-core CPU baseline of
bench.py.  Built by oracle/c/build.sh (gcc) into oracle/_build/; never loaded by the product."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "liboracle_c.so")
_lib = None


def build():
    subprocess.check_call(["bash", os.path.join(_HERE, "c", "build.sh")], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "c", "imgan_oracle.c")
        if not os.path.exists(LIB_PATH) or (os.path.exists(src) and
                                            os.path.getmtime(src) > os.path.getmtime(LIB_PATH)):
            build()
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.oracle_path_batch.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_double] + \
            [ctypes.c_int] * 3 + [ctypes.c_void_p, ctypes.c_void_p]
        _lib.oracle_slic_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_double, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        _lib.oracle_dcd_segment.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                            ctypes.c_int] + [ctypes.c_void_p] * 4
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def slic_run(lab_u8, K, colour_weight, num_iter):
    """-> (labels int32 [H, W], centres int32 [n, 5]) as SLICOracle(lab=..., mode="float64").run()."""
    L = lib()
    lab_u8 = np.ascontiguousarray(lab_u8, dtype=np.uint8)
    H, W = lab_u8.shape[:2]
    S, nx, ny = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    if L.oracle_slic_grid(H, W, int(K), ctypes.byref(S), ctypes.byref(nx), ctypes.byref(ny)):
        raise ValueError("bad SLIC shape")
    n = nx.value * ny.value
    labels = np.empty((H, W), np.int32)
    centres = np.zeros((max(n, 1), 5), np.int32)
    rc = L.oracle_slic_run(_p(lab_u8), H, W, int(K), float(colour_weight), int(num_iter), _p(labels),
                           _p(centres))
    if rc < 0:
        raise RuntimeError("oracle_slic_run failed (%d)" % rc)
    return labels, centres[:n]


def dcd_segment(points_u8, seg_idxs, init_idx, max_rounds=1000, resample_idx=None):
    """-> (assign int32 [N], centres float32 [k, 3], stats uint64 [k, 8], rounds).
    resample_idx: the point each cluster takes when it is empty (None: it keeps its centre)."""
    L = lib()
    pts = np.ascontiguousarray(points_u8, dtype=np.uint8)
    rcs = np.ascontiguousarray(seg_idxs, dtype=np.int32)
    init = np.ascontiguousarray(init_idx, dtype=np.int32)
    N, k = pts.shape[0], init.shape[0]
    assign = np.empty(N, np.int32)
    centres = np.empty((k, 3), np.float32)
    stats = np.empty((k, 8), np.uint64)
    rounds = ctypes.c_int32()
    res = None if resample_idx is None else np.ascontiguousarray(resample_idx, dtype=np.int32)
    rc = L.oracle_dcd_segment(_p(pts), _p(rcs), N, _p(init), None if res is None else _p(res), k,
                              int(max_rounds), _p(assign),
                              _p(centres), _p(stats), ctypes.byref(rounds))
    if rc:
        raise RuntimeError("oracle_dcd_segment failed (%d)" % rc)
    return assign, centres, stats, rounds.value


def path_batch(lab_u8_batch, K, colour_weight, num_iter, k, threads):
    """SLIC + per-superpixel dominant colours of B images, one image per thread.
    -> (labels int32 [B, H, W], threads used, pixel-rounds of the Lloyd loops)."""
    L = lib()
    lab = np.ascontiguousarray(lab_u8_batch, dtype=np.uint8)
    B, H, W = lab.shape[:3]
    labels = np.empty((B, H, W), np.int32)
    rt = ctypes.c_int64()
    used = L.oracle_path_batch(_p(lab), B, H, W, int(K), float(colour_weight), int(num_iter), int(k),
                               int(threads), _p(labels), ctypes.byref(rt))
    if used < 0:
        raise RuntimeError("oracle_path_batch failed (%d)" % used)
    return labels, used, rt.value
