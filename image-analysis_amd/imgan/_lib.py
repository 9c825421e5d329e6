"""ctypes binding of libimgan_hip.so (the C ABI declared in include/imgan.h).

The library is the product: there is no CPU fallback.  If the shared object is
missing or a symbol cannot be bound, importing this module raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
import sys

# IMGAN_LIB: an experiment build of the same library (tools/exp_run.sh); never a fallback.  It
# replaces the product library for every importer, so it is announced on stderr, and bench.py
# records the path it ran with (config.library).
PRODUCT_LIB = os.path.join(_HERE, "libimgan_hip.so")
LIB_PATH = os.environ.get("IMGAN_LIB") or PRODUCT_LIB
if LIB_PATH != PRODUCT_LIB:
    sys.stderr.write("imgan: IMGAN_LIB is set -- running the experiment build %s, not the product "
                     "library\n" % LIB_PATH)

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_i64 = ctypes.c_int64
c_size_t = ctypes.c_size_t
c_double = ctypes.c_double

# name -> (restype, argtypes); mirrors include/imgan.h one to one
SIGNATURES = {
    "imgan_last_error": (ctypes.c_char_p, []),
    "imgan_version": (c_int, []),
    "imgan_lab_tables_host": (c_int, [c_void_p, c_void_p, c_void_p]),
    "imgan_bgr2lab_u8": (c_int, [c_void_p, c_void_p, c_i64, c_void_p]),
    "imgan_bgr2gray_u8": (c_int, [c_void_p, c_void_p, c_i64, c_void_p]),
    "imgan_median7_u8c3": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "imgan_median_u8c3": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "imgan_resize_linear_u8c3": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "imgan_slic_grid_host": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "imgan_slic_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "imgan_slic_init": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                c_void_p, c_void_p, c_void_p]),
    "imgan_slic_bin_centres": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                       c_double, c_void_p, c_void_p]),
    "imgan_slic_assign_accumulate": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                             c_double, c_void_p, c_void_p]),
    "imgan_slic_assign_accumulate_bbox": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                                  c_int, c_double, c_void_p, c_void_p, c_void_p]),
    "imgan_slic_seed_centres": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                        c_void_p]),
    "imgan_slic_assign_accumulate_first": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                                   c_int, c_double, c_void_p, c_void_p, c_void_p]),
    "imgan_slic_finalise_centres": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                            c_void_p]),
    "imgan_slic_label_counts": (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                        c_void_p]),
    "imgan_slic_label_sums": (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p,
                                      c_void_p]),
    "imgan_slic_run": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_double, c_int,
                               c_void_p, c_void_p, c_void_p, c_void_p]),
    "imgan_rag_accumulate": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                     c_void_p, c_i64, c_void_p, c_void_p]),
    "imgan_rag_merge_host_batch": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_double,
                                           c_void_p, c_int]),
    "imgan_rag_merge_host": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_i64, c_double,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p]),
    "imgan_rag_merge_device": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_int, c_int, c_double,
                                       c_void_p, c_void_p, c_void_p]),
    "imgan_relabel_lut": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_i64, c_int, c_void_p]),
    "imgan_paint_means": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_i64, c_int, c_void_p]),
    "imgan_cc_label": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "imgan_enforce_connectivity_scratch_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "imgan_enforce_connectivity": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                           c_void_p]),
    "imgan_dcd_state_bytes": (c_size_t, [c_int, c_int, c_int]),
    "imgan_dcd_kmeans_round": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                       ctypes.c_uint32, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                       c_void_p]),
    "imgan_dcd_rounds": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "imgan_dcd_init_hash": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                    ctypes.c_uint32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "imgan_dcd_bbox_bytes": (c_size_t, [c_int, c_int]),
    "imgan_dcd_segments": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                   ctypes.c_uint32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_int, c_void_p]),
    "imgan_dcd_segments_boxed": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                         ctypes.c_uint32, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_int, c_void_p]),
    "imgan_dcd_list_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "imgan_dcd_segments_listed": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                          ctypes.c_uint32, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "imgan_label_counts": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "imgan_dcd_features": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "imgan_dcd_stats": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                c_void_p, c_void_p, c_void_p, c_void_p]),
    "imgan_cld": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "imgan_cld_scratch_bytes": (c_size_t, [c_int]),
    "imgan_cld_static": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "imgan_cld_static_prepare": (c_int, []),
    "imgan_ehd": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "imgan_ehd_scratch_bytes": (c_size_t, [c_int, c_int, c_int]),
    "imgan_cld_ehd": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "imgan_csd": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "imgan_hmmd_quantise": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_void_p]),
    "imgan_region_descriptors_scratch_bytes": (c_size_t, [c_int, c_int]),
    "imgan_region_descriptors": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "imgan_region_descriptors_listed": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                                c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                c_void_p]),
    "imgan_region_descriptors_large_scratch_bytes": (c_size_t, [c_int, c_int]),
    "imgan_region_descriptors_large": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                               c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_void_p]),
    "imgan_region_shape_scratch_bytes": (c_size_t, [c_int, c_int, c_int]),
    "imgan_region_shape": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                   c_double, c_void_p, c_void_p, c_void_p, c_void_p]),
    "imgan_texture_scratch_bytes": (c_size_t, [c_int, c_int, c_int]),
    "imgan_homogeneous_texture": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p]),
    "imgan_label_bbox": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "imgan_spatial_sectors": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int,
                                      c_int, c_void_p, c_void_p, c_void_p]),
}

E_INVALID, E_HIP, E_RANGE, E_NOMEM = -1, -2, -3, -4


class ImganError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libimgan_hip.so not found at %s -- build it with image-analysis_amd/csrc/build.sh "
            "(there is no CPU fallback)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def last_error():
    msg = lib.imgan_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc):
    """Translate a C-ABI return code into the exception the reference would raise."""
    if rc == 0:
        return
    msg = last_error()
    if rc == E_INVALID:
        if "out of bounds" in msg:
            raise IndexError(msg)
        raise ValueError(msg)
    if rc == E_RANGE:
        raise OverflowError(msg)
    if rc == E_NOMEM:
        raise MemoryError(msg)
    raise ImganError(msg)
