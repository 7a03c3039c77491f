"""ctypes binding of libmdx.so (the C ABI declared in include/mdx.h).

There is NO fallback: if the shared library is missing or a call fails, an
exception is raised.  `build()` compiles it in-tree with hipcc for gfx950
(cross-compiles without a GPU); the built file travels with the source tree.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MDIR_AMD_LIB") or os.path.join(_HERE, "libmdx.so")   # override: A/B builds only
CSRC = os.path.join(_HERE, "csrc")

MDX_DIM_MAJOR, MDX_ROW_MAJOR = 0, 1
MDX_POOL_GEM, MDX_POOL_MAC, MDX_POOL_SPOC = 0, 1, 2
MDX_F32, MDX_F16 = 0, 1
MDX_F32_CHAIN, MDX_F32_SPLIT3, MDX_F32_SPLIT2 = 0, 1, 2
COMPUTE = {"chain": MDX_F32_CHAIN, "exact": MDX_F32_CHAIN, "split3": MDX_F32_SPLIT3, "split2": MDX_F32_SPLIT2}
STORAGE = {"f32": MDX_F32, "f16": MDX_F16}
POOL_KINDS = {"gem": MDX_POOL_GEM, "mac": MDX_POOL_MAC, "spoc": MDX_POOL_SPOC}


class MdxError(RuntimeError):
    """A libmdx call returned a negative status."""


_STATUS_EXC = {-1: ValueError, -2: MdxError, -3: MemoryError, -4: ValueError}

_lib = None
ABI_VERSION = 3        # include/mdx.h MDX_ABI_VERSION this binding was written against


def build(force=False):
    """Compile libmdx.so with hipcc --offload-arch=gfx950 (see csrc/Makefile)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(_HERE, "..", "include", "mdx.h"))
    stale = not os.path.exists(LIB_PATH) or \
        any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", CSRC] + (["-B"] if force else []))
    return LIB_PATH


class JpegInfo(ctypes.Structure):
    """``mdx_jpeg_info`` of include/mdx.h."""
    _fields_ = [("width", ctypes.c_int32), ("height", ctypes.c_int32), ("ncomp", ctypes.c_int32),
                ("hsamp", ctypes.c_int32 * 3), ("vsamp", ctypes.c_int32 * 3),
                ("blocks_w", ctypes.c_int32 * 3), ("blocks_h", ctypes.c_int32 * 3), ("supported", ctypes.c_int32),
                ("block_offset", ctypes.c_int64 * 3), ("nblocks", ctypes.c_int64)]


def _declare(lib):
    i32, i64, f32, p = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p
    pp = ctypes.POINTER(ctypes.c_void_p)
    pi64 = ctypes.POINTER(ctypes.c_int64)
    sig = {
        "mdx_abi_version": (i32, []),
        "mdx_last_error": (ctypes.c_char_p, []),
        "mdx_capture_recover": (i32, [p]),
        "mdx_rmac_workspace": (i64, [i32, i32, i32]),
        "mdx_rmac": (i32, [p, i32, i32, i32, i32, ctypes.POINTER(ctypes.c_int32), i32, f32, p, i64, p, p]),
        "mdx_roipool": (i32, [p, i32, i32, i32, i32, ctypes.POINTER(ctypes.c_int32), i32, i32, f32, f32, p, p]),
        "mdx_region_sum": (i32, [p, i32, i32, i32, f32, p, p]),
        "mdx_pool_l2n": (i32, [p, i32, i32, i32, i32, i32, f32, f32, f32, p, p]),
        "mdx_l2n_rows": (i32, [p, i64, i64, p, f32, p]),
        "mdx_ms_aggregate": (i32, [pp, i32, i64, f32, p, p]),
        "mdx_ms_aggregate_batch": (i32, [pp, i32, i64, i64, f32, p, p]),
        "mdx_pool_multi": (i32, [pp, i32, i32, i32, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), i32, f32, f32, p, p]),
        "mdx_l2n_aggregate": (i32, [p, i32, i64, i64, f32, f32, p, p]),
        "mdx_bn_act": (i32, [p, p, i64, i64, i64, p, p, p, p, f32, i32, p]),
        "mdx_u8_to_chw": (i32, [p, i64, i64, i64, i32, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), p, p]),
        "mdx_clahe_workspace": (i64, [i64, i64, i64, i32, i32]),
        "mdx_clahe_u8_to_chw": (i32, [p, i64, i64, i64, i32, i32, i32, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), p, i64, p, p]),
        "mdx_bilinear_pyramid": (i32, [p, i64, i64, i32, i32, i32, ctypes.POINTER(ctypes.c_double), pp, p]),
        "mdx_resample_u8": (i32, [p, i64, i32, i32, i32, i32, i32, p, p, i32, p, p]),
        "mdx_jpeg_probe": (i32, [p, i64, p]),
        "mdx_jpeg_coefficients": (i32, [p, i64, p, i64, p]),
        "mdx_jpeg_pixels": (i32, [p, p, p, p, p, p]),
        "mdx_index_create": (i32, [pp, p, i64, i64, i32, i64, p]),
        "mdx_index_create_ex": (i32, [pp, p, i64, i64, i32, i64, i32, p]),
        "mdx_index_bytes": (i64, [i64, i64, i32]),
        "mdx_index_create_in": (i32, [pp, p, i64, i64, i32, i64, i32, p, i64, p]),
        "mdx_index_destroy": (i32, [p]),
        "mdx_index_info": (i32, [p, pi64, pi64, pi64, pi64]),
        "mdx_scores_workspace": (i64, [i64, i64]),
        "mdx_scores": (i32, [p, p, i64, i32, p, p, p, i64, p]),
        "mdx_scores_rowmajor": (i32, [p, i64, i64, p, i64, i32, p, p, p, i64, p]),
        "mdx_scores_workspace_ex": (i64, [i64, i64, i32]),
        "mdx_scores_ex": (i32, [p, p, i64, i32, p, p, p, i64, i32, p]),
        "mdx_rank_workspace": (i64, [i64, i64]),
        "mdx_rank_full": (i32, [p, i64, i64, i64, p, p, i64, p]),
        "mdx_rank_full_segments": (i32, [pp, pi64, i32, i64, i64, p, p, i64, p]),
        "mdx_topk": (i32, [p, i64, i64, i64, i64, p, p, p, i64, p]),
        "mdx_rank_of": (i32, [p, i64, i64, p, p, i64, p, p, p]),
        "mdx_rank_positions": (i32, [p, i64, i64, i64, p, p, i64, p, p]),
        "mdx_gather_scores": (i32, [p, i64, i64, p, p, i64, p, p]),
        "mdx_rank_count": (i32, [p, i64, i64, i64, p, p, p, i64, p, p]),
        "mdx_conv1x1_transpose_weights": (i32, [p, i64, i64, p, p]),
        "mdx_conv1x1_bn_act": (i32, [p, p, i64, i64, i64, i64, p, p, p, p, f32, p, i32, p, p]),
        "mdx_gram_f64_workspace": (i64, [i64, i64]),
        "mdx_gram_f64": (i32, [p, i64, i64, p, p, p, i64, p]),
        "mdx_project_f64_workspace": (i64, [i64, i64]),
        "mdx_project_f64": (i32, [p, i64, i64, p, i64, p, p, p, i64, p]),
        "mdx_l2n_cols_f64": (i32, [p, i64, i64, ctypes.c_double, p]),
        "mdx_comm_unique_id": (i32, [p]),
        "mdx_comm_init": (i32, [pp, p, i32, i32]),
        "mdx_comm_destroy": (i32, [p]),
        "mdx_comm_info": (i32, [p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
        "mdx_query_bounds": (i32, [i64, i32, i32, pi64, pi64]),
        "mdx_allgather_scores": (i32, [p, p, i64, pi64, p, p]),
        "mdx_exchange_scores": (i32, [p, p, i64, pi64, p, p]),
        "mdx_p2p_create": (i32, [pp, i32, i32, i64, i64, p]),
        "mdx_p2p_connect": (i32, [p, p]),
        "mdx_p2p_connect_ptrs": (i32, [p, pp]),
        "mdx_p2p_base": (p, [p]),
        "mdx_p2p_bytes": (i64, [p]),
        "mdx_scores_p2p": (i32, [p, p, i64, i32, p, p, p, i64, p]),
        "mdx_p2p_close_step": (i32, [p, pp, p]),
        "mdx_p2p_status": (i32, [p, ctypes.POINTER(ctypes.c_uint32), p]),
        "mdx_p2p_destroy": (i32, [p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype, fn.argtypes = res, args
    return sig


EXPORTS = ("mdx_abi_version", "mdx_last_error", "mdx_capture_recover", "mdx_rmac_workspace", "mdx_rmac", "mdx_roipool", "mdx_region_sum", "mdx_pool_l2n", "mdx_l2n_rows", "mdx_ms_aggregate",
           "mdx_ms_aggregate_batch", "mdx_pool_multi", "mdx_l2n_aggregate", "mdx_bn_act", "mdx_u8_to_chw", "mdx_resample_u8", "mdx_bilinear_pyramid", "mdx_jpeg_probe", "mdx_jpeg_coefficients", "mdx_jpeg_pixels",
           "mdx_index_create", "mdx_index_create_ex", "mdx_index_bytes", "mdx_index_create_in", "mdx_index_destroy", "mdx_index_info", "mdx_scores_workspace",
           "mdx_scores", "mdx_scores_rowmajor", "mdx_scores_workspace_ex", "mdx_scores_ex", "mdx_rank_workspace", "mdx_rank_full", "mdx_rank_full_segments", "mdx_topk", "mdx_rank_of", "mdx_rank_positions",
           "mdx_gather_scores", "mdx_rank_count", "mdx_conv1x1_transpose_weights", "mdx_conv1x1_bn_act", "mdx_clahe_workspace", "mdx_clahe_u8_to_chw", "mdx_gram_f64_workspace", "mdx_gram_f64", "mdx_project_f64_workspace", "mdx_project_f64", "mdx_l2n_cols_f64", "mdx_comm_unique_id", "mdx_comm_init",
           "mdx_comm_destroy", "mdx_comm_info", "mdx_query_bounds", "mdx_allgather_scores", "mdx_exchange_scores",
           "mdx_p2p_create", "mdx_p2p_connect", "mdx_p2p_connect_ptrs", "mdx_p2p_base", "mdx_p2p_bytes", "mdx_scores_p2p", "mdx_p2p_close_step",
           "mdx_p2p_status", "mdx_p2p_destroy")


def lib():
    """The loaded library; raises if it was never built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MdxError("libmdx.so is not built (%s); run `python -c 'import __graft_entry__ as g; "
                           "g.build()'` or `make -C mdir_amd/csrc`" % LIB_PATH)
        # torch first: its wheel carries its own libamdhip64, and libmdx.so must bind to THAT copy of the HIP runtime -- loaded
        # before torch it would pull in /opt/rocm's, and a process with two HIP runtimes loses the device ("no ROCm-capable
        # device is detected" at the first launch)
        import torch  # noqa: F401
        handle = ctypes.CDLL(LIB_PATH)
        _declare(handle)
        if handle.mdx_abi_version() != ABI_VERSION:
            raise MdxError("libmdx.so ABI version %d, this package binds version %d (include/mdx.h MDX_ABI_VERSION): rebuild "
                           "with `make -C mdir_amd/csrc`" % (handle.mdx_abi_version(), ABI_VERSION))
        _lib = handle
    return _lib


def check(status, what=""):
    if status != 0:
        msg = lib().mdx_last_error().decode(errors="replace")
        raise _STATUS_EXC.get(status, MdxError)("%s failed (status %d): %s" % (what or "libmdx", status, msg))
