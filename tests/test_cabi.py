"""The C-ABI library loads on a CPU-only box and exports every symbol include/mdx.h declares.
(No compute calls here: they need a GPU and live in the -m gpu tests.)"""
import os
import re

import pytest

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "mdx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mdx_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_path():
    names = _declared()
    for must in ("mdx_pool_l2n", "mdx_ms_aggregate", "mdx_index_create", "mdx_scores", "mdx_rank_full",
                 "mdx_topk", "mdx_rank_of", "mdx_rank_positions", "mdx_rank_count", "mdx_l2n_rows"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from mdir_amd import _lib
    _lib.build()
    handle = _lib.lib()
    for name in _declared():
        assert hasattr(handle, name), name
    assert set(_declared()) == set(_lib.EXPORTS)
    assert handle.mdx_abi_version() == _lib.ABI_VERSION == 3
    # pure host-side helpers may be called without a GPU
    assert handle.mdx_scores_workspace(70, 2048) == 80 * 2048 * 4
    assert handle.mdx_scores_workspace(0, 2048) == 0
    assert handle.mdx_rank_workspace(1004993, 70) > 16 * 1004993 * 70


def test_argument_errors_do_not_need_a_gpu():
    import ctypes
    from mdir_amd import _lib
    h = _lib.lib()
    rc = h.mdx_pool_l2n(None, 1, 1, 1, 1, 0, 3.0, 1e-6, 1e-6, None, None)
    assert rc == -1 and b"NULL" in h.mdx_last_error()
    with pytest.raises(ValueError):
        _lib.check(rc, "mdx_pool_l2n")
    out = ctypes.c_void_p()
    assert h.mdx_index_create(ctypes.byref(out), ctypes.c_void_p(16), -5, 8, 0, 0, None) == -1


def test_round3_entry_points_check_their_arguments():
    """The entry points added in round 3 refuse bad arguments before touching any device (no GPU here)."""
    import ctypes
    from mdir_amd import _lib
    h = _lib.lib()
    assert h.mdx_gram_f64(None, 4, 4, None, None, None, 0, None) == -1 and b"NULL" in h.mdx_last_error()
    assert h.mdx_gram_f64(ctypes.c_void_p(16), 4, 4, None, ctypes.c_void_p(16), None, 0, None) == -4      # workspace too small
    assert h.mdx_gram_f64_workspace(2048, 20000) >= 2048 * 20000 * 8 and h.mdx_gram_f64_workspace(0, 5) == 0
    assert h.mdx_project_f64(ctypes.c_void_p(16), 4, 4, ctypes.c_void_p(16), 4, None, ctypes.c_void_p(16), None, 0, None) == -4
    assert h.mdx_conv1x1_bn_act(ctypes.c_void_p(16), ctypes.c_void_p(16), 1, 24, 64, 9, None, None, None, None, 1e-5, None, 1,
                                ctypes.c_void_p(16), None) == -1 and b"Cin" in h.mdx_last_error()        # Cin % 16
    assert h.mdx_conv1x1_bn_act(ctypes.c_void_p(16), ctypes.c_void_p(16), 1, 32, 64, 9, ctypes.c_void_p(16), None, None, None, 1e-5,
                                None, 1, ctypes.c_void_p(16), None) == -1                                 # mean without var
    mean = (ctypes.c_float * 3)(0, 0, 0)
    assert h.mdx_clahe_u8_to_chw(ctypes.c_void_p(16), 1, 8, 8, 4, 0, 8, mean, mean, ctypes.c_void_p(16), 1 << 20, ctypes.c_void_p(16), None) == -1
    assert h.mdx_clahe_u8_to_chw(ctypes.c_void_p(16), 1, 8, 8, 4, 8, 8, mean, mean, ctypes.c_void_p(16), 8, ctypes.c_void_p(16), None) == -4
    assert h.mdx_clahe_workspace(2, 10, 10, 8, 8) == 2 * 256 + 2 * 64 * 256 + 8 * 2 * 10 * 10      # L8, LUTs, equalised L8, chroma (a, b)
    lo, hi = ctypes.c_int64(), ctypes.c_int64()
    assert h.mdx_query_bounds(70, 8, 3, ctypes.byref(lo), ctypes.byref(hi)) == 0 and (lo.value, hi.value) == (27, 36)
    assert h.mdx_query_bounds(70, 8, 8, ctypes.byref(lo), ctypes.byref(hi)) == -1
    assert h.mdx_comm_init(None, None, 1, 0) == -1
    assert h.mdx_exchange_scores(None, None, 1, None, None, None) == -1 and h.mdx_comm_destroy(None) == 0


def test_round4_entry_points_check_their_arguments():
    """mdx_scores_ex / mdx_scores_workspace_ex (MDX_F32_SPLIT3) refuse bad arguments before touching a device."""
    import ctypes
    from mdir_amd import _lib
    h = _lib.lib()
    assert h.mdx_scores_workspace_ex(70, 2048, _lib.MDX_F32_SPLIT3) == 80 * 2048 * 6           # three bf16 pieces per padded element
    assert h.mdx_scores_workspace_ex(70, 2048, _lib.MDX_F32_CHAIN) == h.mdx_scores_workspace(70, 2048)
    assert h.mdx_scores_workspace_ex(0, 2048, _lib.MDX_F32_SPLIT3) == 0
    assert h.mdx_scores_ex(None, None, 1, 0, None, None, None, 0, _lib.MDX_F32_SPLIT3, None) == -1 and b"NULL" in h.mdx_last_error()
    assert h.mdx_scores_ex(ctypes.c_void_p(16), ctypes.c_void_p(16), 1, 0, None, ctypes.c_void_p(16), None, 0, 7, None) == -1
    assert b"compute mode" in h.mdx_last_error()


def test_jpeg_pixels_refuses_a_geometry_the_probe_would_not_report():
    """mdx_jpeg_pixels indexes its buffers by the caller's mdx_jpeg_info: anything but the probe's own arithmetic is refused
    before a kernel is launched."""
    import ctypes
    from mdir_amd import _lib
    h = _lib.lib()
    fake = ctypes.c_void_p(16)

    def info(**kw):
        i = _lib.JpegInfo()
        i.width, i.height, i.ncomp, i.supported = 64, 48, 3, 1
        i.hsamp[0], i.vsamp[0] = 2, 2
        for c in (1, 2):
            i.hsamp[c] = i.vsamp[c] = 1
        i.blocks_w[0], i.blocks_h[0], i.blocks_w[1], i.blocks_h[1], i.blocks_w[2], i.blocks_h[2] = 8, 6, 4, 3, 4, 3
        i.block_offset[0], i.block_offset[1], i.block_offset[2], i.nblocks = 0, 48, 60, 72
        for k, v in kw.items():
            if isinstance(v, tuple):
                getattr(i, k)[v[0]] = v[1]
            else:
                setattr(i, k, v)
        return i
    for bad in (dict(nblocks=73), dict(blocks_w=(0, 9)), dict(blocks_h=(2, 4)), dict(block_offset=(1, 47)), dict(hsamp=(1, 2)), dict(vsamp=(0, 4)),
                dict(ncomp=2), dict(width=70000), dict(height=-1), dict(width=65), dict(supported=0)):
        i = info(**bad)
        assert h.mdx_jpeg_pixels(fake, fake, ctypes.byref(i), fake, fake, None) == -1, bad
    assert b"geometry" in h.mdx_last_error() or b"unsupported" in h.mdx_last_error()


def test_ops_refuse_cpu_tensors():
    import torch
    from mdir_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.pool_l2n(torch.zeros(1, 2, 3, 3), "gem")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.rank_full(torch.zeros(2, 5))


def test_late_round4_entry_points_check_their_arguments():
    """mdx_scores_rowmajor, mdx_index_bytes / mdx_index_create_in and mdx_rank_positions refuse what they cannot do before
    touching a device."""
    import ctypes
    from mdir_amd import _lib
    h = _lib.lib()
    P = ctypes.c_void_p
    ws = h.mdx_scores_workspace(70, 2048)
    assert h.mdx_scores_rowmajor(None, 10, 8, P(16), 1, 1, None, P(16), P(16), ws, None) == -1 and b"NULL" in h.mdx_last_error()
    assert h.mdx_scores_rowmajor(P(16), 10, 30, P(16), 1, 1, None, P(16), P(16), ws, None) == -1 and b"multiple of 4" in h.mdx_last_error()
    assert h.mdx_scores_rowmajor(P(18), 10, 32, P(16), 1, 1, None, P(16), P(16), ws, None) == -1           # not 4-byte aligned
    assert h.mdx_scores_rowmajor(P(16), 10, 32, P(16), 70, 1, None, P(16), P(16), 8, None) == -4            # workspace too small
    # an fp32 shard of 1 004 993 x 2048: 62 816 row tiles x 128 k-blocks of 1 KiB + the cell of its maximum
    assert h.mdx_index_bytes(1004993, 2048, 0) == 62816 * 128 * 1024 + 256
    assert h.mdx_index_bytes(1004993, 2048, 1) == 62816 * 64 * 1024 + 256 and h.mdx_index_bytes(0, 8, 0) == 0 and h.mdx_index_bytes(8, 8, 7) == 0
    out = P()
    assert h.mdx_index_create_in(ctypes.byref(out), P(16), 100, 64, 1, 0, 0, P(256), 1024, None) == -4 and b"mdx_index_bytes" in h.mdx_last_error()
    assert h.mdx_index_create_in(ctypes.byref(out), P(16), 100, 64, 1, 0, 0, P(264), 1 << 30, None) == -4   # not at a 256-byte boundary
    assert h.mdx_rank_positions(None, 10, 1, 10, P(16), P(16), 1, P(16), None) == -1
    assert h.mdx_rank_positions(P(16), 10, 1, 5, P(16), P(16), 1, P(16), None) == -1                          # row stride < n
    assert h.mdx_rank_positions(P(16), 10, 1, 10, P(16), P(16), 0, P(16), None) == 0                          # nothing to look up


def test_round6_entry_points_check_their_arguments():
    """The direct-store exchange (mdx_p2p_* / mdx_scores_p2p) refuses bad arguments before touching a device."""
    import ctypes
    from mdir_amd import _lib
    h = _lib.lib()
    out, buf = ctypes.c_void_p(), (ctypes.c_char * 64)()
    assert h.mdx_p2p_create(None, 2, 0, 70, 1000, buf) == -1 and b"NULL" in h.mdx_last_error()
    assert h.mdx_p2p_create(ctypes.byref(out), 2, 2, 70, 1000, buf) == -1 and out.value is None          # rank out of range
    assert h.mdx_p2p_create(ctypes.byref(out), 65, 0, 70, 1000, buf) == -1                               # more than 64 ranks
    assert h.mdx_p2p_create(ctypes.byref(out), 8, 0, 129, 1000, buf) == -1 and b"nq" in h.mdx_last_error()     # at most 128 queries per step
    assert h.mdx_p2p_create(ctypes.byref(out), 8, 0, 70, 0, buf) == -1
    assert h.mdx_p2p_connect(None, buf) == -1 and h.mdx_p2p_connect_ptrs(None, None) == -1
    assert h.mdx_scores_p2p(None, None, 70, 0, None, None, None, 0, None) == -1
    mine = ctypes.c_void_p()
    assert h.mdx_p2p_close_step(None, ctypes.byref(mine), None) == -1
    assert h.mdx_p2p_status(None, None, None) == -1
    assert h.mdx_p2p_destroy(None) == 0 and h.mdx_p2p_base(None) is None and h.mdx_p2p_bytes(None) == 0
