"""ctypes front-end of oracle/chain.c (TEST INFRASTRUCTURE ONLY, see oracle.py).

``build()`` compiles ``liboracle_chain.so`` with gcc through ``oracle/Makefile``;
the loader builds on first use if the file is absent (the GPU box receives the
prebuilt .so with the snapshot).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_chain.so")
_lib = None


def build(force=False):
    if force or not os.path.exists(_SO) or \
            os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "chain.c")):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle_chain.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        i64, p = ctypes.c_int64, ctypes.c_void_p
        _lib.oracle_scores_chain.argtypes = [p, p, i64, i64, i64, p]
        _lib.oracle_gemm_nt_chain.argtypes = [p, p, i64, i64, i64, p]
        _lib.oracle_rank_full.argtypes = [p, i64, i64, p]
        _lib.oracle_rank_of.argtypes = [p, i64, p, i64, p]
        _lib.oracle_desc_key.argtypes = [ctypes.c_float]
        _lib.oracle_desc_key.restype = ctypes.c_uint32
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def scores_chain(vecs, qvecs):
    """``[D,N]`` x ``[D,Q]`` -> ``[Q,N]`` fp32, k-ascending fmaf chain."""
    vecs = np.ascontiguousarray(vecs, dtype=np.float32)
    qvecs = np.ascontiguousarray(qvecs, dtype=np.float32)
    d, n = vecs.shape
    assert qvecs.shape[0] == d
    nq = qvecs.shape[1]
    out = np.empty((nq, n), dtype=np.float32)
    lib().oracle_scores_chain(_ptr(vecs), _ptr(qvecs), n, d, nq, _ptr(out))
    return out


def gemm_nt_chain(a, b):
    """``a [m,d]``, ``b [n,d]`` -> ``[m,n]`` with the same chain per element."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    out = np.empty((a.shape[0], b.shape[0]), dtype=np.float32)
    lib().oracle_gemm_nt_chain(_ptr(a), _ptr(b), a.shape[0], b.shape[0], a.shape[1], _ptr(out))
    return out


def rank_full(sc_qn):
    """``[Q,N]`` fp32 -> ``[Q,N]`` int64: descending score, ascending id on ties."""
    sc_qn = np.ascontiguousarray(sc_qn, dtype=np.float32)
    out = np.empty(sc_qn.shape, dtype=np.int64)
    lib().oracle_rank_full(_ptr(sc_qn), sc_qn.shape[1], sc_qn.shape[0], _ptr(out))
    return out


def rank_of(sc_row, ids):
    sc_row = np.ascontiguousarray(sc_row, dtype=np.float32)
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    out = np.empty(ids.shape, dtype=np.int64)
    lib().oracle_rank_of(_ptr(sc_row), sc_row.shape[0], _ptr(ids), ids.shape[0], _ptr(out))
    return out


def desc_key(x):
    return lib().oracle_desc_key(float(x))
