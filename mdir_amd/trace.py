"""Named ranges for profilers (rocprofv3 --marker-trace) around the stages of the path.

The reference times its stages with wall-clock laps only (``StopWatch``, mdir/tools/stats.py:47-67: the
``extract_descriptors`` / ``compute_score`` laps of cirscore.py).  Those laps are kept (scenario.StopWatch);
with ``MDIR_AMD_ROCTX=1`` the same stage names -- and the similarity / ranking calls inside them -- are also
pushed as roctx ranges, so that a kernel trace can be read stage by stage.  No-ops otherwise.
"""
import contextlib
import ctypes
import os

_lib = None


def _roctx():
    global _lib
    if _lib is None:
        _lib = False
        if os.environ.get("MDIR_AMD_ROCTX") == "1":
            for name in ("libroctx64.so", "librocprofiler-sdk-roctx.so"):
                try:
                    _lib = ctypes.CDLL(name)
                    _lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                    break
                except OSError:
                    _lib = False
    return _lib


@contextlib.contextmanager
def range_(name):
    lib = _roctx()
    if lib:
        lib.roctxRangePushA(name.encode())
    try:
        yield
    finally:
        if lib:
            lib.roctxRangePop()
