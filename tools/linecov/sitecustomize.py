"""Line coverage of the host surface without the `coverage` package (absent from the image): put this directory on
PYTHONPATH and set MDIR_AMD_LINECOV_DIR; every Python process (pytest, the eval.py / bench.py children it starts, torchrun
ranks) then records which lines of mdir_amd/*.py and eval.py ran and writes cov_<pid>.json there at exit.
tools/linecov_report.py merges the files into profiles/r05_host_branches.md.

    MDIR_AMD_LINECOV_DIR=/tmp/cov PYTHONPATH=tools/linecov python -m pytest tests -m "not gpu" -q
"""
import atexit
import json
import os
import sys
import threading

_DIR = os.environ.get("MDIR_AMD_LINECOV_DIR")
if _DIR:
    _ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    _PKG = os.path.join(_ROOT, "mdir_amd") + os.sep
    _EVAL = os.path.join(_ROOT, "eval.py")
    _seen = {}

    def _local(frame, event, arg):
        if event == "line":
            _seen[frame.f_code.co_filename].add(frame.f_lineno)
        return _local

    def _global(frame, event, arg):
        name = frame.f_code.co_filename
        lines = _seen.get(name)
        if lines is None:
            if not (name.startswith(_PKG) or name == _EVAL):
                return None
            lines = _seen[name] = set()
        lines.add(frame.f_lineno)
        return _local

    def _dump():
        sys.settrace(None)
        os.makedirs(_DIR, exist_ok=True)
        with open(os.path.join(_DIR, "cov_%d.json" % os.getpid()), "w") as f:
            json.dump({os.path.relpath(k, _ROOT): sorted(v) for k, v in _seen.items()}, f)

    threading.settrace(_global)
    sys.settrace(_global)
    atexit.register(_dump)
