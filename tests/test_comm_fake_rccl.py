"""The multi-rank logic of the C-ABI exchange (mdx_exchange_scores / mdx_allgather_scores: offsets, counts, query split,
even and uneven shards) on a CPU box: tests/fake_rccl.c stands in for librccl.so.1 with host buffers and the ranks as
threads.  What this cannot cover is RCCL itself; the 1-rank communicator on real RCCL is in tests/test_gpu_round3.py.
Runs in a subprocess: the stand-in must be in the process before libmdx.so first looks for RCCL."""
import os
import subprocess
import sys

from conftest import ROOT

_SCRIPT = r"""
import ctypes, os, subprocess, sys, threading
import numpy as np
root, tmp = sys.argv[1], sys.argv[2]
fake = os.path.join(tmp, "librccl.so.1")
subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-pthread", "-Wl,-soname,librccl.so.1", "-o", fake, os.path.join(root, "tests", "fake_rccl.c")])
ctypes.CDLL(fake, mode=ctypes.RTLD_GLOBAL)
sys.path.insert(0, root)
from mdir_amd import _lib
lib = _lib.lib()
i64 = ctypes.c_int64

# the FIRST communicator calls of the process, from 8 threads at once (ranks-as-threads hosts): RCCL is resolved under
# std::call_once, so every one of them finds the whole symbol table (a plain `static bool tried` let a second thread in
# while the first was still filling it: NULL calls, or "not found" with no message)
gate, first = threading.Barrier(8), []


def first_call():
    ident = (ctypes.c_char * 128)()
    gate.wait()
    first.append((lib.mdx_comm_unique_id(ident), bytes(ident)[:4]))


ts = [threading.Thread(target=first_call) for _ in range(8)]
[t.start() for t in ts]
[t.join(60) for t in ts]
assert first == [(0, b"ZZZZ")] * 8, first


def run(G, nq, widths, call):
    rng = np.random.default_rng(G * 100 + nq)
    blocks = [rng.standard_normal((nq, w)).astype(np.float32) for w in widths]       # block g = S_g [nq, w_g]
    ident = (ctypes.c_char * 128)()
    assert lib.mdx_comm_unique_id(ident) == 0
    out, errs = [None] * G, []

    def rank(r):
        try:
            comm = ctypes.c_void_p()
            assert lib.mdx_comm_init(ctypes.byref(comm), ident, G, r) == 0, lib.mdx_last_error()
            w = (i64 * G)(*widths)
            lo, hi = i64(), i64()
            assert lib.mdx_query_bounds(nq, G, r, ctypes.byref(lo), ctypes.byref(hi)) == 0
            rows = nq if call == "allgather" else hi.value - lo.value
            buf = np.full(max(1, rows * sum(widths)), np.nan, dtype=np.float32)
            fn = lib.mdx_allgather_scores if call == "allgather" else lib.mdx_exchange_scores
            rc = fn(comm, blocks[r].ctypes.data, nq, w, buf.ctypes.data, None)
            assert rc == 0, lib.mdx_last_error()
            out[r] = (buf, lo.value, hi.value)
            assert lib.mdx_comm_destroy(comm) == 0
        except BaseException as exc:
            errs.append((r, repr(exc)))

    threads = [threading.Thread(target=rank, args=(r,)) for r in range(G)]
    [t.start() for t in threads]
    [t.join(60) for t in threads]
    assert not errs and not any(t.is_alive() for t in threads), errs
    for r in range(G):
        buf, lo, hi = out[r]
        rows = nq if call == "allgather" else hi - lo
        o = 0
        for g in range(G):                           # blocks back to back in rank order, block g = [rows, widths[g]]
            got = buf[o:o + rows * widths[g]].reshape(rows, widths[g])
            want = blocks[g] if call == "allgather" else blocks[g][lo:hi]
            assert np.array_equal(got, want), (call, G, nq, widths, r, g)
            o += rows * widths[g]


for call in ("exchange", "allgather"):
    run(1, 5, [7], call)
    run(2, 7, [4, 4], call)                          # even shards: ncclAllGather
    run(3, 7, [5, 4, 4], call)                       # shards differ by a row: grouped send / receive
    run(8, 70, [13, 13, 13, 13, 13, 12, 12, 12], call)     # the node's shape: 70 queries over 8 ranks
    run(4, 3, [6, 6, 5, 5], call)                    # fewer queries than ranks: a rank that owns none
    run(3, 5, [4, 0, 3], call)                       # an empty shard
print("FAKE-RCCL-OK")
"""


def test_exchange_logic_on_fake_rccl(tmp_path):
    proc = subprocess.run([sys.executable, "-c", _SCRIPT, ROOT, str(tmp_path)], text=True, capture_output=True, timeout=600)
    assert proc.returncode == 0 and "FAKE-RCCL-OK" in proc.stdout, (proc.stdout[-2000:], proc.stderr[-4000:])
