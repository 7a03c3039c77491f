"""Full-size ranking on random scores: time per call, look-back status flag, order check."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdir_amd import ops, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1004993
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 70
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
sc = torch.randn((nq, n), generator=g, device=dev) * 0.022
ws = torch.empty(ops.rank_workspace_bytes(n, nq), dtype=torch.uint8, device=dev)
rk = torch.empty((nq, n), dtype=torch.int64, device=dev)
st = 0
for i in range(reps):
    if os.environ.get("DUMP") and i > 0 and st != 0: break
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ops.rank_full(sc, out=rk, workspace=ws)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = _lib.lib().mdx_rank_status(ctypes.c_void_p(ws.data_ptr()), n, nq, None) if hasattr(_lib.lib(), "mdx_rank_status") else 0
    print("call %d: %.3f ms, status %d %s" % (i, dt * 1e3, st, _lib.lib().mdx_last_error().decode() if st else ""), flush=True)
st = st if reps else 0
if st == 0:
  chk = torch.gather(sc, 1, rk)
  print("sorted desc:", bool((chk[:, 1:] <= chk[:, :-1]).all()), " permutation:", bool((torch.sort(rk, dim=1).values[0] == torch.arange(n, device=dev)).all()))
if os.environ.get("DUMP"):
    import numpy as np
    # workspace layout of mdx_rank_onesweep.h: pairs[2] | digit_tot | look | ticket | status
    def up(x): return (x + 255) // 256 * 256
    nblk = (n + 4095) // 4096
    off = 2 * up(n * nq * 8) + up(nq * 4 * 256 * 4)
    look = ws[off:off + nq * nblk * 1024].view(torch.int32).cpu().numpy().astype(np.uint32).reshape(nq, nblk, 256)
    off2 = off + up(nq * nblk * 1024)
    tick = ws[off2:off2 + 128].view(torch.int32).cpu().numpy().reshape(4, 8)
    stat = ws[off2 + 256:off2 + 320].view(torch.int32).cpu().numpy()
    print("tickets per pass x list:\n", tick, "\nstatus", stat)
    state, tag = look >> 30, (look >> 28) & 3
    for p in range(4):
        m = tag == p
        print("pass", p, ": words tagged", int(m.sum()), " prefix", int((m & (state == 2)).sum()), " count-only", int((m & (state == 1)).sum()))
    fq, fj, fb, fd = [int(x) for x in stat[4:8]]
    fp = [i for i in range(4) if stat[i]][0] if any(stat[:4]) else 3
    print("failing pass", fp, "query", fq, ": state of digit", fd, "words around tile", fj)
    for b in range(max(0, fj - 3), min(nblk, fb + 3)):
        print("   tile", b, "state", int(state[fq, b, fd]), "tag", int(tag[fq, b, fd]), "value", int(look[fq, b, fd] & 0xFFFFFFF), " | all digits: states", np.bincount(state[fq, b], minlength=3).tolist(), "tags", np.bincount(tag[fq, b], minlength=4).tolist())
    stale = np.argwhere(tag != fp)
    print("words not tagged with the failing pass:", len(stale), "first", stale[:8].tolist(), "queries", np.unique(stale[:, 0]).tolist(), "tiles", np.unique(stale[:, 1]).tolist()[:20], "digits", np.unique(stale[:, 2]).tolist()[:20])
    # how many tiles of each query of the failing pass were published at all
    pub = ((tag == fp) & (state > 0)).any(axis=2).sum(axis=1)
    print("tiles published per query (pass %d):" % fp, pub.tolist())
