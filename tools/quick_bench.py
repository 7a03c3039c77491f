"""Scratch timing of the two hot kernels (scores + full ranking) on one GPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdir_amd import ops

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1004993
    nq, d = (int(sys.argv[2]) if len(sys.argv) > 2 else 70), 2048
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(0)
    vecs = torch.empty((d, n), dtype=torch.float32, device=dev)
    for s in range(0, n, 65536):
        e = min(n, s + 65536)
        blk = torch.randn((e - s, d), generator=g, device=dev)
        blk /= blk.norm(dim=1, keepdim=True)
        vecs[:, s:e] = blk.t()
    q = vecs[:, torch.randperm(n, device=dev)[:nq]].contiguous()
    t0 = time.time(); ix = ops.DescriptorIndex(vecs, "DN"); torch.cuda.synchronize(); print("index build s", time.time() - t0, ix.device_bytes / 1e9, "GB")
    del vecs
    sc = torch.empty((nq, n), dtype=torch.float32, device=dev)
    ws = torch.empty(ops.rank_workspace_bytes(n, nq), dtype=torch.uint8, device=dev)
    rk = torch.empty((nq, n), dtype=torch.int64, device=dev)
    for name, fn in (("scores", lambda: ix.scores(q, "DN", out=sc)), ("rank_full", lambda: ops.rank_full(sc, out=rk, workspace=ws))):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        reps = 10
        ev[0].record()
        for _ in range(reps): fn()
        ev[1].record(); torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / reps
        print("%-10s %.3f ms" % (name, ms), flush=True)
        if name == "scores":
            print("   TFLOP/s %.1f  (of 157.3)  GB/s %.0f" % (2 * nq * n * d / ms / 1e9, (4 * n * d + 4 * nq * n) / ms / 1e6))
    # sanity: sortedness
    top = rk[:, :5].cpu(); print(top[0].tolist(), sc[0, top[0].to(dev)].tolist())
    chk = torch.gather(sc, 1, rk); print("sorted desc:", bool((chk[:, 1:] <= chk[:, :-1]).all()))

main()
