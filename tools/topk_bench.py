"""top-k timing at 1 M x 70 (sampled-threshold path vs radix-select path: MDX_NO_SAMPLED_TOPK=1)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdir_amd import ops
n, nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1004993, 70
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
sc = torch.randn((nq, n), generator=g, device=dev) * 0.022
ws = torch.empty(ops.rank_workspace_bytes(n, nq), dtype=torch.uint8, device=dev)
for k in (10, 100, 1000):
    for _ in range(3): ops.topk(sc, k, workspace=ws)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): ops.topk(sc, k, workspace=ws)
    b.record(); torch.cuda.synchronize()
    print("n=%d k=%4d: %.3f ms" % (n, k, a.elapsed_time(b) / 10), flush=True)
