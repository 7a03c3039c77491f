import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from mdir_amd import ops
from mdir_amd.sharded import shard_bounds
N, G = 1004993, 8
dev = "cuda:0"
g = torch.Generator(device=dev); g.manual_seed(0)
full = torch.randn((9, N), generator=g, device=dev) * 0.022
blocks, o = [], 0
for r in range(G):
    a, b = shard_bounds(N, G, r)
    blocks.append(full[:, a:b].contiguous())
out = torch.empty((9, N), dtype=torch.int64, device=dev)
ws = torch.empty(ops.rank_workspace_bytes(N, 9), dtype=torch.uint8, device=dev)
for _ in range(20):
    ops.rank_full(full, out=out, workspace=ws)
torch.cuda.synchronize()
for _ in range(20):
    ops.rank_full_segments(blocks, out=out, workspace=ws)
torch.cuda.synchronize()
