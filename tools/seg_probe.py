"""Per-kernel cost of one rank's ranking at G = 8: 9 queries x 1 004 993 rows, dense and as the 16 peer blocks of the exchange
(8 peers x the two chunks of ``chunk_bounds``).  Run under ``rocprofv3 --kernel-trace --stats``."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from mdir_amd import ops
from mdir_amd.sharded import shard_bounds, chunk_bounds
N, G = 1004993, 8
dev = "cuda:0"
g = torch.Generator(device=dev); g.manual_seed(0)
full = torch.randn((9, N), generator=g, device=dev) * 0.022
blocks = []
for r in range(G):
    for a, b in chunk_bounds(*shard_bounds(N, G, r), 2):
        blocks.append(full[:, a:b].contiguous())
out = torch.empty((9, N), dtype=torch.int64, device=dev)
ws = torch.empty(ops.rank_workspace_bytes(N, 9), dtype=torch.uint8, device=dev)
reps = int(os.environ.get("REPS", "20"))
def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
print("dense     us", timed(lambda: ops.rank_full(full, out=out, workspace=ws)))
want = out.clone()
print("16 blocks us", timed(lambda: ops.rank_full_segments(blocks, out=out, workspace=ws)))
print("equal", bool(torch.equal(want, out)))
