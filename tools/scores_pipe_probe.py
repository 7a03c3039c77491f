"""A/B of the exact similarity kernel's two consumer schedules in ONE process: the shipped form (one
s_barrier at the top of every chunk, all 64 + 16 MFMAs behind it) against PIPE (mdx_scores_kernel.h: operands read one k-block
ahead, the stage hand-over in front of a stage's last MFMA step).  MDX_SCORES_PIPE is read per launch, so both run on the same
index, interleaved, on gaussian unit rows and on all-zero operands (the schedule at the full clock).  Scores must be bit-equal.

    python tools/scores_pipe_probe.py [rounds]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mdir_amd import ops

dev = "cuda:0"
N, Q, D = 1004993, 70, 2048
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
g = torch.Generator(device=dev)
g.manual_seed(1)


def timed(ix, q, out, pipe, reps=20):
    os.environ["MDX_SCORES_PIPE"] = pipe
    for _ in range(3):
        ix.scores(q, "ND", out=out)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        ix.scores(q, "ND", out=out)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for name in ("gaussian unit rows", "all zero"):
    if name.startswith("gauss"):
        x = torch.randn((N, D), generator=g, device=dev)
        x /= x.norm(dim=1, keepdim=True)
        q = torch.randn((Q, D), generator=g, device=dev)
        q /= q.norm(dim=1, keepdim=True)
    else:
        x = torch.zeros((N, D), device=dev)
        q = torch.zeros((Q, D), device=dev)
    ix = ops.DescriptorIndex(x, "ND")
    del x
    o0 = torch.empty((Q, N), dtype=torch.float32, device=dev)
    o1 = torch.empty_like(o0)
    for r in range(rounds):
        t0 = timed(ix, q, o0, "0")
        t1 = timed(ix, q, o1, "1")
        print("%-20s round %d: shipped %.3f ms   pipelined %.3f ms   (%+.1f %%)   bit-equal %s"
              % (name, r, t0, t1, 100.0 * (t1 - t0) / t0, bool(torch.equal(o0, o1))), flush=True)
    del ix, o0, o1
    torch.cuda.empty_cache()
