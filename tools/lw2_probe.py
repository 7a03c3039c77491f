"""Round 6, VERDICT item 6: the exact similarity launch (N = 1 004 993, Q = 70, D = 2048 fp32) with the shipped schedule (4 loader
waves, the round-5 pipelined consumer) against the experiment MDX_SCORES_LW2=1 (2 loader waves issuing twice the tiles each, 168
registers per wave, the FULL double-buffered operand set), alternating in ONE process on the same index -- gaussian unit rows and
all-zero operands -- with the outputs compared bit for bit.  -> profiles/r06_scores_schedule.md

The experiment is NOT in the shipped sources: apply tools/ablate/scores_lw2.patch to a scratch copy of mdir_amd/csrc, build it
(`make -C <copy>` -> libmdx.so) and point MDIR_AMD_LIB at that library:
    cp -r mdir_amd/csrc include /tmp/lw2/ ...; (cd /tmp/lw2/mdir_amd/csrc && patch -p1 < tools/ablate/scores_lw2.patch && make)
    MDIR_AMD_LIB=/tmp/lw2/mdir_amd/libmdx.so python tools/lw2_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdir_amd import ops

dev = "cuda:0"
N, Q, D = 1004993, 70, 2048
g = torch.Generator(device=dev); g.manual_seed(1)


def timed(ix, q, out, reps=20):
    for _ in range(3):
        ix.scores(q, "ND", out=out)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        ix.scores(q, "ND", out=out)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for name in ("gaussian unit rows", "all zero"):
    if name.startswith("gauss"):
        x = torch.randn((N, D), generator=g, device=dev); x /= x.norm(dim=1, keepdim=True)
        q = torch.randn((Q, D), generator=g, device=dev); q /= q.norm(dim=1, keepdim=True)
    else:
        x = torch.zeros((N, D), device=dev); q = torch.zeros((Q, D), device=dev)
    ix = ops.DescriptorIndex(x, "ND")
    del x
    outs = {v: torch.empty((Q, N), dtype=torch.float32, device=dev) for v in ("0", "1")}
    for rnd in range(3):
        for v in ("0", "1"):
            os.environ["MDX_SCORES_LW2"] = v
            print("%-20s round %d  LW2=%s  %.3f ms" % (name, rnd, v, timed(ix, q, outs[v])), flush=True)
    print("%-20s bit-equal: %s" % (name, bool(torch.equal(outs["0"], outs["1"]))), flush=True)
    del ix, outs
    torch.cuda.empty_cache()
os.environ.pop("MDX_SCORES_LW2", None)
