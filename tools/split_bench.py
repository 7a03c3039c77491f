#!/usr/bin/env python3
"""Exact chain vs MDX_F32_SPLIT3 on the headline shape (1 004 993 x 70 x 2048): HIP-event times, interleaved, and the
score difference.  python tools/split_bench.py [rows] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from mdir_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else bench.N_ROXFORD + bench.N_DISTRACTORS
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
rows = bench.gen_rows(0, n, dev)
qvecs, qid = bench.gen_queries(n, dev)
ix = ops.DescriptorIndex(rows, "ND")
del rows
sc = {m: torch.empty((bench.NQ, n), dtype=torch.float32, device=dev) for m in ("chain", "split3", "split2")}


def timed(mode):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        ix.scores(qvecs, "DN", out=sc[mode], compute=mode)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for mode in sc:
    ix.scores(qvecs, "DN", out=sc[mode], compute=mode)
torch.cuda.synchronize()
for rnd in range(3):
    print("round %d: " % rnd + "  ".join("%s %.4f ms" % (m, timed(m)) for m in sc), flush=True)
for m in ("split3", "split2"):
    diff = (sc["chain"] - sc[m]).abs()
    print("max |%s - chain| = %.3g, mean %.3g; top-1 equal %s" % (m, float(diff.max()), float(diff.mean()),
          bool((sc["chain"].argmax(1) == sc[m].argmax(1)).all())))
t = timed("split3")
algo = 4.0 * n * bench.DIM + 4.0 * bench.NQ * n
print("split3: %.4f ms = %.2f TB/s of algorithmic bytes (%.3f of 8 TB/s); chain-equivalent %.1f TFLOP/s" % (
    t, algo / t / 1e9, algo / t / 1e9 / 8.0, 2.0 * bench.NQ * n * bench.DIM / t / 1e9))
