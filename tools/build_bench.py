"""Time of mdx_index_create (the re-tiling of the shard into fragment-order tiles) at N = 1 004 993, D = 2048, both source
layouts and both storages; bytes = source read + tiles written."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdir_amd import ops

dev = "cuda:0"
N, D = (int(sys.argv[1]) if len(sys.argv) > 1 else 1004993), 2048
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.randn((N, D), generator=g, device=dev)
for layout, src in (("ND", x), ("DN", x.t().contiguous())):
    for storage in ("f32", "f16"):
        times = []
        for _ in range(4):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ix = ops.DescriptorIndex(src, layout, storage=storage)
            b.record(); torch.cuda.synchronize()
            times.append(a.elapsed_time(b))
            nbytes = N * D * 4 + ix.device_bytes
            del ix
        t = min(times[1:])
        print("%s %s  %.3f ms  %.2f TB/s (read + written)" % (layout, storage, t, nbytes / t / 1e9), flush=True)
