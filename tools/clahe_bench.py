import torch, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdir_amd import ops
u8 = torch.randint(0, 256, (4, 768, 1024, 3), device="cuda", dtype=torch.uint8)
mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
for fn, name in ((lambda: ops.clahe_u8_to_chw(u8, 4, 8, mean, std), "clahe"), (lambda: ops.u8_to_chw(u8, mean, std), "plain")):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): fn()
    b.record(); torch.cuda.synchronize()
    print(name, a.elapsed_time(b) / 20, "ms per batch of 4")
