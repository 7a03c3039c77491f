"""1x1 convolutions of the ResNet101 trunk at 1024x768 (batch 4, the three pyramid scales): MIOpen conv + mdx_bn_act
against mdx_conv1x1_bn_act (when the library has it).  Per shape: microseconds and TFLOP/s."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import mdir_amd  # noqa: F401  (MIOPEN_FIND_MODE)
from mdir_amd import ops

dev = "cuda:0"
B = int(os.environ.get("B", "4"))
# (name, Cin, Cout, H, W, residual)
SHAPES = []
for scale, (H, W) in (("s1", (256, 192)), ("s0.7", (181, 136)), ("s0.5", (128, 96))):
    for lname, planes, div in (("layer1", 64, 1), ("layer2", 128, 2), ("layer3", 256, 4), ("layer4", 512, 8)):
        h, w = -(-H // div), -(-W // div)
        SHAPES.append(("%s %s reduce" % (scale, lname), planes * 4, planes, h, w, False))
        SHAPES.append(("%s %s expand" % (scale, lname), planes, planes * 4, h, w, True))


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3       # us


have = hasattr(ops, "conv1x1_bn_act")
rows = []
tot_ref = tot_new = 0.0
# how often each shape occurs in ResNet101 (blocks per layer: 3, 4, 23, 3)
COUNT = {"layer1": 3, "layer2": 4, "layer3": 23, "layer4": 3}
for name, cin, cout, h, w, res in SHAPES:
    torch.manual_seed(0)
    x = torch.randn(B, cin, h, w, device=dev)
    wt = torch.randn(cout, cin, 1, 1, device=dev) / cin ** 0.5
    mean, var = torch.randn(cout, device=dev) * 0.1, torch.rand(cout, device=dev) + 0.5
    gamma, beta = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1
    idt = torch.randn(B, cout, h, w, device=dev) if res else None
    flops = 2.0 * B * cin * cout * h * w

    def ref():
        return ops.bn_act_(F.conv2d(x, wt), mean, var, gamma, beta, 1e-5, idt, True)
    t_conv = timed(lambda: F.conv2d(x, wt))
    t_ref = timed(ref)
    row = {"shape": name, "cin": cin, "cout": cout, "hw": h * w, "conv_us": round(t_conv, 1), "conv_bn_us": round(t_ref, 1),
           "conv_tflops": round(flops / t_conv / 1e6, 1)}
    n = COUNT[name.split()[1]]
    tot_ref += n * t_ref
    if have:
        wtt = ops.conv1x1_transpose_weights(wt)

        def new():
            return ops.conv1x1_bn_act(x, wtt, mean, var, gamma, beta, 1e-5, idt, True)
        got, want = new(), ref()
        row["max_rel_err"] = float((got - want).abs().max() / want.abs().max())
        t_new = timed(new)
        row["fused_us"] = round(t_new, 1)
        row["fused_tflops"] = round(flops / t_new / 1e6, 1)
        tot_new += n * t_new
    rows.append(row)
    print(json.dumps(row))
print(json.dumps({"sum_over_trunk_1x1_us_per_batch": {"miopen_plus_bn_act": round(tot_ref), "fused": round(tot_new) if have else None},
                  "batch": B}))
