#!/usr/bin/env python3
"""Time every convolution geometry of a BASELINE stack (VGG-11-shaped / ResNet-18-shaped / reference net), one layer at a
time, isolated launches: fwd / dgrad / wgrad kernels with the ABI's HIP-event kernel timer.
usage: tune_stack.py vgg11|resnet18|alexnet [batch] [reps]     (TUNE_OPS=fwd,dgrad,wgrad selects)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cnn_amd import capi
from cnn_amd.stacks import conv_geometries

name = sys.argv[1]
batch = int(sys.argv[2]) if len(sys.argv) > 2 else {"vgg11": 128, "resnet18": 64, "alexnet": 256}[name]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
which = os.environ.get("TUNE_OPS", "fwd,dgrad,wgrad").split(",")
PEAK = 157.3

tot = {}
seen = {}
for (Ci, H, W, Co, k, s, pad) in conv_geometries(name):
    case = (batch, Ci, H, W, Co, k, s, pad)
    if case in seen:
        for op, us in seen[case].items():
            tot[op] = tot.get(op, 0.0) + us
        print(f"== {case}  (same as above)")
        continue
    conv = capi.Conv2d(*case)
    if not os.environ.get("TUNE_NO_AUTOTUNE"):
        conv.autotune()
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand((batch, Ci, H, W), generator=g, device="cuda")
    w = torch.randn((Co, Ci, k, k), generator=g, device="cuda") * 0.1
    b = torch.randn((Co,), generator=g, device="cuda") * 0.1
    y = torch.empty(conv.out_shape(), device="cuda")
    dy = torch.rand(conv.out_shape(), generator=g, device="cuda") * 2 - 1
    dx = torch.empty_like(x)
    flops = 2.0 * batch * Co * conv.Ho * conv.Wo * Ci * k * k
    print(f"== {case}  {flops / 1e9:.1f} GFLOP per pass")
    seen[case] = {}
    for op in which:
        def run():
            if op == "fwd":
                conv.forward(x, w, b, y)
            elif op == "dgrad":
                conv.backward_data(dy, w, dx)
            else:
                conv.backward_weight(x, dy, float(batch))
        run(); run()
        torch.cuda.synchronize()
        capi.kernel_timing(1)
        for _ in range(reps):
            run()
        rep = capi.kernel_timing_report()
        capi.kernel_timing(0)
        op_us = 0.0
        for key, (cnt, ms) in rep.items():
            us = ms / cnt * 1e3
            op_us += us
            print(f"   {op:5s} {us:10.1f} us  {flops / (us * 1e-6) / 1e12:7.2f} TF ({100 * flops / (us * 1e-6) / 1e12 / PEAK:5.1f}%)  {key.split('|')[0]}")
        seen[case][op] = op_us
        tot[op] = tot.get(op, 0.0) + op_us
    del x, y, dy, dx, conv
    torch.cuda.empty_cache()
print("== totals (us):", {k: round(v, 1) for k, v in tot.items()}, " sum", round(sum(tot.values()), 1))
