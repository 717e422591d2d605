#!/usr/bin/env python3
"""Time one Conv2D geometry (fwd / dgrad / wgrad kernels) on the GPU with the ABI's HIP-event kernel timer.
usage: tune_conv.py B Ci H W Co k s pad [reps]      (env CNN_AMD_IGEMM_CFG=<id> forces an igemm tile config)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cnn_amd import capi

case = tuple(int(a) for a in sys.argv[1:9])
reps = int(sys.argv[9]) if len(sys.argv) > 9 else 5
B, Ci, H, W, Co, k, s, pad = case
conv = capi.Conv2d(*case)
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.rand((B, Ci, H, W), generator=g, device="cuda")
w = torch.randn((Co, Ci, k, k), generator=g, device="cuda") * 0.1
b = torch.randn((Co,), generator=g, device="cuda") * 0.1
y = torch.empty(conv.out_shape(), device="cuda")
dy = torch.rand(conv.out_shape(), generator=g, device="cuda") * 2 - 1
dx = torch.empty_like(x)
which = os.environ.get("TUNE_OPS", "fwd,dgrad,wgrad").split(",")


def run():
    if "fwd" in which:
        conv.forward(x, w, b, y)
    if "dgrad" in which:
        conv.backward_data(dy, w, dx)
    if "wgrad" in which:
        conv.backward_weight(x, dy, float(B))


run(); run()
torch.cuda.synchronize()
capi.kernel_timing(1)
for _ in range(reps):
    run()
rep = capi.kernel_timing_report()
flops = 2.0 * B * Co * conv.Ho * conv.Wo * Ci * k * k
nbytes = 4.0 * (x.numel() + y.numel() + w.numel())
for key, (cnt, ms) in rep.items():
    t = ms / cnt / 1e3
    print(f"{ms / cnt * 1e3:10.1f} us  {flops / t / 1e12:7.2f} TF  {nbytes / t / 1e9:8.1f} GB/s  {key.split('|')[0]}")
