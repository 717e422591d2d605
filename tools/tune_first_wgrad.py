#!/usr/bin/env python3
"""Time the first layer's weight gradient (3 -> 16, 3x3, stride 2) alone: plain delta and both pooled-domain variants.
usage: tune_first_wgrad.py [batch] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cnn_amd import capi

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
conv = capi.Conv2d(B, 3, 224, 224, 16, 3, 2, 0)
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.rand((B, 3, 224, 224), generator=g, device="cuda")
w = torch.randn((16, 3, 3, 3), generator=g, device="cuda") * 0.1
b = torch.randn((16,), generator=g, device="cuda") * 0.1
pooled = torch.empty((B, 16, 55, 55), device="cuda")
mask = torch.empty((B, 16, 55, 55), dtype=torch.int32, device="cuda")
conv.relu_maxpool2_forward(x, w, b, pooled, mask)
dpool = torch.rand((B, 16, 55, 55), generator=g, device="cuda") * 2 - 1
dy = torch.rand(conv.out_shape(), generator=g, device="cuda") * 2 - 1
gw = torch.empty((16, 3, 3, 3), device="cuda")
gb = torch.empty((16,), device="cuda")


def run():
    conv.backward_weight(x, dy, float(B), gw, gb)
    conv.backward_weight_pooled2(x, dpool, mask, pooled, float(B), gw, gb)
    conv.backward_weight_pooled2(x, dpool, mask, None, float(B), gw, gb)


run(); run()
torch.cuda.synchronize()
capi.kernel_timing(1)
for _ in range(reps):
    run()
for key, (cnt, ms) in capi.kernel_timing_report().items():
    if "slab" in key:
        continue
    name = key.split("|")[0]
    nbytes = 4.0 * (x.numel() + (dy.numel() if "+pool" not in name else dpool.numel() * (2 if name.endswith("poolm") else 3)))
    print(f"{ms / cnt * 1e3:9.1f} us  {nbytes / (ms / cnt / 1e3) / 1e9:8.1f} GB/s  {name}")
