#!/usr/bin/env python3
"""time ONE convolution geometry (isolated launches, the ABI's HIP-event kernel timer): fwd / dgrad / wgrad.
usage: one_layer.py B Ci H W Co k s pad [reps]     (TUNE_OPS=fwd,dgrad,wgrad; TUNE_NO_AUTOTUNE=1; CNN_AMD_DBG / CNN_AMD_IGEMM_CFG apply)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cnn_amd import capi

case = tuple(int(a) for a in sys.argv[1:9])
reps = int(sys.argv[9]) if len(sys.argv) > 9 else 3
B, Ci, H, W, Co, k, s, pad = case
which = os.environ.get("TUNE_OPS", "fwd,dgrad,wgrad").split(",")
conv = capi.Conv2d(*case)
if not os.environ.get("TUNE_NO_AUTOTUNE"):
    conv.autotune()
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.rand((B, Ci, H, W), generator=g, device="cuda")
w = torch.randn((Co, Ci, k, k), generator=g, device="cuda") * 0.1
b = torch.randn((Co,), generator=g, device="cuda") * 0.1
y = torch.empty(conv.out_shape(), device="cuda")
dy = torch.rand(conv.out_shape(), generator=g, device="cuda") * 2 - 1
dx = torch.empty_like(x)
if os.environ.get("TUNE_ZERO_X"):  # (experiment: data-dependent power / clocks)
    x.zero_(); dy.zero_()
if os.environ.get("TUNE_ZERO_W"):
    w.zero_()
flops = 2.0 * B * Co * conv.Ho * conv.Wo * Ci * k * k
for op in which:
    def run():
        if op == "fwd":
            conv.forward(x, w, b, y)
        elif op == "dgrad":
            conv.backward_data(dy, w, dx)
        elif op == "dgrad_relu":  # the data gradient through the ReLU' mask of the layer in front (relu_below = x: ~all pass)
            conv.backward_data_relu(dy, w, x, dx)
        else:
            conv.backward_weight(x, dy, float(B))
    run(); run()
    torch.cuda.synchronize()
    capi.kernel_timing(1)
    for _ in range(reps):
        run()
    rep = capi.kernel_timing_report()
    capi.kernel_timing(0)
    for key, (cnt, ms) in rep.items():
        us = ms / cnt * 1e3
        print(f"   {op:5s} {us:10.1f} us  {flops / (us * 1e-6) / 1e12:7.2f} TF ({100 * flops / (us * 1e-6) / 1e12 / 157.3:5.1f}%)  {key.split('|')[0]}")
