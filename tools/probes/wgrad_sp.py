#!/usr/bin/env python3
"""conv_wgrad_sp.hip against the register-direct kernel on the 3x3 / stride-1 / pad-1 layers with small planes (kernel + slab reduce).
usage: wgrad_sp.py            (B=64: the ResNet-shaped stack's stages; B=128 VGG=1: the VGG-shaped stack's deep layers)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from cnn_amd import capi

B = int(os.environ.get("B", 64))
CASES = [(512, 7, 7, 512), (256, 14, 14, 256), (128, 28, 28, 128), (64, 56, 56, 64)]
if os.environ.get("VGG"):
    CASES = [(512, 14, 14, 512), (512, 28, 28, 512), (256, 28, 28, 512), (256, 56, 56, 256), (128, 56, 56, 256)]
MODES = [m for m in os.environ.get("MODES", "rd,sp,sp1").split(",")]
for (Ci, H, W, Co) in CASES:
    case = (B, Ci, H, W, Co, 3, 1, 1)
    flops = 2.0 * B * Co * H * W * Ci * 9
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand((B, Ci, H, W), generator=g, device="cuda")
    ref = None
    for mode in MODES:
        capi.set_option("WGRAD_SP", "0" if mode == "rd" else "1")
        capi.set_option("SP_UNIT", "1" if mode == "sp1" else None)
        conv = capi.Conv2d(*case)
        dy = torch.rand(conv.out_shape(), generator=torch.Generator(device="cuda").manual_seed(2), device="cuda") * 2 - 1
        gw, gb = conv.backward_weight(x, dy, float(B))
        torch.cuda.synchronize()
        if ref is None:
            ref = gw.clone()
        err = float((gw - ref).abs().max() / ref.abs().max())
        capi.kernel_timing(1)
        for _ in range(5):
            conv.backward_weight(x, dy, float(B))
        rep = capi.kernel_timing_report()
        capi.kernel_timing(0)
        tot = sum(ms / c for (c, ms) in rep.values()) * 1e3
        line = "  ".join(f"{k.split('|')[0]} {ms / c * 1e3:7.1f} us" for k, (c, ms) in rep.items())
        print(f"{case} {mode:4s} total {tot:7.1f} us = {flops / tot / 1e6:6.1f} TF  ({line})  vs first {err:.1e}", flush=True)
capi.set_option("WGRAD_SP", None)
capi.set_option("SP_UNIT", None)
