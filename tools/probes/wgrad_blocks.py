#!/usr/bin/env python3
"""wgrad_rd on the small planes of the ResNet-shaped stack: pixel-range split (CNN_AMD_RD_BLOCKS = workgroup slots) vs time, slab reduce included.
usage: wgrad_blocks.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from cnn_amd import capi

B = int(os.environ.get("B", 64))
CASES = [(512, 7, 7, 512, (0, 384, 768, 1152)), (256, 14, 14, 256, (0, 96, 192, 288, 384)), (128, 28, 28, 128, (0, 120, 240, 384)), (64, 56, 56, 64, (0, 252, 384, 768))]
if os.environ.get("SWEEP") == "xcd":  # pixel-range counts that are multiples of 8: one XCD's L2 per pixel range
    CASES = [(256, 14, 14, 256, (0, 768, 1536)), (128, 28, 28, 128, (0, 192, 384, 768)), (64, 56, 56, 64, (0, 48, 96, 192, 384, 768)), (512, 7, 7, 512, (0, 3072))]
    if os.environ.get("B") == "128" and os.environ.get("MORE"):
        CASES = [(256, 56, 56, 256, (480, 0, 480, 0)), (64, 112, 112, 128, (504, 0, 504, 0, 1008, 504)), (128, 56, 56, 256, (480, 0, 480, 0))]
    elif os.environ.get("B") == "128":
        CASES = [(512, 14, 14, 512, (0, 3072)), (512, 28, 28, 512, (0, 3072)), (256, 28, 28, 512, (0, 1536, 3072)), (256, 56, 56, 256, (0, 768, 1536)), (128, 56, 56, 256, (0, 384, 768, 1536))]
for (Ci, H, W, Co, slots) in CASES:
    case = (B, Ci, H, W, Co, 3, 1, 1)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand((B, Ci, H, W), generator=g, device="cuda")
    ref = None
    for sl in slots:
        capi.set_option("RD_BLOCKS", None if sl == 0 else str(sl))
        conv = capi.Conv2d(*case)
        dy = torch.rand(conv.out_shape(), generator=g, device="cuda") * 2 - 1
        gw, gb = conv.backward_weight(x, dy, float(B))
        torch.cuda.synchronize()
        capi.kernel_timing(1)
        for _ in range(5):
            conv.backward_weight(x, dy, float(B))
        rep = capi.kernel_timing_report()
        capi.kernel_timing(0)
        if ref is None:
            ref = None
        line = "  ".join(f"{k.split('|')[0]} {ms / c * 1e3:7.1f} us" for k, (c, ms) in rep.items())
        print(case, "slots", sl, line)
    capi.set_option("RD_BLOCKS", None)
