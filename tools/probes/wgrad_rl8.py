#!/usr/bin/env python3
"""wgrad_rd on 56-wide rows: flattened runs of 16 (default) against row-wise runs of 8 (CNN_AMD_RD_RL8=1)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cnn_amd import capi

for case in [(128, 256, 56, 56, 256, 3, 1, 1), (128, 128, 56, 56, 256, 3, 1, 1), (64, 64, 56, 56, 64, 3, 1, 1)]:
    B, Ci, H, W, Co = case[:5]
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand((B, Ci, H, W), generator=g, device="cuda")
    ref = None
    for rl8 in (None, "1"):
        capi.set_option("RD_RL8", rl8)
        conv = capi.Conv2d(*case)
        dy = torch.rand(conv.out_shape(), generator=torch.Generator(device="cuda").manual_seed(2), device="cuda") * 2 - 1
        gw, gb = conv.backward_weight(x, dy, float(B))
        if ref is None: ref = gw.clone()
        torch.cuda.synchronize(); capi.kernel_timing(1)
        for _ in range(4): conv.backward_weight(x, dy, float(B))
        rep = capi.kernel_timing_report(); capi.kernel_timing(0)
        fl = 2.0 * B * Co * H * W * Ci * 9
        print(case, "rl8", rl8, "diff %.1e" % float((gw - ref).abs().max() / ref.abs().max()), "  ".join(f"{k.split('|')[0]} {ms / c * 1e3:8.1f} us {fl / (ms / c) / 1e9:6.1f} TF" for k, (c, ms) in rep.items() if "reduce" not in k))
    capi.set_option("RD_RL8", None)
