#!/usr/bin/env python3
"""wgrad_rd on 14-wide rows: one run of 16 per row (default) against flattened runs of 8 (CNN_AMD_RD_FLAT8=1)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cnn_amd import capi

for case in [(64, 256, 14, 14, 256, 3, 1, 1), (128, 512, 14, 14, 512, 3, 1, 1), (7, 24, 14, 13, 40, 3, 1, 1), (3, 16, 9, 15, 32, 3, 1, 1)]:
    B, Ci, H, W, Co = case[:5]
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand((B, Ci, H, W), generator=g, device="cuda") - 0.5
    ref = None
    for f8 in ("0", "1"):
        capi.set_option("RD_FLAT8", f8)
        conv = capi.Conv2d(*case)
        dy = torch.rand(conv.out_shape(), generator=torch.Generator(device="cuda").manual_seed(2), device="cuda") * 2 - 1
        gw, gb = conv.backward_weight(x, dy, float(B))
        if ref is None: ref = (gw.clone(), gb.clone())
        torch.cuda.synchronize(); capi.kernel_timing(1)
        for _ in range(4): conv.backward_weight(x, dy, float(B))
        rep = capi.kernel_timing_report(); capi.kernel_timing(0)
        fl = 2.0 * B * Co * H * W * Ci * 9
        gwr = conv.backward_weight_im2col(x, dy, float(B))[0]
        print(case, "flat8", f8, "diff %.1e / vs im2col %.1e" % (float((gw - ref[0]).abs().max() / ref[0].abs().max()), float((gw - gwr).abs().max() / gwr.abs().max())),
              "  ".join(f"{k.split('|')[0]} {ms / c * 1e3:8.1f} us {fl / (ms / c) / 1e9:6.1f} TF" for k, (c, ms) in rep.items() if "reduce" not in k))
    capi.set_option("RD_FLAT8", None)
