#!/usr/bin/env python3
"""conv_rows / wgrad_sp determinism stress: the inline-assembly MFMAs carry their own hazard padding (s_nop, acc_settle) -- a missed
hazard would show up as rare differing bits between runs of the same launch.  Every pass of every shape is repeated N times, alone and
with a second stream keeping the chip busy, and compared bit for bit with its first run.  usage: rows_stress.py [reps=40]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cnn_amd import capi

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
SHAPES = [(16, 64, 112, 112, 128, 3, 1, 0), (16, 64, 112, 112, 128, 3, 1, 1), (16, 128, 56, 56, 256, 3, 1, 1), (32, 64, 56, 56, 64, 3, 1, 1),
          (32, 256, 28, 28, 512, 3, 1, 1), (64, 256, 14, 14, 256, 3, 1, 1), (64, 512, 7, 7, 512, 3, 1, 1), (5, 48, 7, 7, 80, 3, 1, 1),
          # round 6: tall units at batch 64, the stride-2 siblings, the runtime-size kernels (any width / any plane size)
          (64, 64, 56, 56, 64, 3, 1, 1), (64, 128, 28, 28, 128, 3, 1, 1), (16, 64, 56, 56, 128, 3, 2, 1), (16, 128, 28, 28, 256, 3, 2, 1),
          (16, 256, 14, 14, 512, 3, 2, 1), (8, 64, 109, 109, 64, 3, 1, 0), (8, 128, 52, 52, 128, 3, 1, 0), (16, 128, 23, 23, 256, 3, 1, 0),
          (4, 32, 222, 222, 32, 3, 1, 0), (8, 64, 50, 50, 96, 3, 1, 1)]
side = torch.cuda.Stream()
noise_a = torch.rand((64 << 20,), device="cuda")  # 256 MB through the library's own ReLU kernel: the memory system and the wave slots stay busy
noise_b = torch.empty_like(noise_a)
bad = 0
for case in SHAPES:
    B, Ci, H, W, Co, k, s, pad = case
    conv = capi.Conv2d(*case)
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.rand((B, Ci, H, W), generator=g, device="cuda") - 0.3
    w = torch.randn((Co, Ci, k, k), generator=g, device="cuda") * 0.1
    b = torch.randn((Co,), generator=g, device="cuda") * 0.1
    dy = torch.rand(conv.out_shape(), generator=g, device="cuda") * 2 - 1
    first = None
    for rep in range(reps):
        busy = rep % 2 == 1
        if busy:  # a second stream hammers the memory system / the CUs meanwhile
            with torch.cuda.stream(side):
                for _ in range(4):
                    capi.relu_forward(noise_a, noise_b)
        y = conv.forward(x, w, b)
        dx = conv.backward_data(dy, w)
        dxr = torch.empty_like(x)
        conv.backward_data_relu(dy, w, x, dxr)
        gw, gb = conv.backward_weight(x, dy, float(B))
        torch.cuda.synchronize()
        cur = [t.clone() for t in (y, dx, dxr, gw, gb)]
        if first is None:
            first = cur
        else:
            for name, a, c in zip(("y", "dx", "dx_relu", "gw", "gb"), first, cur):
                if not torch.equal(a, c):
                    bad += 1
                    print(f"MISMATCH {case} rep {rep} {name}: {(a != c).sum().item()} elements differ, max |d| {(a - c).abs().max().item():.3e}")
    print(f"{case}: {reps} runs identical" if bad == 0 else f"{case}: done ({bad} mismatches so far)")
print("STRESS", "OK" if bad == 0 else f"FAILED ({bad})")
sys.exit(1 if bad else 0)
