#!/usr/bin/env python3
"""Is the first layer's window kernel 'slower in company' (64 us alone, 78 us in the step) -- or is the back-to-back benchmark warm?  Time it alone
with and without a 1 GB fill between launches (the 154 MB input then comes from HBM, not from the 256 MB Infinity Cache the previous launch left it in)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cnn_amd import capi

B = 256
conv = capi.Conv2d(B, 3, 224, 224, 16, 3, 2, 0)
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.rand((B, 3, 224, 224), generator=g, device="cuda")
w = torch.randn((16, 3, 3, 3), generator=g, device="cuda") * 0.1
b = torch.randn((16,), generator=g, device="cuda") * 0.1
pooled = torch.empty((B, 16, 55, 55), device="cuda")
mask = torch.empty((B, 16, 55, 55), dtype=torch.int32, device="cuda")
conv.relu_maxpool2_forward(x, w, b, pooled, mask)
dpool = torch.rand((B, 16, 55, 55), generator=g, device="cuda") * 2 - 1
gw = torch.empty((16, 3, 3, 3), device="cuda"); gb = torch.empty((16,), device="cuda")
dx = torch.empty_like(x)
junk = torch.empty((256 * 1024 * 1024,), device="cuda")
for flush in (False, True, False, True):
    for _ in range(2):
        conv.backward_weight_pooled2(x, dpool, mask, None, float(B), gw, gb)
    torch.cuda.synchronize()
    capi.kernel_timing(1)
    for _ in range(10):
        if flush:
            junk.fill_(1.0)
        conv.backward_weight_pooled2(x, dpool, mask, None, float(B), gw, gb)
        if flush:
            junk.fill_(2.0)
        conv.backward_data_pooled2(dpool, mask, None, w, dx)
    rep = capi.kernel_timing_report(); capi.kernel_timing(0)
    print("flush between launches:" if flush else "back to back:           ", "  ".join(f"{k.split('|')[0]} {ms / c * 1e3:6.1f} us" for k, (c, ms) in rep.items() if "reduce" not in k and "pack" not in k))
