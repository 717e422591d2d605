#!/usr/bin/env python3
"""conv_layer_1's data gradient from the pooled domain, alone: the per-lane-load kernel (DGRAD_POOL_LDS=0) against the LDS-staged
one (default), bit-identity and time per launch.  usage: python tools/time_first_dgrad.py [B=256] [H=224] [W=224]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cnn_amd import capi

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H = int(sys.argv[2]) if len(sys.argv) > 2 else 224
W = int(sys.argv[3]) if len(sys.argv) > 3 else 224
conv = capi.Conv2d(B, 3, H, W, 16, 3, 2, 0)
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.rand((B, 3, H, W), generator=g, device="cuda")
w = torch.randn((16, 3, 3, 3), generator=g, device="cuda") * 0.1
b = torch.randn((16,), generator=g, device="cuda") * 0.1
Ho, Wo = conv.Ho, conv.Wo
pooled = torch.empty((B, 16, Ho // 2, Wo // 2), device="cuda")
mask = torch.empty((B, 16, Ho // 2, Wo // 2), dtype=torch.int32, device="cuda")
conv.relu_maxpool2_forward(x, w, b, pooled, mask)
dpool = torch.rand(pooled.shape, generator=g, device="cuda") * 2 - 1
res = {}
for mode in ("0", "1"):
    capi.set_option("DGRAD_POOL_LDS", mode)
    for pl in (pooled, None):
        dx = torch.full_like(x, 7.0)
        conv.backward_data_pooled2(dpool, mask, pl, w, dx)
        torch.cuda.synchronize()
        capi.kernel_timing(1)
        for _ in range(20):
            conv.backward_data_pooled2(dpool, mask, pl, w, dx)
        rep = capi.kernel_timing_report()
        capi.kernel_timing(0)
        for key, (cnt, ms) in rep.items():
            if "conv_dgrad_pk" in key:
                print(f"DGRAD_POOL_LDS={mode} {'+pool ' if pl is not None else '+poolm'} {ms / cnt * 1e3:7.1f} us   {key.split('|')[0]}")
        res[(mode, pl is None)] = dx.clone()
for pm in (False, True):
    same = torch.equal(res[("0", pm)].view(torch.int32), res[("1", pm)].view(torch.int32))
    print("bit-identical" if same else "MISMATCH", "(+poolm)" if pm else "(+pool)")
    assert same
