#!/usr/bin/env python3
"""The three kernels of conv_layer_1's block, alone (prepared filters, batch 256): forward (with the ablations of CNN_AMD_DBG), data
gradient and weight gradient, each with the int32 and the packed pool mask.  usage: python tools/time_first_block.py [B=256]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cnn_amd import capi

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H = W = 224
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.rand((B, 3, H, W), generator=g, device="cuda")
w = torch.randn((16, 3, 3, 3), generator=g, device="cuda") * 0.1
b = torch.randn((16,), generator=g, device="cuda") * 0.1


def timed(tag, fn, match, n=20):
    fn()
    torch.cuda.synchronize()
    capi.kernel_timing(1)
    for _ in range(n):
        fn()
    rep = capi.kernel_timing_report()
    capi.kernel_timing(0)
    for key, (cnt, ms) in rep.items():
        if match in key:
            print(f"{tag:34s} {ms / cnt * 1e3:7.1f} us   {key.split('|')[0]}")


for packed in (False, True):
    conv = capi.Conv2d(B, 3, H, W, 16, 3, 2, 0)
    if packed:
        conv.set_pool_mask_packed()
    pf, pd = conv.prepared_buffers("cuda")
    capi.prepare_filters([conv], [w], [b], [pf], [pd])
    pooled = torch.empty((B, 16, conv.Ho // 2, conv.Wo // 2), device="cuda")
    mask = torch.empty(conv.pool_mask_bytes(), dtype=torch.uint8, device="cuda")
    dpool = torch.rand(pooled.shape, generator=g, device="cuda") * 2 - 1
    dx = torch.empty_like(x)
    gw, gb = torch.empty_like(w), torch.empty_like(b)
    tag = "packed mask" if packed else "int32 mask "
    for dbg in ((None,) if packed else (None, "1", "2")):
        capi.set_option("DBG", dbg)
        timed(f"forward {tag} DBG={dbg}", lambda: conv.relu_maxpool2_forward(x, None, None, pooled, mask, prepared_fwd=pf), "conv_fwd_pool_pk")
    capi.set_option("DBG", None)
    conv.relu_maxpool2_forward(x, None, None, pooled, mask, prepared_fwd=pf)
    timed(f"data gradient {tag}", lambda: conv.backward_data_pooled2(dpool, mask, None, None, dx, prepared_dgrad=pd), "conv_dgrad_pk")
    timed(f"weight gradient {tag}", lambda: conv.backward_weight_pooled2(x, dpool, mask, None, float(B), gw, gb), "conv_wgrad_win")
