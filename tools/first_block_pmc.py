#!/usr/bin/env python3
"""The three kernels of conv_layer_1's block (packed pool mask: the train step's variants), alone, N launches each, next to a plain
216 MB device copy (108 MB read + 108 MB written: what a streaming kernel's counters look like on this box).  Driven under
rocprofv3 --pmc by tools/pmc_first_block.sh.  usage: python tools/first_block_pmc.py [B=256] [N=6]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cnn_amd import capi

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 6
H = W = 224
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.rand((B, 3, H, W), generator=g, device="cuda")
w = torch.randn((16, 3, 3, 3), generator=g, device="cuda") * 0.1
b = torch.randn((16,), generator=g, device="cuda") * 0.1
conv = capi.Conv2d(B, 3, H, W, 16, 3, 2, 0)
conv.set_pool_mask_packed()
pf, pd = conv.prepared_buffers("cuda")
capi.prepare_filters([conv], [w], [b], [pf], [pd])
pooled = torch.empty((B, 16, conv.Ho // 2, conv.Wo // 2), device="cuda")
mask = torch.empty(conv.pool_mask_bytes(), dtype=torch.uint8, device="cuda")
dpool = torch.rand(pooled.shape, generator=g, device="cuda") * 2 - 1
dx = torch.empty_like(x)
gw, gb = torch.empty_like(w), torch.empty_like(b)
src = torch.rand((27 * 1000 * 1000,), generator=g, device="cuda")  # 108 MB
dst = torch.empty_like(src)
for _ in range(N):
    conv.relu_maxpool2_forward(x, None, None, pooled, mask, prepared_fwd=pf)
    conv.backward_data_pooled2(dpool, mask, None, None, dx, prepared_dgrad=pd)
    conv.backward_weight_pooled2(x, dpool, mask, None, float(B), gw, gb)
    dst.copy_(src)
torch.cuda.synchronize()
