"""conv_layer_1's window kernel (packed mask), alone: default / staging only (WIN_DBG=1) / MFMA only (WIN_DBG=2, non-specialised
instance) / specialisation off / slack variants.  (tools/probes/win_consumer_prof.patch adds WIN_DBG=4..7 -- consumers without any DMA,
without the row hand-over, with every LDS read of a strip up front -- and per-phase counters under -DCNN_WIN_EXPERIMENT=6.)  usage: python tools/probes/win_phases.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cnn_amd import capi

B, H, W = 256, 224, 224
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.rand((B, 3, H, W), generator=g, device="cuda")
w = torch.randn((16, 3, 3, 3), generator=g, device="cuda") * 0.1
b = torch.randn((16,), generator=g, device="cuda") * 0.1
conv = capi.Conv2d(B, 3, H, W, 16, 3, 2, 0)
conv.set_pool_mask_packed()
pooled = torch.empty((B, 16, 55, 55), device="cuda")
mask = torch.empty(conv.pool_mask_bytes(), dtype=torch.uint8, device="cuda")
conv.relu_maxpool2_forward(x, w, b, pooled, mask)
dpool = torch.rand(pooled.shape, generator=g, device="cuda") * 2 - 1
gw, gb = torch.empty_like(w), torch.empty_like(b)
for opts in ({}, {"WIN_DBG": "1"}, {"WIN_DBG": "2"}, {"WIN_SPEC": "0"}, {"WIN_SPEC": "0", "WIN_DBG": "1"}, {"WIN_SLACK": "4"}, {"WIN_SLACK": "28"}, {}):
    for k, v in opts.items():
        capi.set_option(k, v)
    f = lambda: conv.backward_weight_pooled2(x, dpool, mask, None, float(B), gw, gb)
    f(); torch.cuda.synchronize()
    capi.kernel_timing(1)
    for _ in range(20):
        f()
    rep = capi.kernel_timing_report()
    capi.kernel_timing(0)
    for key, (cnt, ms) in rep.items():
        if "conv_wgrad_win" in key:
            print(f"{str(opts):44s} {ms / cnt * 1e3:7.1f} us")
    for k in opts:
        capi.set_option(k, None)
