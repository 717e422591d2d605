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
for dbg in (None, "4"):
    capi.set_option("WIN_DBG", dbg)
    print("WIN_DBG", dbg, file=sys.stderr)
    for _ in range(2):
        conv.backward_weight_pooled2(x, dpool, mask, None, float(B), gw, gb)
    torch.cuda.synchronize()
