import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnn_amd import capi
case = (64, 64, 112, 112, 128, 3, 1, 0)
conv = capi.Conv2d(*case)
g = torch.Generator(device="cuda").manual_seed(3)
x = torch.rand((case[0], 64, 112, 112), generator=g, device="cuda")
w = torch.randn((128, 64, 3, 3), generator=g, device="cuda") * 0.1
b = torch.randn((128,), generator=g, device="cuda") * 0.1
dy = torch.rand(conv.out_shape(), generator=g, device="cuda") * 2 - 1
for name, fn in (("fwd", lambda s: conv.forward(x * s, w, b * s)), ("dgrad", lambda s: conv.backward_data(dy * s, w))):
    a1 = fn(1.0).clone(); a2 = fn(1.0).clone(); a3 = fn(2.0).clone()
    d12 = (a1 != a2); d13 = (a1 * 2 != a3)
    print(name, "repeat mismatches", int(d12.sum()), "of", a1.numel(), "| linearity mismatches", int(d13.sum()),
          "max abs diff", float((a1 * 2 - a3).abs().max()))
    if d13.any():
        idx = d13.nonzero()[:8]
        print(" first mismatching indices", idx.tolist())
        # distribution over w coordinate
        ws = d13.nonzero()[:, 3]
        print(" mismatch count by w (first 8 / last 8):", torch.bincount(ws, minlength=a1.shape[3])[:8].tolist(), torch.bincount(ws, minlength=a1.shape[3])[-8:].tolist())
        hs = d13.nonzero()[:, 2]
        print(" mismatch count by h (first 8 / last 8):", torch.bincount(hs, minlength=a1.shape[2])[:8].tolist(), torch.bincount(hs, minlength=a1.shape[2])[-8:].tolist())
