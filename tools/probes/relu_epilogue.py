"""does the fused ReLU epilogue cost anything? forward vs forward+relu, data gradient vs data gradient + ReLU' on one geometry
usage: PYTHONPATH=. python tools/probes/relu_epilogue.py B Ci H W Co k s pad"""
import sys

import torch

from cnn_amd import capi

case = tuple(int(a) for a in sys.argv[1:9]) if len(sys.argv) >= 9 else (128, 512, 14, 14, 512, 3, 1, 1)
B, Ci, H, W, Co, k, s, pad = case
conv = capi.Conv2d(*case)
conv.autotune()
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.rand((B, Ci, H, W), generator=g, device="cuda")
w = torch.randn((Co, Ci, k, k), generator=g, device="cuda") * 0.1
b = torch.randn((Co,), generator=g, device="cuda") * 0.1
y = torch.empty(conv.out_shape(), device="cuda")
y2 = torch.empty_like(y)
dy = torch.rand(conv.out_shape(), generator=g, device="cuda") * 2 - 1
dx = torch.empty_like(x)
mask = torch.relu(x - 0.5)
flops = 2.0 * B * Co * conv.Ho * conv.Wo * Ci * k * k
runs = {"fwd": lambda: conv.forward(x, w, b, y), "fwd+relu": lambda: conv.forward_relu(x, w, b, y, y2),
        "dgrad": lambda: conv.backward_data(dy, w, dx), "dgrad+relu": lambda: conv.backward_data_relu(dy, w, mask, dx)}
for name, run in runs.items():
    run(); run()
    torch.cuda.synchronize()
    capi.kernel_timing(1)
    for _ in range(5):
        run()
    rep = capi.kernel_timing_report()
    capi.kernel_timing(0)
    for key, (cnt, ms) in rep.items():
        print(f"{name:11s} {ms / cnt * 1e3:8.1f} us  {flops / (ms / cnt) / 1e9:6.1f} TF  {key}")
