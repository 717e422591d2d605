"""BatchNorm2D kernel times / achieved HBM rate at the BN sites of AlexNet(batch_norm=true), B=256."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnn_amd import capi

B = int(os.environ.get("B", 256))
SHAPES = [(16, 111, 111), (32, 27, 27), (64, 13, 13), (128, 6, 6), (64, 112, 112)]
if os.environ.get("SHAPES") == "resnet":  # B=64: the BN sites of the ResNet-18-shaped stack
    SHAPES = [(64, 56, 56), (128, 28, 28), (256, 14, 14), (512, 7, 7)]
for (C, H, W) in SHAPES:
    x = torch.randn((B, C, H, W), device="cuda")
    y, dy = torch.empty_like(x), torch.randn_like(x)
    gm, bt = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    mm, mv = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    gg, gb = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    bn = capi.BatchNorm2d(B, C, H, W)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for it in range(3):
        bn.forward(x, gm, bt, mm, mv, y, True); bn.backward(x, dy, gm, gg, gb)
    torch.cuda.synchronize()
    n = 20
    ev[0].record()
    for _ in range(n): bn.forward(x, gm, bt, mm, mv, y, True)
    ev[1].record()
    for _ in range(n): bn.backward(x, dy, gm, gg, gb)
    ev[2].record()
    torch.cuda.synchronize()
    tf, tb = ev[0].elapsed_time(ev[1]) / n * 1e-3, ev[1].elapsed_time(ev[2]) / n * 1e-3
    byt = x.numel() * 4
    print(f"B{B} C{C} {H}x{W}: fwd {tf*1e6:8.1f} us  {16*x.numel()/tf/1e9:7.0f} GB/s(alg 16B/el) | bwd {tb*1e6:8.1f} us {20*x.numel()/tb/1e9:7.0f} GB/s(alg 20B/el)  tensor {byt/1e6:.1f} MB")
