"""conv_layer_2's forward alone (prepared filters, ReLU-only output), kernel-timer averages"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from cnn_amd import capi
c = capi.Conv2d(256, 16, 55, 55, 32, 3, 2, 0)
x = torch.rand((256, 16, 55, 55), device="cuda"); w = torch.rand((32, 16, 3, 3), device="cuda") * 0.1; b = torch.zeros(32, device="cuda")
yr = torch.empty((256, 32, 27, 27), device="cuda")
pf, pd = c.prepared_buffers("cuda"); capi.prepare_filters([c], [w], [b], [pf], [pd])
for _ in range(3): c.forward_prepared(x, pf, b, None, yr)
torch.cuda.synchronize(); capi.kernel_timing(1)
for _ in range(20): c.forward_prepared(x, pf, b, None, yr)
torch.cuda.synchronize()
for k, (n, ms) in capi.kernel_timing_report().items(): print(f"{ms/n*1e3:8.1f} us x{n:3d}  {k}")
