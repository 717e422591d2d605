"""conv_layer_3 / conv_layer_4 kernels alone (prepared filters, fused ReLU paths), kernel-timer averages.
CNN_AMD_FWD_RD_DBG=1 / CNN_AMD_DGRAD_RD_DBG=1 / CNN_AMD_RD_DBG=9 add the per-phase cycle counts of workgroup 0."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from cnn_amd import capi
for case in ((256, 32, 27, 27, 64, 3, 2, 0), (256, 64, 13, 13, 128, 3, 2, 0)):
    c = capi.Conv2d(*case)
    B, Ci, H, W, Co = case[:5]
    Ho, Wo = (H - 3) // 2 + 1, (W - 3) // 2 + 1
    x = torch.rand((B, Ci, H, W), device="cuda"); w = torch.rand((Co, Ci, 3, 3), device="cuda") * 0.1; b = torch.zeros(Co, device="cuda")
    dy = torch.rand((B, Co, Ho, Wo), device="cuda"); dx = torch.empty_like(x); yr = torch.empty_like(dy)
    gw, gb = torch.empty_like(w), torch.empty_like(b)
    pf, pd = c.prepared_buffers("cuda"); capi.prepare_filters([c], [w], [b], [pf], [pd])
    def run():
        c.forward_prepared(x, pf, b, None, yr)
        c.backward_data_relu(dy, None, x, dx, prepared_dgrad=pd)
        c.backward_weight(x, dy, float(B), gw, gb)
    for _ in range(3): run()
    torch.cuda.synchronize(); capi.kernel_timing(1)
    for _ in range(10): run()
    torch.cuda.synchronize()
    for k, (n, ms) in capi.kernel_timing_report().items(): print(f"{ms/n*1e3:8.1f} us x{n:3d}  {k}")
    capi.kernel_timing(0)
