"""the linear head alone: fused forward + loss (+ dx + ReLU') and the fused backward kernels, kernel-timer averages.
usage: linear_time.py [B=256] [in=4608]      (out = 3)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from cnn_amd import capi
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n_in = int(sys.argv[2]) if len(sys.argv) > 2 else 4608
n_out = 3
x = torch.rand((B, n_in), device="cuda") - 0.3; w = torch.rand((n_in, n_out), device="cuda") * 0.1; b = torch.zeros(n_out, device="cuda")
labels = (torch.arange(B, device="cuda") % 3).to(torch.int32)
logits, delta, terms = torch.empty((B, n_out), device="cuda"), torch.empty((B, n_out), device="cuda"), torch.empty(B, device="cuda")
gw, gb, dx = torch.empty_like(w), torch.empty_like(b), torch.empty_like(x)
lib = capi.load()
def run():
    capi.check(lib.cnn_linear_forward_softmax_xent(capi._ptr(x), capi._ptr(w), capi._ptr(b), capi._ptr(labels), capi._ptr(logits), None,
                                                   capi._ptr(delta), capi._ptr(terms), B, n_in, n_out, capi._stream()), "fwd")
    capi.linear_backward(x, delta, w, float(B), gw, gb, dx, relu_below=True)
    # the train step's pair: head with dx + ReLU', then the weight / bias gradient alone
    capi.check(lib.cnn_linear_forward_softmax_xent_dx(capi._ptr(x), capi._ptr(w), capi._ptr(b), capi._ptr(labels), capi._ptr(logits), None,
                                                      capi._ptr(delta), capi._ptr(terms), capi._ptr(dx), 1, B, n_in, n_out, capi._stream()), "fwd+dx")
    capi.check(lib.cnn_linear_backward(capi._ptr(x), capi._ptr(delta), capi._ptr(w), capi._ptr(gw), capi._ptr(gb), None, B, n_in, n_out, float(B),
                                       capi._stream()), "wb")
for _ in range(5): run()
torch.cuda.synchronize(); capi.kernel_timing(1)
for _ in range(20): run()
torch.cuda.synchronize()
for k, (n, ms) in capi.kernel_timing_report().items(): print(f"{ms/n*1e3:8.1f} us x{n:3d}  {k}")
