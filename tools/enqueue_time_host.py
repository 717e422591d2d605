"""How long does the host take to ENQUEUE one train step of the C++ Layer API (ctypes -> architectures::Sequential::train_step -> HIP
launches) against the GPU time of the step?  If the two are close the step is host-bound and kernel work no longer shows."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cnn_amd import hostapi

B = 256
net = hostapi.HostAlexNet(3)
net.set_params((np.random.RandomState(1).standard_normal(net.n_params) * 0.1).astype(np.float32))
x = torch.rand((B, 3, 224, 224), device="cuda")
labels = (torch.arange(B, device="cuda") % 3).to(torch.int32)
for _ in range(10):
    net.train_step(x, labels, 1e-3)
torch.cuda.synchronize()
for n in (20, 100, 400):
    t0 = time.perf_counter()
    for _ in range(n):
        net.train_step(x, labels, 1e-3)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"n={n:4d}  enqueue {1e3*(t1-t0)/n:.4f} ms/step   total {1e3*(t2-t0)/n:.4f} ms/step   (host ahead by {1e3*(t2-t1):.2f} ms at the end)")
net.close()
