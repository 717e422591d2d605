"""How long does the host take to ENQUEUE one train step (ctypes + HIP launches) vs the GPU time of the step?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnn_amd.pynet import AlexNetHip

B = 256
net = AlexNetHip(B, 3)
net.load_params((torch.randn(net.n_params) * 0.1).numpy())
x = torch.rand((B, 3, 224, 224), device="cuda")
labels = torch.randint(0, 3, (B,), dtype=torch.int32, device="cuda")
for _ in range(5):
    net.train_step(x, labels, 1e-3)
torch.cuda.synchronize()
n = 50
t0 = time.perf_counter()
for _ in range(n):
    net.train_step(x, labels, 1e-3)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/n:.3f} ms/step   total {1e3*(t2-t0)/n:.3f} ms/step")
