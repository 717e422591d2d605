import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from cnn_amd import hostapi, capi
B=256
x=torch.rand((B,3,224,224),device="cuda")
labels=(np.arange(B)%3).astype(np.int32)
if len(sys.argv) > 1 and sys.argv[1] == "pool":
    hostapi.load().cnnh_set_fuse_pool_block(1)
net=hostapi.HostAlexNet(3)
rng=np.random.default_rng(0)
net.set_params((rng.standard_normal(net.n_params)*0.02).astype(np.float32))
for _ in range(5): net.train_step_device(x, labels, 1e-3)
torch.cuda.synchronize()
t0=time.perf_counter()
K=30
for _ in range(K): net.train_step_device(x, labels, 1e-3)
torch.cuda.synchronize()
dt=(time.perf_counter()-t0)/K
print("host mirror train_step_device: %.3f ms/step  %.0f img/s" % (dt*1e3, B/dt))
