import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
from cnn_amd import hostapi
B = 256
net = hostapi.HostAlexNet(3)
net.set_params((np.random.RandomState(1).standard_normal(net.n_params) * 0.1).astype(np.float32))
x = torch.rand((B, 3, 224, 224), device="cuda")
labels = (torch.arange(B, device="cuda") % 3).to(torch.int32)
for _ in range(6):
    net.train_step(x, labels, 1e-3)
torch.cuda.synchronize()
net.close()
