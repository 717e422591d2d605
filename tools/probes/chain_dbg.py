import numpy as np, torch
from cnn_amd import hostapi, capi
capi.set_option("CHAIN_DBG", "1")
B=256
x=torch.rand(B,3,224,224,device="cuda"); l=(torch.arange(B,device="cuda")%3).int()
net=hostapi.HostAlexNet(3)
for i in range(4):
    net.train_step(x,l,1e-3)
    torch.cuda.synchronize()
    print("---- step", i, flush=True)
net.close()
