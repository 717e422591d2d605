import sys; sys.path.insert(0,"/root/repo")
import torch
from cnn_amd import capi
c=capi.Conv2d(256,16,55,55,32,3,2,0)
dy=torch.rand((256,32,27,27),device="cuda"); w=torch.rand((32,16,3,3),device="cuda")*0.1
rb=torch.rand((256,16,55,55),device="cuda")-0.3; dx=torch.empty_like(rb)
for _ in range(3):
    c.backward_data(dy,w,dx); c.backward_data_relu(dy,w,rb,dx)
torch.cuda.synchronize(); capi.kernel_timing(1)
for _ in range(10):
    c.backward_data(dy,w,dx); c.backward_data_relu(dy,w,rb,dx)
for k,(n,ms) in capi.kernel_timing_report().items(): print(f"{ms/n*1e3:8.1f} us  {k}")
