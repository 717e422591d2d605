import sys; sys.path.insert(0,"/root/repo")
import torch
from cnn_amd import capi
c=capi.Conv2d(256,3,224,224,16,3,2,0)
x=torch.rand((256,3,224,224),device="cuda"); w=torch.rand((16,3,3,3),device="cuda")*0.1; b=torch.rand(16,device="cuda")
pooled=torch.empty((256,16,55,55),device="cuda"); mask=torch.empty((256,16,55,55),dtype=torch.int32,device="cuda")
for _ in range(3): c.relu_maxpool2_forward(x,w,b,pooled,mask)
torch.cuda.synchronize(); capi.kernel_timing(1)
for _ in range(20): c.relu_maxpool2_forward(x,w,b,pooled,mask)
for k,(n,ms) in capi.kernel_timing_report().items(): print(f"{ms/n*1e3:8.1f} us  {k}")
