"""conv_layer_2's data gradient alone (plain / fused ReLU', unprepared / prepared filters), kernel-timer averages"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from cnn_amd import capi
c=capi.Conv2d(256,16,55,55,32,3,2,0)
dy=torch.rand((256,32,27,27),device="cuda"); w=torch.rand((32,16,3,3),device="cuda")*0.1; b=torch.zeros(32,device="cuda")
rb=torch.rand((256,16,55,55),device="cuda")-0.3; dx=torch.empty_like(rb)
pf,pd=c.prepared_buffers("cuda"); capi.prepare_filters([c],[w],[b],[pf],[pd])
def run():
    c.backward_data(dy,w,dx); c.backward_data_relu(dy,w,rb,dx); c.backward_data_relu(dy,None,rb,dx,prepared_dgrad=pd)
for _ in range(3): run()
torch.cuda.synchronize(); capi.kernel_timing(1)
for _ in range(10): run()
for k,(n,ms) in capi.kernel_timing_report().items(): print(f"{ms/n*1e3:8.1f} us x{n:3d}  {k}")
