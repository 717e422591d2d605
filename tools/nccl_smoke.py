"""RCCL sanity check on one GPU: the collective calls bench.py makes for N > 1, with a 1-rank group."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from cnn_amd import dp
from cnn_amd.pynet import AlexNetHip

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
net = AlexNetHip(8, 3)
net.load_params((torch.randn(net.n_params) * 0.1).numpy())
x = torch.rand((8, 3, 224, 224), device="cuda"); labels = (torch.arange(8, device="cuda") % 3).to(torch.int32)
net.forward(x); net.loss_backward_seed(labels); net.backward(net.delta)
g0 = net.grads.clone()
dist.all_reduce(net.grads)            # what dp.allreduce_grads does for world > 1
dist.barrier(); torch.cuda.synchronize()
assert torch.equal(g0, net.grads)
t = torch.tensor([1.5], device="cuda", dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
net.update(1e-3, 1.0)
print("NCCL_OK", float(t.item()))
dist.destroy_process_group()
