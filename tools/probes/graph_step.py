"""Experiment: the train step replayed from a HIP graph (G steps per graph, deferred conv1 dgrad flushed at the graph's end)
vs the eager stream launches bench.py times.  Usage: python tools/probes/graph_step.py [G] [replays]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch

from cnn_amd.pynet import AlexNetHip

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4
R = int(sys.argv[2]) if len(sys.argv) > 2 else 50
B = 256
net = AlexNetHip(B, 3, defer_input_grad=True, fuse_pool=True)
rs = np.random.RandomState(1234)
p0 = (rs.standard_normal(net.n_params) * 0.1).astype(np.float32)
net.load_params(p0)
x = torch.rand((B, 3, 224, 224), device="cuda")
labels = (torch.arange(B, device="cuda") % 3).to(torch.int32)


def eager(n):
    for _ in range(n):
        net.train_step(x, labels, 1e-3)
    net.flush()
    torch.cuda.synchronize()


eager(8)
t0 = time.perf_counter(); eager(G * R); te = time.perf_counter() - t0
print("eager  %.1f us/step" % (te / (G * R) * 1e6))

net.load_params(p0)
eager(2 * G)
ref = net.params.clone()
net.load_params(p0)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    eager(G)  # same parity / buffer state as the capture start
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    for _ in range(G):
        net.train_step(x, labels, 1e-3)
    net.flush()
torch.cuda.synchronize()
g.replay()
torch.cuda.synchronize()
print("params identical to eager after 2G steps:", torch.equal(ref, net.params))
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(R):
    g.replay()
torch.cuda.synchronize()
tg = time.perf_counter() - t0
print("graph  %.1f us/step (G=%d)" % (tg / (G * R) * 1e6, G))
