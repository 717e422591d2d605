#!/usr/bin/env python3
"""Run-to-run determinism of whole train steps at the BASELINE batches (the C++ Layer API, default settings: side streams, deferred kernels,
recorded reductions, fused tails): the same N steps twice from the same parameters on fresh containers -- parameters and loss must be
bit-identical.  An unordered pair of kernels (a missing event) shows up here as differing bits on some run.  Odd runs share the chip with a
memory-bound kernel on another stream (other timings, other interleavings).  usage: step_determinism.py [steps=8] [runs=3]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from cnn_amd import hostapi
from cnn_amd import stacks as S

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
bad = 0
for which, B in (("alexnet", 256), ("vgg11", 32), ("resnet18", 64)):
    spec = S.STACKS[which]()
    layout = S.walk(spec, 3, 224, 224)
    p0 = S.he_init(layout, 5)
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.rand((B, 3, 224, 224), generator=g, device="cuda")
    labels = (torch.arange(B, device="cuda") % 3).to(torch.int32)
    ref = None
    from cnn_amd import capi
    side = torch.cuda.Stream()
    noise_a = torch.rand((32 << 20,), device="cuda")
    noise_b = torch.empty_like(noise_a)
    for r in range(runs):
        net = hostapi.HostSequential(spec, (3, 224, 224))
        net.set_params(p0)
        losses = []
        for _ in range(steps):
            if r % 2 == 1:
                with torch.cuda.stream(side):
                    capi.relu_forward(noise_a, noise_b)
            net.train_step(x, labels, 1e-3)
            if r % 2 == 0:
                losses.append(net.last_loss())  # (every other run without an observer between the steps: other stream interleavings)
        p = net.get_params()
        d = net.input_delta((B, 3, 224, 224))
        net.close()
        if ref is None:
            ref = (p, d)
        else:
            same = np.array_equal(p.view(np.uint32), ref[0].view(np.uint32)) and np.array_equal(d.view(np.uint32), ref[1].view(np.uint32))
            bad += not same
            print(f"{which} B{B}: run {r} vs run 0 after {steps} steps: {'bit-identical' if same else 'DIFFERENT  <-- FAIL'}")
print("DETERMINISM", "OK" if bad == 0 else f"FAILED ({bad})")
sys.exit(1 if bad else 0)
