#!/usr/bin/env python3
"""Sweep the implicit-GEMM tile configurations (CNN_AMD_IGEMM_CFG) over the convolution shapes of a stack: which config the
planner should pick per shape.  usage: sweep_igemm.py vgg11|resnet18 [batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cnn_amd import capi
from cnn_amd.stacks import conv_geometries

name = sys.argv[1]
batch = int(sys.argv[2]) if len(sys.argv) > 2 else {"vgg11": 128, "resnet18": 64}[name]
CFGS = [None] + list(range(200, 227)) + [0, 1, 2, 3, 4, 20, 21, 22, 23]
seen = set()
for (Ci, H, W, Co, k, s, pad) in conv_geometries(name):
    case = (batch, Ci, H, W, Co, k, s, pad)
    if case in seen or Ci < 16:
        continue
    seen.add(case)
    conv = capi.Conv2d(*case)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand((batch, Ci, H, W), generator=g, device="cuda")
    w = torch.randn((Co, Ci, k, k), generator=g, device="cuda") * 0.1
    b = torch.randn((Co,), generator=g, device="cuda") * 0.1
    y = torch.empty(conv.out_shape(), device="cuda")
    dy = torch.rand(conv.out_shape(), generator=g, device="cuda") * 2 - 1
    dx = torch.empty_like(x)
    flops = 2.0 * batch * Co * conv.Ho * conv.Wo * Ci * k * k
    for op in ("fwd", "dgrad"):
        res = []
        for cfg in CFGS:
            capi.set_option("IGEMM_CFG", None if cfg is None else str(cfg))
            try:
                run = (lambda: conv.forward(x, w, b, y)) if op == "fwd" else (lambda: conv.backward_data(dy, w, dx))
                run()
                torch.cuda.synchronize()
                capi.kernel_timing(1)
                run(); run()
                rep = capi.kernel_timing_report()
                capi.kernel_timing(0)
            except Exception as e:  # config not applicable to this shape
                capi.kernel_timing(0)
                continue
            for key, (cnt, ms) in rep.items():
                if "prep" in key:
                    continue
                res.append((ms / cnt, cfg, key.split("|")[0]))
        capi.set_option("IGEMM_CFG", None)
        res.sort(key=lambda r: r[0])
        dflt = [r for r in res if r[1] is None][0]
        print(f"{case} {op}: default {dflt[0] * 1e3:.0f} us {flops / dflt[0] / 1e9:.1f} TF {dflt[2]} | best: " +
              "; ".join(f"cfg {r[1]} {r[0] * 1e3:.0f} us {flops / r[0] / 1e9:.1f} TF {r[2]}" for r in res[:3]))
    del x, y, dy, dx, conv
    torch.cuda.empty_cache()
