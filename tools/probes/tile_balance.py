#!/usr/bin/env python3
"""Tile-count quantisation of the implicit GEMM at batch 64 (profiles/NOTEBOOK.md 4.30): time forward / data gradient of the ResNet-shaped stack's 3x3 layers with
the tuned tile and with the 13- / 7-block tiles (CNN_AMD_IGEMM_CFG), and check the results against the default tile's.
usage: tile_balance.py [cfg ids ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from cnn_amd import capi

CFGS = [None] + [int(a) for a in sys.argv[1:]]
B = int(os.environ.get("B", 64))
SHAPES = [(64, 56, 56, 64, 3, 1, 1), (128, 28, 28, 128, 3, 1, 1), (256, 14, 14, 256, 3, 1, 1), (512, 7, 7, 512, 3, 1, 1)]
if os.environ.get("SHAPES") == "vgg":  # B=128: the north-star layer and the VGG-shaped stack's
    SHAPES = [(64, 112, 112, 128, 3, 1, 0), (64, 112, 112, 128, 3, 1, 1), (128, 56, 56, 256, 3, 1, 1), (256, 56, 56, 256, 3, 1, 1), (256, 28, 28, 512, 3, 1, 1),
              (512, 28, 28, 512, 3, 1, 1), (512, 14, 14, 512, 3, 1, 1)]
if os.environ.get("SHAPES") == "first":  # the VGG-shaped stack's first layer
    SHAPES = [(3, 224, 224, 64, 3, 1, 1)]
for (Ci, H, W, Co, k, s, pad) in SHAPES:
    case = (B, Ci, H, W, Co, k, s, pad)
    conv = capi.Conv2d(*case)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand((B, Ci, H, W), generator=g, device="cuda")
    w = torch.randn((Co, Ci, k, k), generator=g, device="cuda") * 0.1
    b = torch.randn((Co,), generator=g, device="cuda") * 0.1
    dy = torch.rand(conv.out_shape(), generator=g, device="cuda") * 2 - 1
    flops = 2.0 * B * Co * conv.Ho * conv.Wo * Ci * k * k
    ref = {}
    for op in ("fwd", "dgrad"):
        for cfg in CFGS:
            capi.set_option("IGEMM_CFG", None if cfg is None else str(cfg))
            out = torch.empty(conv.out_shape(), device="cuda") if op == "fwd" else torch.empty_like(x)
            run = (lambda: conv.forward(x, w, b, out)) if op == "fwd" else (lambda: conv.backward_data(dy, w, out))
            try:
                run()
                torch.cuda.synchronize()
                capi.kernel_timing(1)
                for _ in range(5):
                    run()
                rep = capi.kernel_timing_report()
                capi.kernel_timing(0)
            except Exception as e:
                capi.kernel_timing(0)
                print(f"   {op} cfg {cfg}: {str(e)[:100]}")
                continue
            if cfg is None:
                ref[op] = out.clone()
            err = float((out - ref[op]).abs().max() / ref[op].abs().max())
            for key, (cnt, ms) in rep.items():
                if "prep" in key:
                    continue
                print(f"{case} {op:5s} cfg {str(cfg):>4s} {ms / cnt * 1e3:8.1f} us {flops / (ms / cnt) / 1e9:7.1f} TF  diff {err:.1e}  {key.split('|')[0]}")
    capi.set_option("IGEMM_CFG", None)
