"""Time the BatchNorm2D kernels on the BN sites of a stack (default: the ResNet-18-shaped stack at B=64) and print a
digest of every output, so that two builds of the library can be compared bit for bit:
    python tools/bn_time.py                      (the in-tree library)
    CNN_AMD_LIB=/path/to/other.so python tools/bn_time.py
"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cnn_amd import capi  # noqa: E402

if os.environ.get("CNN_AMD_LIB"):
    capi.LIB_PATH = os.environ["CNN_AMD_LIB"]

SHAPES = [(64, 64, 112, 112), (64, 64, 56, 56), (64, 128, 28, 28), (64, 256, 14, 14), (64, 512, 7, 7), (5, 7, 33, 37)]


def timed(fn, n=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def digest(*ts):
    h = hashlib.sha1()
    for t in ts:
        h.update(t.detach().cpu().numpy().tobytes())
    return h.hexdigest()[:12]


for shp in SHAPES:
    B, C, H, W = shp
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand(shp, device="cuda", generator=g) * 2 - 1
    dy0 = torch.rand(shp, device="cuda", generator=g) * 2 - 1
    gamma = torch.rand(C, device="cuda", generator=g) + 0.5
    beta = torch.rand(C, device="cuda", generator=g) - 0.5
    mm, mv = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    y = torch.empty_like(x)
    gg, gb = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    bn = capi.BatchNorm2d(B, C, H, W)
    bn.forward(x, gamma, beta, mm, mv, y)
    dy = dy0.clone()
    bn.backward(x, dy, gamma, gg, gb)
    dg = digest(y, bn.saved_mean, bn.saved_var, mm, mv, dy, gg, gb)
    tf = timed(lambda: bn.forward(x, gamma, beta, mm, mv, y))
    tb = timed(lambda: bn.backward(x, dy, gamma, gg, gb))
    mb = x.numel() * 4 / 1e6
    print(f"{shp}: fwd {tf:7.1f} us ({4 * mb / tf :5.2f} TB/s)  bwd {tb:7.1f} us ({5 * mb / tb:5.2f} TB/s)  digest {dg}")
