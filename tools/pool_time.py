"""Time MaxPool2D(2,2) forward / backward (+ReLU) alone on the pool sites of the stacks (isolated launches, back to back)."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cnn_amd import capi  # noqa: E402

SHAPES = [(64, 64, 112, 112), (128, 64, 224, 224), (128, 128, 112, 112), (128, 256, 56, 56), (128, 512, 28, 28), (128, 512, 14, 14), (256, 16, 111, 111), (3, 5, 17, 9), (2, 3, 7, 66)]


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


for shp in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.rand(shp, device="cuda", generator=g) - 0.3
    y, mask = capi.maxpool_forward(x, 2, 2)
    dy = torch.rand(y.shape, device="cuda", generator=g)
    dx = torch.empty_like(x)
    mb = x.numel() * 4 / 1e6
    tf = timed(lambda: capi.maxpool_forward(x, 2, 2))
    tb = timed(lambda: capi.maxpool_backward(dy, mask, shp, 2, 2, dx))
    tr = timed(lambda: capi.maxpool_backward_relu(dy, mask, y, shp, 2, 2, dx))
    h = hashlib.sha1()
    for t in (y, mask, dx, capi.maxpool_backward(dy, mask, shp, 2, 2)):
        h.update(t.cpu().numpy().tobytes())
    print(f"{shp}: fwd {tf:7.1f} us ({1.5 * mb / tf:5.2f} TB/s)  bwd {tb:7.1f} us ({1.5 * mb / tb:5.2f} TB/s)  "
          f"bwd+relu {tr:7.1f} us ({1.75 * mb / tr:5.2f} TB/s)  digest {h.hexdigest()[:10]}")
