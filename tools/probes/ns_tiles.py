import sys, os, json
sys.path.insert(0, "/root/repo")
import torch, bench
from cnn_amd import capi
print(json.dumps(bench.conv_ns_bench(torch, capi)))
for rd in ("1", "0"):
    for cfg in (None, 200, 227, 228, 229):
        capi.set_option("FWD_RD", rd); capi.set_option("DGRAD_RD", rd)
        capi.set_option("IGEMM_CFG", None if cfg is None else str(cfg))
        case = (256, 64, 112, 112, 128, 3, 1, 0)
        conv = capi.Conv2d(*case)
        g = torch.Generator(device="cuda").manual_seed(1)
        x = torch.rand((256, 64, 112, 112), generator=g, device="cuda"); w = torch.randn((128, 64, 3, 3), generator=g, device="cuda") * 0.1
        b = torch.randn((128,), generator=g, device="cuda") * 0.1
        y = torch.empty(conv.out_shape(), device="cuda"); dy = torch.rand(conv.out_shape(), generator=g, device="cuda"); dx = torch.empty_like(x)
        for _ in range(2): conv.forward(x, w, b, y); conv.backward_data(dy, w, dx)
        torch.cuda.synchronize(); capi.kernel_timing(1)
        for _ in range(3): conv.forward(x, w, b, y); conv.backward_data(dy, w, dx)
        rep = capi.kernel_timing_report(); capi.kernel_timing(0)
        fl = 2.0 * 256 * 128 * 110 * 110 * 64 * 9
        for k, (c, ms) in rep.items():
            if ms / c > 1: print("rd", rd, "cfg", cfg, k.split("|")[0], round(ms / c, 3), "ms", round(fl / (ms / c) / 1e9 / 157.3 * 100, 1), "%")
        del x, y, dy, dx
