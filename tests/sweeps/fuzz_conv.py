#!/usr/bin/env python3
"""Random-geometry sweep of the DEFAULT dispatch of cnn_conv2d_forward / _backward_data(_relu) / _backward_weight against the CPU oracle
(oracle/pyoracle.py; conv2d.cpp:69-199): 3x3 layers of stride 1 / 2 and padding 0 / 1 with random batch, channel counts and plane sizes --
the shapes between the test suite's hand-picked cases, where the runtime-size kernels of round 6 (conv_rows_any, wgrad_sp_any) and the
per-width instances meet.  Tensor-normalised 1e-4 like tests/util.py; prints which kernel served each pass.
usage: fuzz_conv.py [cases=60] [seed=1] [widths]
widths: plane widths of the per-width kernel instances (7 / 14 / 28 / 56 / 112 / 110, the reference net's 55 / 27 / 13 and 224-wide first
layers with 3 channels) with random heights, batches and channel counts: partial tiles, ragged row groups and odd batches of conv_rows,
conv_rows_s2, wgrad_sp, wgrad_sp2, the first-layer and stem kernels"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from cnn_amd import capi
from oracle import pyoracle as O

widths = "widths" in sys.argv[1:]
batches = "batches" in sys.argv[1:]  # square BASELINE planes (7 / 14 / 28 / 56, pad 1, stride 1 / 2) at batches of 1 .. 140: units per workgroup, ragged last units, tall units
wild = "wild" in sys.argv[1:]  # anything the desc admits: odd k 1..7, strides 1..4, padding 0..4 (also > k/2), 1..40 channels (odd counts, 1, 3), planes 1..40; geometries
#                                  without an output pixel must be REFUSED (a status, not a crash)
sys.argv = [a for a in sys.argv if a not in ("widths", "batches", "wild")]
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def rel(a, ref):
    return float(np.abs(np.asarray(a, np.float64) - ref).max() / max(np.abs(ref).max(), 1e-30))


worst, bad = 0.0, 0
for it in range(n_cases):
    s = 1 if rs.rand() < 0.75 else 2
    pad = int(rs.randint(0, 2))
    B = int(rs.randint(1, 4))
    Ci, Co = int(rs.choice([3, 8, 16, 24, 32, 40, 64, 72, 96])), int(rs.choice([8, 16, 24, 32, 48, 64, 80, 128]))
    H, W = int(rs.randint(5, 64)), int(rs.randint(5, 64))
    if rs.rand() < 0.3:
        W = H
    k = 3
    if widths:
        W = int(rs.choice([7, 14, 28, 56, 112, 110, 55, 27, 13, 224]))
        B = int(rs.randint(1, 7))
        if W == 224:  # first layers: 3 channels
            Ci = 3
            k, s, pad, Co = [(3, 2, 0, 16), (7, 2, 3, int(rs.choice([8, 64, 72]))), (3, 1, 1, int(rs.choice([16, 64])))][int(rs.randint(0, 3))]
            H = int(rs.choice([224, 64, 37, 96]))
            B = int(rs.randint(1, 4))
        else:
            H = W if rs.rand() < 0.6 else int(rs.randint(max(3, W // 8), W + 9))
            if W >= 110:
                H = int(rs.randint(3, 30))
            if W in (55, 27, 13):
                s, pad = 2, 0
            elif W == 110:
                s, pad = 1, int(rs.choice([0, 1, 2]) if False else rs.randint(0, 2))
            else:
                pad = 1 if rs.rand() < 0.85 else 0
            Ci, Co = int(rs.choice([16, 32, 40, 64, 72, 128, 136])), int(rs.choice([32, 48, 64, 128, 160]))
            if W >= 110:
                Ci, Co = min(Ci, 64), min(Co, 128)
        if (H + 2 * pad - k) // s + 1 < 1:
            H = k
    if batches:
        k, pad = 3, 1
        W = H = int(rs.choice([7, 7, 14, 14, 28, 56]))
        s = 1 if rs.rand() < 0.7 else 2
        Ci, Co = int(rs.choice([32, 64, 128])), int(rs.choice([32, 64, 128, 192]))
        B = int(rs.randint(1, 141))
        while B * H * W * Ci * Co > 6e9:
            B = max(1, B // 2)
    if wild:
        k, s, pad = int(rs.choice([1, 3, 5, 7])), int(rs.randint(1, 5)), int(rs.randint(0, 5))
        B, Ci, Co = int(rs.randint(1, 4)), int(rs.randint(1, 41)), int(rs.randint(1, 41))
        H, W = int(rs.randint(1, 41)), int(rs.randint(1, 41))
        if (H + 2 * pad - k) // s + 1 < 1 or (W + 2 * pad - k) // s + 1 < 1 or H + 2 * pad < k or W + 2 * pad < k:
            try:
                conv = capi.Conv2d(B, Ci, H, W, Co, k, s, pad)
                xd = torch.zeros((B, Ci, H, W), device="cuda")
                conv.forward(xd, torch.zeros((Co, Ci, k, k), device="cuda"), torch.zeros((Co,), device="cuda"))
                print(f"{str((B, Ci, H, W, Co, k, s, pad)):42s} no output pixel, and the call was ACCEPTED   <-- FAIL")
                bad += 1
            except (capi.CnnAmdError, AssertionError, ValueError, RuntimeError) as e:
                print(f"{str((B, Ci, H, W, Co, k, s, pad)):42s} refused: {str(e)[:90]}")
            continue
    case = (B, Ci, H, W, Co, k, s, pad)
    x = rs.rand(B, Ci, H, W).astype(np.float32)
    w = (rs.standard_normal((Co, Ci, k, k)) * 0.1).astype(np.float32)
    b = (rs.standard_normal(Co) * 0.1).astype(np.float32)
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    dy = (rs.rand(B, Co, Ho, Wo) * 2 - 1).astype(np.float32)
    xp = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad))) if pad else x
    y_ref = O.conv2d_forward(xp, w, b, s)
    gw_ref, gb_ref, dxp = O.conv2d_backward(xp, dy, w, s)
    dx_ref = dxp[:, :, pad:pad + H, pad:pad + W] if pad else dxp
    conv = capi.Conv2d(*case)
    if os.environ.get("FUZZ_AUTOTUNE"):  # let the library measure and pin the implicit-GEMM tile of this geometry first (cnn_conv2d_autotune)
        conv.autotune()
    xd, wd, bd, dyd = (torch.from_numpy(t).cuda() for t in (x, w, b, dy))
    # every output lives between two guard zones of a larger allocation: a kernel that stores outside its tensor shows up there
    GUARD, CANARY = 4096, -12345.5
    guards = []

    MIS = int(os.environ.get("FUZZ_MISALIGN", "0"))  # every tensor MIS floats off a 16-byte boundary: plain pointers are all the C ABI asks for

    def guarded(shape):
        n = int(np.prod(shape))
        big = torch.full((n + 2 * GUARD + MIS,), CANARY, dtype=torch.float32, device="cuda")
        guards.append((big[MIS:], n))
        return big[GUARD + MIS:GUARD + MIS + n].view(*shape)

    if MIS:
        def shifted(t):
            big = torch.empty((t.numel() + MIS,), dtype=torch.float32, device="cuda")
            big[MIS:] = t.reshape(-1)
            return big[MIS:].view(*t.shape)

        xd, wd, bd, dyd = shifted(xd), shifted(wd), shifted(bd), shifted(dyd)

    capi.kernel_timing(1)
    y = conv.forward(xd, wd, bd, y=guarded((B, Co, Ho, Wo)))
    dx = conv.backward_data(dyd, wd, dx=guarded((B, Ci, H, W)))
    relu_in = capi.relu_forward((xd - 0.5).contiguous())
    if MIS:
        relu_in = shifted(relu_in)
    dxm = guarded((B, Ci, H, W))
    conv.backward_data_relu(dyd, wd, relu_in, dxm)
    gw, gb = conv.backward_weight(xd, dyd, float(B), gw=guarded((Co, Ci, k, k)), gb=guarded((Co,)))
    torch.cuda.synchronize()
    overrun = sum(int((big[:GUARD] != CANARY).sum().item()) + int((big[GUARD + n:] != CANARY).sum().item()) for big, n in guards)
    names = [k.split("|")[0] for k in capi.kernel_timing_report() if not any(t in k for t in ("prep", "reduce", "relu_f", "pack"))]
    capi.kernel_timing(0)
    errs = {"y": rel(y.cpu().numpy(), y_ref), "dx": rel(dx.cpu().numpy(), dx_ref),
            "dx_relu": rel(dxm.cpu().numpy(), np.where(relu_in.cpu().numpy() <= 0, np.float32(0), dx_ref)),
            "gw": rel(gw.cpu().numpy(), gw_ref), "gb": rel(gb.cpu().numpy(), gb_ref)}
    if overrun:
        errs["STORES OUTSIDE AN OUTPUT TENSOR (guard floats changed)"] = float(overrun)
    e = max(errs.values())
    worst = max(worst, e)
    flag = "" if e <= 1e-4 else "   <-- FAIL " + str({k: f"{v:.2e}" for k, v in errs.items() if v > 1e-4})
    bad += e > 1e-4
    print(f"{str(case):42s} {e:.2e}  {' '.join(sorted(set(names)))}{flag}")
print(f"FUZZ {'OK' if bad == 0 else 'FAILED'}: {n_cases} geometries, worst tensor-normalised error {worst:.2e}, {bad} above 1e-4")
sys.exit(1 if bad else 0)
