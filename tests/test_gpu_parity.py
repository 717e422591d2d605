"""GPU parity: every HIP entry point (called through the C ABI) against the CPU oracle on the same seeded inputs.

Bars (north_star): bit-exact for MaxPool values/argmax/backward, ReLU and SGD; tensor-normalised 1e-4 for conv /
linear activations and gradients (tests/util.py: max|a-b| <= 1e-4 * max|ref|).
"""
import json
import os

import numpy as np
import pytest

from oracle import pyoracle as O
from tests.util import REL_TOL, assert_close, assert_close_arbitrated, normal_scaled, rel_err, uniform01, uniform_pm1

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a device"
    from cnn_amd import capi

    arch = capi.load().cnn_amd_device_arch().decode()
    assert arch == "gfx950", f"built for gfx950, running on {arch}"
    return torch


def dev(T, a):
    return T.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


# (B, Ci, H, W, Co, k, s, pad): SURVEY 8(c) cases + every reference-net layer + the north-star shape
CONV_CASES = [
    (2, 3, 9, 9, 4, 3, 2, 0),
    (2, 16, 55, 55, 32, 3, 2, 0),
    (3, 8, 12, 12, 8, 3, 1, 0),
    (2, 8, 9, 9, 4, 1, 2, 0),
    (2, 5, 13, 11, 7, 5, 2, 0),
    (2, 3, 224, 224, 16, 3, 2, 0),   # conv_layer_1 (alexnet.cpp:12)
    (3, 32, 27, 27, 64, 3, 2, 0),    # conv_layer_3
    (5, 64, 13, 13, 128, 3, 2, 0),   # conv_layer_4 (tiles span several images)
    (2, 64, 112, 112, 128, 3, 1, 0), # north-star shape at B=2
    (2, 6, 10, 10, 40, 3, 1, 1),     # padding extension
    (1, 20, 17, 19, 130, 3, 3, 0),   # stride 3, Co not a tile multiple
    (2, 7, 8, 8, 5, 3, 2, 1),        # stride 2 with padding
    # BASELINE configs 4 / 5 layer classes (VGG-11 / ResNet-18 shaped) at a small batch
    (1, 3, 64, 64, 8, 7, 2, 3),      # ResNet stem: 7x7 stride 2 pad 3 (49 taps: LDS fallback chain)
    (2, 8, 14, 14, 16, 1, 2, 0),     # ResNet downsample: 1x1 stride 2
    (2, 16, 28, 28, 40, 3, 2, 1),    # ResNet stage entry: 3x3 stride 2 pad 1 (masked whole-image staging + tap skipping)
    (2, 3, 32, 32, 16, 3, 1, 1),     # VGG first layer: 3 -> C, stride 1 pad 1
    (2, 64, 14, 14, 128, 3, 1, 1),   # VGG deep layer: 3x3 stride 1 pad 1, small image
    (1, 3, 224, 224, 64, 7, 2, 3),   # the ResNet stem at full resolution (row staging of 230-float rows)
    (3, 3, 38, 44, 72, 7, 2, 3),     # conv_stem.hip: two channel blocks (64 + 8), 19 x 22 outputs: partial row group, one pixel tile
    (2, 3, 20, 260, 32, 7, 2, 3),    # ... 130 output columns: two column blocks of 128
    (2, 64, 14, 14, 128, 3, 2, 1),   # stage entry of the ResNet-shaped stack: the 16x16x4 stride-2 data gradient with pad 1
    (3, 64, 9, 11, 128, 3, 2, 1),    # ... odd sizes: first / last padded row and column fall outside the image
    (2, 3, 22, 264, 72, 7, 2, 3),    # ... W % 8 == 0: also the stem weight-gradient kernel, 132 columns (128 + 4), 64 + 8 channels
    (3, 32, 9, 11, 64, 3, 1, 0),     # stride-1 register-direct data gradient: one tile of 32 channels, borders everywhere
    (2, 64, 8, 7, 128, 3, 1, 0),     # ... two tiles per wave
    (3, 3, 70, 130, 24, 3, 1, 1),    # thin-input data gradient (conv_dgrad_thin.hip): three column segments, ragged last band
    (2, 3, 9, 5, 7, 3, 1, 0),        # ... without padding, image smaller than one patch
    (2, 32, 28, 30, 64, 3, 2, 0),    # stride-2 register-direct data gradient on even sizes (last input row / column uncovered)
    (1, 96, 9, 12, 128, 3, 2, 0),    # ... three 32-channel groups, tiny image
    (2, 128, 28, 28, 256, 1, 2, 0),  # conv_1x1.hip: the ResNet-shaped stack's downsample layer (1x1 stride 2) at a small batch
    (3, 32, 9, 11, 48, 1, 2, 0),     # ... odd sizes (last row / column off the sampled grid stay 0), Co below one row block, one K chunk
    (2, 64, 7, 7, 32, 1, 1, 0),      # ... stride 1 (no zero fill), two K chunks
    (2, 160, 10, 9, 144, 1, 2, 0),   # ... ragged everywhere: two ci blocks (128 + 32), two co blocks (128 + 16), five K chunks
]


def _conv_inputs(case, seed):
    B, Ci, H, W, Co, k, s, pad = case
    x = uniform01(seed, (B, Ci, H, W))
    w = normal_scaled(seed + 1, (Co, Ci, k, k))
    b = normal_scaled(seed + 2, (Co,))
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    dy = uniform_pm1(seed + 3, (B, Co, Ho, Wo))
    return x, w, b, dy


def _oracle_conv(case, x, w, b, dy):
    """oracle results; padding = reference conv on the zero-padded input (Tensor3D::pad, data_format.cpp:139-150)"""
    B, Ci, H, W, Co, k, s, pad = case
    xp = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad))) if pad else x
    y = O.conv2d_forward(xp, w, b, s)
    gw, gb, dxp = O.conv2d_backward(xp, dy, w, s)
    dx = dxp[:, :, pad : pad + H, pad : pad + W] if pad else dxp
    return y, gw, gb, dx


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "B%d_%dx%dx%d_to%d_k%ds%dp%d" % c)
def test_conv2d_mfma_vs_oracle(T, case):
    from cnn_amd import capi

    x, w, b, dy = _conv_inputs(case, 100)
    y_ref, gw_ref, gb_ref, dx_ref = _oracle_conv(case, x, w, b, dy)
    conv = capi.Conv2d(*case)
    xd, wd, bd, dyd = dev(T, x), dev(T, w), dev(T, b), dev(T, dy)
    assert_close(host(conv.forward(xd, wd, bd)), y_ref, REL_TOL, "forward")
    gw, gb = conv.backward_weight(xd, dyd, float(case[0]))
    assert_close(host(gw), gw_ref, REL_TOL, "weight grad")
    assert_close(host(gb), gb_ref, REL_TOL, "bias grad")
    assert_close(host(conv.backward_data(dyd, wd)), dx_ref, REL_TOL, "data grad")


@pytest.mark.parametrize("case", CONV_CASES[:5] + CONV_CASES[9:], ids=lambda c: "B%d_%dx%dx%d_to%d_k%ds%dp%d" % c)
def test_conv2d_im2col_fallback_vs_oracle(T, case):
    from cnn_amd import capi

    x, w, b, dy = _conv_inputs(case, 200)
    y_ref, gw_ref, gb_ref, dx_ref = _oracle_conv(case, x, w, b, dy)
    conv = capi.Conv2d(*case)
    xd, wd, bd, dyd = dev(T, x), dev(T, w), dev(T, b), dev(T, dy)
    assert_close(host(conv.forward_im2col(xd, wd, bd)), y_ref, REL_TOL, "im2col forward")
    gw, gb = conv.backward_weight_im2col(xd, dyd, float(case[0]))
    assert_close(host(gw), gw_ref, REL_TOL, "im2col weight grad")
    assert_close(host(gb), gb_ref, REL_TOL, "im2col bias grad")
    assert_close(host(conv.backward_data_im2col(dyd, wd)), dx_ref, REL_TOL, "im2col data grad")


RD_CASES = [
    (1, 1, 3, 3, 1, 3, 1, 0),        # one output pixel
    (1, 4, 5, 5, 3, 3, 2, 0),        # whole tensor smaller than one window: guarded path only
    (2, 7, 20, 37, 33, 3, 1, 0),     # Co = 33 (two row tiles), 64 columns (two tiles)
    (3, 18, 21, 35, 16, 3, 2, 0),    # 163 columns: six tiles -> 3-tile groups, ragged row tails
    (2, 40, 9, 9, 70, 3, 1, 0),      # runs of 8 pixels, 361 columns -> 4-tile groups
    (4, 12, 19, 19, 8, 3, 2, 0),     # Wo = 9: a full run and a 1-pixel tail per row, 109 columns -> 4 tiles
    (1, 2, 40, 70, 5, 3, 2, 0),      # one image: the last rows take the guarded path
    # padding 1 (VGG / ResNet-shaped stacks): border rows fetched from the centre row and masked, per-window live ranges
    (1, 1, 3, 3, 1, 3, 1, 1),        # 3x3 image: every tap row / column leaves the image somewhere
    (2, 7, 20, 37, 33, 3, 1, 1),     # stride 1: rows of 16 + 16 + 5 pixels, two row tiles
    (3, 18, 21, 35, 16, 3, 2, 1),    # stride 2, odd sizes: window starts at column -2, last pixel inside
    (2, 5, 8, 8, 40, 3, 2, 1),       # stride 2, even sizes (column 8 = first one outside), runs of 8
    (2, 40, 9, 9, 70, 3, 1, 1),      # runs of 8 pixels (Wo = 9 -> a 1-pixel tail run whose window starts inside)
    (2, 3, 30, 34, 64, 3, 1, 1),     # first VGG layer shape class: 27 columns, one tile
    (3, 130, 7, 7, 96, 3, 1, 1),     # 1170 columns: 37 tiles -> groups of 5 with dead tiles, three row tiles
    (2, 5, 28, 28, 40, 3, 1, 1),     # flattened runs (W = 28: a run of 16 crosses a row end; 784 = 49 whole runs per image)
    (1, 6, 19, 21, 8, 3, 1, 1),      # ... ragged: 399 pixels = 24 runs + 15, wraps at every position
    (2, 3, 56, 56, 16, 3, 1, 1),     # ... 27 columns, rows of 3.5 runs
]


@pytest.mark.parametrize("slow", [False, True], ids=["pipelined", "guarded"])
@pytest.mark.parametrize("case", RD_CASES + [CONV_CASES[1], CONV_CASES[6], CONV_CASES[7]], ids=lambda c: "B%d_%dx%dx%d_to%d_k%ds%dp%d" % c)
def test_conv2d_wgrad_register_direct(T, case, slow, lib_option):
    """conv_wgrad_rd.hip: the pipelined over-reading path and the guarded path agree with the oracle and, bit for bit,
    with each other (same MFMA order); the LDS-staged kernel (CNN_AMD_WGRAD_RD=0) is held to the same tolerance"""
    from cnn_amd import capi

    x, w, b, dy = _conv_inputs(case, 400)
    _, gw_ref, gb_ref, _ = _oracle_conv(case, x, w, b, dy)
    lib_option("WGRAD_SP", "0")  # (this test is about conv_wgrad_rd.hip; the small-plane kernel has test_conv2d_wgrad_small_planes)
    conv = capi.Conv2d(*case)
    xd, dyd = dev(T, x), dev(T, dy)
    if slow:
        lib_option("RD_SLOW", "1")
    gw, gb = conv.backward_weight(xd, dyd, float(case[0]))
    assert_close(host(gw), gw_ref, REL_TOL, "weight grad")
    assert_close(host(gb), gb_ref, REL_TOL, "bias grad")
    if slow:
        lib_option("RD_SLOW", None)
        gw2, gb2 = conv.backward_weight(xd, dyd, float(case[0]))
        assert np.array_equal(host(gw), host(gw2)) and np.array_equal(host(gb), host(gb2))
        lib_option("WGRAD_RD", "0")
        gw3, gb3 = conv.backward_weight(xd, dyd, float(case[0]))
        assert_close(host(gw3), gw_ref, REL_TOL, "LDS-staged weight grad")
        assert_close(host(gb3), gb_ref, REL_TOL, "LDS-staged bias grad")


SP_CASES = [
    (2, 64, 7, 7, 64, 3, 1, 1),      # 7x7: sample pairs, whole 64-channel blocks copied as they lie (16-byte DMA)
    (3, 64, 7, 7, 128, 3, 1, 1),     # ... odd batch: the last stage's second sample is staged as zeros; two co tiles
    (5, 40, 7, 7, 70, 3, 1, 1),      # ... partial channel tiles (4-byte DMA), three stages
    (2, 64, 14, 14, 128, 3, 1, 1),   # 14x14: two row blocks of 7, left / right halves of a row on the two k-slots
    (1, 24, 14, 14, 100, 3, 1, 1),   # ... partial tiles
    (2, 64, 28, 28, 64, 3, 1, 1),    # 28x28: 14 row blocks of 2, 16-byte DMA of 28-float rows
    (1, 20, 28, 28, 72, 3, 1, 1),
    (2, 64, 56, 56, 64, 3, 1, 1),    # 56x56: one output row per stage, four segments per half row
    (1, 12, 56, 56, 40, 3, 1, 1),
    (2, 32, 28, 14, 32, 3, 1, 1),    # planes need not be square: 28 rows of 14
    (1, 32, 14, 28, 64, 3, 1, 1),    # ... 14 rows of 28
    (2, 64, 3, 56, 64, 3, 1, 1),     # ... three rows of 56: every stage has a halo row outside the image
    (70, 64, 7, 7, 64, 3, 1, 1),     # more sample pairs than one workgroup per tile: several stages per pixel range
    (2, 64, 4, 112, 64, 3, 1, 1),    # 112-wide planes: 32-channel ci tiles, the two waves of a tile split a row's segments
    (1, 40, 6, 112, 100, 3, 1, 1),   # ... partial tiles (32 + 8 input channels, 64 + 36 output channels)
    (2, 64, 5, 112, 128, 3, 1, 0),   # ... pad 0 (the north-star geometry, conv2d.cpp:41-42): dy rows of 110 floats staged contiguously
    (1, 24, 9, 112, 70, 3, 1, 0),    # ... partial tiles
]


@pytest.mark.parametrize("unit", [0, 1], ids=["dma16", "dma4"])
@pytest.mark.parametrize("case", SP_CASES, ids=lambda c: "B%d_%dx%dx%d_to%d_k%ds%dp%d" % c)
def test_conv2d_wgrad_small_planes(T, case, unit, lib_option):
    """conv_wgrad_sp.hip (LDS-staged output-stationary kernel for 7x7 .. 56x56 planes, cpu/src/conv2d.cpp:117-159) against the
    oracle, with 16-byte and with 4-byte DMA staging, on pointers that are and are not 16-byte aligned"""
    from cnn_amd import capi

    lib_option("WGRAD_SP", "1")  # (also lifts the 32-channel floor of the default dispatch)
    if unit:
        lib_option("SP_UNIT", "1")
    x, w, b, dy = _conv_inputs(case, 430)
    _, gw_ref, gb_ref, _ = _oracle_conv(case, x, w, b, dy)
    conv = capi.Conv2d(*case)
    xd, dyd = dev(T, x), dev(T, dy)
    capi.kernel_timing(1)
    gw, gb = conv.backward_weight(xd, dyd, float(case[0]))
    T.cuda.synchronize()
    rep = capi.kernel_timing_report()
    capi.kernel_timing(0)
    assert any(k.startswith("wgrad_sp<") for k in rep), list(rep)
    assert_close(host(gw), gw_ref, REL_TOL, "weight grad")
    assert_close(host(gb), gb_ref, REL_TOL, "bias grad")
    if unit == 0:
        # the same tensors one float further: no 16-byte alignment -> the 4-byte DMA instance, same sums in the same order
        xs = T.empty(xd.numel() + 1, device="cuda")[1:].view_as(xd).copy_(xd)
        dys = T.empty(dyd.numel() + 1, device="cuda")[1:].view_as(dyd).copy_(dyd)
        assert xs.data_ptr() % 16 != 0
        gw2, gb2 = conv.backward_weight(xs, dys, float(case[0]))
        assert np.array_equal(host(gw), host(gw2)) and np.array_equal(host(gb), host(gb2))
    lib_option("WGRAD_SP", "0")
    gw3, gb3 = conv.backward_weight(xd, dyd, float(case[0]))
    assert_close(host(gw3), gw_ref, REL_TOL, "register-direct weight grad")


SP2_CASES = [
    (2, 64, 56, 56, 128, 3, 2, 1),   # stage entry of the ResNet-shaped stack: 28 row pairs, 32-channel ci tiles, two waves split a row's four segments
    (1, 40, 56, 56, 72, 3, 2, 1),    # ... partial tiles (32 + 8 input channels, 64 + 8 output channels)
    (2, 128, 28, 28, 64, 3, 2, 1),   # 28-wide: rows of 14 pixels = two segments, seven row pairs per plane
    (3, 24, 28, 28, 100, 3, 2, 1),   # ... partial tiles, odd batch
    (2, 64, 14, 14, 128, 3, 2, 1),   # 14-wide: the whole plane per stage, 7 output rows in 8 row slots (the last one selected away)
    (5, 72, 14, 14, 40, 3, 2, 1),    # ... partial tiles, more stages than workgroups per tile need
    (2, 32, 27, 27, 64, 3, 2, 0),    # the reference's own geometry (pad 0, conv2d.cpp:41-42): rows of 13 = segments of 7 + 6, 13 rows in pairs
    (3, 64, 13, 13, 128, 3, 2, 0),   # ... rows of 6, the whole plane per stage
    (70, 64, 14, 14, 64, 3, 2, 1),   # several stages per workgroup
]


@pytest.mark.parametrize("case", SP2_CASES, ids=lambda c: "B%d_%dx%dx%d_to%d_k%ds%dp%d" % c)
def test_conv2d_wgrad_small_planes_stride2(T, case, lib_option):
    """conv_wgrad_sp2.hip (the stride-2 sibling of the LDS-staged output-stationary weight gradient; cpu/src/conv2d.cpp:117-159 with the
    window walk of conv2d.cpp:76-77) against the oracle, on 16-byte aligned and unaligned tensors, and against the kernel it replaces"""
    from cnn_amd import capi

    lib_option("WGRAD_SP2", "2")  # (every instance, any channel count)
    x, w, b, dy = _conv_inputs(case, 470)
    _, gw_ref, gb_ref, _ = _oracle_conv(case, x, w, b, dy)
    conv = capi.Conv2d(*case)
    xd, dyd = dev(T, x), dev(T, dy)
    capi.kernel_timing(1)
    gw, gb = conv.backward_weight(xd, dyd, float(case[0]))
    T.cuda.synchronize()
    rep = capi.kernel_timing_report()
    capi.kernel_timing(0)
    assert any(k.startswith("wgrad_sp2<") for k in rep), list(rep)
    assert_close(host(gw), gw_ref, REL_TOL, "weight grad")
    assert_close(host(gb), gb_ref, REL_TOL, "bias grad")
    # the same tensors one float further: the DMA sources need 4-byte alignment only -- same sums in the same order
    xs = T.empty(xd.numel() + 1, device="cuda")[1:].view_as(xd).copy_(xd)
    dys = T.empty(dyd.numel() + 1, device="cuda")[1:].view_as(dyd).copy_(dyd)
    assert xs.data_ptr() % 16 != 0
    gw2, gb2 = conv.backward_weight(xs, dys, float(case[0]))
    assert np.array_equal(host(gw), host(gw2)) and np.array_equal(host(gb), host(gb2))
    # a tensor whose neighbours in memory are NaN: nothing outside it may reach a sum (rows staged past a plane's end are selected away)
    big = T.full((xd.numel() + 4096,), float("nan"), device="cuda")
    xn = big[2048 : 2048 + xd.numel()].view_as(xd).copy_(xd)
    bigd = T.full((dyd.numel() + 4096,), float("nan"), device="cuda")
    dyn = bigd[2048 : 2048 + dyd.numel()].view_as(dyd).copy_(dyd)
    gw4, gb4 = conv.backward_weight(xn, dyn, float(case[0]))
    assert np.array_equal(host(gw), host(gw4)) and np.array_equal(host(gb), host(gb4))
    lib_option("WGRAD_SP2", "0")
    gw3, gb3 = conv.backward_weight(xd, dyd, float(case[0]))
    assert_close(host(gw3), gw_ref, REL_TOL, "the replaced kernel's weight grad")


SPA_CASES = [
    (2, 64, 9, 23, 64, 3, 1, 0),      # pad 0 (conv2d.cpp:41-42): 21 x 7 outputs = one column block of 21, 4 row pairs (the last one a single row)
    (2, 40, 30, 50, 72, 3, 1, 1),     # pad 1: 50 outputs = two blocks of 28 (the second: 22 live columns), partial channel tiles, rows above / below the plane
    (1, 64, 7, 109, 64, 3, 1, 0),     # 107 outputs: four blocks of 28 (23 live in the last), odd row count
    (1, 32, 5, 222, 32, 3, 1, 0),     # 220 outputs: eight blocks, widths that are multiples of 4 (no fix-ups)
    (1, 32, 6, 223, 40, 3, 1, 1),     # 223 in / out: odd width, the straddling units of x AND dy fixed up
    (3, 64, 21, 21, 64, 3, 1, 0),     # 19 outputs: one block of 21, odd batch
    (2, 128, 13, 13, 64, 3, 1, 1),    # two ci tiles
    (5, 32, 3, 96, 32, 3, 1, 0),      # one output row (every pair's second row lies below the plane)
    (40, 32, 12, 12, 32, 3, 1, 1),    # many stages per workgroup: the stage walk crosses row pairs, column blocks and samples
    # 21-column blocks start anywhere modulo 4: the straddling unit of the LAST block is cut relative to that block's first column
    # (found by tests/sweeps/fuzz_conv.py late in round 6: the fix-up width had been taken from the row length alone)
    (2, 32, 32, 32, 32, 3, 1, 0),     # 30 outputs = 21 + 9: three floats of the last block's third unit lie behind the row
    (2, 96, 10, 42, 32, 3, 1, 0),     # 40 outputs = 21 + 19 (a row length that IS a multiple of 4: one float to zero all the same)
    (1, 96, 21, 41, 48, 3, 1, 0),     # 39 = 21 + 18
    (3, 72, 60, 60, 64, 3, 1, 1),     # 60 = 21 + 21 + 18, pad 1
    (3, 32, 25, 59, 64, 3, 1, 1),     # 59 = 21 + 21 + 17
    (1, 64, 6, 100, 64, 3, 1, 1),     # 100 = 4 x 21 + 16: the last block ends on a unit (nothing to fix), the row length does not matter
]


@pytest.mark.parametrize("case", SPA_CASES, ids=lambda c: "B%d_%dx%dx%d_to%d_k%ds%dp%d" % c)
def test_conv2d_wgrad_any_size_vs_oracle(T, case, lib_option):
    """conv_wgrad_sp_any.hip (round 6: the runtime-size member of the LDS-staged output-stationary weight gradient, any plane size,
    cpu/src/conv2d.cpp:117-159 / 41-42) against the oracle, on aligned and unaligned tensors, with NaN around the tensors (everything
    outside them is a zero in LDS, never a value), one workgroup per tile walking every stage, and against the kernel it replaces"""
    from cnn_amd import capi

    lib_option("WGRAD_SP_ANY", "2")  # (also lifts the 32-channel floor of the default dispatch)
    x, w, b, dy = _conv_inputs(case, 475)
    _, gw_ref, gb_ref, _ = _oracle_conv(case, x, w, b, dy)
    conv = capi.Conv2d(*case)
    xd, dyd = dev(T, x), dev(T, dy)
    capi.kernel_timing(1)
    gw, gb = conv.backward_weight(xd, dyd, float(case[0]))
    T.cuda.synchronize()
    rep = capi.kernel_timing_report()
    capi.kernel_timing(0)
    assert any(k.startswith("wgrad_sp_any<") for k in rep), list(rep)
    assert_close(host(gw), gw_ref, REL_TOL, "weight grad")
    assert_close(host(gb), gb_ref, REL_TOL, "bias grad")
    xs = T.empty(xd.numel() + 1, device="cuda")[1:].view_as(xd).copy_(xd)
    dys = T.empty(dyd.numel() + 1, device="cuda")[1:].view_as(dyd).copy_(dyd)
    assert xs.data_ptr() % 16 != 0
    gw2, gb2 = conv.backward_weight(xs, dys, float(case[0]))
    assert np.array_equal(host(gw), host(gw2)) and np.array_equal(host(gb), host(gb2))
    big = T.full((xd.numel() + 8192,), float("nan"), device="cuda")
    xn = big[4096 : 4096 + xd.numel()].view_as(xd).copy_(xd)
    bigd = T.full((dyd.numel() + 8192,), float("nan"), device="cuda")
    dyn = bigd[4096 : 4096 + dyd.numel()].view_as(dyd).copy_(dyd)
    gw4, gb4 = conv.backward_weight(xn, dyn, float(case[0]))
    assert np.array_equal(host(gw), host(gw4)) and np.array_equal(host(gb), host(gb4))
    lib_option("SP_BLOCKS", "1")  # one pixel range per tile: a single workgroup walks every stage
    gw5, gb5 = conv.backward_weight(xd, dyd, float(case[0]))
    assert_close(host(gw5), gw_ref, REL_TOL, "weight grad, one workgroup per tile")
    assert_close(host(gb5), gb_ref, REL_TOL, "bias grad, one workgroup per tile")
    lib_option("SP_BLOCKS", None)
    lib_option("WGRAD_SP_ANY", "0")
    gw3, gb3 = conv.backward_weight(xd, dyd, float(case[0]))
    assert_close(host(gw3), gw_ref, REL_TOL, "the replaced kernel's weight grad")


@pytest.mark.parametrize("case", [SP_CASES[i] for i in (0, 2, 3, 5, 7)] + [(4, 16, 55, 55, 32, 3, 2, 0), (3, 64, 13, 13, 128, 3, 2, 0), (2, 32, 27, 27, 64, 3, 2, 0)],
                         ids=lambda c: "B%d_%dx%dx%d_to%d_k%ds%dp%d" % c)
def test_flat_slab_reduction_is_bit_identical_to_the_workgroup_one(T, case, lib_option):
    """slab_reduce_flat (round 5: one thread per four elements walks the slabs itself) keeps the summation order of the workgroup-per-32-
    elements kernels (eight slot partials, then the partials in order): weight and bias gradients bit for bit, every weight-gradient family"""
    from cnn_amd import capi

    x, w, b, dy = _conv_inputs(case, 910)
    conv = capi.Conv2d(*case)
    xd, dyd = dev(T, x), dev(T, dy)
    gw1, gb1 = conv.backward_weight(xd, dyd, float(case[0]))
    lib_option("REDUCE_OLD", "1")
    gw0, gb0 = conv.backward_weight(xd, dyd, float(case[0]))
    assert np.array_equal(host(gw1).view(np.uint32), host(gw0).view(np.uint32))
    assert np.array_equal(host(gb1).view(np.uint32), host(gb0).view(np.uint32))


ROWS_CASES = [
    (2, 64, 7, 112, 128, 3, 1, 0),   # the north-star geometry at a small height: forward 112-wide pad 0 (128-channel tile), data gradient 110-wide pad 2
    (1, 40, 9, 112, 104, 3, 1, 0),   # ... partial output-channel tiles (104 of 128 forward, 40 of 64 backward), 5 / 13 channel chunks
    (2, 64, 6, 112, 64, 3, 1, 1),    # pad 1 (VGG conv2 class): halo rows above / below, both tap columns that leave a row; 64-channel tile (2 x 2 waves)
    (3, 32, 5, 112, 40, 3, 1, 1),    # ... ragged row blocks (5 rows in blocks of 4 / 2), four chunks
    (2, 64, 56, 56, 64, 3, 1, 1),    # 56-wide planes: super-rows of two rows (112 flat pixels), 64-channel tiles both ways
    (2, 64, 8, 56, 128, 3, 1, 1),    # ... 128-channel tile forward (4 x 1 waves), 64-channel tile backward
    (1, 32, 12, 56, 72, 3, 1, 1),    # ... partial tiles, ragged units (12 rows in units of 8 / 4)
    (2, 128, 28, 28, 128, 3, 1, 1),  # 28-wide planes: super-rows of four rows, one per workgroup unit (7 units per plane)
    (1, 72, 10, 28, 200, 3, 1, 1),   # ... partial tiles, ragged units (10 rows in units of 4)
    (2, 64, 14, 14, 128, 3, 1, 1),   # 14x14 planes whole: one super-row of 196 pixels (13 blocks, split 7 + 6 between two waves)
    (3, 40, 14, 14, 72, 3, 1, 1),    # ... partial tiles
    (4, 64, 7, 7, 128, 3, 1, 1),     # 7x7 planes packed: the planes of two samples as one super-row of 98 pixels, every out-of-plane tap a lane mask
    (5, 48, 7, 7, 80, 3, 1, 1),      # ... odd batch (the last unit holds one sample), partial tiles, three / five 16-channel chunks
    (1, 32, 7, 7, 32, 3, 1, 1),      # ... one sample, two chunks
]


@pytest.mark.parametrize("case", ROWS_CASES, ids=lambda c: "B%d_%dx%dx%d_to%d_k%ds%dp%d" % c)
def test_conv2d_row_kernel_vs_oracle(T, case, lib_option):
    """conv_rows.hip (round 5: LDS-staged 3x3 / stride-1 forward and data gradient of wide planes, conv2d.cpp:69-92 / 168-199) against the
    oracle, and against the implicit GEMM it replaces on these geometries (CNN_AMD_CONV_ROWS=0)"""
    from cnn_amd import capi

    x, w, b, dy = _conv_inputs(case, 440)
    y_ref, _, _, dx_ref = _oracle_conv(case, x, w, b, dy)
    conv = capi.Conv2d(*case)
    xd, wd, bd, dyd = dev(T, x), dev(T, w), dev(T, b), dev(T, dy)
    capi.kernel_timing(1)
    y = conv.forward(xd, wd, bd)
    dx = conv.backward_data(dyd, wd)
    T.cuda.synchronize()
    names = [k.split("|")[0] for k in capi.kernel_timing_report()]
    capi.kernel_timing(0)
    assert sum(n.startswith("conv_rows<") for n in names) == 2, names
    assert_close(host(y), y_ref, REL_TOL, "row kernel forward")
    assert_close(host(dx), dx_ref, REL_TOL, "row kernel data gradient")
    # the unit walk: with one workgroup per output-channel tile for the whole launch every workgroup walks ALL units
    # of the case -- the cross-unit prefetch, the "stage 0 already waited for in front of the previous unit's stores" rule, the accumulator
    # reset and the ragged last unit of a plane in the MIDDLE of a walk (conv_rows.hip's unit loop) -- against the oracle, with the
    # ReLU / ReLU' epilogues of the fused steps on the same walk
    lib_option("ROWS_BLOCKS", "1")
    capi.kernel_timing(1)
    y = conv.forward(xd, wd, bd)
    dx = conv.backward_data(dyd, wd)
    y2, r2 = T.full_like(y, 7.0), T.full_like(y, 7.0)
    conv.forward_relu(xd, wd, bd, y2, r2)
    relu_in = capi.relu_forward(xd - 0.5)  # the output of a ReLU layer in front (half of it blocked)
    dxm = T.full_like(xd, 7.0)
    conv.backward_data_relu(dyd, wd, relu_in, dxm)
    T.cuda.synchronize()
    names = [k.split("|")[0] for k in capi.kernel_timing_report()]
    capi.kernel_timing(0)
    assert sum(n.startswith("conv_rows<") for n in names) == 4, names
    assert_close(host(y), y_ref, REL_TOL, "row kernel forward, one workgroup walks every unit")
    assert_close(host(dx), dx_ref, REL_TOL, "row kernel data gradient, one workgroup walks every unit")
    assert_close(host(y2), y_ref, REL_TOL, "row kernel forward + ReLU, pre-activation")
    assert np.array_equal(host(r2), np.where(host(y2) >= 0, host(y2), np.float32(0)))  # relu.cpp:25 on the same sums
    assert_close(host(dxm), np.where(host(relu_in) <= 0, np.float32(0), dx_ref), REL_TOL, "row kernel data gradient + ReLU'")
    if case[3] in (7, 14):
        # the producer-wave variant of the small planes' instances (round 6: a fifth wave issues the stage DMA; default for 7x7) against its
        # twin without one, with one workgroup walking every unit: same MFMAs on the same operands, bit for bit
        lib_option("ROWS_PROD", "0" if case[3] == 7 else "2")
        y3, dx3 = conv.forward(xd, wd, bd), conv.backward_data(dyd, wd)
        assert np.array_equal(host(y3).view(np.uint32), host(y).view(np.uint32))
        assert np.array_equal(host(dx3).view(np.uint32), host(dx).view(np.uint32))
        lib_option("ROWS_PROD", None)
    lib_option("ROWS_BLOCKS", None)
    lib_option("CONV_ROWS", "0")
    assert_close(host(conv.forward(xd, wd, bd)), y_ref, REL_TOL, "implicit GEMM forward")
    assert_close(host(conv.backward_data(dyd, wd)), dx_ref, REL_TOL, "implicit GEMM data gradient")


ANY_CASES = [
    (2, 32, 9, 23, 40, 3, 1, 0),      # pad 0 (conv2d.cpp:41-42): 23-wide planes, class pitch 28: rows of 21 outputs, 7 rows = one unit; data gradient 21-wide pad 2
    (2, 32, 30, 50, 64, 3, 1, 1),     # pad 1: 50-wide, balanced units of 8 / 7 rows, four channel chunks, both edge columns
    (1, 32, 7, 109, 72, 3, 1, 0),     # 109-wide (class pitch 112): units of 4 rows (5 output rows = 3 + 2), partial channel tiles (72 = 64 + 8)
    (2, 32, 5, 222, 32, 3, 1, 0),     # 222-wide (class pitch 224): units of 2 rows of 220 outputs (27.5 blocks), data gradient 220-wide pad 2 -> 222
    (1, 32, 6, 223, 40, 3, 1, 1),     # ... odd width with pad 1: rows of 223 outputs, the row's last 16-byte unit straddles its end
    (3, 40, 23, 21, 48, 3, 1, 1),     # 21-wide pad 1: 21 rows per unit would be 441 pixels -- 23 rows = 12 + 11, odd batch
    (2, 64, 52, 52, 64, 3, 1, 0),     # the reference-style pad-0 VGG shapes: 52 -> 50
    (1, 32, 100, 9, 32, 3, 1, 1),     # tall narrow planes: 9 columns, 25 rows per unit
    (2, 32, 3, 96, 32, 3, 1, 0),      # one output row
    (3, 20, 11, 37, 24, 3, 1, 1),     # channel counts that are no multiples of 8 / 16: the last stage's planes behind the tensor's last channel are staged as zeros
    (2, 9, 14, 30, 17, 3, 1, 0),      # ... 9 -> 17 channels (data gradient 17 -> 9: below the kernel's floor of 16 output channels, stays on the generic path)
]


@pytest.mark.parametrize("case", ANY_CASES, ids=lambda c: "B%d_%dx%dx%d_to%d_k%ds%dp%d" % c)
def test_conv2d_row_kernel_any_width_vs_oracle(T, case, lib_option):
    """conv_rows_any.hip (round 6: the runtime-width member of the row-kernel family -- 3x3 / stride-1 layers of ANY plane size up to 224
    columns, conv2d.cpp:41-42; forward conv2d.cpp:69-92, data gradient conv2d.cpp:168-199) against the oracle: default walk, one workgroup
    walking every unit, the ReLU / ReLU' epilogues, and against the kernels it replaces (ROWS_ANY=0)"""
    from cnn_amd import capi

    x, w, b, dy = _conv_inputs(case, 455)
    y_ref, _, _, dx_ref = _oracle_conv(case, x, w, b, dy)
    conv = capi.Conv2d(*case)
    xd, wd, bd, dyd = dev(T, x), dev(T, w), dev(T, b), dev(T, dy)
    for blocks in (None, "1"):
        lib_option("ROWS_BLOCKS", blocks)
        capi.kernel_timing(1)
        y = conv.forward(xd, wd, bd)
        dx = conv.backward_data(dyd, wd)
        y2, r2 = T.full_like(y, 7.0), T.full_like(y, 7.0)
        conv.forward_relu(xd, wd, bd, y2, r2)
        relu_in = capi.relu_forward(xd - 0.5)
        dxm = T.full_like(xd, 7.0)
        conv.backward_data_relu(dyd, wd, relu_in, dxm)
        T.cuda.synchronize()
        names = [k.split("|")[0] for k in capi.kernel_timing_report()]
        capi.kernel_timing(0)
        assert sum(n.startswith("conv_rows_any<") for n in names) == (4 if case[1] >= 16 else 2), names
        assert_close(host(y), y_ref, REL_TOL, "any-width forward")
        assert_close(host(dx), dx_ref, REL_TOL, "any-width data gradient")
        assert_close(host(y2), y_ref, REL_TOL, "any-width forward + ReLU, pre-activation")
        assert np.array_equal(host(r2), np.where(host(y2) >= 0, host(y2), np.float32(0)))
        assert_close(host(dxm), np.where(host(relu_in) <= 0, np.float32(0), dx_ref), REL_TOL, "any-width data gradient + ReLU'")
    lib_option("ROWS_BLOCKS", None)
    # tensors whose neighbours in memory are NaN: nothing outside them reaches a sum (rows / columns staged past an edge are selected away)
    big = T.full((xd.numel() + 8192,), float("nan"), device="cuda")
    xn = big[4096 : 4096 + xd.numel()].view_as(xd).copy_(xd)
    assert np.array_equal(host(conv.forward(xn, wd, bd)).view(np.uint32), host(y).view(np.uint32))
    bigd = T.full((dyd.numel() + 8192,), float("nan"), device="cuda")
    dyn = bigd[4096 : 4096 + dyd.numel()].view_as(dyd).copy_(dyd)
    assert np.array_equal(host(conv.backward_data(dyn, wd)).view(np.uint32), host(dx).view(np.uint32))
    lib_option("ROWS_ANY", "0")
    assert_close(host(conv.forward(xd, wd, bd)), y_ref, REL_TOL, "the replaced kernel's forward")
    assert_close(host(conv.backward_data(dyd, wd)), dx_ref, REL_TOL, "the replaced kernel's data gradient")


TALL_CASES = [
    (2, 64, 56, 56, 64, 3, 1, 1),    # 56-wide planes as four units of 14 rows: four co waves x seven super-rows of two rows (49 tiles per wave)
    (3, 32, 28, 56, 48, 3, 1, 1),    # ... 28 rows (two units per plane), partial tiles, odd batch
    (2, 128, 28, 28, 128, 3, 1, 1),  # 28-wide planes as four units of 7 rows: one super-row of 196 pixels (13 blocks, the last one a quarter full)
    (3, 72, 14, 28, 200, 3, 1, 1),   # ... 14 rows (two units per plane), partial tiles
]


@pytest.mark.parametrize("case", TALL_CASES, ids=lambda c: "B%d_%dx%dx%d_to%d_k%ds%dp%d" % c)
def test_conv2d_row_kernel_tall_units_vs_oracle(T, case, lib_option):
    """conv_rows.hip's TALL units (round 6: the batch-64 layers of the ResNet-shaped stack as 256 units instead of 448; conv2d.cpp:69-92 /
    168-199) against the oracle: forced here (ROWS_TALL=2: at these batch sizes the planner would not pick them), as one unit per workgroup
    and with one workgroup walking every unit, with the ReLU / ReLU' epilogues; bit-identical to the short units (same sums, same order)"""
    from cnn_amd import capi

    x, w, b, dy = _conv_inputs(case, 445)
    y_ref, _, _, dx_ref = _oracle_conv(case, x, w, b, dy)
    conv = capi.Conv2d(*case)
    xd, wd, bd, dyd = dev(T, x), dev(T, w), dev(T, b), dev(T, dy)
    y0, dx0 = conv.forward(xd, wd, bd), conv.backward_data(dyd, wd)
    lib_option("ROWS_TALL", "2")
    for blocks in (None, "1"):
        lib_option("ROWS_BLOCKS", blocks)
        capi.kernel_timing(1)
        y = conv.forward(xd, wd, bd)
        dx = conv.backward_data(dyd, wd)
        y2, r2 = T.full_like(y, 7.0), T.full_like(y, 7.0)
        conv.forward_relu(xd, wd, bd, y2, r2)
        relu_in = capi.relu_forward(xd - 0.5)
        dxm = T.full_like(xd, 7.0)
        conv.backward_data_relu(dyd, wd, relu_in, dxm)
        T.cuda.synchronize()
        names = [k.split("|")[0] for k in capi.kernel_timing_report()]
        capi.kernel_timing(0)
        tall = [n for n in names if n.startswith("conv_rows<") and ",r14>" in n or ",r7>" in n]
        assert len(tall) == 4, names
        assert_close(host(y), y_ref, REL_TOL, "tall units forward")
        assert_close(host(dx), dx_ref, REL_TOL, "tall units data gradient")
        assert_close(host(y2), y_ref, REL_TOL, "tall units forward + ReLU, pre-activation")
        assert np.array_equal(host(r2), np.where(host(y2) >= 0, host(y2), np.float32(0)))
        assert_close(host(dxm), np.where(host(relu_in) <= 0, np.float32(0), dx_ref), REL_TOL, "tall units data gradient + ReLU'")
        assert np.array_equal(host(y).view(np.uint32), host(y0).view(np.uint32))
        assert np.array_equal(host(dx).view(np.uint32), host(dx0).view(np.uint32))


STEM_DGRAD_CASES = [
    (2, 3, 64, 64, 8, 7, 2, 3),      # the 7x7 / stride-2 / pad-3 stem (ResNet-shaped stack): 32 x 32 grid positions, even channel count
    (3, 3, 38, 44, 72, 7, 2, 3),     # ... 19 x 22 positions: the last wave of an image ragged, odd batch
    (1, 3, 37, 45, 5, 7, 2, 3),      # ... odd planes (the last row / column of 2 x 2 blocks half outside), odd channel count (tail channel)
]


@pytest.mark.parametrize("case", STEM_DGRAD_CASES, ids=lambda c: "B%d_%dx%dx%d_to%d_k%ds%dp%d" % c)
def test_conv2d_stem_data_gradient_packed_vs_oracle(T, case, lib_option):
    """conv_dgrad_thin_s2_pk7 (round 6): the data gradient of the 3 -> Co, 7x7, stride-2, pad-3 stem (conv2d.cpp:168-199) on packed fp32
    FMAs with the filters re-packed into 32-byte rows -- against the oracle, with and without the ReLU' epilogue, through the per-call
    packing (workspace) and through the prepared image; bit-identical to the scalar-operand kernel (same products, same order)"""
    from cnn_amd import capi

    x, w, b, dy = _conv_inputs(case, 611)
    _, _, _, dx_ref = _oracle_conv(case, x, w, b, dy)
    conv = capi.Conv2d(*case)
    xd, wd, dyd = dev(T, x), dev(T, w), dev(T, dy)
    capi.kernel_timing(1)
    dx = conv.backward_data(dyd, wd)
    relu_in = capi.relu_forward(xd - 0.5)
    dxm = T.full_like(xd, 7.0)
    conv.backward_data_relu(dyd, wd, relu_in, dxm)
    T.cuda.synchronize()
    names = [k.split("|")[0] for k in capi.kernel_timing_report()]
    capi.kernel_timing(0)
    assert sum(n.startswith("conv_dgrad_thin_pk<3,k7s2>") for n in names) == 2 and "thin_pack_k7" in names, names
    assert_close(host(dx), dx_ref, REL_TOL, "packed stem data gradient")
    assert_close(host(dxm), np.where(host(relu_in) <= 0, np.float32(0), dx_ref), REL_TOL, "packed stem data gradient + ReLU'")
    lib_option("DGRAD_THIN_PK", "0")
    capi.kernel_timing(1)
    dx0 = capi.Conv2d(*case).backward_data(dyd, wd)
    T.cuda.synchronize()
    names = [k.split("|")[0] for k in capi.kernel_timing_report()]
    capi.kernel_timing(0)
    assert "conv_dgrad_thin<3,k7s2>" in names, names
    assert np.array_equal(host(dx).view(np.uint32), host(dx0).view(np.uint32))


S2_CASES = [
    (3, 16, 55, 55, 32, 3, 2, 0),     # conv_layer_2 of the reference net (alexnet.cpp:17): 27x27 outputs flat-packed in units of 7 rows; data gradient in 28x28 domains
    (2, 32, 27, 27, 64, 3, 2, 0),     # conv_layer_3: 13x13 outputs, units of 7 rows (ragged last unit of 6)
    (5, 64, 13, 13, 128, 3, 2, 0),    # conv_layer_4: 6x6 outputs, the planes of two samples per unit (odd batch: the last unit holds one)
    (2, 24, 55, 55, 40, 3, 2, 0),     # ... partial channel tiles (40 of 64 forward, 24 of 32 backward), 3 / 5 channel chunks
    (3, 40, 13, 13, 72, 3, 2, 0),
    (2, 64, 56, 56, 128, 3, 2, 1),    # stage entries of the ResNet-shaped stack: pad 1 (halo row above / column left of the image)
    (2, 128, 28, 28, 256, 3, 2, 1),
    (3, 256, 14, 14, 512, 3, 2, 1),   # ... 7x7 outputs, two samples per unit, odd batch
    (1, 72, 14, 14, 136, 3, 2, 1),    # ... partial tiles, one sample
    (2, 8, 28, 28, 24, 3, 2, 1),      # ... one channel chunk
]


@pytest.mark.parametrize("case", S2_CASES, ids=lambda c: "B%d_%dx%dx%d_to%d_k%ds%dp%d" % c)
def test_conv2d_stride2_row_kernel_vs_oracle(T, case, lib_option):
    """conv_rows_s2.hip (round 6: LDS-staged 3x3 / stride-2 forward and data gradient, conv2d.cpp:69-92 / 168-199 with the reference's
    default stride) against the oracle: default dispatch, then with one workgroup walking every unit of the launch (the unit loop's
    cross-unit prefetch and ragged last units in the middle of a walk), with the fused ReLU / ReLU' epilogues; and against the kernels it
    replaces on these geometries (CNN_AMD_CONV_S2=0)"""
    from cnn_amd import capi

    x, w, b, dy = _conv_inputs(case, 540)
    y_ref, _, _, dx_ref = _oracle_conv(case, x, w, b, dy)
    conv = capi.Conv2d(*case)
    xd, wd, bd, dyd = dev(T, x), dev(T, w), dev(T, b), dev(T, dy)
    relu_in = capi.relu_forward(xd - 0.5)  # the output of a ReLU layer in front (half of it blocked)
    lib_option("CONV_S2", "2")  # (every instance, also those the default dispatch leaves to the register-direct kernels)
    for blocks in (None, "1"):
        lib_option("S2_BLOCKS", blocks)
        capi.kernel_timing(1)
        y = T.full(conv.out_shape(), 7.0, device="cuda")
        conv.forward(xd, wd, bd, y)
        dx = T.full_like(xd, 7.0)
        conv.backward_data(dyd, wd, dx)
        y2, r2 = T.full_like(y, 7.0), T.full_like(y, 7.0)
        conv.forward_relu(xd, wd, bd, y2, r2)
        dxm = T.full_like(xd, 7.0)
        conv.backward_data_relu(dyd, wd, relu_in, dxm)
        T.cuda.synchronize()
        names = [k.split("|")[0] for k in capi.kernel_timing_report()]
        capi.kernel_timing(0)
        assert sum(n.startswith("conv_s2<") for n in names) == 4, names
        tag = "default grid" if blocks is None else "one workgroup walks every unit"
        assert_close(host(y), y_ref, REL_TOL, f"stride-2 row kernel forward, {tag}")
        assert_close(host(dx), dx_ref, REL_TOL, f"stride-2 row kernel data gradient, {tag}")
        assert_close(host(y2), y_ref, REL_TOL, f"stride-2 row kernel forward + ReLU, pre-activation, {tag}")
        assert np.array_equal(host(r2), np.where(host(y2) >= 0, host(y2), np.float32(0)))  # relu.cpp:25 on the same sums
        assert_close(host(dxm), np.where(host(relu_in) <= 0, np.float32(0), dx_ref), REL_TOL, f"stride-2 row kernel data gradient + ReLU', {tag}")
    lib_option("S2_BLOCKS", None)
    lib_option("CONV_S2", "0")
    assert_close(host(conv.forward(xd, wd, bd)), y_ref, REL_TOL, "replaced forward kernel")
    assert_close(host(conv.backward_data(dyd, wd)), dx_ref, REL_TOL, "replaced data-gradient kernel")


@pytest.mark.parametrize("case", [(8, 64, 112, 112, 128, 3, 1, 0), (16, 128, 28, 28, 128, 3, 1, 1), (9, 48, 7, 7, 80, 3, 1, 1)],
                         ids=lambda c: "B%d_%dx%dx%d_to%d_k%ds%dp%d" % c)
def test_inline_asm_mfma_kernels_are_bit_reproducible(T, case):
    """conv_rows.hip issues its MFMAs as inline assembly and pads the hazards itself (s_nop 1, acc_settle): a missed hazard would show as
    rare differing bits between two runs of one launch.  Forward, both data gradients and the weight gradient, eight runs each, every
    second one beside a second stream that keeps the chip busy (tools/rows_stress.py is the long form)"""
    from cnn_amd import capi

    x, w, b, dy = _conv_inputs(case, 977)
    conv = capi.Conv2d(*case)
    xd, wd, bd, dyd = dev(T, x), dev(T, w), dev(T, b), dev(T, dy)
    side, noise = T.cuda.Stream(), T.rand((64 << 20,), device="cuda")  # (256 MB: a few passes of the library's own ReLU kernel as the load)
    noise_out = T.empty_like(noise)
    first = None
    for rep in range(8):
        if rep % 2:
            with T.cuda.stream(side):
                for _ in range(3):
                    capi.relu_forward(noise, noise_out)
        dxr = T.empty_like(xd)
        conv.backward_data_relu(dyd, wd, xd, dxr)
        cur = [conv.forward(xd, wd, bd), conv.backward_data(dyd, wd), dxr, *conv.backward_weight(xd, dyd, float(case[0]))]
        T.cuda.synchronize()
        cur = [t.clone() for t in cur]
        if first is None:
            first = cur
        else:
            assert all(T.equal(a, c) for a, c in zip(first, cur)), f"run {rep} differs from run 0"


def test_u8_batch_stager_is_bit_identical_to_the_reference_conversion(T, golden_dir):
    """row n4, cnn_batch_stager_create_u8: the bytes of a real input (the six images behind the reference's own Grad-CAM pictures and the
    three README images, tests/golden/*_images_u8.*) are uploaded AS BYTES and converted on the device to the fp32 planar batch --
    bit-identical to Tensor3D::read_from_opencv_mat (data_format.cpp:13-23: data[c*H*W + i] = byte * 1.f / 255), hence to what the fp32
    stager uploads: two nets trained from the two staging paths end with the same bits"""
    from cnn_amd import capi, hostapi

    imgs = np.concatenate([np.load(os.path.join(golden_dir, "gradcam_kat_images_u8.npz"))["images"],
                           np.load(os.path.join(golden_dir, "readme_kat_images_u8.npy"))])
    assert imgs.shape == (9, 224, 224, 3) and imgs.dtype == np.uint8  # [B][H][W][3] bytes, as cv::Mat holds them
    B = imgs.shape[0]
    # the reference's expression in float32: uchar -> int -> * 1.f -> / 255 (int -> float)
    want = (imgs.astype(np.float32) * np.float32(1.0) / np.float32(255)).transpose(0, 3, 1, 2).copy()
    assert len(np.unique(imgs)) > 200  # (nearly every table entry is exercised)
    st8 = capi.BatchStager(u8_shape=(B, 224, 224), depth=2)
    st32 = capi.BatchStager(B * 3 * 224 * 224 * 4, depth=2)
    labels = (T.arange(B, device="cuda") % 3).to(T.int32)
    nets = [hostapi.HostAlexNet(3), hostapi.HostAlexNet(3)]
    for net in nets:
        net.load_checkpoint(os.path.join(golden_dir, "readme_kat_checkpoint.model"))
    for rep in range(3):  # more submits than slots: the slots are reused
        host8, slot8 = st8.acquire()
        host8[:] = imgs.reshape(-1)
        dev8 = st8.submit(slot8)
        st8.wait(slot8)
        got = T.empty((B, 3, 224, 224), device="cuda")
        capi.check(capi.load().cnn_memcpy_d2d(got.data_ptr(), dev8, got.numel() * 4, capi._stream()), "cnn_memcpy_d2d")
        nets[0].train_step_ptr(dev8, labels, B, 224, 224, 1e-3)
        st8.release(slot8)
        assert np.array_equal(got.cpu().numpy(), want), f"submit {rep}: the device conversion is not bit-identical"
        host32, slot32 = st32.acquire()
        host32[:] = want.reshape(-1)
        dev32 = st32.submit(slot32)
        st32.wait(slot32)
        nets[1].train_step_ptr(dev32, labels, B, 224, 224, 1e-3)
        st32.release(slot32)
    T.cuda.synchronize()
    assert nets[0].last_loss() == nets[1].last_loss() and np.array_equal(nets[0].get_params(), nets[1].get_params())
    for net in nets:
        net.close()
    st8.close()
    st32.close()


def test_conv2d_wgrad_ignores_non_finite_unused_columns(T):
    """W = 56, stride 2: input column 55 is read by no output pixel (conv2d.cpp:127-146 never touches it), so an Inf
    there must not leak into the gradient through a zero-weighted over-read"""
    from cnn_amd import capi

    case = (2, 4, 10, 56, 6, 3, 2, 0)
    x, w, b, dy = _conv_inputs(case, 410)
    _, gw_ref, gb_ref, _ = _oracle_conv(case, x, w, b, dy)
    x2 = x.copy()
    x2[:, :, :, 55] = np.inf
    x2[:, :, 9, :] = np.inf  # row 9 likewise: Ho = 4 reads rows 0..8
    gw, gb = capi.Conv2d(*case).backward_weight(dev(T, x2), dev(T, dy), float(case[0]))
    assert np.all(np.isfinite(host(gw)))
    assert_close(host(gw), gw_ref, REL_TOL, "weight grad")


def test_conv2d_dgrad_uncovered_rows_are_zero(T):
    from cnn_amd import capi

    case = (2, 3, 224, 224, 16, 3, 2, 0)  # row/col 223 is never covered (conv2d.cpp:168,183)
    x, w, b, dy = _conv_inputs(case, 7)
    dx = host(capi.Conv2d(*case).backward_data(dev(T, dy), dev(T, w)))
    assert np.all(dx[:, :, 223, :] == 0) and np.all(dx[:, :, :, 223] == 0) and np.any(dx[:, :, 222, :] != 0)


@pytest.mark.parametrize("shape,k,step", [((2, 16, 111, 111), 2, 2), ((3, 5, 7, 7), 2, 2), ((2, 4, 9, 10), 3, 2),
                                          ((2, 3, 8, 8), 3, 1), ((1, 2, 6, 6), 2, 3)])
def test_maxpool_bit_exact(T, shape, k, step):
    from cnn_amd import capi

    x = uniform_pm1(5, shape)
    # deliberate ties, signed zeros and NaNs (SURVEY H4 / pool2d.cpp:67-75)
    flat = x.reshape(-1)
    rs = np.random.RandomState(1)
    flat[rs.choice(flat.size, flat.size // 6, replace=False)] = 0.5
    flat[rs.choice(flat.size, flat.size // 20, replace=False)] = -0.0
    flat[rs.choice(flat.size, flat.size // 20, replace=False)] = 0.0
    flat[rs.choice(flat.size, flat.size // 50, replace=False)] = np.nan
    y_ref, m_ref = O.maxpool_forward(x, k, step)
    y, m = capi.maxpool_forward(dev(T, x), k, step)
    assert np.array_equal(host(m), m_ref), "argmax mask must be bit-exact"
    assert np.array_equal(host(y).view(np.uint32), y_ref.view(np.uint32)), "pooled values must be bit-exact"
    y2, none = capi.maxpool_forward(dev(T, x), k, step, record_mask=False)  # the no_grad path
    assert none is None and np.array_equal(host(y2).view(np.uint32), y_ref.view(np.uint32))
    dy = uniform_pm1(6, y_ref.shape)
    dx_ref = O.maxpool_backward(dy, m_ref, shape, k, step)
    dx = capi.maxpool_backward(dev(T, dy), m, shape, k, step)
    assert np.array_equal(host(dx).view(np.uint32), dx_ref.view(np.uint32))


@pytest.mark.parametrize("n", [1, 3, 4, 1000, 16 * 111 * 111 * 2 + 1])
def test_relu_bit_exact(T, n):
    from cnn_amd import capi

    x = uniform_pm1(8, (n,))
    x[:: 7] = -0.0
    x[1 :: 11] = 0.0
    x[2 :: 13] = np.nan
    x[3 :: 17] = -np.inf
    y_ref = O.relu_forward(x)
    y = capi.relu_forward(dev(T, x))
    assert np.array_equal(host(y).view(np.uint32), y_ref.view(np.uint32))
    d = uniform_pm1(9, (n,))
    d_ref = O.relu_backward(y_ref, d)
    dd = capi.relu_backward(y, dev(T, d))
    assert np.array_equal(host(dd).view(np.uint32), d_ref.view(np.uint32))
    # unaligned views take the scalar path
    if n > 8:
        xo = dev(T, np.concatenate([[0.0], x]).astype(np.float32))[1:]
        assert np.array_equal(host(capi.relu_forward(xo.contiguous())).view(np.uint32), y_ref.view(np.uint32))


@pytest.mark.parametrize("B,n_in,n_out", [(4, 4608, 3), (3, 100, 10), (5, 33, 17), (37, 130, 8), (256, 4608, 3), (1, 7, 1), (19, 25088, 3), (33, 6000, 2)])
def test_linear_vs_oracle(T, B, n_in, n_out):
    from cnn_amd import capi

    x, w, b = uniform01(10, (B, n_in)), normal_scaled(11, (n_in, n_out)), normal_scaled(12, (n_out,))
    dy = uniform_pm1(13, (B, n_out))
    y_ref = O.linear_forward(x, w, b)
    gw_ref, gb_ref, dx_ref = O.linear_backward(x, dy, w)
    xd, wd = dev(T, x), dev(T, w)
    assert_close(host(capi.linear_forward(xd, wd, dev(T, b))), y_ref, REL_TOL, "linear forward")
    gw, gb, dx = capi.linear_backward(xd, dev(T, dy), wd, float(B))
    assert_close(host(gw), gw_ref, REL_TOL, "linear gW")
    assert_close(host(gb), gb_ref, REL_TOL, "linear gb")
    assert_close(host(dx), dx_ref, REL_TOL, "linear dx")


def test_sgd_bit_exact_and_scaled(T):
    from cnn_amd import capi

    n = 111267
    p, g = normal_scaled(14, (n,)), uniform_pm1(15, (n,))
    out = capi.sgd_update(dev(T, p), dev(T, g), 1e-3)
    assert np.array_equal(host(out).view(np.uint32), O.sgd_update(p, g, 1e-3).view(np.uint32))
    out = capi.sgd_update(dev(T, p), dev(T, g), 1e-3, 0.125)  # power-of-two scale is exact
    assert np.array_equal(host(out).view(np.uint32), O.sgd_update(p, g * np.float32(0.125), 1e-3).view(np.uint32))


def test_softmax_xent_vs_oracle(T):
    from cnn_amd import capi

    B, n = 300, 3
    logits = (uniform_pm1(16, (B, n)) * 12).astype(np.float32)
    logits[0] = [100, 0, -100]  # clamped exp (func.cpp:7-11)
    logits[1] = [0, 0, 0]
    labels = (np.arange(B) % n).astype(np.int32)
    labels[0] = 0
    p_ref = O.softmax(logits)
    loss_ref, d_ref = O.cross_entropy_backward(p_ref, labels)
    probs, delta, loss = capi.softmax_xent(dev(T, logits), dev(T, labels))
    assert np.allclose(host(probs), p_ref, rtol=1e-5, atol=1e-7)
    assert np.allclose(host(delta), d_ref, rtol=1e-5, atol=1e-6)
    # sample 0 has p == 0 for a non-label class: the reference's log(p)*y gives -inf*0 = NaN (func.cpp:65) -- kept
    assert np.isnan(loss_ref) and np.isnan(float(host(loss)[0]))
    logits[0] = [1, 2, 3]
    p_ref = O.softmax(logits)
    loss_ref, d_ref = O.cross_entropy_backward(p_ref, labels)
    probs, delta, loss = capi.softmax_xent(dev(T, logits), dev(T, labels))
    assert np.allclose(host(delta), d_ref, rtol=1e-5, atol=1e-6)
    assert np.isclose(float(host(loss)[0]) / B, loss_ref, rtol=1e-5)


def test_readme_known_answer_on_gpu(T, golden_dir):
    """the reference's published inference result (README.md:92) through the HIP path"""
    from cnn_amd.pynet import AlexNetHip

    imgs = np.load(os.path.join(golden_dir, "readme_kat_images_u8.npy"))
    exp = json.load(open(os.path.join(golden_dir, "readme_kat_expected.json")))
    x = np.ascontiguousarray((imgs.astype(np.float32) * np.float32(1.0) / np.float32(255)).transpose(0, 3, 1, 2))
    net = AlexNetHip(3, 3)
    net.load_checkpoint(os.path.join(golden_dir, "readme_kat_checkpoint.model"))
    logits = host(net.forward(dev(T, x), record=False))
    probs = O.softmax(logits)
    assert probs.argmax(axis=1).tolist() == exp["argmax"]
    assert np.allclose(probs.max(axis=1), exp["prob"], atol=3e-6), probs
    onet = O.Net(3, 3)
    onet.load_checkpoint(os.path.join(golden_dir, "readme_kat_checkpoint.model"))
    assert_close(logits, onet.forward(x), REL_TOL, "logits vs oracle")


def _count_decision_flips(net, onet, relu_layers, tag):
    """the discrete decisions of the forward pass -- ReLU pass / block (relu.cpp:25) and the MaxPool argmax (pool2d.cpp:67-82) --
    agree with the oracle's except where a pre-activation lies within rounding distance of zero / of its window neighbour: bounded
    at the level the stack tests use (2e-5 of the elements, at least 2)"""
    flipped = total = 0
    for l in relu_layers:
        got, want = host(net.relu_out[l]), np.maximum(onet.conv_out(l), 0)
        flipped += int(np.count_nonzero((got <= 0) != (want <= 0)))
        total += got.size
    assert flipped <= max(2, 2e-5 * total), (tag, "ReLU flips", flipped, total)
    mm = int(np.count_nonzero((host(net.pool_mask_int32()) & 0x7FFFFFFF) != onet.pool_mask()))  # (bit 31: the fused kernel's "pooled <= 0" mark)
    assert mm <= max(2, 2e-5 * onet.pool_mask().size), (tag, "pool argmax mismatches", mm, onet.pool_mask().size)


def test_whole_net_train_steps_vs_oracle(T):
    """config 1 plumbing: two full train steps (cnn.cpp:79-90) at B=4, 224x224: every activation, the pool mask,
    every gradient and the post-SGD weights against the oracle."""
    from cnn_amd.pynet import AlexNetHip

    B = 4
    x = uniform01(20, (B, 3, 224, 224))
    labels = (np.arange(B) % 3).astype(np.int32)
    onet = O.Net(B, 3)
    p0 = normal_scaled(21, (onet.n_params,))
    onet.params[:] = p0
    net = AlexNetHip(B, 3)
    assert net.n_params == onet.n_params == 111267
    net.load_params(p0)
    xd, ld = dev(T, x), dev(T, labels)
    for step in range(2):
        net.forward(xd)
        net.loss_backward_seed(ld)
        ologits = onet.forward(x)
        oprobs = O.softmax(ologits)
        oloss, odelta = O.cross_entropy_backward(oprobs, labels)
        for l in range(4):
            assert_close(host(net.conv_out[l]), onet.conv_out(l), REL_TOL, f"step{step} conv{l} out")
        # pool argmax: bit-exact is only meaningful on identical inputs -> checked op-level above; here the
        # conv outputs differ in the last bits, so compare values and allow rare near-tie flips in the mask
        assert_close(host(net.pool_out), onet.pool_out(), REL_TOL, f"step{step} pool out")
        _count_decision_flips(net, onet, (0, 1, 2, 3), f"step{step}")
        assert_close(host(net.logits), ologits, REL_TOL, f"step{step} logits")
        assert np.isclose(float(host(net.loss_sum)[0]) / B, oloss, rtol=1e-4)
        net.backward(net.delta)
        onet.backward(odelta)
        # gradients: 1e-4 tensor-normalised, fp64-arbitrated, with the oracle's ReLU' / MaxPool' decisions taken from the HIP forward
        # tensors (tests/util.py, oracle.pyoracle.SeqNet.backward) -- the round-1 tolerance of 2e-4 is gone
        s32, s64 = _synced_oracle_backward(net, onet.params, x, labels)
        for l in range(4):
            ci = _ALEX_DELTA_IDX[l]
            assert_close_arbitrated(host(net.d_conv[l]), s32.deltas[ci], s64.deltas[ci], REL_TOL, 2.0, f"step{step} d_conv{l}")
        g = host(net.grads)
        for name, lo, hi in _param_slices(net):
            assert_close_arbitrated(g[lo:hi], s32.grads[lo:hi], s64.grads[lo:hi], REL_TOL, 2.0, f"step{step} grad {name}")
        net.update(1e-3)
        onet.update(1e-3)
        assert_close(host(net.params), onet.params, REL_TOL, f"step{step} params")
        onet.params[:] = host(net.params)  # the next step starts from identical parameters


def test_bench_configuration_train_steps_vs_oracle(T):
    """the configuration bench.py runs (pool-fused first block, deferred conv1 data gradient, fused ReLU backward,
    prepared filters, register-direct kernels) against the oracle: two full train steps at B=4"""
    from cnn_amd.pynet import AlexNetHip

    B = 4
    x = uniform01(22, (B, 3, 224, 224))
    labels = (np.arange(B) % 3).astype(np.int32)
    onet = O.Net(B, 3)
    p0 = normal_scaled(23, (onet.n_params,))
    onet.params[:] = p0
    net = AlexNetHip(B, 3, defer_input_grad=True, fuse_pool=True)
    assert net.fuse_pool and net.defer_dx0
    net.load_params(p0)
    xd, ld = dev(T, x), dev(T, labels)
    for step in range(2):
        p_before = onet.params.copy()
        net.train_step(xd, ld, 1e-3)
        net.flush()
        ologits = onet.forward(x)
        oloss, odelta = O.cross_entropy_backward(O.softmax(ologits), labels)
        onet.backward(odelta)
        assert_close(host(net.pool_out), onet.pool_out(), REL_TOL, f"step{step} pool out")
        _count_decision_flips(net, onet, (1, 2, 3), f"step{step}")
        for l in (1, 2, 3):  # (pre-activations are not materialised in this configuration: compare the ReLU outputs)
            assert_close(host(net.relu_out[l]), np.maximum(onet.conv_out(l), 0), REL_TOL, f"step{step} relu{l} out")
        assert_close(host(net.logits), ologits, REL_TOL, f"step{step} logits")
        assert np.isclose(float(host(net.loss_sum)[0]) / B, oloss, rtol=1e-4)
        s32, s64 = _synced_oracle_backward(net, p_before, x, labels, pooled_domain=True)
        for l in (0, 2, 3):
            ci = _ALEX_DELTA_IDX[l]
            assert_close_arbitrated(host(net.d_conv[l]), s32.deltas[ci], s64.deltas[ci], REL_TOL, 2.0, f"step{step} d_conv{l}")
        g = host(net.grads)
        for name, lo, hi in _param_slices(net):
            assert_close_arbitrated(g[lo:hi], s32.grads[lo:hi], s64.grads[lo:hi], REL_TOL, 2.0, f"step{step} grad {name}")
        onet.update(1e-3)
        assert_close(host(net.params), onet.params, REL_TOL, f"step{step} params")
        onet.params[:] = host(net.params)


@pytest.mark.parametrize("B", [16, 256])
def test_config_batch_train_steps_vs_oracle(T, B):
    """BASELINE configs[0] (batch 16) and configs[1] (batch 256: the configuration `value` is quoted on -- VERDICT r5 weak 1b) at their
    stated sizes: the reference net, 224x224 -- two full train steps through the C++ Layer classes with their DEFAULT settings
    (pool-fused first block, fused step tail after the first pass: exactly what bench.py times) against the oracle, gradient for
    gradient: every layer's get_output(), loss, the discrete decisions, every gradient tensor, the delta w.r.t. the input and the
    bit-exact SGD step"""
    O.set_threads(0)  # (checker only: order-preserving thread split, bit-identical to one thread)
    try:
        _config_batch_train_steps_vs_oracle(T, B)
    finally:
        O.set_threads(1)


def _config_batch_train_steps_vs_oracle(T, B):
    import torch

    from cnn_amd import hostapi, stacks as S

    x = uniform01(24, (B, 3, 224, 224))
    labels = (np.arange(B) % 3).astype(np.int32)
    spec = S.alexnet()
    p0 = normal_scaled(25, (111267,))
    net = hostapi.HostAlexNet(3)
    net.set_params(p0)
    xd, ld = torch.from_numpy(x).cuda(), torch.from_numpy(labels).cuda()
    names = hostapi._layer_names(spec)
    params = p0
    for step in range(2):  # (the second step runs the prepared / pool-fused kernels and the fused tail)
        net.train_step(xd, ld, 1e-3)
        loss = net.last_loss()
        g = net.get_grads()
        o32, o64 = O.SeqNet(spec), O.SeqNet(spec, f64=True)
        outs = {}
        for o, f64 in ((o32, False), (o64, True)):
            o.params[:] = params
            logits = o.forward(x)
            oloss, odelta = O.cross_entropy_backward(O.softmax(logits, f64=f64), labels, f64=f64)
            outs[f64] = (oloss, odelta)
        assert abs(loss - outs[False][0]) <= 1e-4 * max(1.0, abs(outs[False][0])), (loss, outs[False][0])
        masks_from, flipped, total = {}, 0, 0
        for idx, e in enumerate(o32.layers):
            got = net.layer_output(names[idx], (B,) + e["out"]).reshape(o32.acts[idx].shape)
            assert_close_arbitrated(got, o32.acts[idx], o64.acts[idx], REL_TOL, 2.0, f"step{step} {names[idx]} output")
            if e["kind"] == "relu":
                masks_from[idx] = got
                flipped += int(np.count_nonzero((got <= 0) != (o32.acts[idx] <= 0)))
                total += got.size
                if o32.layers[idx + 1]["kind"] == "pool":
                    masks_from[idx + 1] = got
        assert flipped <= max(2, 2e-5 * total), (flipped, total)
        for o, f64 in ((o32, False), (o64, True)):
            o.backward(outs[f64][1], masks_from=masks_from)
        off = 0
        for idx, e in enumerate(o32.layers):
            if e["params"]:
                sl = slice(off, off + e["params"])
                assert_close_arbitrated(g[sl], o32.grads[sl], o64.grads[sl], REL_TOL, 2.0, f"step{step} grad {names[idx]}")
                off += e["params"]
        dx = net.input_delta((B, 3, 224, 224))
        assert_close_arbitrated(dx, o32.deltas[0], o64.deltas[0], REL_TOL, 2.0, f"step{step} delta w.r.t. the input")
        got = net.get_params()
        assert np.array_equal(got, O.sgd_update(params, g, 1e-3)), f"step{step}: p - lr*g is not bit-exact"
        params = got
    net.close()


# cnn_amd.stacks.alexnet(): conv1 relu1 pool conv2 relu2 conv3 relu3 conv4 relu4 linear.  ReLU::backward masks the upstream layer's
# delta IN PLACE (relu.cpp:37-39), so what a convolution's delta_output holds after the backward pass is the delta the ReLU in front of
# it handed on: SeqNet.deltas[] of that ReLU layer (conv3 -> relu2 = index 4, conv4 -> relu3 = 6); conv1 / conv2 have no ReLU in front.
_ALEX_DELTA_IDX = (0, 3, 4, 6)


def _synced_oracle_backward(net, params, x, labels, pooled_domain=False):
    """fp32 and fp64 oracle backward passes of the reference net at `params`, with the ReLU' / MaxPool' decisions taken from the HIP
    net's forward tensors of the same step (so both sides differentiate the same piecewise-linear function)"""
    from cnn_amd import stacks as S

    out = []
    masks = {4: host(net.relu_out[1]), 6: host(net.relu_out[2]), 8: host(net.relu_out[3]), 2: host(net.pool_mask_int32()) & 0x7FFFFFFF}
    if not pooled_domain:
        masks[1] = host(net.relu_out[0])  # (pool-fused runs do not materialise relu_layer_1's output)
    for f64 in (False, True):
        o = O.SeqNet(S.alexnet(), f64=f64)
        o.params[:] = params
        logits = o.forward(x)
        _, delta = O.cross_entropy_backward(O.softmax(logits, f64=f64), labels, f64=f64)
        o.backward(delta, masks_from=masks)
        out.append(o)
    return out


def _param_slices(net):
    out = []
    for l in range(4):
        out.append((f"conv{l}.w", net.w_off[l], net.b_off[l]))
        out.append((f"conv{l}.b", net.b_off[l], net.b_off[l] + net.CHANS[l + 1]))
    out.append(("linear.w", net.lw_off, net.lb_off))
    out.append(("linear.b", net.lb_off, net.n_params))
    return out


def test_full_size_northstar_properties(T):
    """BASELINE config 2 at its full size (B=256, 64->128, 112x112): the oracle would need minutes, so check
    size-independent properties: implicit-GEMM == im2col fallback, and per-sample agreement with the oracle on a
    3-sample slice (conv is per-sample independent, conv2d.cpp:69)."""
    from cnn_amd import capi

    case = (256, 64, 112, 112, 128, 3, 1, 0)
    conv = capi.Conv2d(*case)
    g = T.Generator(device="cuda").manual_seed(3)
    x = T.rand((256, 64, 112, 112), generator=g, device="cuda")
    w = T.randn((128, 64, 3, 3), generator=g, device="cuda") * 0.1
    b = T.randn((128,), generator=g, device="cuda") * 0.1
    y = conv.forward(x, w, b)
    y2 = conv.forward_im2col(x, w, b)
    scale = float(y2.abs().max())
    assert float((y - y2).abs().max()) <= REL_TOL * scale
    idx = [0, 131, 255]
    y_ref = O.conv2d_forward(host(x[idx]), host(w), host(b), 1)
    assert_close(host(y[idx]), y_ref, REL_TOL, "north-star forward slice")
    dy = T.rand(y.shape, generator=g, device="cuda") * 2 - 1
    del y2
    gw, gb = conv.backward_weight(x, dy, 256.0)
    gw2, gb2 = conv.backward_weight_im2col(x, dy, 256.0)
    assert float((gw - gw2).abs().max()) <= REL_TOL * float(gw2.abs().max())
    assert float((gb - gb2).abs().max()) <= REL_TOL * float(gb2.abs().max())
    dx = conv.backward_data(dy, w)
    _, _, dx_ref = O.conv2d_backward(host(x[idx]), host(dy[idx]), host(w), 1, need=(False, False, True))
    assert_close(host(dx[idx]), dx_ref, REL_TOL, "north-star dgrad slice")
    # linearity in dy: dgrad(2*dy) == 2*dgrad(dy) exactly (power-of-two scaling commutes with fp32 rounding)
    assert T.equal(conv.backward_data(dy * 2, w), dx * 2)


@pytest.mark.parametrize("case", [(64, 64, 112, 112, 128, 3, 1, 0), (64, 16, 55, 55, 32, 3, 2, 0), (96, 32, 27, 27, 64, 3, 2, 0)],
                         ids=lambda c: "B%d_%dx%dx%d_to%d_k%ds%dp%d" % c)
def test_conv_determinism_and_exact_linearity(T, case):
    """race screen for the double-buffered DMA kernels: run-to-run bit equality, and f(2x) == 2 f(x) bit-for-bit
    (power-of-two scaling commutes with every fp32 rounding, so any mismatch is a stale LDS read)."""
    from cnn_amd import capi

    conv = capi.Conv2d(*case)
    g = T.Generator(device="cuda").manual_seed(11)
    B, Ci, H, W, Co, k, s, pad = case
    x = T.rand((B, Ci, H, W), generator=g, device="cuda")
    w = T.randn((Co, Ci, k, k), generator=g, device="cuda") * 0.1
    b = T.randn((Co,), generator=g, device="cuda") * 0.1
    dy = T.rand(conv.out_shape(), generator=g, device="cuda") * 2 - 1
    for rep in range(3):
        y1, y2, y3 = conv.forward(x, w, b).clone(), conv.forward(x, w, b).clone(), conv.forward(x * 2, w, b * 2).clone()
        assert T.equal(y1, y2) and T.equal(y1 * 2, y3), f"forward rep {rep}"
        d1, d2, d3 = conv.backward_data(dy, w).clone(), conv.backward_data(dy, w).clone(), conv.backward_data(dy * 2, w).clone()
        assert T.equal(d1, d2) and T.equal(d1 * 2, d3), f"dgrad rep {rep}"
        g1, b1 = conv.backward_weight(x, dy, float(B))
        g1, b1 = g1.clone(), b1.clone()
        g2, b2 = conv.backward_weight(x, dy, float(B))
        assert T.equal(g1, g2) and T.equal(b1, b2), f"wgrad rep {rep}"


def test_conv2d_combined_backward_matches_separate_calls(T):
    """cnn_conv2d_backward (weight gradient on the side stream || data gradient) == the two separate entry points"""
    from cnn_amd import capi

    for case in [(4, 16, 55, 55, 32, 3, 2, 0), (3, 64, 13, 13, 128, 3, 2, 0), (2, 3, 224, 224, 16, 3, 2, 0)]:
        x, w, b, dy = _conv_inputs(case, 300)
        conv = capi.Conv2d(*case)
        xd, wd, dyd = dev(T, x), dev(T, w), dev(T, dy)
        gw1, gb1 = conv.backward_weight(xd, dyd, float(case[0]))
        dx1 = conv.backward_data(dyd, wd)
        for _ in range(3):
            gw2, gb2, dx2 = conv.backward(xd, dyd, wd, float(case[0]))  # defer_join=False: ordered on return
            T.cuda.synchronize()
            assert T.equal(gw1, gw2) and T.equal(gb1, gb2) and T.equal(dx1, dx2)


@pytest.mark.parametrize("case", [CONV_CASES[i] for i in (0, 1, 5, 6, 7, 8, 9, 10)] + ROWS_CASES, ids=lambda c: "B%d_%dx%dx%d_to%d_k%ds%dp%d" % c)
def test_conv2d_forward_relu_fusion_is_bit_identical(T, case):
    """cnn_conv2d_forward_relu == cnn_conv2d_forward + cnn_relu_forward, both outputs, every kernel family"""
    from cnn_amd import capi

    x, w, b, _ = _conv_inputs(case, 300)
    conv = capi.Conv2d(*case)
    xd, wd, bd = dev(T, x), dev(T, w), dev(T, b)
    y_sep = conv.forward(xd, wd, bd)
    r_sep = capi.relu_forward(y_sep)
    y_f, r_f = T.full_like(y_sep, 7.0), T.full_like(y_sep, 7.0)
    conv.forward_relu(xd, wd, bd, y_f, r_f)
    assert T.equal(y_f, y_sep) and T.equal(r_f, r_sep)
    assert (r_f >= 0).all() and (r_f == 0).any() and (r_f > 0).any()
    if conv.relu_only_supported():  # y = NULL: only the ReLU output is written
        pf, pd = conv.prepared_buffers("cuda")
        capi.prepare_filters([conv], [wd], [bd], [pf], [pd])
        r_only = T.full_like(y_sep, 7.0)
        conv.forward_prepared(xd, pf, bd, None, r_only)
        assert T.equal(r_only, r_sep)


@pytest.mark.parametrize("shape,k,step", [((2, 16, 111, 111), 2, 2), ((3, 5, 7, 7), 2, 2), ((2, 4, 9, 10), 3, 2), ((2, 3, 8, 8), 3, 1)])
def test_maxpool_backward_relu_fusion_is_bit_identical(T, shape, k, step):
    """cnn_maxpool2d_backward_relu(pooled) == cnn_maxpool2d_backward + cnn_relu_backward(relu_out) where the pool's
    input is that ReLU's output (zeros, ties and NaN included)"""
    from cnn_amd import capi

    pre = uniform_pm1(310, shape)
    pre[0, 0, :4, :4] = -1.0      # an all-zero window after ReLU: argmax value 0 -> delta masked
    pre[-1, -1, 0, 0] = np.nan    # NaN survives ReLU (NaN >= 0 is false -> 0)? relu.cpp:25 gives 0; keep it as data
    relu_out = capi.relu_forward(dev(T, pre))
    relu_np = host(relu_out)
    relu_np[0, 1, 2, 2] = np.nan  # a NaN activation inside the pool input
    relu_out = dev(T, relu_np)
    B, C, H, W = shape
    Ho, Wo = (H - k) // step + 1, (W - k) // step + 1
    pooled = T.empty((B, C, Ho, Wo), device="cuda")
    mask = T.empty((B, C, Ho, Wo), dtype=T.int32, device="cuda")
    L = capi.load()
    capi.check(L.cnn_maxpool2d_forward(capi._ptr(relu_out), capi._ptr(pooled), capi._ptr(mask), B, C, H, W, k, step, None), "pool fwd")
    dy = dev(T, uniform_pm1(311, (B, C, Ho, Wo)))
    d_sep = capi.maxpool_backward(dy, mask, shape, k, step)
    capi.relu_backward(relu_out, d_sep)
    d_f = T.full(shape, 7.0, device="cuda")
    capi.maxpool_backward_relu(dy, mask, pooled, shape, k, step, d_f)
    a, b = host(d_f), host(d_sep)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert (a[0, 0, :4, :4] == 0).all()


def test_whole_net_fused_equals_unfused(T):
    """pynet with the fused Conv+ReLU / Pool+ReLU kernels gives bit-identical parameters, gradients and activations"""
    from cnn_amd.pynet import AlexNetHip

    B = 4
    x = dev(T, uniform01(320, (B, 3, 224, 224)))
    labels = dev(T, (np.arange(B) % 3).astype(np.int32))
    nets = [AlexNetHip(B, 3, fuse=f) for f in (True, False)]
    p0 = normal_scaled(321, (nets[0].n_params,))
    for n in nets:
        n.load_params(p0)
        for _ in range(2):
            n.train_step(x, labels, 1e-3)
    a, b = nets
    assert T.equal(a.params, b.params) and T.equal(a.grads, b.grads)
    for l in range(4):
        assert T.equal(a.conv_out[l], b.conv_out[l]) and T.equal(a.relu_out[l], b.relu_out[l])
        assert T.equal(a.d_conv[l], b.d_conv[l])
    assert T.equal(a.d_pool, b.d_pool)


def test_prepared_filters_match_plain_calls(T):
    """cnn_conv2d_prepare_filters + *_prepared == the plain calls, bit for bit, for every kernel family of the net
    (packed first layer, packed 16-channel dgrad, implicit GEMM with and without whole-image staging)"""
    from cnn_amd import capi

    cases = [(3, 3, 224, 224, 16, 3, 2, 0), (3, 16, 55, 55, 32, 3, 2, 0), (3, 32, 27, 27, 64, 3, 2, 0), (3, 64, 13, 13, 128, 3, 2, 0),
             (2, 8, 12, 12, 8, 3, 1, 0), (2, 6, 10, 10, 40, 3, 1, 1)]
    convs = [capi.Conv2d(*c) for c in cases]
    data = [_conv_inputs(c, 500 + 7 * i) for i, c in enumerate(cases)]
    xs, ws, bs = [dev(T, d[0]) for d in data], [dev(T, d[1]) for d in data], [dev(T, d[2]) for d in data]
    dys = [dev(T, d[3]) for d in data]
    bufs = [c.prepared_buffers() for c in convs]
    capi.prepare_filters(convs, ws, bs, [b[0] for b in bufs], [b[1] for b in bufs])
    for c, x, w, b, dy, (pf, pd) in zip(convs, xs, ws, bs, dys, bufs):
        y0 = c.forward(x, w, b)
        r0 = capi.relu_forward(y0)
        y1, r1 = T.full_like(y0, 7.0), T.full_like(y0, 7.0)
        c.forward_prepared(x, pf, b, y1, r1)
        assert T.equal(y0, y1) and T.equal(r0, r1)
        y2 = T.full_like(y0, 7.0)
        c.forward_prepared(x, pf, b, y2, None)
        assert T.equal(y0, y2)
        dx0 = c.backward_data(dy, w)
        dx1 = T.full_like(dx0, 7.0)
        c.backward_data_prepared(dy, pd, dx1)
        assert T.equal(dx0, dx1)
        gw0, gb0, dx2 = c.backward(x, dy, w, 3.0)
        gw1, gb1, dx3 = T.empty_like(gw0), T.empty_like(gb0), T.full_like(dx0, 7.0)
        c.backward_prepared(x, dy, pd, 3.0, gw1, gb1, dx3)
        T.cuda.synchronize()
        assert T.equal(gw0, gw1) and T.equal(gb0, gb1) and T.equal(dx2, dx3) and T.equal(dx0, dx3)


def test_deferred_input_gradient_is_bit_identical(T):
    """pynet(defer_input_grad=True) launches conv_layer_1's data gradient one forward pass later on a second stream;
    after flush() every tensor, including that gradient, equals the in-order run bit for bit"""
    from cnn_amd.pynet import AlexNetHip

    B = 4
    x = dev(T, uniform01(330, (B, 3, 224, 224)))
    labels = dev(T, (np.arange(B) % 3).astype(np.int32))
    nets = [AlexNetHip(B, 3, defer_input_grad=d) for d in (True, False)]
    p0 = normal_scaled(331, (nets[0].n_params,))
    for n in nets:
        n.load_params(p0)
    for step in range(3):
        for n in nets:
            n.train_step(x, labels, 1e-3)
        a, b = nets
        assert a.pending_dx0 is not None  # still to be launched
        a.flush()
        T.cuda.synchronize()
        assert T.equal(a.params, b.params) and T.equal(a.grads, b.grads) and T.equal(a.d_conv[0], b.d_conv[0]), step
        assert T.equal(a.d_pool, b.d_pool) and T.equal(a.d_conv[1], b.d_conv[1])


@pytest.mark.parametrize("shape", [(2, 224, 224), (3, 37, 41), (1, 9, 9), (2, 12, 20), (5, 7, 5)], ids=lambda s: "B%d_%dx%d" % s)
@pytest.mark.parametrize("prepared", [False, True], ids=["plain", "prepared"])
def test_conv_relu_maxpool_fusion_is_bit_identical(T, shape, prepared):
    """cnn_conv2d_relu_maxpool2_forward == cnn_conv2d_forward_relu + cnn_maxpool2d_forward (pooled values AND argmax mask),
    including windows with ties (ReLU zeros), odd output sizes (last conv row / column outside every window) and NaN / -0"""
    from cnn_amd import capi

    B, H, W = shape
    case = (B, 3, H, W, 16, 3, 2, 0)
    x, w, b, _ = _conv_inputs(case, 500)
    x = x - 0.5  # negative pre-activations: plenty of all-zero windows after the ReLU (first-maximum tie rule)
    if H > 20:
        x[0, :, 4:9, 4:9] = np.nan
        x[-1, 1, 10, 11] = np.inf
    conv = capi.Conv2d(*case)
    assert conv.relu_maxpool2_supported()
    xd, wd, bd = dev(T, x), dev(T, w), dev(T, b)
    Ho, Wo = (H - 3) // 2 + 1, (W - 3) // 2 + 1
    y, r = T.empty((B, 16, Ho, Wo), device="cuda"), T.empty((B, 16, Ho, Wo), device="cuda")
    conv.forward_relu(xd, wd, bd, y, r)
    pooled_ref, mask_ref = capi.maxpool_forward(r, 2, 2)
    pooled = T.full_like(pooled_ref, 7.0)
    mask = T.full_like(mask_ref, -1)
    prep = None
    if prepared:
        pf, pd = conv.prepared_buffers("cuda")
        capi.prepare_filters([conv], [wd], [bd], [pf], [pd])
        prep = pf
    conv.relu_maxpool2_forward(xd, wd, bd, pooled, mask, prepared_fwd=prep)
    assert np.array_equal(host(pooled).view(np.uint32), host(pooled_ref).view(np.uint32))
    # the argmax part is cnn_maxpool2d_forward's mask bit for bit; bit 31 marks exactly the windows whose pooled value is <= 0
    assert np.array_equal(host(mask) & 0x7FFFFFFF, host(mask_ref))
    assert np.array_equal(host(mask) < 0, host(pooled_ref) <= 0)
    pooled2 = T.full_like(pooled_ref, 7.0)  # no_grad path: mask = NULL
    conv.relu_maxpool2_forward(xd, wd, bd, pooled2, None, prepared_fwd=prep)
    assert np.array_equal(host(pooled2).view(np.uint32), host(pooled_ref).view(np.uint32))


def test_conv_relu_maxpool_fusion_rejects_other_layers(T):
    from cnn_amd import capi

    conv = capi.Conv2d(2, 16, 20, 20, 32, 3, 2, 0)
    assert not conv.relu_maxpool2_supported()
    with pytest.raises(capi.CnnAmdError):
        conv.relu_maxpool2_forward(T.zeros((2, 16, 20, 20), device="cuda"), T.zeros((32, 16, 3, 3), device="cuda"),
                                   T.zeros(32, device="cuda"), T.zeros((2, 32, 4, 4), device="cuda"))


@pytest.mark.parametrize("shape", [(2, 224, 224), (3, 37, 41), (1, 9, 9), (2, 12, 20), (5, 7, 5), (2, 30, 26)], ids=lambda s: "B%d_%dx%d" % s)
def test_conv_backward_from_pooled_domain_is_bit_identical(T, shape):
    """cnn_conv2d_backward_{weight,data}_pooled2 == cnn_maxpool2d_backward_relu + cnn_conv2d_backward_{weight,data}"""
    from cnn_amd import capi

    B, H, W = shape
    case = (B, 3, H, W, 16, 3, 2, 0)
    x, w, b, _ = _conv_inputs(case, 600)
    x = x - 0.5
    conv = capi.Conv2d(*case)
    xd, wd, bd = dev(T, x), dev(T, w), dev(T, b)
    Ho, Wo = (H - 3) // 2 + 1, (W - 3) // 2 + 1
    PHo, PWo = Ho // 2, Wo // 2
    pooled = T.empty((B, 16, PHo, PWo), device="cuda")
    mask = T.empty((B, 16, PHo, PWo), dtype=T.int32, device="cuda")
    conv.relu_maxpool2_forward(xd, wd, bd, pooled, mask)
    dpool = dev(T, uniform_pm1(601, (B, 16, PHo, PWo)))
    dy = capi.maxpool_backward_relu(dpool, mask, pooled, (B, 16, Ho, Wo), 2, 2)
    gw_ref, gb_ref = conv.backward_weight(xd, dy, float(B))
    dx_ref = conv.backward_data(dy, wd)
    gw, gb = T.full_like(gw_ref, 7.0), T.full_like(gb_ref, 7.0)
    conv.backward_weight_pooled2(xd, dpool, mask, pooled, float(B), gw, gb)
    assert np.array_equal(host(gw), host(gw_ref)) and np.array_equal(host(gb), host(gb_ref))
    dx = T.full_like(dx_ref, 7.0)
    conv.backward_data_pooled2(dpool, mask, pooled, wd, dx)
    assert np.array_equal(host(dx), host(dx_ref))
    pf, pd = conv.prepared_buffers("cuda")
    capi.prepare_filters([conv], [wd], [bd], [pf], [pd])
    dx2 = T.full_like(dx_ref, 7.0)
    conv.backward_data_pooled2(dpool, mask, pooled, None, dx2, prepared_dgrad=pd)
    assert np.array_equal(host(dx2), host(dx_ref))
    # pooled = None with the MARKED mask of the fused forward kernel (bit 31 = pooled <= 0): the ReLU mask needs no tensor at all ...
    gw3, gb3, dx3 = T.full_like(gw_ref, 7.0), T.full_like(gb_ref, 7.0), T.full_like(dx_ref, 7.0)
    conv.backward_weight_pooled2(xd, dpool, mask, None, float(B), gw3, gb3)
    conv.backward_data_pooled2(dpool, mask, None, wd, dx3)
    assert np.array_equal(host(gw3), host(gw_ref)) and np.array_equal(host(gb3), host(gb_ref)) and np.array_equal(host(dx3), host(dx_ref))
    # ... and with a plain mask (cnn_maxpool2d_forward's) once the caller has applied the ReLU mask to dpool itself
    plain_mask = (mask & 0x7FFFFFFF).to(T.int32)
    dmasked = T.where(pooled <= 0, T.zeros_like(dpool), dpool)
    gw4, gb4, dx4 = T.full_like(gw_ref, 7.0), T.full_like(gb_ref, 7.0), T.full_like(dx_ref, 7.0)
    conv.backward_weight_pooled2(xd, dmasked, plain_mask, None, float(B), gw4, gb4)
    conv.backward_data_pooled2(dmasked, plain_mask, None, wd, dx4)
    assert np.array_equal(host(gw4), host(gw_ref)) and np.array_equal(host(gb4), host(gb_ref)) and np.array_equal(host(dx4), host(dx_ref))


@pytest.mark.parametrize("shape", [(2, 224, 224), (16, 224, 224), (3, 37, 41), (1, 9, 9), (2, 12, 20), (5, 7, 5), (2, 30, 26), (3, 64, 132)],
                         ids=lambda s: "B%d_%dx%d" % s)
def test_packed_pool_mask_block_is_bit_identical(T, shape):
    """CNN_CONV2D_POOL_MASK_PACKED (include/cnn_amd.h): the one-byte pool mask of the fused first block.  The forward call writes the
    same pooled tensor, its mask unpacks to the int32 form bit for bit (bit 31 included), and the pooled-domain gradient calls return
    the same bits from either form -- on ragged shapes (pitch > row length, odd output sizes, last rows of the allocation) too"""
    from cnn_amd import capi

    B, H, W = shape
    case = (B, 3, H, W, 16, 3, 2, 0)
    x, w, b, _ = _conv_inputs(case, 620)
    x = x - 0.5
    conv = capi.Conv2d(*case)
    packed = capi.Conv2d(*case)
    assert packed.pool_mask_packed_supported()
    packed.set_pool_mask_packed()
    xd, wd, bd = dev(T, x), dev(T, w), dev(T, b)
    Ho, Wo = (H - 3) // 2 + 1, (W - 3) // 2 + 1
    PHo, PWo = Ho // 2, Wo // 2
    pitch = (PWo + 3) & ~3
    assert packed.pool_mask_bytes() == B * 16 * PHo * pitch + 64 and conv.pool_mask_bytes() == B * 16 * PHo * PWo * 4
    pooled, pooled8 = T.empty((B, 16, PHo, PWo), device="cuda"), T.full((B, 16, PHo, PWo), 7.0, device="cuda")
    mask = T.empty((B, 16, PHo, PWo), dtype=T.int32, device="cuda")
    mask8 = T.full((packed.pool_mask_bytes(),), 0x55, dtype=T.uint8, device="cuda")
    conv.relu_maxpool2_forward(xd, wd, bd, pooled, mask)
    packed.relu_maxpool2_forward(xd, wd, bd, pooled8, mask8)
    assert np.array_equal(host(pooled8).view(np.uint32), host(pooled).view(np.uint32))
    m8 = host(mask8)
    rows = m8[:B * 16 * PHo * pitch].reshape(B, 16, PHo, pitch)
    assert np.all(rows[..., PWo:] == 0x55) and np.all(m8[B * 16 * PHo * pitch:] == 0x55)  # pad bytes / slack are never written
    assert np.all((rows[..., :PWo] & 0x7C) == 0)
    unpacked = T.full_like(mask, -2)
    packed.pool_mask_unpack(mask8, unpacked)
    assert np.array_equal(host(unpacked), host(mask))
    dpool = dev(T, uniform_pm1(621, (B, 16, PHo, PWo)))
    gw_ref, gb_ref, dx_ref = T.empty_like(wd), T.empty_like(bd), T.empty((B, 3, H, W), device="cuda")
    conv.backward_weight_pooled2(xd, dpool, mask, None, float(B), gw_ref, gb_ref)
    conv.backward_data_pooled2(dpool, mask, None, wd, dx_ref)
    gw, gb, dx = T.full_like(wd, 7.0), T.full_like(bd, 7.0), T.full_like(dx_ref, 7.0)
    packed.backward_weight_pooled2(xd, dpool, mask8, None, float(B), gw, gb)
    packed.backward_data_pooled2(dpool, mask8, None, wd, dx)
    u32 = lambda t: host(t).view(np.uint32)
    assert np.array_equal(u32(gw), u32(gw_ref)) and np.array_equal(u32(gb), u32(gb_ref)) and np.array_equal(u32(dx), u32(dx_ref))
    # the prepared / fused-SGD members of the family
    pf, pd = packed.prepared_buffers("cuda")
    capi.prepare_filters([packed], [wd], [bd], [pf], [pd])
    dx2 = T.full_like(dx_ref, 7.0)
    packed.backward_data_pooled2(dpool, mask8, None, None, dx2, prepared_dgrad=pd)
    assert np.array_equal(u32(dx2), u32(dx_ref))
    w2, b2, w3, b3 = wd.clone(), bd.clone(), wd.clone(), bd.clone()
    ga, gb_a, gc, gb_c = T.empty_like(wd), T.empty_like(bd), T.empty_like(wd), T.empty_like(bd)
    conv.backward_weight_pooled2_sgd(xd, dpool, mask, None, float(B), ga, gb_a, w2, b2, 0.05, 1.0, None, None)
    packed.backward_weight_pooled2_sgd(xd, dpool, mask8, None, float(B), gc, gb_c, w3, b3, 0.05, 1.0, None, None)
    assert np.array_equal(u32(w3), u32(w2)) and np.array_equal(u32(b3), u32(b2)) and np.array_equal(u32(gc), u32(ga))
    # pooled must be NULL with the packed form (bit 7 IS the ReLU mask), and the older kernels of the block do not read it
    with pytest.raises(capi.CnnAmdError):
        packed.backward_weight_pooled2(xd, dpool, mask8, pooled, float(B), gw, gb)
    with capi.option("DGRAD_POOL_LDS", 0):
        assert not packed.pool_mask_packed_supported()
        with pytest.raises(capi.CnnAmdError):
            packed.backward_data_pooled2(dpool, mask8, None, wd, dx)


@pytest.mark.parametrize("shape", [(256, 224, 224), (2, 224, 224), (3, 37, 41), (5, 7, 5)], ids=lambda s: "B%d_%dx%d" % s)
@pytest.mark.parametrize("scale", [1.0, 0.5])
def test_first_layer_weight_gradient_sgd_prepare_fusion_is_bit_identical(T, shape, scale):
    """cnn_conv2d_backward_weight_pooled2_sgd == cnn_conv2d_backward_weight_pooled2 + cnn_sgd_update (on w and bias) +
    cnn_conv2d_prepare_filters: gradients, updated parameters and both filter images, bit for bit"""
    from cnn_amd import capi

    B, H, W = shape
    case = (B, 3, H, W, 16, 3, 2, 0)
    x, w, b, _ = _conv_inputs(case, 610)
    conv = capi.Conv2d(*case)
    xd, wd, bd = dev(T, x - 0.5), dev(T, w), dev(T, b)
    Ho, Wo = (H - 3) // 2 + 1, (W - 3) // 2 + 1
    PHo, PWo = Ho // 2, Wo // 2
    pooled = T.empty((B, 16, PHo, PWo), device="cuda")
    mask = T.empty((B, 16, PHo, PWo), dtype=T.int32, device="cuda")
    conv.relu_maxpool2_forward(xd, wd, bd, pooled, mask)
    dpool = dev(T, uniform_pm1(611, (B, 16, PHo, PWo)))
    lr = 0.05
    # reference: three calls
    gw_ref, gb_ref = T.empty_like(wd), T.empty_like(bd)
    conv.backward_weight_pooled2(xd, dpool, mask, pooled, float(B), gw_ref, gb_ref)
    w_ref, b_ref = wd.clone(), bd.clone()
    capi.sgd_update(w_ref.view(-1), gw_ref.view(-1), lr, scale)
    capi.sgd_update(b_ref, gb_ref, lr, scale)
    pf_ref, pd_ref = conv.prepared_buffers("cuda")
    pf_ref.zero_(); pd_ref.zero_()
    capi.prepare_filters([conv], [w_ref], [b_ref], [pf_ref], [pd_ref])
    # fused
    gw, gb, w2, b2 = T.full_like(wd, 7.0), T.full_like(bd, 7.0), wd.clone(), bd.clone()
    pf, pd = conv.prepared_buffers("cuda")
    pf.zero_(); pd.zero_()
    conv.backward_weight_pooled2_sgd(xd, dpool, mask, pooled, float(B), gw, gb, w2, b2, lr, scale, pf, pd)
    T.cuda.synchronize()
    u32 = lambda t: host(t).view(np.uint32)
    assert np.array_equal(u32(gw), u32(gw_ref)) and np.array_equal(u32(gb), u32(gb_ref))
    assert np.array_equal(u32(w2), u32(w_ref)) and np.array_equal(u32(b2), u32(b_ref))
    assert not np.array_equal(u32(w2), u32(wd))
    nf, nd = 28 * 16, 16 * 32  # floats of the two images (conv_direct.hip: [tap | bias][co], [co][32])
    assert np.array_equal(host(pf).view(np.uint32).ravel()[:nf], host(pf_ref).view(np.uint32).ravel()[:nf])
    assert np.array_equal(host(pd).view(np.uint32).ravel()[:nd], host(pd_ref).view(np.uint32).ravel()[:nd])
    # the images are usable: forward from the new image == forward with the new raw filters
    y1 = T.empty_like(pooled); y2 = T.empty_like(pooled)
    conv.relu_maxpool2_forward(xd, None, None, y1, None, prepared_fwd=pf)
    conv.relu_maxpool2_forward(xd, w_ref, b_ref, y2, None)
    assert T.equal(y1, y2)


@pytest.mark.parametrize("defer", [False, True], ids=["in_order", "deferred_dx0"])
def test_pool_fused_net_is_bit_identical(T, defer):
    """pynet(fuse_pool=True): conv_layer_1 / relu_layer_1 / max_pool_1 as one forward kernel and their backward pass from
    the pooled domain; after every step the parameters, gradients, pool tensors and every delta that is still materialised
    equal the kernel-per-layer run bit for bit (different inputs per step: the two pool buffer sets must not mix)"""
    from cnn_amd.pynet import AlexNetHip

    B = 4
    labels = dev(T, (np.arange(B) % 3).astype(np.int32))
    nets = [AlexNetHip(B, 3, defer_input_grad=defer, fuse_pool=fp) for fp in (True, False)]
    assert nets[0].fuse_pool and not nets[1].fuse_pool
    p0 = normal_scaled(341, (nets[0].n_params,))
    for n in nets:
        n.load_params(p0)
    for step in range(4):
        x = dev(T, uniform01(340 + step, (B, 3, 224, 224)))
        for n in nets:
            n.train_step(x, labels, 1e-3)
        a, b = nets
        for n in nets:
            n.flush()
        T.cuda.synchronize()
        assert T.equal(a.params, b.params) and T.equal(a.grads, b.grads), step
        am = a.pool_mask_int32()  # (the fused block keeps its mask packed, one byte per window, where the library supports it)
        assert T.equal(a.pool_out, b.pool_out) and T.equal(am & 0x7FFFFFFF, b.pool_mask)
        assert T.equal(am < 0, b.pool_out <= 0)  # (bit 31: relu_layer_1's backward mask rides in the pool mask)
        assert T.equal(a.d_conv[0], b.d_conv[0]) and T.equal(a.logits, b.logits)
        assert T.equal(a.d_conv[1], b.d_conv[1])  # d(pool output): conv_layer_2's plain data gradient in both nets


@pytest.mark.parametrize("case", [CONV_CASES[i] for i in (0, 1, 2, 5, 6, 7, 8, 9, 10, 11, 13)] + [(3, 32, 9, 11, 64, 3, 1, 0), (2, 32, 28, 30, 64, 3, 2, 0), (1, 96, 9, 12, 128, 3, 2, 0), (2, 64, 14, 14, 128, 3, 2, 1), (3, 64, 9, 11, 128, 3, 2, 1)] + ROWS_CASES, ids=lambda c: "B%d_%dx%dx%d_to%d_k%ds%dp%d" % c)
def test_conv2d_backward_data_relu_fusion_is_bit_identical(T, case):
    """cnn_conv2d_backward_data_relu == cnn_conv2d_backward_data + cnn_relu_backward, every kernel family, plain and prepared"""
    from cnn_amd import capi

    x, w, b, dy = _conv_inputs(case, 700)
    relu_below = np.maximum(x - 0.5, 0).astype(np.float32)  # a ReLU output with plenty of exact zeros
    relu_below[0, 0, 0, :3] = [np.nan, -0.0, 0.0]
    conv = capi.Conv2d(*case)
    wd, dyd, rd = dev(T, w), dev(T, dy), dev(T, relu_below)
    dx_ref = conv.backward_data(dyd, wd)
    capi.relu_backward(rd, dx_ref)
    dx = T.full_like(dx_ref, 7.0)
    conv.backward_data_relu(dyd, wd, rd, dx)
    assert np.array_equal(host(dx).view(np.uint32), host(dx_ref).view(np.uint32))
    pf, pd = conv.prepared_buffers("cuda")
    capi.prepare_filters([conv], [wd], [dev(T, b)], [pf], [pd])
    dx2 = T.full_like(dx_ref, 7.0)
    conv.backward_data_relu(dyd, None, rd, dx2, prepared_dgrad=pd)
    assert np.array_equal(host(dx2).view(np.uint32), host(dx_ref).view(np.uint32))


FWD_FAMILY_CASES = [
    (3, 32, 27, 27, 64, 3, 2, 0), (5, 64, 13, 13, 128, 3, 2, 0), (2, 16, 55, 55, 32, 3, 2, 0), (3, 32, 9, 11, 64, 3, 1, 0),
    (2, 64, 8, 7, 128, 3, 1, 0), (1, 16, 5, 4, 16, 3, 1, 0), (7, 16, 7, 9, 48, 3, 2, 0), (2, 32, 28, 30, 80, 3, 2, 0),
]


@pytest.mark.parametrize("family", ["lds", "m16"])
@pytest.mark.parametrize("case", FWD_FAMILY_CASES, ids=lambda c: "B%d_%dx%dx%d_to%d_k%ds%dp%d" % c)
def test_conv2d_forward_kernel_families(T, case, family, lib_option):
    """conv_fwd_rd.hip has two forward kernels (LDS filter slice + 32x32x2 tiles | filters in registers + 16x16x4 tiles,
    picked by layer size): each one forced on every shape, from the reference filter layout and from the prepared images,
    pre-activation and fused ReLU outputs, against the oracle"""
    from cnn_amd import capi

    lib_option("FWD_M16", "1" if family == "m16" else "0")
    x, w, b, dy = _conv_inputs(case, 740)
    y_ref = _oracle_conv(case, x, w, b, dy)[0]
    conv = capi.Conv2d(*case)
    xd, wd, bd = dev(T, x), dev(T, w), dev(T, b)
    y = conv.forward(xd, wd, bd)
    assert_close(host(y), y_ref, REL_TOL, "forward")
    pf, pd = conv.prepared_buffers("cuda")
    capi.prepare_filters([conv], [wd], [bd], [pf], [pd])
    y2, yr = T.full_like(y, 7.0), T.full_like(y, 7.0)
    conv.forward_prepared(xd, pf, bd, y2, yr)
    assert np.array_equal(host(y2).view(np.uint32), host(y).view(np.uint32)), "prepared == unprepared, bit for bit"
    relu_ref = host(y).copy()
    relu_ref[~(relu_ref >= 0)] = 0  # relu.cpp:21-26
    assert np.array_equal(host(yr).view(np.uint32), relu_ref.view(np.uint32))
    if conv.relu_only_supported():
        yr2 = T.full_like(y, 7.0)
        conv.forward_prepared(xd, pf, bd, None, yr2)
        assert np.array_equal(host(yr2).view(np.uint32), relu_ref.view(np.uint32))


@pytest.mark.parametrize("case", [(2, 16, 13, 13, 32, 3, 2, 0), (3, 16, 28, 27, 64, 3, 2, 0), (1, 16, 111, 111, 32, 3, 2, 0),
                                  (2, 16, 9, 10, 128, 3, 2, 0), (2, 24, 12, 12, 32, 3, 2, 0), (3, 16, 12, 14, 32, 3, 2, 0), (5, 16, 5, 6, 32, 3, 2, 0)], ids=lambda c: "B%d_%dx%dx%d_to%d_k%ds%dp%d" % c)
def test_conv2d_dgrad_register_direct_opt_in_tiles(T, case, lib_option):
    """conv_dgrad_rd.hip behind CNN_AMD_DGRAD_RD32=1: the Co = 32 instantiation and, for Ci = 16, the tile that packs two
    parity classes into one 32-row MFMA operand -- oracle parity plus bit-identity of the fused ReLU' epilogue"""
    from cnn_amd import capi

    lib_option("DGRAD_RD32", "1")
    x, w, b, dy = _conv_inputs(case, 730)
    _, _, _, dx_ref = _oracle_conv(case, x, w, b, dy)
    relu_below = np.maximum(x - 0.3, 0).astype(np.float32)
    conv = capi.Conv2d(*case)
    wd, dyd, rd = dev(T, w), dev(T, dy), dev(T, relu_below)
    dx = conv.backward_data(dyd, wd)
    assert_close(host(dx), dx_ref, REL_TOL, "data grad")
    masked = dx.clone()
    capi.relu_backward(rd, masked)
    pf, pd = conv.prepared_buffers("cuda")
    capi.prepare_filters([conv], [wd], [dev(T, b)], [pf], [pd])
    for prepared in (None, pd):
        dx2 = T.full_like(dx, 7.0)
        conv.backward_data_relu(dyd, None if prepared is not None else wd, rd, dx2, prepared_dgrad=prepared)
        assert np.array_equal(host(dx2).view(np.uint32), host(masked).view(np.uint32))


@pytest.mark.parametrize("B,n_in,n_out", [(4, 4608, 3), (3, 70, 5), (2, 33, 20)])
def test_linear_backward_relu_fusion_is_bit_identical(T, B, n_in, n_out):
    from cnn_amd import capi

    x = np.maximum(uniform_pm1(710, (B, n_in)), 0).astype(np.float32)
    x[0, :3] = [np.nan, -0.0, 0.0]
    w, dy = normal_scaled(711, (n_in, n_out)), uniform_pm1(712, (B, n_out))
    xd, wd, dyd = dev(T, x), dev(T, w), dev(T, dy)
    gw0, gb0, dx0 = capi.linear_backward(xd, dyd, wd, float(B))
    capi.relu_backward(xd, dx0)
    gw1, gb1, dx1 = capi.linear_backward(xd, dyd, wd, float(B), relu_below=True)
    for a, b2 in ((gw0, gw1), (gb0, gb1), (dx0, dx1)):  # (bit patterns: the NaN in x makes a NaN gradient row)
        assert np.array_equal(host(a).view(np.uint32), host(b2).view(np.uint32))


def test_pool_fused_net_full_batch_is_bit_identical(T):
    """BASELINE configs[1] at its full batch (256 x 3 x 224 x 224): the bench configuration (pool-fused block, deferred conv1
    data gradient, fused ReLU backward, register-direct kernels) against the kernel-per-layer sequence, bit for bit, and the
    loss against a finite range (the oracle needs ~4 s per image batch of 16 here, so it is not run at this size)"""
    from cnn_amd.pynet import AlexNetHip

    B = 256
    g = T.Generator(device="cuda").manual_seed(77)
    x = T.rand((B, 3, 224, 224), generator=g, device="cuda")
    labels = dev(T, (np.arange(B) % 3).astype(np.int32))
    fused = AlexNetHip(B, 3, defer_input_grad=True, fuse_pool=True)
    plain = AlexNetHip(B, 3, fuse=False)
    p0 = normal_scaled(342, (fused.n_params,))
    for n in (fused, plain):
        n.load_params(p0)
    for step in range(2):
        for n in (fused, plain):
            n.train_step(x, labels, 1e-3)
            n.flush()
        T.cuda.synchronize()
        assert T.equal(fused.params, plain.params) and T.equal(fused.grads, plain.grads), step
        assert T.equal(fused.pool_out, plain.pool_out) and T.equal(fused.pool_mask_int32() & 0x7FFFFFFF, plain.pool_mask_int32() & 0x7FFFFFFF)
        assert T.equal(fused.d_conv[0], plain.d_conv[0]) and T.equal(fused.logits, plain.logits)
    loss = float(fused.loss_sum.item()) / B
    assert np.isfinite(loss) and 0.0 < loss < 20.0


@pytest.mark.parametrize("B,n_in,n_out", [(256, 4608, 3), (5, 70, 8), (3, 33, 1)])
def test_linear_forward_softmax_xent_fusion_is_bit_identical(T, B, n_in, n_out):
    """cnn_linear_forward_softmax_xent + cnn_loss_from_terms == cnn_linear_forward + cnn_softmax_xent (logits, probs, delta, loss)"""
    from cnn_amd import capi

    x = dev(T, uniform_pm1(720, (B, n_in)))
    w, b = dev(T, normal_scaled(721, (n_in, n_out))), dev(T, normal_scaled(722, (n_out,)))
    labels = dev(T, (np.arange(B) % n_out).astype(np.int32))
    logits0 = capi.linear_forward(x, w, b)
    probs0, delta0, loss0 = T.empty_like(logits0), T.empty_like(logits0), T.zeros(1, device="cuda")
    lib = capi.load()
    capi.check(lib.cnn_softmax_xent(capi._ptr(logits0), capi._ptr(labels), capi._ptr(probs0), capi._ptr(delta0), capi._ptr(loss0), B, n_out,
                                    capi._stream()), "cnn_softmax_xent")
    logits1, probs1, delta1 = T.full_like(logits0, 7.0), T.full_like(logits0, 7.0), T.full_like(logits0, 7.0)
    terms, loss1 = T.zeros(B, device="cuda"), T.zeros(1, device="cuda")
    capi.check(lib.cnn_linear_forward_softmax_xent(capi._ptr(x), capi._ptr(w), capi._ptr(b), capi._ptr(labels), capi._ptr(logits1),
                                                   capi._ptr(probs1), capi._ptr(delta1), capi._ptr(terms), B, n_in, n_out, capi._stream()),
               "cnn_linear_forward_softmax_xent")
    capi.check(lib.cnn_loss_from_terms(capi._ptr(terms), capi._ptr(loss1), B, capi._stream()), "cnn_loss_from_terms")
    for a, c in ((logits0, logits1), (probs0, probs1), (delta0, delta1), (loss0, loss1)):
        assert np.array_equal(host(a).view(np.uint32), host(c).view(np.uint32))


@pytest.mark.parametrize("relu", [0, 1], ids=["plain", "relu_below"])
@pytest.mark.parametrize("B,n_in,n_out", [(256, 4608, 3), (7, 4608, 3), (5, 70, 8), (3, 33, 1), (4, 9216, 3), (3, 25088, 3), (2, 7000, 3)])
def test_linear_head_with_data_gradient_is_bit_identical(T, B, n_in, n_out, relu):
    """cnn_linear_forward_softmax_xent_dx (forward + loss head + the layer's data gradient in ONE kernel) and the parameter-only
    cnn_linear_backward(dx = NULL) against the unfused trio cnn_linear_forward_softmax_xent + cnn_linear_backward(_relu): logits,
    probabilities, delta, loss terms, dx, gW and gb bit for bit -- the 4608 -> 3 head with everything in registers, and the generic
    widths (linear.cpp:33-90, func.cpp:16-73)"""
    from cnn_amd import capi

    lib = capi.load()
    xh = uniform_pm1(820, (B, n_in))
    if relu:
        xh = np.maximum(xh, 0)  # a ReLU layer's output: zeros where its backward mask applies
    x = dev(T, xh)
    w, b = dev(T, normal_scaled(821, (n_in, n_out))), dev(T, normal_scaled(822, (n_out,)))
    labels = dev(T, (np.arange(B) % n_out).astype(np.int32))
    mk = lambda *shape: T.full(shape, 7.0, device="cuda")
    logits0, probs0, delta0, terms0 = mk(B, n_out), mk(B, n_out), mk(B, n_out), mk(B)
    capi.check(lib.cnn_linear_forward_softmax_xent(capi._ptr(x), capi._ptr(w), capi._ptr(b), capi._ptr(labels), capi._ptr(logits0),
                                                   capi._ptr(probs0), capi._ptr(delta0), capi._ptr(terms0), B, n_in, n_out, capi._stream()),
               "cnn_linear_forward_softmax_xent")
    gw0, gb0, dx0 = capi.linear_backward(x, delta0, w, float(B), relu_below=bool(relu))
    logits1, probs1, delta1, terms1, dx1 = mk(B, n_out), mk(B, n_out), mk(B, n_out), mk(B), mk(B, n_in)
    capi.check(lib.cnn_linear_forward_softmax_xent_dx(capi._ptr(x), capi._ptr(w), capi._ptr(b), capi._ptr(labels), capi._ptr(logits1),
                                                      capi._ptr(probs1), capi._ptr(delta1), capi._ptr(terms1), capi._ptr(dx1), relu, B, n_in, n_out,
                                                      capi._stream()), "cnn_linear_forward_softmax_xent_dx")
    gw1, gb1 = mk(n_in, n_out), mk(n_out)
    capi.check(lib.cnn_linear_backward(capi._ptr(x), capi._ptr(delta1), capi._ptr(w), capi._ptr(gw1), capi._ptr(gb1), None, B, n_in, n_out,
                                       float(B), capi._stream()), "cnn_linear_backward")
    for name, a, c in (("logits", logits0, logits1), ("probs", probs0, probs1), ("delta", delta0, delta1), ("loss terms", terms0, terms1),
                       ("dx", dx0, dx1), ("gW", gw0, gw1), ("gb", gb0, gb1)):
        assert np.array_equal(host(a).view(np.uint32), host(c).view(np.uint32)), name


@pytest.mark.parametrize("shape", [(1, 64, 13, 13), (4, 64, 13, 13), (3, 128, 6, 6), (2, 16, 111, 111), (5, 7, 3, 5)], ids=lambda s: "B%d_C%d_%dx%d" % s)
def test_grad_cam_matches_oracle_bit_for_bit(T, shape):
    """cnn_grad_cam == the restatement of alexnet.cpp:107-140: same summation orders, so the normalised map and the 8-bit picture
    are identical (also with negative maps, a constant plane, and a NaN in element 0)"""
    from cnn_amd import capi
    from oracle import pyoracle as O

    rng = np.random.default_rng(11)
    f = (rng.standard_normal(shape) * 0.5 + 0.1).astype(np.float32)
    for variant in range(3):
        g = f.copy()
        if variant == 1:
            g[0, 0] = -3.0  # a plane that drives sample 0 negative in places
        if variant == 2:
            g[0, :, 0, 0] = np.nan
        cam_o, img_o = O.grad_cam(g)
        cam, img = capi.grad_cam(dev(T, g))
        got = host(cam)
        same = (got.view(np.uint32) == cam_o.view(np.uint32)) | (np.isnan(got) & np.isnan(cam_o))  # (NaN payloads are not compared)
        assert same.all(), (variant, int((~same).sum()))
        assert np.array_equal(host(img), img_o), variant


def _sweep_cases(n, seed):
    rs = np.random.RandomState(seed)
    cases = []
    for _ in range(n):
        s_ = 1 if rs.rand() < 0.75 else 2
        pad = int(rs.randint(0, 2))
        B = int(rs.randint(1, 4))
        Ci, Co = int(rs.choice([3, 8, 16, 24, 32, 40, 64, 72, 96])), int(rs.choice([8, 16, 24, 32, 48, 64, 80, 128]))
        H, W = int(rs.randint(5, 64)), int(rs.randint(5, 64))
        if rs.rand() < 0.3:
            W = H
        cases.append((B, Ci, H, W, Co, 3, s_, pad))
    return cases


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_conv2d_random_geometries_default_dispatch_vs_oracle(T, seed):
    """the DEFAULT dispatch of all three passes on 40 random 3x3 geometries per seed (stride 1 / 2, padding 0 / 1, planes of 5 .. 63 pixels,
    3 .. 96 -> 8 .. 128 channels; conv2d.cpp:69-199 accepts every one of them) against the oracle: the shapes BETWEEN the hand-picked cases
    above, where the runtime-size kernels (conv_rows_any, wgrad_sp_any) meet the per-width instances and the generic kernels.  The same
    generator as tests/sweeps/fuzz_conv.py, which found the wgrad_sp_any fix-up width bug on 30- / 39- / 40- / 59- / 60-wide outputs"""
    from cnn_amd import capi

    served = set()
    for case in _sweep_cases(40, seed):
        x, w, b, dy = _conv_inputs(case, 7000 + seed)
        y_ref, gw_ref, gb_ref, dx_ref = _oracle_conv(case, x, w, b, dy)
        conv = capi.Conv2d(*case)
        xd, wd, bd, dyd = dev(T, x), dev(T, w), dev(T, b), dev(T, dy)
        capi.kernel_timing(1)
        y = conv.forward(xd, wd, bd)
        dx = conv.backward_data(dyd, wd)
        relu_in = capi.relu_forward(xd - 0.5)
        dxm = T.full_like(xd, 7.0)
        conv.backward_data_relu(dyd, wd, relu_in, dxm)
        gw, gb = conv.backward_weight(xd, dyd, float(case[0]))
        T.cuda.synchronize()
        served |= {k.split("|")[0].split("<")[0] for k in capi.kernel_timing_report()}
        capi.kernel_timing(0)
        tag = "sweep B%d %dx%dx%d->%d s%d p%d" % (case[0], case[1], case[2], case[3], case[4], case[6], case[7])
        assert_close(host(y), y_ref, REL_TOL, tag + " forward")
        assert_close(host(dx), dx_ref, REL_TOL, tag + " data gradient")
        assert_close(host(dxm), np.where(host(relu_in) <= 0, np.float32(0), dx_ref), REL_TOL, tag + " data gradient + ReLU'")
        assert_close(host(gw), gw_ref, REL_TOL, tag + " weight gradient")
        assert_close(host(gb), gb_ref, REL_TOL, tag + " bias gradient")
    assert {"conv_rows_any", "wgrad_sp_any"} <= served, served
