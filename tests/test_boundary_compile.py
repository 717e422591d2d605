"""The drop-in claim of north_star ("keeping the existing Layer::forward/backward C++ API and Tensor struct so it drops into
cpu/src unchanged"), checked with a compiler instead of greps:
  * CPU box: where /root/reference exists (this container; not the GPU box), the reference's OWN cpu/src/func.cpp and
    cpu/src/alexnet.cpp compile against cnn_amd/host/include (the latter up to its OpenCV-typed grad_cam, :95 -- OpenCV is not
    part of this build); tests/caller/plain_list_caller.cpp -- this repo's own caller of the public Layer API: stand-alone
    layers in a vector built from a run-time spec, no arena, no fusion wiring -- compiles too;
  * GPU box: that caller is linked BESIDE libcnn_amd_host.so, trains two steps through the device layers, and its checkpoint
    equals the arena-based AlexNet's byte for byte."""
import ctypes as C
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

from tests.util import uniform01

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = ["-I" + os.path.join(ROOT, "cnn_amd", "host", "include"), "-I" + os.path.join(ROOT, "include")]
TU = os.path.join(ROOT, "tests", "caller", "plain_list_caller.cpp")
SPEC = ("conv conv_layer_1 3 16 3 2; relu relu_layer_1; pool max_pool_1 2 2; conv conv_layer_2 16 32 3 2; relu relu_layer_2; "
        "conv conv_layer_3 32 64 3 2; relu relu_layer_3; conv conv_layer_4 64 128 3 2; relu relu_layer_4; linear linear_1 4608 3")
REF = "/root/reference/cpu/src"

needs_gxx = pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")


def _syntax(path, extra=()):
    return subprocess.run(["g++", "-std=c++17", "-fsyntax-only", *INC, *extra, path], capture_output=True, text=True)


@needs_gxx
def test_plain_list_caller_compiles_against_the_host_headers():
    out = _syntax(TU, ["-Wall", "-Wextra", "-Wno-unused-parameter"])
    assert out.returncode == 0, out.stderr[-3000:]


@needs_gxx
@pytest.mark.skipif(not os.path.exists(REF), reason="the reference tree only exists in the build container")
def test_reference_sources_compile_against_the_host_headers():
    # func.cpp: relies on <cfloat> / <cmath> arriving through data_format.h (the reference gets them from <opencv2/core.hpp>)
    out = _syntax(os.path.join(REF, "func.cpp"))
    assert out.returncode == 0, out.stderr[-3000:]
    # alexnet.cpp: everything the reference DEFINES on AlexNet (:10-90) must match a declaration in architectures.h; the first
    # and only complaint allowed is the OpenCV return type of grad_cam at :95
    out = _syntax(os.path.join(REF, "alexnet.cpp"))
    lines = sorted({int(m.group(1)) for m in re.finditer(r"alexnet\.cpp:(\d+):\d+: error", out.stderr)})
    assert lines and lines[0] >= 95, out.stderr[-3000:]
    assert "'cv'" in out.stderr or "cv::" in out.stderr


@pytest.mark.gpu
@needs_gxx
def test_plain_list_caller_drops_in_on_the_device_layers(tmp_path, golden_dir):
    import torch  # noqa: F401  (one HIP runtime per process: torch's first)

    from cnn_amd import capi, hostapi

    capi.load()
    hostapi.load()
    libdir = os.path.join(ROOT, "cnn_amd", "lib")
    lib = str(tmp_path / "libplain_caller.so")
    cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", *INC, TU, "-L" + libdir, "-lcnn_amd_host", "-lcnn_amd", "-Wl,-rpath," + libdir, "-o", lib]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-4000:]
    caller = C.CDLL(lib)
    caller.plain_list_train.restype = C.c_float
    caller.plain_list_train.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_float, C.POINTER(C.c_int), C.c_char_p]
    B, steps, lr = 3, 2, 1e-3
    x = uniform01(70, (B, 3, 224, 224))
    labels = np.array([1, 0, 2], np.int32)
    ckpt = os.path.join(golden_dir, "readme_kat_checkpoint.model")
    saved = str(tmp_path / "plain_caller.model")
    predict = np.zeros(B, np.int32)
    mean_loss = caller.plain_list_train(SPEC.encode(), ckpt.encode(), x.ctypes.data_as(C.POINTER(C.c_float)),
                                        labels.ctypes.data_as(C.POINTER(C.c_int)), B, 224, 224, 3, steps, lr,
                                        predict.ctypes.data_as(C.POINTER(C.c_int)), saved.encode())
    assert mean_loss >= 0
    # the same two iterations through this build's own AlexNet (flat arena, fused + prepared kernels)
    net = hostapi.HostAlexNet(3)
    net.load_checkpoint(ckpt)
    losses = [net.train_step_host(x, labels, lr)[0] for _ in range(steps)]
    got = np.fromfile(saved, dtype=np.float32)
    want = net.get_params()
    assert got.size == want.size == 111267
    assert abs(mean_loss - float(np.mean(losses))) <= 1e-6 * max(1.0, abs(mean_loss))
    # same kernels, same order of operations (fusions and prepared filters are bit-identical): the checkpoints agree exactly
    assert np.array_equal(got, want), float(np.abs(got - want).max())
    assert predict.min() >= 0 and predict.max() < 3
    net.close()
