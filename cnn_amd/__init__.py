"""cnn_amd -- MI355X (gfx950) implementation of the hermosayhl/CNN layer hot path.

cnn_amd/csrc   hand-written HIP kernels + the C ABI (include/cnn_amd.h) -> cnn_amd/lib/libcnn_amd.so
cnn_amd/host   C++17 mirror of the reference's Tensor3D / Layer classes on top of that ABI
cnn_amd/capi   ctypes plumbing used by tests/ and bench.py (no CPU fallback: it raises if the .so is missing)
"""
from . import capi  # noqa: F401
