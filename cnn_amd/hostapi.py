"""ctypes front end of libcnn_amd_host.so -- the C++17 mirror of the reference's Layer/AlexNet API
(cnn_amd/host).  Used by tests/ and bench.py to drive exactly the code path a reference user links against."""
import ctypes as C
import os

import numpy as np

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcnn_amd_host.so")
_lib = None

_F = C.POINTER(C.c_float)
_I = C.POINTER(C.c_int)
SIGNATURES = {
    "cnnh_net_create": (C.c_void_p, [C.c_int, C.c_void_p, C.c_void_p]),
    "cnnh_net_create_ex": (C.c_void_p, [C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "cnnh_net_destroy": (None, [C.c_void_p]),
    "cnnh_net_num_params": (C.c_size_t, [C.c_void_p]),
    "cnnh_net_params_device": (C.c_void_p, [C.c_void_p]),
    "cnnh_net_grads_device": (C.c_void_p, [C.c_void_p]),
    "cnnh_set_stream": (None, [C.c_void_p]),
    "cnnh_set_no_grad": (None, [C.c_int]),
    "cnnh_set_fuse_layers": (None, [C.c_int]),
    "cnnh_set_fuse_pool_block": (None, [C.c_int]),
    "cnnh_net_set_params": (None, [C.c_void_p, _F]),
    "cnnh_net_get_params": (None, [C.c_void_p, _F]),
    "cnnh_net_get_grads": (None, [C.c_void_p, _F]),
    "cnnh_net_load_checkpoint": (C.c_int, [C.c_void_p, C.c_char_p]),
    "cnnh_net_save_checkpoint": (None, [C.c_void_p, C.c_char_p]),
    "cnnh_net_forward_host": (None, [C.c_void_p, _F, C.c_int, C.c_int, C.c_int, _F]),
    "cnnh_net_train_step_host": (C.c_float, [C.c_void_p, _F, _I, C.c_int, C.c_int, C.c_int, C.c_float, _F]),
    "cnnh_net_train_step_device": (C.c_float, [C.c_void_p, C.c_void_p, _I, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]),
    "cnnh_net_update": (None, [C.c_void_p, C.c_float, C.c_float]),
    "cnnh_net_layer_output": (C.c_int, [C.c_void_p, C.c_char_p, _F, C.c_size_t]),
}


def load():
    global _lib
    if _lib is None:
        capi.load()  # libcnn_amd.so first (the host library links against it)
        if not os.path.exists(LIB_PATH):
            raise capi.CnnAmdError(f"{LIB_PATH} is missing: run __graft_entry__.build()")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _fp(a):
    return a.ctypes.data_as(_F)


class HostAlexNet:
    """architectures::AlexNet (cnn_amd/host) behind a handle."""

    def __init__(self, classes=3, params=None, grads=None, batch_norm=False):
        self.lib = load()
        self.classes = classes
        self._keep = (params, grads)  # torch tensors that own the arenas, if any
        p = C.c_void_p(params.data_ptr()) if params is not None else None
        g = C.c_void_p(grads.data_ptr()) if grads is not None else None
        self.h = C.c_void_p(self.lib.cnnh_net_create_ex(classes, p, g, 1 if batch_norm else 0))
        self.n_params = int(self.lib.cnnh_net_num_params(self.h))

    def close(self):
        if self.h:
            self.lib.cnnh_net_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, flat):
        flat = np.ascontiguousarray(flat, dtype=np.float32)
        assert flat.size == self.n_params
        self.lib.cnnh_net_set_params(self.h, _fp(flat))

    def get_params(self):
        out = np.empty(self.n_params, np.float32)
        self.lib.cnnh_net_get_params(self.h, _fp(out))
        return out

    def get_grads(self):
        out = np.empty(self.n_params, np.float32)
        self.lib.cnnh_net_get_grads(self.h, _fp(out))
        return out

    def load_checkpoint(self, path):
        if self.lib.cnnh_net_load_checkpoint(self.h, str(path).encode()) != 0:
            raise FileNotFoundError(path)

    def save_checkpoint(self, path):
        self.lib.cnnh_net_save_checkpoint(self.h, str(path).encode())

    def forward_host(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        B, _, H, W = x.shape
        out = np.empty((B, self.classes), np.float32)
        self.lib.cnnh_net_forward_host(self.h, _fp(x), B, H, W, _fp(out))
        return out

    def train_step_host(self, x, labels, lr):
        x = np.ascontiguousarray(x, dtype=np.float32)
        labels = np.ascontiguousarray(labels, dtype=np.int32)
        B, _, H, W = x.shape
        probs = np.empty((B, self.classes), np.float32)
        loss = self.lib.cnnh_net_train_step_host(self.h, _fp(x), labels.ctypes.data_as(_I), B, H, W, float(lr), _fp(probs))
        return float(loss), probs

    def train_step_device(self, x_dev, labels, lr, do_update=True):
        labels = np.ascontiguousarray(labels, dtype=np.int32)
        B, _, H, W = x_dev.shape
        return float(self.lib.cnnh_net_train_step_device(self.h, C.c_void_p(x_dev.data_ptr()), labels.ctypes.data_as(_I), B, H,
                                                         W, float(lr), 1 if do_update else 0))

    def update(self, lr, grad_scale=1.0):
        self.lib.cnnh_net_update(self.h, float(lr), float(grad_scale))

    def layer_output(self, name, shape):
        out = np.empty(shape, np.float32)
        rc = self.lib.cnnh_net_layer_output(self.h, name.encode(), _fp(out), out.size)
        if rc != 0:
            raise KeyError(f"layer {name}: rc={rc}")
        return out
