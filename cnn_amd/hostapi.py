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
    "cnnh_seq_create": (C.c_void_p, [C.c_int, C.c_int]),
    "cnnh_seq_add_conv": (None, [C.c_void_p, C.c_char_p] + [C.c_int] * 5),
    "cnnh_seq_add_bn": (None, [C.c_void_p, C.c_char_p, C.c_int]),
    "cnnh_seq_add_relu": (None, [C.c_void_p, C.c_char_p]),
    "cnnh_seq_add_dropout": (None, [C.c_void_p, C.c_char_p, C.c_float]),
    "cnnh_seq_add_pool": (None, [C.c_void_p, C.c_char_p, C.c_int, C.c_int]),
    "cnnh_seq_add_linear": (None, [C.c_void_p, C.c_char_p, C.c_int, C.c_int]),
    "cnnh_seq_finalize": (None, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "cnnh_stack_create": (C.c_void_p, [C.c_char_p, C.c_int, C.c_int]),
    "cnnh_net_describe": (C.c_size_t, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "cnnh_net_set_comm": (None, [C.c_void_p, C.c_void_p, C.c_int]),
    "cnnh_net_allreduce_gradients": (None, [C.c_void_p]),
    "cnnh_net_update_auto": (None, [C.c_void_p, C.c_float]),
    "cnnh_net_destroy": (None, [C.c_void_p]),
    "cnnh_net_num_params": (C.c_size_t, [C.c_void_p]),
    "cnnh_net_params_device": (C.c_void_p, [C.c_void_p]),
    "cnnh_net_grads_device": (C.c_void_p, [C.c_void_p]),
    "cnnh_set_stream": (None, [C.c_void_p]),
    "cnnh_set_no_grad": (None, [C.c_int]),
    "cnnh_set_fuse_layers": (None, [C.c_int]),
    "cnnh_set_fuse_pool_block": (None, [C.c_int]),
    "cnnh_set_input_gradient": (None, [C.c_int]),
    "cnnh_net_set_params": (None, [C.c_void_p, _F]),
    "cnnh_net_get_params": (None, [C.c_void_p, _F]),
    "cnnh_net_get_grads": (None, [C.c_void_p, _F]),
    "cnnh_net_load_checkpoint": (C.c_int, [C.c_void_p, C.c_char_p]),
    "cnnh_net_save_checkpoint": (None, [C.c_void_p, C.c_char_p]),
    "cnnh_net_forward_host": (None, [C.c_void_p, _F, C.c_int, C.c_int, C.c_int, _F]),
    "cnnh_net_train_step_host": (C.c_float, [C.c_void_p, _F, _I, C.c_int, C.c_int, C.c_int, C.c_float, _F]),
    "cnnh_net_train_step_device": (C.c_float, [C.c_void_p, C.c_void_p, _I, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]),
    "cnnh_net_update": (None, [C.c_void_p, C.c_float, C.c_float]),
    "cnnh_net_train_step_device_loss": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float]),
    "cnnh_net_last_loss": (C.c_float, [C.c_void_p]),
    "cnnh_net_flush": (None, [C.c_void_p]),
    "cnnh_net_input_delta": (C.c_int, [C.c_void_p, _F, C.c_size_t]),
    "cnnh_net_layer_output": (C.c_int, [C.c_void_p, C.c_char_p, _F, C.c_size_t]),
    "cnnh_net_grad_cam": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, _F, C.c_size_t]),
}


def load():
    global _lib
    if _lib is None:
        capi.load()  # libcnn_amd.so first (the host library links against it)
        if os.environ.get("CNN_AMD_LIB"):
            raise capi.CnnAmdError("CNN_AMD_LIB selects another build of the C ABI; libcnn_amd_host.so is linked against the in-tree libcnn_amd.so")
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


class HostNet:
    """architectures::Sequential / AlexNet (cnn_amd/host) behind a handle: the C++ Layer API a reference user links against."""

    h = None

    def _adopt(self, handle, classes, params, grads):
        self.classes = classes
        self._keep = (params, grads)  # torch tensors that own the arenas, if any
        self.h = C.c_void_p(handle)
        self.n_params = int(self.lib.cnnh_net_num_params(self.h))

    def describe(self):
        """[(layer name, parameter count)] in layer order"""
        n = int(self.lib.cnnh_net_describe(self.h, None, 0))
        buf = C.create_string_buffer(n)
        self.lib.cnnh_net_describe(self.h, buf, n)
        return [(ln.rsplit(":", 1)[0], int(ln.rsplit(":", 1)[1])) for ln in buf.value.decode().splitlines()]

    def set_comm(self, comm, world):
        """comm: the void* of cnn_comm_init_rank / cnn_comm_init_all (int or ctypes.c_void_p), or None"""
        self.lib.cnnh_net_set_comm(self.h, comm, int(world))

    def allreduce_gradients(self):
        self.lib.cnnh_net_allreduce_gradients(self.h)

    def update_auto(self, lr):
        """Sequential::update_gradients(lr): all-reduce + lr/world when a communicator is set"""
        self.lib.cnnh_net_update_auto(self.h, float(lr))

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

    def train_step(self, x_dev, labels_dev, lr):
        """Sequential::train_step: forward, device-side softmax / cross-entropy, backward, [all-reduce], SGD -- nothing is read
        back and the host never blocks.  x_dev float32 [B][C][H][W], labels_dev int32 [B], both device tensors."""
        B, _, H, W = x_dev.shape
        self.lib.cnnh_net_train_step_device_loss(self.h, C.c_void_p(x_dev.data_ptr()), C.c_void_p(labels_dev.data_ptr()), B, H, W,
                                                 float(lr))

    def train_step_ptr(self, x_ptr, labels_dev, B, H, W, lr):
        """train_step on a raw device pointer (e.g. a capi.BatchStager slot)"""
        self.lib.cnnh_net_train_step_device_loss(self.h, C.c_void_p(x_ptr), C.c_void_p(labels_dev.data_ptr()), B, H, W, float(lr))

    def last_loss(self):
        return float(self.lib.cnnh_net_last_loss(self.h))

    def input_delta(self, shape):
        """the delta with respect to the network input of the last backward pass (the first layer's data gradient)"""
        out = np.empty(shape, np.float32)
        rc = self.lib.cnnh_net_input_delta(self.h, _fp(out), out.size)
        if rc != 0:
            raise KeyError(f"input_delta: rc={rc}")
        return out

    def flush(self):
        """Sequential::flush_deferred(): the data gradient train_step deferred into the next pass is launched / ordered now"""
        self.lib.cnnh_net_flush(self.h)

    def layer_output(self, name, shape):
        out = np.empty(shape, np.float32)
        rc = self.lib.cnnh_net_layer_output(self.h, name.encode(), _fp(out), out.size)
        if rc == 3:  # std::runtime_error from Layer::get_output(): the tensor was fused away and its parameters are gone
            raise RuntimeError(f"layer {name}: get_output() of a tensor the last forward pass did not write, after its parameters were overwritten")
        if rc != 0:
            raise KeyError(f"layer {name}: rc={rc}")
        return out


    def grad_cam(self, layer_name, shape):
        """Sequential::grad_cam (alexnet.cpp:95-142) after a forward pass with gradients enabled; shape = (B, H, W) of the layer's
        feature map.  Returns (uint8 image [H][W] -- the reference's cv::Mat --, normalised cam [B][H][W])"""
        B, H, W = shape
        img = np.empty((H, W), np.uint8)
        cam = np.empty((B, H, W), np.float32)
        rc = self.lib.cnnh_net_grad_cam(self.h, layer_name.encode(), img.ctypes.data_as(C.c_void_p), img.size, _fp(cam), cam.size)
        if rc != 0:
            raise KeyError(f"grad_cam({layer_name}): rc={rc}")
        return img, cam


class HostAlexNet(HostNet):
    """architectures::AlexNet: the reference's fixed list (alexnet.cpp:10-33)"""

    def __init__(self, classes=3, params=None, grads=None, batch_norm=False):
        self.lib = load()
        p = C.c_void_p(params.data_ptr()) if params is not None else None
        g = C.c_void_p(grads.data_ptr()) if grads is not None else None
        self._adopt(self.lib.cnnh_net_create_ex(classes, p, g, 1 if batch_norm else 0), classes, params, grads)


def _layer_names(spec):
    """the naming scheme of sequential.cpp's StackBuilder / the reference (conv_layer_N, bn_layer_N, relu_layer_N, max_pool_N)"""
    names, n_conv, n_pool, n_lin = [], 0, 0, 0
    for item in spec:
        kind = item[0]
        if kind == "conv":
            n_conv += 1
            names.append(f"conv_layer_{n_conv}")
        elif kind == "bn":
            names.append(f"bn_layer_{n_conv}")
        elif kind == "relu":
            names.append(f"relu_layer_{n_conv}")
        elif kind == "pool":
            n_pool += 1
            names.append(f"max_pool_{n_pool}")
        elif kind == "dropout":
            names.append(f"dropout_layer_{n_conv}")
        else:
            n_lin += 1
            names.append(f"linear_{n_lin}")
    return names


class HostSequential(HostNet):
    """architectures::Sequential built from a cnn_amd.stacks spec (any list of the reference's layer types)"""

    def __init__(self, spec, in_shape=(3, 224, 224), params=None, grads=None):
        from . import stacks

        self.lib = load()
        self.spec = list(spec)
        self.layout = stacks.walk(self.spec, *in_shape)
        self.names = _layer_names(self.spec)
        classes = self.layout[-1]["out"][0]
        h = C.c_void_p(self.lib.cnnh_seq_create(in_shape[0], classes))
        for name, item, ent in zip(self.names, self.spec, self.layout):
            nm = name.encode()
            if item[0] == "conv":
                self.lib.cnnh_seq_add_conv(h, nm, ent["in"][0], item[1], item[2], item[3], item[4])
            elif item[0] == "bn":
                self.lib.cnnh_seq_add_bn(h, nm, ent["in"][0])
            elif item[0] == "relu":
                self.lib.cnnh_seq_add_relu(h, nm)
            elif item[0] == "pool":
                self.lib.cnnh_seq_add_pool(h, nm, item[1], item[2])
            elif item[0] == "dropout":
                self.lib.cnnh_seq_add_dropout(h, nm, float(item[1]))
            else:
                self.lib.cnnh_seq_add_linear(h, nm, ent["n_in"], ent["n_out"])
        p = C.c_void_p(params.data_ptr()) if params is not None else None
        g = C.c_void_p(grads.data_ptr()) if grads is not None else None
        self.lib.cnnh_seq_finalize(h, p, g)
        self._adopt(h.value, classes, params, grads)
        assert self.n_params == sum(e["params"] for e in self.layout)


class HostStack(HostNet):
    """architectures::build_vgg11 / build_resnet18 (sequential.cpp) -- the C++ side's own builders of the BASELINE stacks"""

    def __init__(self, which, classes=3, batch_norm=False):
        self.lib = load()
        h = self.lib.cnnh_stack_create(which.encode(), classes, 1 if batch_norm else 0)
        if not h:
            raise KeyError(which)
        hh = C.c_void_p(h)
        self.lib.cnnh_seq_finalize(hh, None, None)
        self._adopt(h, classes, None, None)
