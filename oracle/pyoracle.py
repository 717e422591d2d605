"""ctypes front end of oracle/cnn_oracle.c (TEST INFRASTRUCTURE -- see that file's header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "build", "liboracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "cnn_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _declare(_lib)
    return _lib


def _declare(L):
    for suf, real in (("", C.c_float), ("_f64", C.c_double)):
        P = C.POINTER(real)
        I = C.c_int
        IP = C.POINTER(C.c_int)

        def sig(name, res, *args):
            f = getattr(L, name + suf)
            f.restype = res
            f.argtypes = list(args)

        sig("oracle_conv2d_forward", None, P, P, P, P, I, I, I, I, I, I, I)
        sig("oracle_conv2d_backward", None, P, P, P, P, P, P, I, I, I, I, I, I, I)
        sig("oracle_sgd_update", None, P, P, C.c_size_t, real)
        sig("oracle_set_threads", None, I)
        sig("oracle_get_threads", I)
        sig("oracle_maxpool_forward", None, P, P, IP, I, I, I, I, I, I)
        sig("oracle_maxpool_backward", None, P, IP, P, I, I, I, I, I, I)
        sig("oracle_relu_forward", None, P, P, C.c_size_t)
        sig("oracle_relu_backward", None, P, P, C.c_size_t)
        sig("oracle_dropout_forward", None, P, P, I, I, I, real, I)
        sig("oracle_dropout_backward", None, P, I, I, I, real)
        sig("oracle_grad_cam", None, P, I, I, I, I, P, C.c_void_p)
        sig("oracle_linear_forward", None, P, P, P, P, I, I, I)
        sig("oracle_linear_backward", None, P, P, P, P, P, P, I, I, I)
        sig("oracle_batchnorm_forward", None, P, P, P, P, P, P, P, P, P, I, I, I, I, real, real, I)
        sig("oracle_batchnorm_backward", None, P, P, P, P, P, P, P, I, I, I, I, real)
        sig("oracle_softmax", None, P, P, I, I)
        sig("oracle_cross_entropy_backward", real, P, IP, P, I, I)
        sig("oracle_net_create", C.c_void_p, I, I, I, I)
        sig("oracle_net_destroy", None, C.c_void_p)
        sig("oracle_net_num_params", C.c_size_t, C.c_void_p)
        sig("oracle_net_params", P, C.c_void_p)
        sig("oracle_net_grads", P, C.c_void_p)
        sig("oracle_net_linear_in", I, C.c_void_p)
        sig("oracle_net_forward", P, C.c_void_p, P)
        sig("oracle_net_backward", None, C.c_void_p, P)
        sig("oracle_net_update", None, C.c_void_p, real)
        sig("oracle_net_train_step", real, C.c_void_p, P, IP, real, P)
        sig("oracle_net_conv_out", P, C.c_void_p, I)
        sig("oracle_net_relu_out", P, C.c_void_p, I)
        sig("oracle_net_pool_out", P, C.c_void_p)
        sig("oracle_net_pool_mask", IP, C.c_void_p)
        sig("oracle_net_d_conv", P, C.c_void_p, I)


def _dt(f64):
    return (np.float64, C.c_double, "_f64") if f64 else (np.float32, C.c_float, "")


def _p(a, ct):
    return None if a is None else a.ctypes.data_as(C.POINTER(ct))


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def conv_out_dim(h, k, s):
    return (h - k) // s + 1


def set_threads(n):
    """threads of the convolution loop nests (0 = all host cores).  Results are bit-identical for every count (the split never
    reorders a single element's accumulation); the timed cpu_baseline uses 1 like the single-threaded reference."""
    lib().oracle_set_threads(int(n))


def get_threads():
    return int(lib().oracle_get_threads())


def _pad_hw(x, p):
    """Tensor3D::pad(p) (data_format.cpp:139-150) on every sample / channel: zero border of p rows / columns"""
    return x if p == 0 else np.pad(x, ((0, 0), (0, 0), (p, p), (p, p)))


def conv2d_forward_padded(x, w, bias, stride, pad, f64=False):
    """the padding extension's oracle: the reference convolution on the Tensor3D::pad(p)-ed input"""
    return conv2d_forward(_pad_hw(x, pad), w, bias, stride, f64=f64)


def conv2d_backward_padded(x, dy, w, stride, pad, f64=False, need=(True, True, True)):
    """... and its backward pass with the data gradient cropped back to the unpadded frame"""
    gw, gb, dx = conv2d_backward(_pad_hw(x, pad), dy, w, stride, f64=f64, need=need)
    if dx is not None and pad:
        dx = np.ascontiguousarray(dx[:, :, pad:-pad, pad:-pad])
    return gw, gb, dx


def conv2d_forward(x, w, bias, stride, f64=False):
    dt, ct, suf = _dt(f64)
    x, w, bias = _c(x, dt), _c(w, dt), _c(bias, dt)
    B, Ci, H, W = x.shape
    Co, _, k, _ = w.shape
    y = np.empty((B, Co, conv_out_dim(H, k, stride), conv_out_dim(W, k, stride)), dt)
    getattr(lib(), "oracle_conv2d_forward" + suf)(_p(x, ct), _p(w, ct), _p(bias, ct), _p(y, ct), B, Ci, H, W, Co, k, stride)
    return y


def conv2d_backward(x, dy, w, stride, f64=False, need=(True, True, True)):
    dt, ct, suf = _dt(f64)
    x, dy, w = _c(x, dt), _c(dy, dt), _c(w, dt)
    B, Ci, H, W = x.shape
    Co, _, k, _ = w.shape
    gw = np.empty_like(w) if need[0] else None
    gb = np.empty(Co, dt) if need[1] else None
    dx = np.empty_like(x) if need[2] else None
    getattr(lib(), "oracle_conv2d_backward" + suf)(
        _p(x, ct), _p(dy, ct), _p(w, ct), _p(gw, ct), _p(gb, ct), _p(dx, ct), B, Ci, H, W, Co, k, stride
    )
    return gw, gb, dx


def maxpool_forward(x, k, step, record_mask=True, f64=False):
    dt, ct, suf = _dt(f64)
    x = _c(x, dt)
    B, Cc, H, W = x.shape
    Ho, Wo = (H - k) // step + 1, (W - k) // step + 1
    y = np.empty((B, Cc, Ho, Wo), dt)
    mask = np.zeros((B, Cc, Ho, Wo), np.int32) if record_mask else None
    getattr(lib(), "oracle_maxpool_forward" + suf)(_p(x, ct), _p(y, ct), _p(mask, C.c_int), B, Cc, H, W, k, step)
    return y, mask


def maxpool_backward(dy, mask, in_shape, k, step, f64=False):
    dt, ct, suf = _dt(f64)
    dy = _c(dy, dt)
    mask = _c(mask, np.int32)
    B, Cc, H, W = in_shape
    dx = np.empty(in_shape, dt)
    getattr(lib(), "oracle_maxpool_backward" + suf)(_p(dy, ct), _p(mask, C.c_int), _p(dx, ct), B, Cc, H, W, k, step)
    return dx


def relu_forward(x, f64=False):
    dt, ct, suf = _dt(f64)
    x = _c(x, dt)
    y = np.empty_like(x)
    getattr(lib(), "oracle_relu_forward" + suf)(_p(x, ct), _p(y, ct), x.size)
    return y


def relu_backward(y, dy, f64=False):
    """Returns the masked delta (the reference masks in place, relu.cpp:37-39)."""
    dt, ct, suf = _dt(f64)
    y = _c(y, dt)
    dy = np.array(dy, dtype=dt, order="C", copy=True)
    getattr(lib(), "oracle_relu_backward" + suf)(_p(y, ct), _p(dy, ct), y.size)
    return dy


def dropout_forward(x, p, training=True, f64=False):
    dt, ct, suf = _dt(f64)
    x = _c(x, dt)
    B, Cc, H, W = x.shape
    y = np.empty_like(x)
    getattr(lib(), "oracle_dropout_forward" + suf)(_p(x, ct), _p(y, ct), B, Cc, H * W, ct(p), 1 if training else 0)
    return y


def dropout_backward(dy, p, f64=False):
    dt, ct, suf = _dt(f64)
    d = np.array(dy, dtype=dt, order="C", copy=True)
    B, Cc, H, W = d.shape
    getattr(lib(), "oracle_dropout_backward" + suf)(_p(d, ct), B, Cc, H * W, ct(p))
    return d


def grad_cam(feature, f64=False):
    """(normalised cam [B][H][W], uint8 image [H][W] of plane 0) -- alexnet.cpp:107-140"""
    dt, ct, suf = _dt(f64)
    f = _c(feature, dt)
    B, Cc, H, W = f.shape
    cam = np.empty((B, H, W), dtype=dt)
    img = np.empty((H, W), dtype=np.uint8)
    getattr(lib(), "oracle_grad_cam" + suf)(_p(f, ct), B, Cc, H, W, _p(cam, ct), img.ctypes.data_as(C.c_void_p))
    return cam, img


def linear_forward(x, w, bias, f64=False):
    dt, ct, suf = _dt(f64)
    x, w, bias = _c(x, dt), _c(w, dt), _c(bias, dt)
    B = x.shape[0]
    n_in, n_out = w.shape
    x2 = x.reshape(B, n_in)
    y = np.empty((B, n_out), dt)
    getattr(lib(), "oracle_linear_forward" + suf)(_p(x2, ct), _p(w, ct), _p(bias, ct), _p(y, ct), B, n_in, n_out)
    return y


def linear_backward(x, dy, w, f64=False):
    dt, ct, suf = _dt(f64)
    x, dy, w = _c(x, dt), _c(dy, dt), _c(w, dt)
    B = x.shape[0]
    n_in, n_out = w.shape
    gw = np.empty_like(w)
    gb = np.empty(n_out, dt)
    dx = np.empty((B, n_in), dt)
    getattr(lib(), "oracle_linear_backward" + suf)(
        _p(x.reshape(B, n_in), ct), _p(dy, ct), _p(w, ct), _p(gw, ct), _p(gb, ct), _p(dx, ct), B, n_in, n_out
    )
    return gw, gb, dx.reshape(x.shape)


def batchnorm_forward(x, gamma, beta, moving_mean, moving_var, eps=1e-5, momentum=0.1, training=True, f64=False):
    """-> (y, norm, saved_mean, saved_var, new_moving_mean, new_moving_var)   (batchnorm2d.cpp:24-95)"""
    dt, ct, suf = _dt(f64)
    x = _c(x, dt)
    B, Cc, H, W = x.shape
    gamma, beta = _c(gamma, dt), _c(beta, dt)
    mm = np.array(moving_mean, dtype=dt, order="C", copy=True)
    mv = np.array(moving_var, dtype=dt, order="C", copy=True)
    y, norm = np.empty_like(x), np.empty_like(x)
    sm, sv = np.zeros(Cc, dt), np.zeros(Cc, dt)
    getattr(lib(), "oracle_batchnorm_forward" + suf)(_p(x, ct), _p(y, ct), _p(norm, ct), _p(gamma, ct), _p(beta, ct), _p(mm, ct),
                                                     _p(mv, ct), _p(sm, ct), _p(sv, ct), B, Cc, H, W, ct(eps), ct(momentum),
                                                     1 if training else 0)
    return y, norm, sm, sv, mm, mv


def batchnorm_backward(x, dy, gamma, saved_mean, saved_var, eps=1e-5, f64=False):
    """-> (dx, ggamma, gbeta); the reference overwrites dy in place (batchnorm2d.cpp:149-155)"""
    dt, ct, suf = _dt(f64)
    x, gamma = _c(x, dt), _c(gamma, dt)
    d = np.array(dy, dtype=dt, order="C", copy=True)
    B, Cc, H, W = x.shape
    gg, gb = np.zeros(Cc, dt), np.zeros(Cc, dt)
    getattr(lib(), "oracle_batchnorm_backward" + suf)(_p(x, ct), _p(d, ct), _p(gamma, ct), _p(_c(saved_mean, dt), ct),
                                                      _p(_c(saved_var, dt), ct), _p(gg, ct), _p(gb, ct), B, Cc, H, W, ct(eps))
    return d, gg, gb


def sgd_update(p, g, lr, f64=False):
    dt, ct, suf = _dt(f64)
    p = np.array(p, dtype=dt, order="C", copy=True)
    g = _c(g, dt)
    getattr(lib(), "oracle_sgd_update" + suf)(_p(p, ct), _p(g, ct), p.size, ct(lr))
    return p


def softmax(logits, f64=False):
    dt, ct, suf = _dt(f64)
    logits = _c(logits, dt)
    B, n = logits.shape
    out = np.empty_like(logits)
    getattr(lib(), "oracle_softmax" + suf)(_p(logits, ct), _p(out, ct), B, n)
    return out


def cross_entropy_backward(probs, labels, f64=False):
    dt, ct, suf = _dt(f64)
    probs = _c(probs, dt)
    labels = _c(labels, np.int32)
    B, n = probs.shape
    delta = np.empty_like(probs)
    loss = getattr(lib(), "oracle_cross_entropy_backward" + suf)(_p(probs, ct), _p(labels, C.c_int), _p(delta, ct), B, n)
    return float(loss), delta


class Net:
    """The reference network (alexnet.cpp:10-33, batch_norm=false) on the oracle."""

    def __init__(self, batch, classes=3, H=224, W=224, f64=False):
        self.dt, self.ct, self.suf = _dt(f64)
        self.B, self.classes, self.H, self.W = batch, classes, H, W
        self._L = lib()
        self._h = C.c_void_p(self._f("oracle_net_create")(batch, classes, H, W))
        self.n_params = self._f("oracle_net_num_params")(self._h)
        self.lin_in = self._f("oracle_net_linear_in")(self._h)
        self.chans = [3, 16, 32, 64, 128]
        self.conv_in_hw, self.conv_out_hw = [], []
        h, w = H, W
        for l in range(4):
            self.conv_in_hw.append((h, w))
            h, w = conv_out_dim(h, 3, 2), conv_out_dim(w, 3, 2)
            self.conv_out_hw.append((h, w))
            if l == 0:
                h, w = (h - 2) // 2 + 1, (w - 2) // 2 + 1
        self.pool_out_hw = self.conv_in_hw[1]

    def _f(self, name):
        return getattr(self._L, name + self.suf)

    def __del__(self):
        try:
            self._f("oracle_net_destroy")(self._h)
        except Exception:
            pass

    def _view(self, ptr, shape):
        n = int(np.prod(shape))
        return np.ctypeslib.as_array(ptr, shape=(n,)).reshape(shape)

    @property
    def params(self):
        return self._view(self._f("oracle_net_params")(self._h), (self.n_params,))

    @property
    def grads(self):
        return self._view(self._f("oracle_net_grads")(self._h), (self.n_params,))

    def load_checkpoint(self, path):
        raw = np.fromfile(path, dtype=np.float32)
        assert raw.size == self.n_params, (raw.size, self.n_params)
        self.params[:] = raw.astype(self.dt)

    def forward(self, x):
        self._x = _c(x, self.dt)
        p = self._f("oracle_net_forward")(self._h, _p(self._x, self.ct))
        return self._view(p, (self.B, self.classes)).copy()

    def backward(self, delta):
        d = _c(delta, self.dt)
        self._f("oracle_net_backward")(self._h, _p(d, self.ct))

    def update(self, lr):
        self._f("oracle_net_update")(self._h, self.ct(lr))

    def train_step(self, x, labels, lr):
        self._x = _c(x, self.dt)
        lab = _c(labels, np.int32)
        probs = np.empty((self.B, self.classes), self.dt)
        loss = self._f("oracle_net_train_step")(self._h, _p(self._x, self.ct), _p(lab, C.c_int), self.ct(lr), _p(probs, self.ct))
        return float(loss), probs

    def conv_out(self, l):
        return self._view(self._f("oracle_net_conv_out")(self._h, l), (self.B, self.chans[l + 1]) + self.conv_out_hw[l]).copy()

    def relu_out(self, l):
        return self._view(self._f("oracle_net_relu_out")(self._h, l), (self.B, self.chans[l + 1]) + self.conv_out_hw[l]).copy()

    def pool_out(self):
        return self._view(self._f("oracle_net_pool_out")(self._h), (self.B, 16) + self.pool_out_hw).copy()

    def pool_mask(self):
        return self._view(self._f("oracle_net_pool_mask")(self._h), (self.B, 16) + self.pool_out_hw).copy()

    def d_conv(self, l):
        return self._view(self._f("oracle_net_d_conv")(self._h, l), (self.B, self.chans[l]) + self.conv_in_hw[l]).copy()


class BnNet:
    """AlexNet(batch_norm=true) (alexnet.cpp:10-33 with the BatchNorm2D layers of :13,17,20,23) composed from the
    oracle's layer functions; parameters are one flat vector in checkpoint order (conv w, conv b, then gamma, beta,
    moving_mean, moving_var of the following BN, ..., linear W, linear b)."""

    CH = [3, 16, 32, 64, 128]

    def __init__(self, classes=3, H=224, W=224):
        self.classes = classes
        self.slices, off = {}, 0

        def take(name, n):
            nonlocal off
            self.slices[name] = slice(off, off + n)
            off += n

        h, w = H, W
        for l in range(4):
            ci, co = self.CH[l], self.CH[l + 1]
            take(f"w{l}", co * ci * 9), take(f"b{l}", co)
            for nm in ("gamma", "beta", "mm", "mv"):
                take(f"{nm}{l}", co)
            h, w = conv_out_dim(h, 3, 2), conv_out_dim(w, 3, 2)
            if l == 0:
                h, w = (h - 2) // 2 + 1, (w - 2) // 2 + 1
        self.lin_in = 128 * h * w
        take("lw", self.lin_in * classes), take("lb", classes)
        self.n_params = off
        self.params = np.zeros(off, np.float32)
        self.grads = np.zeros(off, np.float32)
        for l in range(4):  # batchnorm2d.cpp:18-20
            self.params[self.slices[f"gamma{l}"]] = 1

    def p(self, name):
        return self.params[self.slices[name]]

    def forward(self, x, training=True):
        B = x.shape[0]
        cur = _c(x, np.float32)
        self.t = {"in": [], "conv": [], "relu": [], "sm": [], "sv": []}
        for l in range(4):
            ci, co = self.CH[l], self.CH[l + 1]
            self.t["in"].append(cur)
            c = conv2d_forward(cur, self.p(f"w{l}").reshape(co, ci, 3, 3), self.p(f"b{l}"), 2)
            y, _, sm, sv, mm, mv = batchnorm_forward(c, self.p(f"gamma{l}"), self.p(f"beta{l}"), self.p(f"mm{l}"), self.p(f"mv{l}"),
                                                     training=training)
            if training:
                self.p(f"mm{l}")[:] = mm
                self.p(f"mv{l}")[:] = mv
            r = relu_forward(y)
            self.t["conv"].append(c), self.t["relu"].append(r), self.t["sm"].append(sm), self.t["sv"].append(sv)
            cur = r
            if l == 0:
                cur, self.t["mask"] = maxpool_forward(r, 2, 2)
        self.t["flat"] = cur.reshape(B, -1)
        return linear_forward(self.t["flat"], self.p("lw").reshape(self.lin_in, self.classes), self.p("lb"))

    def train_step(self, x, labels, lr):
        logits = self.forward(x, training=True)
        probs = softmax(logits)
        loss, delta = cross_entropy_backward(probs, labels)
        g = self.grads
        g[:] = 0
        gw, gb, d = linear_backward(self.t["flat"], delta, self.p("lw").reshape(self.lin_in, self.classes))
        g[self.slices["lw"]], g[self.slices["lb"]] = gw.ravel(), gb
        d = d.reshape(self.t["relu"][3].shape)
        for l in (3, 2, 1, 0):
            ci, co = self.CH[l], self.CH[l + 1]
            if l == 0:
                d = maxpool_backward(d, self.t["mask"], self.t["relu"][0].shape, 2, 2)
            d = relu_backward(self.t["relu"][l], d)
            d, gg, gbeta = batchnorm_backward(self.t["conv"][l], d, self.p(f"gamma{l}"), self.t["sm"][l], self.t["sv"][l])
            g[self.slices[f"gamma{l}"]], g[self.slices[f"beta{l}"]] = gg, gbeta
            gw, gb, d = conv2d_backward(self.t["in"][l], d, self.p(f"w{l}").reshape(co, ci, 3, 3), 2)
            g[self.slices[f"w{l}"]], g[self.slices[f"b{l}"]] = gw.ravel(), gb
        self.params[:] = sgd_update(self.params, g, lr)  # the moving_* slots have zero gradient
        return loss, probs


class SeqNet:
    """Any strictly sequential list of the reference's layer types (the container of alexnet.cpp:35-65 for an arbitrary
    layers_sequence) composed from the oracle's layer functions; `spec` is a cnn_amd.stacks-style list of tuples
    ("conv", Co, k, s, pad) | ("bn",) | ("relu",) | ("pool", k, step) | ("linear", out).  Parameters / gradients are one flat
    vector in checkpoint order (alexnet.cpp:73-74: conv w then b, BN gamma beta moving_mean moving_var, linear W then b).
    Keeps every layer's forward output (self.acts[i]) and input delta (self.deltas[i]) of the last step for per-layer checks."""

    def __init__(self, spec, in_shape=(3, 224, 224), f64=False):
        self.spec = list(spec)
        self.f64 = f64
        self.dt = np.float64 if f64 else np.float32
        C_, H, W = in_shape
        self.layers, off = [], 0
        for item in self.spec:
            kind = item[0]
            ent = {"kind": kind, "in": (C_, H, W), "off": off}
            if kind == "conv":
                _, co, k, s, p = item
                ent.update(Co=co, k=k, s=s, pad=p, n=co * C_ * k * k + co)
                C_, H, W = co, (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
            elif kind == "bn":
                ent.update(n=4 * C_)
            elif kind == "relu":
                ent.update(n=0)
            elif kind == "dropout":
                ent.update(p=item[1], n=0)
            elif kind == "pool":
                _, k, st = item
                ent.update(k=k, step=st, n=0)
                H, W = (H - k) // st + 1, (W - k) // st + 1
            elif kind == "linear":
                ent.update(n_in=C_ * H * W, n_out=item[1], n=C_ * H * W * item[1] + item[1])
                C_, H, W = item[1], 1, 1
            else:
                raise ValueError(kind)
            ent["out"] = (C_, H, W)
            ent["params"] = ent["n"]  # (same key as cnn_amd.stacks.walk)
            off += ent["n"]
            self.layers.append(ent)
        self.n_params = off
        self.classes = C_
        self.params = np.zeros(off, self.dt)
        self.grads = np.zeros(off, self.dt)
        for e in self.layers:  # batchnorm2d.cpp:18-20: gamma = 1, everything else 0
            if e["kind"] == "bn":
                self.params[e["off"] : e["off"] + e["in"][0]] = 1

    def _p(self, e, a=None):
        a = self.params if a is None else a
        return a[e["off"] : e["off"] + e["n"]]

    def forward(self, x, training=True):
        f64 = self.f64
        cur = _c(x, self.dt)
        B = cur.shape[0]
        self.acts, self.inputs, self.aux = [], [], []
        for e in self.layers:
            self.inputs.append(cur)
            k = e["kind"]
            aux = None
            if k == "conv":
                ci, co, kk = e["in"][0], e["Co"], e["k"]
                p = self._p(e)
                cur = conv2d_forward_padded(cur, p[: co * ci * kk * kk].reshape(co, ci, kk, kk), p[co * ci * kk * kk :], e["s"], e["pad"], f64=f64)
            elif k == "bn":
                c = e["in"][0]
                p = self._p(e)
                y, _, sm, sv, mm, mv = batchnorm_forward(cur, p[:c], p[c : 2 * c], p[2 * c : 3 * c], p[3 * c :], training=training, f64=f64)
                if training:
                    p[2 * c : 3 * c], p[3 * c :] = mm, mv
                aux = (sm, sv)
                cur = y
            elif k == "relu":
                cur = relu_forward(cur, f64=f64)
            elif k == "dropout":
                cur = dropout_forward(cur, e["p"], training=training, f64=f64)
            elif k == "pool":
                cur, aux = maxpool_forward(cur, e["k"], e["step"], f64=f64)
            else:
                p = self._p(e)
                ni, no = e["n_in"], e["n_out"]
                cur = linear_forward(cur.reshape(B, ni), p[: ni * no].reshape(ni, no), p[ni * no :], f64=f64)
            self.acts.append(cur)
            self.aux.append(aux)
        return cur

    def backward(self, delta, masks_from=None):
        """masks_from (optional): {layer index: tensor} -- the ReLU OUTPUT (for a ReLU layer) or the pool INPUT (for a MaxPool2D
        layer) of ANOTHER implementation's forward pass on the same inputs.  ReLU::backward (relu.cpp:37) and
        MaxPool2D::backward (pool2d.cpp:105) then take their pass / route decisions from those tensors instead of this net's
        own: the two backward passes being compared make the same discrete decisions, so what remains is continuous in the
        inputs and can be held to a tight tolerance (a pre-activation within rounding distance of 0 otherwise resolves
        differently in any two fp32 implementations and moves whole gradients by O(1 / (B*H*W)))."""
        f64 = self.f64
        masks_from = masks_from or {}
        g = self.grads
        g[:] = 0
        d = _c(delta, self.dt)
        self.deltas = [None] * len(self.layers)
        for idx in range(len(self.layers) - 1, -1, -1):
            e, xin = self.layers[idx], self.inputs[idx]
            k = e["kind"]
            if k == "linear":
                p = self._p(e)
                ni, no = e["n_in"], e["n_out"]
                gw, gb, d = linear_backward(xin.reshape(xin.shape[0], ni), d, p[: ni * no].reshape(ni, no), f64=f64)
                self._p(e, g)[: ni * no], self._p(e, g)[ni * no :] = gw.ravel(), gb
                d = d.reshape(xin.shape)
            elif k == "dropout":
                d = dropout_backward(d, e["p"], f64=f64)
            elif k == "relu":
                d = relu_backward(masks_from.get(idx, self.acts[idx]), d, f64=f64)
            elif k == "pool":
                if idx in masks_from and masks_from[idx].dtype == np.int32:
                    mask = masks_from[idx]  # the other implementation's argmax mask itself
                elif idx in masks_from:
                    mask = maxpool_forward(masks_from[idx], e["k"], e["step"], f64=f64)[1]  # ... or its pool INPUT
                else:
                    mask = self.aux[idx]
                d = maxpool_backward(d, mask, xin.shape, e["k"], e["step"], f64=f64)
            elif k == "bn":
                c = e["in"][0]
                p = self._p(e)
                d, gg, gbeta = batchnorm_backward(xin, d, p[:c], self.aux[idx][0], self.aux[idx][1], f64=f64)
                self._p(e, g)[:c], self._p(e, g)[c : 2 * c] = gg, gbeta
            else:
                ci, co, kk = e["in"][0], e["Co"], e["k"]
                p = self._p(e)
                gw, gb, d = conv2d_backward_padded(xin, d, p[: co * ci * kk * kk].reshape(co, ci, kk, kk), e["s"], e["pad"], f64=f64)
                self._p(e, g)[: co * ci * kk * kk], self._p(e, g)[co * ci * kk * kk :] = gw.ravel(), gb
            self.deltas[idx] = d
        return d

    def train_step(self, x, labels, lr):
        """one iteration of cnn.cpp:79-90; -> (loss, probs)"""
        logits = self.forward(x, training=True)
        probs = softmax(logits, f64=self.f64)
        loss, delta = cross_entropy_backward(probs, labels, f64=self.f64)
        self.backward(delta)
        self.params[:] = sgd_update(self.params, self.grads, lr, f64=self.f64)  # BN moving_* slots have zero gradient
        return loss, probs
