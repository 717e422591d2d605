"""Backward of the oracle == derivative of its (README-pinned) forward, checked in fp64 by central
differences, and fp32 oracle == fp64 oracle within rounding.  This is the pin the backward path has."""
import numpy as np
import pytest

from oracle import pyoracle as O
from tests.util import rel_err, uniform_pm1


def _num_grad(f, x, eps=1e-6):
    g = np.zeros_like(x)
    it = np.nditer(x, flags=["multi_index"])
    for _ in it:
        i = it.multi_index
        old = x[i]
        x[i] = old + eps
        fp = f()
        x[i] = old - eps
        fm = f()
        x[i] = old
        g[i] = (fp - fm) / (2 * eps)
    return g


@pytest.mark.parametrize("k,s,H,W", [(3, 2, 9, 8), (3, 1, 6, 7), (5, 2, 11, 9), (1, 2, 5, 5)])
def test_conv_backward_is_gradient(k, s, H, W):
    B, Ci, Co = 2, 3, 4
    x = uniform_pm1(1, (B, Ci, H, W)).astype(np.float64)
    w = uniform_pm1(2, (Co, Ci, k, k)).astype(np.float64)
    b = uniform_pm1(3, (Co,)).astype(np.float64)
    y0 = O.conv2d_forward(x, w, b, s, f64=True)
    r = uniform_pm1(4, y0.shape).astype(np.float64)  # L = sum(r*y)/1 ; the layer divides gw,gb by B itself
    loss = lambda: float((O.conv2d_forward(x, w, b, s, f64=True) * r).sum())
    gw, gb, dx = O.conv2d_backward(x, r, w, s, f64=True)
    assert rel_err(gw * B, _num_grad(loss, w)) < 1e-7
    assert rel_err(gb * B, _num_grad(loss, b)) < 1e-7
    assert rel_err(dx, _num_grad(loss, x)) < 1e-7


def test_linear_backward_is_gradient():
    B, n_in, n_out = 3, 10, 4
    x = uniform_pm1(5, (B, n_in)).astype(np.float64)
    w = uniform_pm1(6, (n_in, n_out)).astype(np.float64)
    b = uniform_pm1(7, (n_out,)).astype(np.float64)
    r = uniform_pm1(8, (B, n_out)).astype(np.float64)
    loss = lambda: float((O.linear_forward(x, w, b, f64=True) * r).sum())
    gw, gb, dx = O.linear_backward(x, r, w, f64=True)
    assert rel_err(gw * B, _num_grad(loss, w)) < 1e-7
    assert rel_err(gb * B, _num_grad(loss, b)) < 1e-7
    assert rel_err(dx, _num_grad(loss, x)) < 1e-7


def test_pool_relu_backward_is_gradient():
    x = uniform_pm1(9, (2, 3, 7, 7)).astype(np.float64)
    y, m = O.maxpool_forward(x, 2, 2, f64=True)
    r = uniform_pm1(10, y.shape).astype(np.float64)
    loss = lambda: float((O.maxpool_forward(x, 2, 2, f64=True)[0] * r).sum())
    assert rel_err(O.maxpool_backward(r, m, x.shape, 2, 2, f64=True), _num_grad(loss, x, 1e-7)) < 1e-6
    loss = lambda: float((O.relu_forward(x, f64=True) * 1.5).sum())
    d = O.relu_backward(O.relu_forward(x, f64=True), np.full(x.shape, 1.5), f64=True)
    assert rel_err(d, _num_grad(loss, x, 1e-7)) < 1e-6


def test_whole_net_gradient_fp64_and_fp32_agree():
    """tiny whole-net step (64x64 input): analytic grads of the fp64 oracle vs finite differences on a
    sample of parameters, and the fp32 oracle against the fp64 one."""
    B, Hh = 2, 64
    x = ((uniform_pm1(11, (B, 3, Hh, Hh)) + 1) / 2).astype(np.float64)
    labels = np.array([0, 2], np.int32)
    net = O.Net(B, 3, Hh, Hh, f64=True)
    rs = np.random.RandomState(0)
    p0 = rs.standard_normal(net.n_params) * 0.1
    net.params[:] = p0

    def loss_at(p):
        net.params[:] = p
        probs = O.softmax(net.forward(x), f64=True)
        return O.cross_entropy_backward(probs, labels, f64=True)[0]

    probs = O.softmax(net.forward(x), f64=True)
    _, delta = O.cross_entropy_backward(probs, labels, f64=True)
    net.backward(delta)
    g = net.grads.copy()
    idx = rs.choice(net.n_params, 60, replace=False)
    num = np.zeros(60)
    for j, i in enumerate(idx):
        p = p0.copy(); p[i] += 1e-6; lp = loss_at(p)
        p[i] -= 2e-6; lm = loss_at(p)
        num[j] = (lp - lm) / 2e-6
    assert rel_err(g[idx], num) < 1e-5
    net32 = O.Net(B, 3, Hh, Hh)
    net32.params[:] = p0.astype(np.float32)
    probs32 = O.softmax(net32.forward(x.astype(np.float32)))
    _, d32 = O.cross_entropy_backward(probs32, labels)
    net32.backward(d32)
    net.params[:] = p0.astype(np.float32).astype(np.float64)
    probs = O.softmax(net.forward(x.astype(np.float32).astype(np.float64)), f64=True)
    _, delta = O.cross_entropy_backward(probs, labels, f64=True)
    net.backward(delta)
    assert rel_err(net32.grads, net.grads) < 1e-5


def test_batchnorm_backward_is_gradient():
    """BatchNorm2D: oracle backward == derivative of its own training-mode forward, in fp64"""
    B, C, H, W = 3, 4, 5, 6
    x = uniform_pm1(60, (B, C, H, W)).astype(np.float64)
    gamma = (uniform_pm1(61, (C,)) + 1.5).astype(np.float64)
    beta = uniform_pm1(62, (C,)).astype(np.float64)
    r = uniform_pm1(63, (B, C, H, W)).astype(np.float64)
    zeros = np.zeros(C)

    def loss():
        return float((O.batchnorm_forward(x, gamma, beta, zeros, zeros, f64=True)[0] * r).sum())

    y, norm, sm, sv, mm, mv = O.batchnorm_forward(x, gamma, beta, zeros, zeros, f64=True)
    dx, gg, gb = O.batchnorm_backward(x, r, gamma, sm, sv, f64=True)
    assert rel_err(dx, _num_grad(loss, x)) < 1e-6
    assert rel_err(gg, _num_grad(loss, gamma)) < 1e-7 and rel_err(gb, _num_grad(loss, beta)) < 1e-7
    # forward semantics: biased variance, moving stats start at 0 and move by momentum 0.1 (batchnorm2d.cpp:20,78-80)
    assert np.allclose(sm, x.mean(axis=(0, 2, 3))) and np.allclose(sv, x.var(axis=(0, 2, 3)))
    assert np.allclose(mm, 0.1 * sm) and np.allclose(mv, 0.1 * sv)
    assert np.allclose(y, gamma[None, :, None, None] * norm + beta[None, :, None, None])
    # eval mode uses the moving statistics (batchnorm2d.cpp:82-93)
    ye = O.batchnorm_forward(x, gamma, beta, mm, mv, training=False, f64=True)[0]
    ref = gamma[None, :, None, None] * (x - mm[None, :, None, None]) / np.sqrt(mv[None, :, None, None] + 1e-5) + beta[None, :, None, None]
    assert np.allclose(ye, ref, rtol=1e-6)
