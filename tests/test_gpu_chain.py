"""GPU parity of the sample-resident chain kernels (csrc/conv_chain.hip, round 4): [Conv2D + ReLU] x n -> LinearLayer -> loss head in ONE
kernel and the data gradients of the same convolutions in ONE kernel, through the C ABI -- against the CPU oracle on the same seeded
inputs (1e-4 tensor-normalised, north_star) AND bit for bit against the per-layer entry points they replace.
Reference: conv2d.cpp:69-92,168-199, relu.cpp:21-26,35-40, linear.cpp:33-43,73-90, func.cpp:16-73."""
import numpy as np
import pytest

from oracle import pyoracle as O
from tests.util import assert_close, normal_scaled, uniform01

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a device"
    return torch


def dev(T, a):
    return T.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def bits(t):
    return host(t).view(np.uint32)


# (n layers, B, H0, W0): the reference net's own chain (16x55x55 behind the pool) and its tails; odd / even sizes so that every border
# case of the stride-2 data gradient (uncovered last row / column, single-element last column) occurs; B not a multiple of anything
CHAINS = [
    (3, 3, 55, 55),
    (3, 2, 57, 54),
    (2, 5, 27, 27),
    (2, 3, 28, 30),
    (1, 7, 13, 13),
    (1, 4, 14, 11),
    (3, 33, 55, 55),
]


def _build(T, n, B, H0, W0, seed):
    from cnn_amd import capi

    convs, ws, bs = [], [], []
    H, W = H0, W0
    for l in range(n):
        ci = 128 >> (n - l)
        c = capi.Conv2d(B, ci, H, W, 2 * ci, 3, 2, 0)
        convs.append(c)
        ws.append(normal_scaled(seed + 10 * l, (2 * ci, ci, 3, 3)))
        bs.append(normal_scaled(seed + 10 * l + 1, (2 * ci,)))
        H, W = capi.conv_out_dim(H, 3, 2), capi.conv_out_dim(W, 3, 2)
    lin_in = 128 * H * W
    lw = normal_scaled(seed + 100, (lin_in, 3), 0.05)
    lb = normal_scaled(seed + 101, (3,))
    x = uniform01(seed + 102, (B, 128 >> n, H0, W0))
    labels = (np.arange(B) % 3).astype(np.int32)
    return convs, ws, bs, lw, lb, x, labels, lin_in


@pytest.mark.parametrize("chain", CHAINS, ids=lambda c: "n%d_B%d_%dx%d" % c)
def test_chain_kernels_vs_oracle_and_per_layer_calls(T, chain):
    from cnn_amd import capi

    n, B, H0, W0 = chain
    lib = capi.load()
    convs, ws, bs, lw, lb, x, labels, lin_in = _build(T, n, B, H0, W0, 4100 + 7 * n)
    if not capi.conv_chain_supported(convs, lin_in, 3):
        pytest.skip("chain not covered on this device / with these switches")
    xd, labd = dev(T, x), dev(T, labels)
    wd, bd = [dev(T, w) for w in ws], [dev(T, b) for b in bs]
    lwd, lbd = dev(T, lw), dev(T, lb)
    prep = [c.prepared_buffers() for c in convs]
    capi.prepare_filters(convs, wd, bd, [p[0] for p in prep], [p[1] for p in prep])
    mk = lambda *shape: T.full(shape, 7.0, device="cuda")
    shapes = [c.out_shape() for c in convs]

    # ---- the oracle's chain (fp32, the reference's loop nests) ----
    act, a = [], x
    for l in range(n):
        a = O.relu_forward(O.conv2d_forward(a, ws[l], bs[l], 2))
        act.append(a)
    logits_o = O.linear_forward(a.reshape(B, -1), lw, lb)
    probs_o = O.softmax(logits_o)
    _, delta_o = O.cross_entropy_backward(probs_o, labels)

    # ---- per-layer entry points ----
    a0, cur = [], xd
    for l in range(n):
        y = mk(*shapes[l])
        convs[l].forward_prepared(cur, prep[l][0], bd[l], None, y)
        a0.append(y)
        cur = y
    logits0, probs0, delta0, terms0, dxh0 = mk(B, 3), mk(B, 3), mk(B, 3), mk(B), mk(B, lin_in)
    capi.check(lib.cnn_linear_forward_softmax_xent_dx(capi._ptr(cur), capi._ptr(lwd), capi._ptr(lbd), capi._ptr(labd), capi._ptr(logits0),
                                                      capi._ptr(probs0), capi._ptr(delta0), capi._ptr(terms0), capi._ptr(dxh0), 1, B, lin_in, 3,
                                                      capi._stream()), "cnn_linear_forward_softmax_xent_dx")
    dx0, dy = [None] * n, dxh0
    for l in range(n - 1, -1, -1):
        inp = a0[l - 1] if l > 0 else xd
        dxl = mk(*inp.shape)
        if l > 0:
            convs[l].backward_data_relu(dy, None, a0[l - 1], dxl, prepared_dgrad=prep[l][1])
        else:
            convs[l].backward_data_prepared(dy, prep[l][1], dxl)
        dx0[l] = dxl
        dy = dxl

    # ---- the chain kernels ----
    a1 = [mk(*shapes[l]) for l in range(n)]
    logits1, probs1, delta1, terms1, dxh1 = mk(B, 3), mk(B, 3), mk(B, 3), mk(B), mk(B, lin_in)
    capi.conv_chain_forward_loss(convs, xd, [p[0] for p in prep], bd, a1, lwd, lbd, labd, logits1, probs1, delta1, terms1, dxh1)
    dx1 = [mk(*(a1[l - 1].shape if l > 0 else xd.shape)) for l in range(n)]
    capi.conv_chain_backward_data(convs, dxh1, [p[1] for p in prep], [a1[l - 1] if l > 0 else None for l in range(n)], dx1)
    T.cuda.synchronize()

    # bit for bit against the calls they replace
    pairs = [(f"relu {l}", a0[l], a1[l]) for l in range(n)] + [("logits", logits0, logits1), ("probs", probs0, probs1), ("delta", delta0, delta1),
                                                               ("loss terms", terms0, terms1), ("head dx", dxh0, dxh1)]
    pairs += [(f"dx {l}", dx0[l], dx1[l]) for l in range(n)]
    for name, p, q in pairs:
        assert np.array_equal(bits(p), bits(q)), name
    # against the oracle (north_star's tolerance): the forward chain end to end ...
    for l in range(n):
        assert_close(host(a1[l]), act[l], what=f"chain: relu(conv) of layer {l}")
    assert_close(host(logits1), logits_o, what="chain: logits")
    assert_close(host(probs1), probs_o, what="chain: probabilities")
    assert_close(host(delta1), delta_o, what="chain: loss delta")
    # ... and the data-gradient chain layer by layer: ReLU::backward (relu.cpp:37) is discontinuous in the forward activations, so an
    # activation within rounding distance of 0 resolves differently in any two fp32 implementations and re-routes a whole delta
    # element; the flipped decisions are counted, and the oracle's backward chain then runs with the masks and the
    # incoming delta the device produced -- every layer's arithmetic is still checked on its own
    flips = sum(int(np.count_nonzero((host(a1[l]) > 0) != (act[l] > 0))) for l in range(n))
    total = sum(act[l].size for l in range(n))
    assert flips <= max(2, int(2e-5 * total)), (flips, total)
    g_act = [host(t) for t in a1]
    _, _, dlin_g = O.linear_backward(g_act[-1].reshape(B, -1), host(delta1), lw)
    assert_close(host(dxh1).reshape(act[-1].shape), O.relu_backward(g_act[-1], dlin_g.reshape(act[-1].shape)), what="chain: d(linear input)")
    dy_g = host(dxh1).reshape(act[-1].shape)
    for l in range(n - 1, -1, -1):
        inp = g_act[l - 1] if l > 0 else x
        _, _, dxl = O.conv2d_backward(inp, dy_g, ws[l], 2, need=(False, False, True))
        if l > 0:
            dxl = O.relu_backward(g_act[l - 1], dxl)
        assert_close(host(dx1[l]), dxl, what=f"chain: data gradient of layer {l}")
        dy_g = host(dx1[l])


def test_chain_rejects_other_geometries(T):
    from cnn_amd import capi

    ok = [capi.Conv2d(4, 32, 27, 27, 64, 3, 2, 0), capi.Conv2d(4, 64, 13, 13, 128, 3, 2, 0)]
    assert capi.conv_chain_supported(ok, 128 * 36, 3) and capi.conv_chain_supported(ok)
    assert not capi.conv_chain_supported(ok, 128 * 36, 4)      # other head widths stay on the per-layer kernels
    assert not capi.conv_chain_supported(ok, 128 * 35, 3)
    assert not capi.conv_chain_supported([capi.Conv2d(4, 32, 27, 27, 64, 3, 2, 0), capi.Conv2d(4, 64, 12, 13, 128, 3, 2, 0)], 0, 0)  # not consecutive
    assert not capi.conv_chain_supported([capi.Conv2d(4, 32, 27, 27, 64, 3, 1, 0), capi.Conv2d(4, 64, 25, 25, 128, 3, 2, 0)], 0, 0)  # stride 1
    assert not capi.conv_chain_supported([capi.Conv2d(4, 16, 27, 27, 64, 3, 2, 0)], 0, 0)                                           # channels
    with capi.option("NO_CHAIN", "1"):
        assert not capi.conv_chain_supported(ok)


ALEXNET_OUTPUTS = [("conv_layer_1", (16, 111, 111)), ("relu_layer_1", (16, 111, 111)), ("max_pool_1", (16, 55, 55)),
                   ("conv_layer_2", (32, 27, 27)), ("relu_layer_2", (32, 27, 27)), ("conv_layer_3", (64, 13, 13)),
                   ("relu_layer_3", (64, 13, 13)), ("conv_layer_4", (128, 6, 6)), ("relu_layer_4", (128, 6, 6)), ("linear_1", (3, 1, 1))]


@pytest.fixture
def lib_option():
    from cnn_amd import capi

    saved = {}

    def set_(name, value):
        if name not in saved:
            saved[name] = capi.get_option(name)
        capi.set_option(name, value)

    yield set_
    for name, old in saved.items():
        capi.set_option(name, old)


def test_train_step_with_chain_kernels_is_bit_identical_to_the_per_layer_step(T, lib_option):
    """architectures::Sequential::train_step of the reference net with the chain kernels (forward chain / data-gradient chain over the
    last 3, 2, 1 convolutions, and each direction alone) against the per-layer step: after every one of four steps the loss, the
    parameters, the gradients, every layer's get_output() (alexnet.cpp:97,105) and the delta with respect to the input image, bit for
    bit -- the chain changes WHERE the kernel boundaries are, not one product or sum"""
    from cnn_amd import hostapi

    B = 5
    x = uniform01(4300, (B, 3, 224, 224))
    labels = (np.arange(B) % 3).astype(np.int32)
    p0 = normal_scaled(4301, (111267,))
    xd, ld = dev(T, x), dev(T, labels)
    lib_option("CHAIN_MIN_B", "1")

    def run(fwd_n, bwd_n):
        lib_option("CHAIN_FWD_N", str(fwd_n))
        lib_option("CHAIN_BWD_N", str(bwd_n))
        net = hostapi.HostAlexNet(3)
        net.set_params(p0)
        trace = []
        for step in range(4):
            net.train_step(xd, ld, 1e-3)
            outs = [net.layer_output(name, (B,) + shp) for name, shp in ALEXNET_OUTPUTS] if step in (1, 3) else None
            dx = net.input_delta((B, 3, 224, 224)) if step != 2 else None  # (step 2: the deferred kernel stays pending into step 3)
            trace.append((net.last_loss(), net.get_params(), net.get_grads(), outs, dx))
        net.close()
        return trace

    base = run(0, 0)
    for fwd_n, bwd_n in ((3, 3), (2, 2), (1, 1), (3, 0), (0, 3), (3, 2), (2, 3)):
        got = run(fwd_n, bwd_n)
        for step, (a, b) in enumerate(zip(base, got)):
            tag = f"chain fwd {fwd_n} / bwd {bwd_n}, step {step}"
            assert a[0] == b[0], (tag, a[0], b[0])
            assert np.array_equal(a[1], b[1]), tag + ": parameters"
            assert np.array_equal(a[2], b[2]), tag + ": gradients"
            if a[3] is not None:
                for (name, _), u, v in zip(ALEXNET_OUTPUTS, a[3], b[3]):
                    assert np.array_equal(u.view(np.uint32), v.view(np.uint32)), f"{tag}: get_output({name})"
            if a[4] is not None:
                assert np.array_equal(a[4].view(np.uint32), b[4].view(np.uint32)), tag + ": delta w.r.t. the input"


def test_full_batch_train_steps_with_chain_kernels_vs_oracle(T, lib_option):
    """BASELINE configs[1] at its own batch (256: one sample per compute unit, the chain kernels' design point) through the C++ classes
    with both chains switched on (they are opt-in: profiles/NOTEBOOK.md section 4.27), three steps, against the CPU oracle: the loss and the
    parameters after SGD within 1e-4"""
    from cnn_amd import hostapi
    from tests.util import REL_TOL

    lib_option("CHAIN_FWD_N", "3")
    lib_option("CHAIN_BWD_N", "3")
    B = 256
    x = uniform01(4400, (B, 3, 224, 224))
    labels = (np.arange(B) % 3).astype(np.int32)
    onet = O.Net(B, 3)
    p0 = normal_scaled(4401, (onet.n_params,))
    onet.params[:] = p0
    net = hostapi.HostAlexNet(3)
    net.set_params(p0)
    xd, ld = dev(T, x), dev(T, labels)
    for step in range(3):
        net.train_step(xd, ld, 1e-3)
        oloss, _ = onet.train_step(x, labels, 1e-3)
        loss = net.last_loss()
        assert abs(loss - oloss) <= 1e-4 * max(1.0, abs(oloss)), (step, loss, oloss)
        assert_close(net.get_params(), onet.params, REL_TOL, f"B=256 chain step {step}: parameters after SGD")
    assert hostapi.load().cnnh_net_chain_layers(net.h, 1) == 3 and hostapi.load().cnnh_net_chain_layers(net.h, 0) == 3  # the chains DID run
    net.close()
