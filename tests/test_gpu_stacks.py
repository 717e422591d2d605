"""GPU parity of the BASELINE configs[3] / [4] workloads: the VGG-11-shaped and ResNet-18-shaped layer lists (cnn_amd/stacks.py,
cnn_amd/host/src/sequential.cpp) run through the C++ Layer API (architectures::Sequential -> C ABI -> HIP kernels) against the same
lists composed from the oracle's layer functions (oracle.pyoracle.SeqNet), plus full-batch properties the oracle is too slow for."""
import numpy as np
import pytest

from cnn_amd import stacks as S
from oracle import pyoracle as O
from tests.util import (REL_TOL, assert_close, assert_close_arbitrated, assert_close_derived_bound, assert_noise_of_exact_zero, he_init, rel_err,
                        uniform01)

pytestmark = pytest.mark.gpu

# the softmax-Jacobian bound of the stacks' 3-element linear.b gradient (see below) grows without limit when that gradient itself is
# tiny (round 5 recorded a vacuous 80.4): whatever the derivation says, the check never goes above 10 x north_star's tolerance
DERIVED_BOUND_CAP = 1e-3


@pytest.fixture(scope="module")
def T():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a device"
    from cnn_amd import capi

    arch = capi.load().cnn_amd_device_arch().decode()
    assert arch == "gfx950", f"built for gfx950, running on {arch}"
    O.set_threads(0)  # (checker only: order-preserving thread split, bit-identical to 1 thread)
    yield torch
    O.set_threads(1)


def _slices(layout):
    """(name, lo, hi, layer index) of every parameter block in the flat checkpoint-order arena"""
    out, off, n = [], 0, 0
    for idx, e in enumerate(layout):
        if e["kind"] == "conv":
            n += 1
            nw = e["Co"] * e["in"][0] * e["k"] ** 2
            out += [(f"conv{n}.w", off, off + nw, idx), (f"conv{n}.b", off + nw, off + e["params"], idx)]
        elif e["kind"] == "bn":
            c = e["in"][0]
            out += [(f"bn{n}.gamma", off, off + c, idx), (f"bn{n}.beta", off + c, off + 2 * c, idx)]
        elif e["kind"] == "linear":
            nw = e["n_in"] * e["n_out"]
            out += [("linear.w", off, off + nw, idx), ("linear.b", off + nw, off + e["params"], idx)]
        off += e["params"]
    return out


@pytest.mark.parametrize("which", ["vgg11", "resnet18"])
def test_cpp_builders_match_the_python_layer_lists(T, which):
    """architectures::build_vgg11 / build_resnet18 (sequential.cpp) == cnn_amd/stacks.py (what the oracle is composed from)"""
    from cnn_amd import hostapi

    spec = S.STACKS[which]()
    layout = S.walk(spec)
    net = hostapi.HostStack(which, 3, batch_norm=(which == "resnet18"))
    want = list(zip(hostapi._layer_names(spec), [e["params"] for e in layout]))
    assert net.describe() == want
    assert net.n_params == sum(e["params"] for e in layout)
    net.close()


# (name, input resolution, batch): the full 224x224 geometry at the smallest batch the task names, plus a ragged low-resolution
# case whose spatial sizes are odd everywhere (border handling of every padded / strided layer)
STACK_CASES = [("vgg11", 224, 2), ("resnet18", 224, 2), ("vgg11", 75, 3), ("resnet18", 97, 3)]


@pytest.mark.parametrize("which,res,B", STACK_CASES, ids=lambda v: str(v))
def test_stack_train_steps_vs_oracle(T, which, res, B):
    """two full train steps (cnn.cpp:79-90) of the whole stack: logits, loss, every parameter gradient and the post-SGD
    parameters against the oracle; gradients fp64-arbitrated (tests/util.py)"""
    from cnn_amd import hostapi

    spec = S.STACKS[which]()
    in_shape = (3, res, res)
    onet = O.SeqNet(spec, in_shape)
    onet64 = O.SeqNet(spec, in_shape, f64=True)
    p0 = he_init(onet.layers, 40 + res)
    onet.params[:] = p0
    onet64.params[:] = p0
    net = hostapi.HostSequential(spec, in_shape)
    assert net.n_params == onet.n_params
    net.set_params(p0)
    x = uniform01(41 + res, (B,) + in_shape)
    labels = (np.arange(B) % 3).astype(np.int32)
    xd = T.from_numpy(x).cuda()
    lr = 1e-3
    names = hostapi._layer_names(spec)
    for step in range(2):
        loss = net.train_step_device(xd, labels, lr, do_update=False)
        g = net.get_grads()
        p_before = net.get_params()
        ologits = onet.forward(x)
        oloss, odelta = O.cross_entropy_backward(O.softmax(ologits), labels)
        l64 = onet64.forward(x)
        _, d64 = O.cross_entropy_backward(O.softmax(l64, f64=True), labels, f64=True)
        logits = net.layer_output("linear_1", (B, 3))
        assert_close_arbitrated(logits, ologits, l64, REL_TOL, 2.0, f"{which} step{step} logits")
        assert abs(loss - oloss) <= 1e-4 * max(1.0, abs(oloss)), (loss, oloss)
        # forward: every ReLU output and every pool output through Layer::get_output() (alexnet.cpp:97,105 contract)
        masks_from, flipped, total = {}, 0, 0
        for idx, e in enumerate(onet.layers):
            if e["kind"] not in ("relu", "pool"):
                continue
            got = net.layer_output(names[idx], (B,) + e["out"])
            assert_close_arbitrated(got, onet.acts[idx], onet64.acts[idx], REL_TOL, 2.0, f"{which} step{step} {names[idx]} output")
            if e["kind"] == "relu":
                masks_from[idx] = got
                flipped += int(np.count_nonzero((got <= 0) != (onet.acts[idx] <= 0)))
                total += got.size
                if idx + 1 < len(onet.layers) and onet.layers[idx + 1]["kind"] == "pool":
                    masks_from[idx + 1] = got  # the pool's input IS this ReLU's output
        # the discrete decisions (ReLU pass / block) agree except for pre-activations within rounding distance of zero
        assert flipped <= max(2, 2e-5 * total), (flipped, total)
        # backward: the oracle takes its ReLU' / MaxPool' decisions from the HIP forward tensors (SeqNet.backward), so both sides
        # differentiate the SAME piecewise-linear function; 1e-4 tensor-normalised, fp64-arbitrated (tests/util.py)
        onet.backward(odelta, masks_from=masks_from)
        onet64.backward(d64, masks_from=masks_from)
        for name, lo, hi, idx in _slices(onet.layers):
            tol = REL_TOL
            if name == "linear.b":
                # = mean_b(softmax(z_b) - onehot_b), a 3-element tensor: a logit error dz moves it by up to |dz| / 2 (the softmax
                # Jacobian diag(p) - p p^T has infinity-norm <= 1/2), and the logits themselves are only held to REL_TOL * max|z|
                # (above) -- the implicit-GEMM tiles the tuner pins differ from box to box, and with them the last bits of z
                tol = min(DERIVED_BOUND_CAP, max(REL_TOL, 0.5 * REL_TOL * float(np.abs(ologits).max()) / float(np.abs(onet.grads[lo:hi]).max())))
            if name.endswith(".b") and name.startswith("conv") and idx + 1 < len(onet.layers) and onet.layers[idx + 1]["kind"] == "bn":
                # exactly 0 in exact arithmetic (BatchNorm2D removes the channel mean behind this bias): both sides hold rounding
                # noise -- bounded against the layer's weight gradient instead of compared with each other (tests/util.py)
                wlo = lo - onet.layers[idx]["Co"] * onet.layers[idx]["in"][0] * onet.layers[idx]["k"] ** 2
                assert_noise_of_exact_zero(g[lo:hi], onet.grads[lo:hi], float(np.abs(onet.grads[wlo:lo]).max()), REL_TOL,
                                           f"{which} step{step} grad {name} (exact zero in front of BatchNorm2D)")
                continue
            if tol > REL_TOL:
                # the one check of the suite held to a DERIVED bound instead of north_star's 1e-4: recorded under its own branch name, so
                # that tests/test_zz_parity_margins.py and the session's summary list it instead of filtering it out
                assert_close_derived_bound(g[lo:hi], onet.grads[lo:hi], tol, f"{which} step{step} grad {name} (softmax-Jacobian bound)")
                continue
            assert_close_arbitrated(g[lo:hi], onet.grads[lo:hi], onet64.grads[lo:hi], tol, 2.0, f"{which} step{step} grad {name}")
        off = 0
        for e in onet.layers:  # BatchNorm2D moving statistics, updated by the forward pass (batchnorm2d.cpp:78-79)
            if e["kind"] == "bn":
                c = e["in"][0]
                sl = slice(off + 2 * c, off + 4 * c)
                assert_close_arbitrated(p_before[sl], onet.params[sl], onet64.params[sl], REL_TOL, 2.0, f"{which} step{step} moving statistics")
            off += e["params"]
        net.update(lr, 1.0)
        got = net.get_params()
        # the SGD step itself is bit-exact given the gradients (test_sgd_bit_exact_and_scaled): check it on the HIP gradients
        assert np.array_equal(got, O.sgd_update(p_before, g, lr)), f"{which} step{step}: p - lr*g is not bit-exact"
        # the second step starts from IDENTICAL parameters again (prepared filters, fused paths): without this the last-bit
        # differences of step 1's update would be amplified through 8-17 layers
        onet.params[:] = got
        onet64.params[:] = got
    net.close()


def _decisive_bn_params(spec, in_shape, x, p0, rel_margin=1e-2):
    """conv -> BatchNorm2D -> ReLU stacks: parameters for which NO ReLU decision can differ between two fp32 implementations -- every
    channel in front of a ReLU is >= margin everywhere (three of four channels) or <= -margin everywhere (every fourth), set through its
    beta on the fp64 oracle's own forward pass; margin = rel_margin x the layer's largest pre-activation.  (BatchNorm2D re-centres
    every layer, so the shifts do not accumulate through the stack -- without it they do, and the gradients become differences of huge
    terms: the VGG-shaped stack takes _exact_forward_params instead.)  The linear layer is rescaled so that the logits stay O(1) (a
    saturated softmax would make every gradient vanish).  -> (params fp32, {relu layer index: margin})"""
    net = O.SeqNet(spec, in_shape, f64=True)
    net.params[:] = p0
    cur = np.asarray(x, np.float64)
    margins = {}
    for idx, e in enumerate(net.layers):
        kind = e["kind"]
        nxt = net.layers[idx + 1]["kind"] if idx + 1 < len(net.layers) else None
        p = net._p(e)
        if kind == "conv":
            ci, co, kk = e["in"][0], e["Co"], e["k"]
            cur = O.conv2d_forward_padded(cur, p[: co * ci * kk * kk].reshape(co, ci, kk, kk), p[co * ci * kk * kk :], e["s"], e["pad"], f64=True)
            assert nxt == "bn"
        elif kind == "bn":
            c = e["in"][0]
            y = O.batchnorm_forward(cur, p[:c], p[c : 2 * c], p[2 * c : 3 * c], p[3 * c :], training=True, f64=True)[0]
            if nxt == "relu":
                lo, hi = y.min(axis=(0, 2, 3)), y.max(axis=(0, 2, 3))
                m = rel_margin * float(np.abs(y).max())
                sh = np.where(np.arange(c) % 4 == 3, -m - hi, m - lo)
                margins[idx + 1] = m
                p[c : 2 * c] += sh
                y = y + sh[None, :, None, None]
            cur = y
        elif kind == "relu":
            cur = O.relu_forward(cur, f64=True)
        elif kind == "pool":
            cur = O.maxpool_forward(cur, e["k"], e["step"], f64=True)[0]
        else:
            ni, no = e["n_in"], e["n_out"]
            z = O.linear_forward(cur.reshape(cur.shape[0], ni), p[: ni * no].reshape(ni, no), p[ni * no :], f64=True)
            p *= 2.0 / max(float(np.abs(z).max()), 1e-30)
    return net.params.astype(np.float32), margins


def _exact_forward_params(spec, in_shape, B, seed):
    """conv -> ReLU (-> MaxPool2D) stacks: an input and parameters for which the forward pass is EXACT in fp32 whatever the summation
    order -- inputs on a 2^-4 grid, four filter entries of +-1 per output channel (everything else 0), biases on the same grid: every
    product and partial sum is a multiple of 2^-4 below 4^layers = 2^16 (20 of fp32's 24 bits).  Two implementations then hold
    bit-identical activations, so every ReLU decision (zeros included) and every MaxPool2D argmax (ties included: first maximum wins,
    pool2d.cpp:67-75) is the same, with a natural mix of passing and blocked units.  The linear layer holds ordinary floats, scaled for
    O(1) logits.  -> (x, params fp32)"""
    rs = np.random.RandomState(seed)
    x = (rs.randint(0, 16, size=(B,) + tuple(in_shape)) / 16.0).astype(np.float32)
    net = O.SeqNet(spec, in_shape, f64=True)
    cur = x.astype(np.float64)
    for e in net.layers:
        p = net._p(e)
        if e["kind"] == "conv":
            ci, co, kk = e["in"][0], e["Co"], e["k"]
            w = np.zeros((co, ci * kk * kk))
            for o in range(co):
                w[o, rs.choice(ci * kk * kk, 4, replace=False)] = rs.choice([-1.0, 1.0], 4)
            p[: co * ci * kk * kk] = w.ravel()
            p[co * ci * kk * kk :] = rs.randint(-2, 3, size=co) / 16.0
            cur = O.conv2d_forward_padded(cur, w.reshape(co, ci, kk, kk), p[co * ci * kk * kk :], e["s"], e["pad"], f64=True)
            assert float(np.abs(cur).max()) * 16 < 2 ** 23, "forward no longer exact in fp32"
        elif e["kind"] == "relu":
            cur = O.relu_forward(cur, f64=True)
        elif e["kind"] == "pool":
            cur = O.maxpool_forward(cur, e["k"], e["step"], f64=True)[0]
        elif e["kind"] == "linear":
            ni, no = e["n_in"], e["n_out"]
            p[: ni * no] = rs.standard_normal(ni * no)
            p[ni * no :] = rs.standard_normal(no) * 0.1
            z = O.linear_forward(cur.reshape(B, ni), p[: ni * no].reshape(ni, no), p[ni * no :], f64=True)
            p *= 2.0 / max(float(np.abs(z).max()), 1e-30)
        else:
            raise ValueError(e["kind"])
    return x, net.params.astype(np.float32)


@pytest.mark.parametrize("which", ["vgg11", "resnet18"])
def test_stack_gradients_vs_the_oracles_own_backward(T, which):
    """the deep-stack gradients against a backward pass the oracle runs ON ITS OWN decisions (no masks_from),
    at north_star's plain 1e-4 (no fp64 arbitration), on inputs engineered so that no discrete decision can differ:
    VGG-shaped stack -- a forward pass that is exact in fp32 (_exact_forward_params): the test asserts BIT-IDENTICAL ReLU / pool outputs;
    ResNet-shaped stack -- every channel in front of a ReLU keeps a margin from 0 (_decisive_bn_params): the test asserts the margin on
    the oracle's pre-activations and identical ReLU decisions on both sides."""
    from cnn_amd import hostapi

    B, res = 2, 224
    spec = S.STACKS[which]()
    in_shape = (3, res, res)
    onet = O.SeqNet(spec, in_shape)
    exact = which == "vgg11"
    if exact:
        x, p0 = _exact_forward_params(spec, in_shape, B, 63)
        margins = {}
    else:
        x = uniform01(61, (B,) + in_shape)
        p0, margins = _decisive_bn_params(spec, in_shape, x, he_init(onet.layers, 62))
    onet.params[:] = p0
    net = hostapi.HostSequential(spec, in_shape)
    net.set_params(p0)
    labels = np.array([0, 2], np.int32)
    loss = net.train_step_device(T.from_numpy(x).cuda(), labels, 1e-3, do_update=False)
    g = net.get_grads()
    ologits = onet.forward(x)
    oloss, odelta = O.cross_entropy_backward(O.softmax(ologits), labels)
    assert abs(loss - oloss) <= 1e-4 * max(1.0, abs(oloss)), (loss, oloss)
    assert float(np.abs(odelta).max()) > 1e-2, "saturated softmax: the gradients would vanish"
    names = hostapi._layer_names(spec)
    for idx, e in enumerate(onet.layers):
        if e["kind"] not in ("relu", "pool"):
            continue
        got = net.layer_output(names[idx], (B,) + e["out"])
        if exact:
            assert np.array_equal(got, onet.acts[idx]), f"{names[idx]}: the engineered forward pass is not bit-identical"
        elif e["kind"] == "relu":
            pre = onet.acts[idx - 1]  # (the BatchNorm2D output in front)
            assert float(np.abs(pre).min()) >= 0.5 * margins[idx], (names[idx], float(np.abs(pre).min()), margins[idx])
            assert np.array_equal(got <= 0, onet.acts[idx] <= 0), names[idx]
            assert_close(got, onet.acts[idx], REL_TOL, f"{which} own-backward {names[idx]} output")
    if exact:
        frac = [float(np.mean(onet.acts[i] <= 0)) for i, e in enumerate(onet.layers) if e["kind"] == "relu"]
        assert 0.1 < min(frac) and max(frac) < 0.9, frac  # (a natural mix of passing and blocked units in every layer)
    onet.backward(odelta)  # its own ReLU' and MaxPool' decisions
    for name, lo, hi, idx in _slices(onet.layers):
        if name.endswith(".b") and name.startswith("conv") and idx + 1 < len(onet.layers) and onet.layers[idx + 1]["kind"] == "bn":
            wlo = lo - onet.layers[idx]["Co"] * onet.layers[idx]["in"][0] * onet.layers[idx]["k"] ** 2
            assert_noise_of_exact_zero(g[lo:hi], onet.grads[lo:hi], float(np.abs(onet.grads[wlo:lo]).max()), REL_TOL,
                                       f"{which} own-backward grad {name} (exact zero in front of BatchNorm2D)")
            continue
        if name == "linear.b":
            tol = min(DERIVED_BOUND_CAP, max(REL_TOL, 0.5 * REL_TOL * float(np.abs(ologits).max()) / float(np.abs(onet.grads[lo:hi]).max())))
            if tol > REL_TOL:
                assert_close_derived_bound(g[lo:hi], onet.grads[lo:hi], tol, f"{which} own-backward grad {name} (softmax-Jacobian bound)")
                continue
        if name.endswith(".beta") and not exact:
            # all-pass / all-block channels make one more gradient an exact zero: behind this ReLU sits a 1x1 convolution followed by a
            # BatchNorm2D, whose input delta sums to 0 over every channel plane (batchnorm2d.cpp:129-147) -- and a 1x1 filter passes
            # that property on to every input channel (no border taps), through a ReLU' that passes (or blocks) whole channels
            nxt = [j for j in range(idx + 1, len(onet.layers)) if onet.layers[j]["kind"] == "conv"][:1]
            if nxt and onet.layers[nxt[0]]["k"] == 1 and onet.layers[nxt[0] + 1]["kind"] == "bn":
                assert_noise_of_exact_zero(g[lo:hi], onet.grads[lo:hi], float(np.abs(onet.grads[lo - (hi - lo) : lo]).max()), REL_TOL,
                                           f"{which} own-backward grad {name} (exact zero in front of a 1x1 convolution + BatchNorm2D)")
                continue
        assert_close(g[lo:hi], onet.grads[lo:hi], REL_TOL, f"{which} own-backward grad {name}")
    net.close()


@pytest.mark.parametrize("which", ["vgg11", "resnet18"])
def test_stack_full_batch_fused_equals_unfused(T, which):
    """BASELINE batch (128 / 64 per GPU): the container with fuse_layers (fused Conv+ReLU, ReLU' in the data gradients, prepared
    filters) is bit-identical to one-kernel-per-layer-call, and finite"""
    from cnn_amd import hostapi

    B = S.DEFAULT_BATCH[which]
    spec = S.STACKS[which]()
    layout = S.walk(spec)
    p0 = he_init(layout, 50)
    x = T.rand((B, 3, 224, 224), generator=T.Generator(device="cuda").manual_seed(5), device="cuda")
    labels = (np.arange(B) % 3).astype(np.int32)
    res = []
    for fuse in (1, 0):
        hostapi.load().cnnh_set_fuse_layers(fuse)
        try:
            net = hostapi.HostSequential(spec)
            net.set_params(p0)
            losses = [net.train_step_device(x, labels, 1e-3) for _ in range(2)]  # 2nd step runs from prepared filters
            res.append((losses, net.get_grads(), net.get_params()))
            net.close()
        finally:
            hostapi.load().cnnh_set_fuse_layers(1)
    (l1, g1, p1), (l0, g0, p0_) = res
    assert np.all(np.isfinite(g1)) and np.all(np.isfinite(p1)) and np.all(np.isfinite(l1))
    assert l1 == l0
    assert np.array_equal(g1, g0) and np.array_equal(p1, p0_)


# every distinct convolution geometry of the two stacks at the BASELINE batch: MFMA path == im2col functional fallback
def _distinct_geoms():
    seen, out = set(), []
    for which in ("vgg11", "resnet18"):
        for g in S.conv_geometries(which):
            key = (S.DEFAULT_BATCH[which],) + g
            if key not in seen:
                seen.add(key)
                out.append(key)
    return out


@pytest.mark.parametrize("case", _distinct_geoms(), ids=lambda c: "B%d_%dx%dx%d_to%d_k%ds%dp%d" % c)
def test_stack_layers_full_batch_mfma_equals_im2col(T, case):
    from cnn_amd import capi

    B, Ci, H, W, Co, k, s, pad = case
    g = T.Generator(device="cuda").manual_seed(7)
    conv = capi.Conv2d(*case)
    x = T.rand((B, Ci, H, W), generator=g, device="cuda")
    w = T.randn((Co, Ci, k, k), generator=g, device="cuda") * float(np.sqrt(2.0 / (Ci * k * k)))
    b = T.randn((Co,), generator=g, device="cuda") * 0.1
    dy = T.rand(conv.out_shape(), generator=g, device="cuda") * 2 - 1

    def err(a, ref):  # tests/util.rel_err on the device (the tensors are up to 1.6 GB)
        return float((a - ref).abs().max() / ref.abs().max())

    y = conv.forward(x, w, b)
    yr = conv.forward_im2col(x, w, b)
    assert err(y, yr) <= REL_TOL
    del y, yr
    dx = conv.backward_data(dy, w)
    dxr = conv.backward_data_im2col(dy, w)
    assert err(dx, dxr) <= REL_TOL
    del dx, dxr
    gw, gb = conv.backward_weight(x, dy, float(B))
    gwr, gbr = conv.backward_weight_im2col(x, dy, float(B))
    assert err(gw, gwr) <= REL_TOL
    assert err(gb, gbr) <= REL_TOL


@pytest.mark.parametrize("cfg", [201, 227, 200, 202, 228, 229])
@pytest.mark.parametrize("case", [(16, 64, 60, 60, 128, 3, 1, 0), (12, 64, 61, 61, 64, 3, 1, 1), (16, 128, 59, 59, 64, 3, 1, 1), (9, 64, 62, 58, 128, 3, 1, 0)],
                         ids=lambda c: "B%d_%dx%dx%d_to%d_k%ds%dp%d" % c)
def test_dma_kernel_ragged_rows_equal_im2col(T, case, cfg, lib_option):
    """the double-buffered DMA kernel with 16-byte row staging on rows that are NOT whole 16-byte units (run_mode 3 with a ragged
    last unit: the north-star data gradient reads dy rows of 110 floats with pad 2): every tile family the tuner may pin, forced,
    forward (padded layers) and data gradient, against the im2col fallback -- and bit-identical to the per-row 4-byte staging
    (IGEMM_RAGGED_ROWS=0), which moves the same values in the same order.  Shapes include a tensor whose very last row ends the
    allocation (the one unit that must not be fetched 16 bytes wide)."""
    from cnn_amd import capi

    B, Ci, H, W, Co, k, s, pad = case
    g = T.Generator(device="cuda").manual_seed(9)
    x = T.rand((B, Ci, H, W), generator=g, device="cuda")
    w = T.randn((Co, Ci, k, k), generator=g, device="cuda") * float(np.sqrt(2.0 / (Ci * k * k)))
    b = T.randn((Co,), generator=g, device="cuda") * 0.1
    lib_option("IGEMM_CFG", str(cfg))
    lib_option("FWD_RD", "0")
    lib_option("DGRAD_RD", "0")
    conv = capi.Conv2d(*case)
    dy = T.rand(conv.out_shape(), generator=g, device="cuda") * 2 - 1

    def err(a, ref):
        return float((a - ref).abs().max() / ref.abs().max())

    res = {}
    for ragged in ("1", "0"):
        lib_option("IGEMM_RAGGED_ROWS", ragged)
        y = T.full(conv.out_shape(), 7.0, device="cuda")
        conv.forward(x, w, b, y)
        dx = T.full_like(x, 7.0)
        conv.backward_data(dy, w, dx)
        res[ragged] = (y, dx)
    assert err(res["1"][0], conv.forward_im2col(x, w, b)) <= REL_TOL
    assert err(res["1"][1], conv.backward_data_im2col(dy, w)) <= REL_TOL
    assert T.equal(res["1"][0], res["0"][0]) and T.equal(res["1"][1], res["0"][1])


@pytest.mark.parametrize("case,cfg", [((64, 128, 28, 28, 128, 3, 1, 1), 230), ((64, 128, 28, 28, 128, 3, 1, 1), 233), ((64, 256, 14, 14, 256, 3, 1, 1), 231),
                                      ((64, 256, 14, 14, 256, 3, 1, 1), 234), ((64, 512, 7, 7, 512, 3, 1, 1), 232), ((64, 512, 7, 7, 512, 3, 1, 1), 235)],
                         ids=lambda c: str(c).replace(" ", ""))
def test_split_k_wide_tiles_equal_im2col(T, case, cfg, lib_option):
    """the wide implicit-GEMM tiles with the channel range split over blockIdx.z (small planes at batch 64): partial tensors + split_reduce
    against the im2col fallback, forward and data gradient, and the epilogues split_reduce takes over from the unsplit kernel
    (ReLU output, ReLU-only output, ReLU' mask) bit-identical to the separate kernels on the same sums"""
    from cnn_amd import capi

    B, Ci, H, W, Co, k, s, pad = case
    g = T.Generator(device="cuda").manual_seed(11)
    x = T.rand((B, Ci, H, W), generator=g, device="cuda") - 0.3
    w = T.randn((Co, Ci, k, k), generator=g, device="cuda") * float(np.sqrt(2.0 / (Ci * k * k)))
    b = T.randn((Co,), generator=g, device="cuda") * 0.1
    lib_option("IGEMM_CFG", str(cfg))
    lib_option("DGRAD_RD", "0")
    lib_option("CONV_ROWS", "0")  # (this test is about the implicit GEMM's tiles; conv_rows.hip has test_conv2d_row_kernel_vs_oracle)
    conv = capi.Conv2d(*case)
    dy = T.rand(conv.out_shape(), generator=g, device="cuda") * 2 - 1

    def err(a, ref):
        return float((a - ref).abs().max() / ref.abs().max())

    capi.kernel_timing(1)
    y = conv.forward(x, w, b)
    dx = conv.backward_data(dy, w) if s == 1 else None
    names = [key.split("|")[0] for key in capi.kernel_timing_report()]
    capi.kernel_timing(0)
    assert any(",k/" in n and n.endswith("/fwd") for n in names) and any(n.startswith("split_reduce") for n in names), names
    assert err(y, conv.forward_im2col(x, w, b)) <= REL_TOL
    y_f, r_f = T.full_like(y, 7.0), T.full_like(y, 7.0)
    conv.forward_relu(x, w, b, y_f, r_f)
    assert T.equal(y_f, y) and T.equal(r_f, capi.relu_forward(y))
    pf, pd = conv.prepared_buffers("cuda")
    capi.prepare_filters([conv], [w], [b], [pf], [pd])
    r_only = T.full_like(y, 7.0)
    conv.forward_prepared(x, pf, b, None, r_only)
    assert T.equal(r_only, r_f)
    if dx is not None:
        assert any(",k/" in n and n.endswith("/dgrad") for n in names), names
        assert err(dx, conv.backward_data_im2col(dy, w)) <= REL_TOL
        relu_in = capi.relu_forward(x)  # the output of a ReLU layer in front: dx is masked where it is 0
        dxm = T.full_like(x, 7.0)
        conv.backward_data_relu(dy, w, relu_in, dxm)
        assert T.equal(dxm, T.where(relu_in <= 0, T.zeros_like(dx), dx))


@pytest.mark.parametrize("case,cfgs", [((64, 64, 56, 56, 64, 3, 1, 1), (228,)), ((64, 128, 28, 28, 128, 3, 1, 1), (229, 230)), ((64, 256, 14, 14, 256, 3, 1, 1), (231, 234)),
                                       ((64, 512, 7, 7, 512, 3, 1, 1), (232, 235)), ((128, 512, 14, 14, 512, 3, 1, 1), (228, 229))],
                         ids=lambda c: str(c).replace(" ", ""))
def test_wide_and_split_tiles_at_the_stacks_full_sizes(T, case, cfgs, lib_option):
    """BASELINE configs[3] / [4] at their stated batch sizes: the layers on which the tuner pins the wide / split-K tiles (profiles/NOTEBOOK.md 4.30), every such
    tile forced, forward and data gradient against the rule-based tile of the same kernel family (a different summation order: 1e-5
    tensor-normalised) -- the full-size launches (242 workgroups of 64 x 832, 32 tiles x 8 channel ranges ...) that the small-batch oracle tests
    above do not reach -- plus linearity of the data gradient in dy at full size"""
    from cnn_amd import capi

    B, Ci, H, W, Co, k, s, pad = case
    g = T.Generator(device="cuda").manual_seed(13)
    x = T.rand((B, Ci, H, W), generator=g, device="cuda") - 0.4
    w = T.randn((Co, Ci, k, k), generator=g, device="cuda") * float(np.sqrt(2.0 / (Ci * k * k)))
    b = T.randn((Co,), generator=g, device="cuda") * 0.1
    lib_option("DGRAD_RD", "0")
    lib_option("CONV_ROWS", "0")  # (the implicit GEMM's tiles)
    conv = capi.Conv2d(*case)
    dy = T.rand(conv.out_shape(), generator=g, device="cuda") * 2 - 1

    def err(a, ref):
        return float((a - ref).abs().max() / ref.abs().max())

    lib_option("IGEMM_AUTOTUNE", "0")
    y0, dx0 = conv.forward(x, w, b), conv.backward_data(dy, w)
    for cfg in cfgs:
        lib_option("IGEMM_CFG", str(cfg))
        capi.kernel_timing(1)
        y, dx = conv.forward(x, w, b), conv.backward_data(dy, w)
        names = [key.split("|")[0] for key in capi.kernel_timing_report()]
        capi.kernel_timing(0)
        assert any(n.startswith("igemm_dma_kernel<16,2,13,") for n in names), names
        assert err(y, y0) <= 1e-5 and err(dx, dx0) <= 1e-5, (cfg, err(y, y0), err(dx, dx0))
        dx2 = conv.backward_data(dy * 2.0, w)  # exact in floating point: every product and partial sum doubles
        assert T.equal(dx2, dx * 2.0)
        # ... and an ORACLE slice of the full-size launch: forward and data gradient are per-sample independent
        # (conv2d.cpp:69,175), so three samples of the batch -- first, middle, last: different tiles / channel-range partials -- are
        # compared with the oracle directly instead of transitively through the im2col fallback
        sel = [0, B // 2, B - 1]
        xs, dys = x[sel].cpu().numpy(), dy[sel].cpu().numpy()
        xp = np.pad(xs, ((0, 0), (0, 0), (pad, pad), (pad, pad)))
        wn = w.cpu().numpy()
        y_ref = O.conv2d_forward(xp, wn, b.cpu().numpy(), s)
        dx_ref = O.conv2d_backward(xp, dys, wn, s)[2][:, :, pad : pad + H, pad : pad + W]
        assert_close(y[sel].cpu().numpy(), y_ref, REL_TOL, f"cfg {cfg} full-size forward, oracle slice")
        assert_close(dx[sel].cpu().numpy(), dx_ref, REL_TOL, f"cfg {cfg} full-size data gradient, oracle slice")


@pytest.mark.parametrize("case", [(64, 64, 56, 56, 64, 3, 1, 1), (64, 128, 28, 28, 128, 3, 1, 1), (64, 256, 14, 14, 256, 3, 1, 1), (64, 512, 7, 7, 512, 3, 1, 1),
                                  (128, 512, 14, 14, 512, 3, 1, 1), (128, 128, 56, 56, 256, 3, 1, 1), (128, 64, 112, 112, 128, 3, 1, 1)],
                         ids=lambda c: str(c).replace(" ", ""))
def test_row_kernel_at_the_stacks_full_sizes_vs_oracle(T, case):
    """the DEFAULT dispatch at BASELINE configs[3] / [4]'s full batch: these layers run on conv_rows.hip, whose
    workgroups walk several (sample, row block) units each at these sizes (448 .. 1792 units on 256 CUs) -- forward, data gradient and
    the data gradient with the ReLU' epilogue (relu.cpp:37 fused, the form the stacks' steps launch) against an ORACLE slice: both
    passes are per-sample independent (conv2d.cpp:69,175), so five samples -- first, last, and three inside, i.e. first / middle /
    last units of different workgroups' walks -- are compared with the oracle directly"""
    from cnn_amd import capi

    B, Ci, H, W, Co, k, s, pad = case
    g = T.Generator(device="cuda").manual_seed(23)
    x = T.rand((B, Ci, H, W), generator=g, device="cuda") - 0.4
    w = T.randn((Co, Ci, k, k), generator=g, device="cuda") * float(np.sqrt(2.0 / (Ci * k * k)))
    b = T.randn((Co,), generator=g, device="cuda") * 0.1
    conv = capi.Conv2d(*case)
    dy = T.rand(conv.out_shape(), generator=g, device="cuda") * 2 - 1
    relu_in = capi.relu_forward(x)
    capi.kernel_timing(1)
    y = conv.forward(x, w, b)
    dx = conv.backward_data(dy, w)
    dxm = T.full_like(x, 7.0)
    conv.backward_data_relu(dy, w, relu_in, dxm)
    T.cuda.synchronize()
    names = [key.split("|")[0] for key in capi.kernel_timing_report()]
    capi.kernel_timing(0)
    assert sum(n.startswith("conv_rows<") for n in names) == 3, names
    sel = [0, 1, B // 2 - 1, B // 2, B - 1]
    xs, dys = x[sel].cpu().numpy(), dy[sel].cpu().numpy()
    xp = np.pad(xs, ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    wn = w.cpu().numpy()
    y_ref = O.conv2d_forward(xp, wn, b.cpu().numpy(), s)
    dx_ref = O.conv2d_backward(xp, dys, wn, s)[2][:, :, pad : pad + H, pad : pad + W]
    assert_close(y[sel].cpu().numpy(), y_ref, REL_TOL, "default dispatch, full-size forward, oracle slice")
    assert_close(dx[sel].cpu().numpy(), dx_ref, REL_TOL, "default dispatch, full-size data gradient, oracle slice")
    assert_close(dxm[sel].cpu().numpy(), np.where(xs <= 0, np.float32(0), dx_ref), REL_TOL, "default dispatch, full-size data gradient + ReLU', oracle slice")
    # the samples the slice does not reach: the ReLU' form is the plain form masked (same sums, bit for bit), and nothing was left unwritten
    assert T.equal(dxm, T.where(relu_in <= 0, T.zeros_like(dx), dx))


@pytest.mark.parametrize("case", [(64, 64, 56, 56, 64, 3, 1, 1), (64, 128, 28, 28, 128, 3, 1, 1), (64, 256, 14, 14, 256, 3, 1, 1), (64, 512, 7, 7, 512, 3, 1, 1),
                                  (128, 512, 14, 14, 512, 3, 1, 1)], ids=lambda c: str(c).replace(" ", ""))
def test_small_plane_wgrad_at_the_stacks_full_sizes_vs_oracle(T, case):
    """conv_wgrad_sp.hip at BASELINE configs[3] / [4]'s full batch (256 workgroups, 4 .. 256 pixel ranges): the weight gradient couples
    all samples (conv2d.cpp:148), so the oracle slice is taken through dy -- a delta that is zero except on three samples (first,
    middle, last: three different pixel ranges) makes the full-size launch equal to the oracle's gradient of those three samples x 3/B"""
    from cnn_amd import capi

    B, Ci, H, W, Co, k, s, pad = case
    g = T.Generator(device="cuda").manual_seed(19)
    x = T.rand((B, Ci, H, W), generator=g, device="cuda") - 0.3
    conv = capi.Conv2d(*case)
    sel = [0, B // 2, B - 1]
    dy = T.zeros(conv.out_shape(), device="cuda")
    dy[sel] = T.rand((3,) + tuple(conv.out_shape()[1:]), generator=g, device="cuda") * 2 - 1
    capi.kernel_timing(1)
    gw, gb = conv.backward_weight(x, dy, float(B))
    names = [key.split("|")[0] for key in capi.kernel_timing_report()]
    capi.kernel_timing(0)
    assert any(n.startswith("wgrad_sp<") for n in names), names
    xp = np.pad(x[sel].cpu().numpy(), ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    gw_ref, gb_ref, _ = O.conv2d_backward(xp, dy[sel].cpu().numpy(), np.zeros((Co, Ci, k, k), np.float32), s)
    assert_close(gw.cpu().numpy(), gw_ref * np.float32(3.0 / B), REL_TOL, "full-size weight gradient, oracle slice")
    assert_close(gb.cpu().numpy(), gb_ref * np.float32(3.0 / B), REL_TOL, "full-size bias gradient, oracle slice")
    # all samples live: the gradient is linear in dy (exact: every product and partial sum doubles)
    dyf = T.rand(conv.out_shape(), generator=g, device="cuda") * 2 - 1
    gw1, gb1 = conv.backward_weight(x, dyf, float(B))
    gw1, gb1 = gw1.clone(), gb1.clone()
    gw2, gb2 = conv.backward_weight(x, dyf * 2.0, float(B))
    assert T.equal(gw2, gw1 * 2.0) and T.equal(gb2, gb1 * 2.0)


@pytest.mark.parametrize("case", [(64, 64, 56, 56, 128, 3, 2, 1), (64, 256, 14, 14, 512, 3, 2, 1), (64, 128, 28, 28, 256, 3, 2, 1)], ids=lambda c: str(c).replace(" ", ""))
def test_stride2_stage_entries_at_full_size_vs_oracle(T, case):
    """the 3x3 / stride-2 stage entries of BASELINE configs[4] at its per-GPU batch, DEFAULT dispatch (round 6: conv_rows_s2.hip forward /
    data gradient where they are the faster kernels, conv_wgrad_sp2.hip for the weight gradient; the window walk of conv2d.cpp:76-77):
    forward, data gradient and data gradient + ReLU' against an oracle slice of five samples (per-sample independent passes,
    conv2d.cpp:69,175); the weight gradient -- it couples all samples (conv2d.cpp:148) -- through a delta that is zero except on three
    samples, which makes the full-size launch (64 .. 8 pixel ranges per tile) equal to the oracle's gradient of those samples x 3/B"""
    from cnn_amd import capi

    B, Ci, H, W, Co, k, s, pad = case
    g = T.Generator(device="cuda").manual_seed(29)
    x = T.rand((B, Ci, H, W), generator=g, device="cuda") - 0.4
    w = T.randn((Co, Ci, k, k), generator=g, device="cuda") * float(np.sqrt(2.0 / (Ci * k * k)))
    b = T.randn((Co,), generator=g, device="cuda") * 0.1
    conv = capi.Conv2d(*case)
    dy = T.rand(conv.out_shape(), generator=g, device="cuda") * 2 - 1
    relu_in = capi.relu_forward(x)
    capi.kernel_timing(1)
    y = conv.forward(x, w, b)
    dx = conv.backward_data(dy, w)
    dxm = T.full_like(x, 7.0)
    conv.backward_data_relu(dy, w, relu_in, dxm)
    T.cuda.synchronize()
    names = [key.split("|")[0] for key in capi.kernel_timing_report()]
    capi.kernel_timing(0)
    assert any(n.startswith("conv_s2<") and n.endswith("/fwd") for n in names), names
    sel = [0, 1, B // 2 - 1, B // 2, B - 1]
    xs, dys = x[sel].cpu().numpy(), dy[sel].cpu().numpy()
    xp = np.pad(xs, ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    wn = w.cpu().numpy()
    y_ref = O.conv2d_forward(xp, wn, b.cpu().numpy(), s)
    dx_ref = O.conv2d_backward(xp, dys, wn, s)[2][:, :, pad : pad + H, pad : pad + W]
    assert_close(y[sel].cpu().numpy(), y_ref, REL_TOL, "full-size stride-2 forward, oracle slice")
    assert_close(dx[sel].cpu().numpy(), dx_ref, REL_TOL, "full-size stride-2 data gradient, oracle slice")
    assert_close(dxm[sel].cpu().numpy(), np.where(xs <= 0, np.float32(0), dx_ref), REL_TOL, "full-size stride-2 data gradient + ReLU', oracle slice")
    assert T.equal(dxm, T.where(relu_in <= 0, T.zeros_like(dx), dx))
    # weight gradient
    sel3 = [0, B // 2, B - 1]
    dz = T.zeros(conv.out_shape(), device="cuda")
    dz[sel3] = dy[sel3]
    capi.kernel_timing(1)
    gw, gb = conv.backward_weight(x, dz, float(B))
    names = [key.split("|")[0] for key in capi.kernel_timing_report()]
    capi.kernel_timing(0)
    assert any(n.startswith("wgrad_sp2<") for n in names), names
    xp3 = np.pad(x[sel3].cpu().numpy(), ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    gw_ref, gb_ref, _ = O.conv2d_backward(xp3, dz[sel3].cpu().numpy(), np.zeros((Co, Ci, k, k), np.float32), s)
    assert_close(gw.cpu().numpy(), gw_ref * np.float32(3.0 / B), REL_TOL, "full-size stride-2 weight gradient, oracle slice")
    assert_close(gb.cpu().numpy(), gb_ref * np.float32(3.0 / B), REL_TOL, "full-size stride-2 bias gradient, oracle slice")
    gw1, gb1 = conv.backward_weight(x, dy, float(B))
    gw1, gb1 = gw1.clone(), gb1.clone()
    gw2, gb2 = conv.backward_weight(x, dy * 2.0, float(B))
    assert T.equal(gw2, gw1 * 2.0) and T.equal(gb2, gb1 * 2.0)


@pytest.mark.parametrize("case", [(7, 24, 14, 13, 40, 3, 1, 1), (3, 16, 9, 15, 32, 3, 1, 1), (5, 32, 14, 14, 64, 3, 1, 1)], ids=str)
def test_wgrad_flattened_runs_of_8_equal_im2col(T, case, lib_option):
    """the weight gradient's flattened runs of 8 for rows of 9 .. 15 pixels (CNN_AMD_RD_FLAT8=1, a measurement switch: profiles/NOTEBOOK.md 9) against the im2col
    fallback, the guarded head / tail chunks included"""
    from cnn_amd import capi

    B, Ci, H, W, Co, k, s, pad = case
    g = T.Generator(device="cuda").manual_seed(17)
    x = T.rand((B, Ci, H, W), generator=g, device="cuda") - 0.5
    lib_option("RD_FLAT8", "1")
    lib_option("WGRAD_SP", "0")  # (round 5: 14-wide planes with >= 32 channels go to conv_wgrad_sp.hip by default)
    conv = capi.Conv2d(*case)
    dy = T.rand(conv.out_shape(), generator=g, device="cuda") * 2 - 1
    capi.kernel_timing(1)
    gw, gb = conv.backward_weight(x, dy, float(B))
    names = [key.split("|")[0] for key in capi.kernel_timing_report()]
    capi.kernel_timing(0)
    assert any(n.endswith(",8,p1,flat>") for n in names), names
    gwr, gbr = conv.backward_weight_im2col(x, dy, float(B))
    assert float((gw - gwr).abs().max() / gwr.abs().max()) <= REL_TOL and float((gb - gbr).abs().max() / gbr.abs().max()) <= REL_TOL


def test_dropout_layer_and_a_list_that_uses_it(T):
    """Dropout (dropout.cpp; row n4): the two kernels bit-exact against the oracle, and a layer list that contains the layer --
    the position the reference's own (commented-out) line alexnet.cpp:28 puts it: behind a convolution -- through the C++
    container, train step and no_grad forward"""
    from cnn_amd import capi, hostapi

    x = uniform01(120, (3, 10, 7, 5)) * 2 - 1
    xd = T.from_numpy(x).cuda()
    for p in (0.4, 0.5, 0.05):
        assert np.array_equal(capi.dropout_forward(xd, p, True).cpu().numpy(), O.dropout_forward(x, p, True))
        assert np.array_equal(capi.dropout_forward(xd, p, False).cpu().numpy(), O.dropout_forward(x, p, False))
        assert np.array_equal(capi.dropout_backward(xd.clone(), p).cpu().numpy(), O.dropout_backward(x, p))
    spec = [("conv", 8, 3, 2, 0), ("dropout", 0.4), ("relu",), ("pool", 2, 2), ("conv", 12, 3, 1, 1), ("relu",), ("linear", 3)]
    in_shape = (3, 31, 29)
    onet = O.SeqNet(spec, in_shape)
    p0 = he_init(onet.layers, 121)
    onet.params[:] = p0
    net = hostapi.HostSequential(spec, in_shape)
    net.set_params(p0)
    B = 4
    xb = uniform01(122, (B,) + in_shape)
    labels = (np.arange(B) % 3).astype(np.int32)
    loss = net.train_step_device(T.from_numpy(xb).cuda(), labels, 1e-2)
    oloss, _ = onet.train_step(xb, labels, 1e-2)
    assert abs(loss - oloss) <= 1e-5 * max(1.0, abs(oloss))
    assert_close(net.get_params(), onet.params, REL_TOL, "params after a step through Dropout")
    hostapi.load().cnnh_set_no_grad(1)
    try:
        logits = net.forward_host(xb)
    finally:
        hostapi.load().cnnh_set_no_grad(0)
    assert_close(logits, onet.forward(xb, training=False), REL_TOL, "no_grad forward (x * (1 - p))")
    net.close()


PARTIAL_SPECS = [
    [("conv", 32, 5, 1, 0), ("relu",), ("conv", 24, 5, 2, 2), ("relu",), ("linear", 3)],
    [("conv", 24, 5, 1, 2), ("bn",), ("relu",), ("pool", 2, 2), ("conv", 24, 1, 1, 0), ("bn",), ("relu",), ("pool", 2, 2), ("linear", 3)],
]


@pytest.mark.parametrize("spec", PARTIAL_SPECS + ["alexnet", "alexnet_bn"], ids=["plain", "bn_pool", "alexnet", "alexnet_bn"])
def test_a_smaller_batch_after_full_ones_is_processed_as_what_it_is(T, spec):
    """three full train steps, then one with a batch SMALLER than the first call's (the layers' buffers are sized by that one: conv2d.cpp:47-52):
    loss and every gradient of the small step against the oracle run on the same sequence.  Found by tests/sweeps/fuzz_nets.py late in round 6:
    forward() / backward() handed on the whole buffer's views (like the reference's `return this->output`, conv2d.cpp:93 / :201), so every layer
    behind the first one also walked the stale samples behind the batch -- convolution gradients off by 60 - 90 %, BatchNorm2D statistics over a
    stale sample.  (The reference itself cannot run such a step: its loss glue indexes the labels by the returned vector.)"""
    from cnn_amd import hostapi

    in_shape, B = (3, 31, 29), 3
    if isinstance(spec, str):  # the reference net itself (alexnet.cpp:10-33): pool-fused first block, fused step tail, deferred input gradient
        spec, in_shape, B = S.alexnet(3, batch_norm=spec.endswith("_bn")), (3, 224, 224), 4
    onet = O.SeqNet(spec, in_shape)
    p0 = he_init(onet.layers, 7)
    onet.params[:] = p0
    x = uniform01(611, (B,) + in_shape)
    labels = (np.arange(B) % 3).astype(np.int32)
    xd, ld = T.from_numpy(x).cuda(), T.from_numpy(labels).cuda()
    net = hostapi.HostSequential(spec, in_shape)
    net.set_params(p0)
    for _ in range(3):
        onet.train_step(x, labels, 1e-3)
        net.train_step(xd, ld, 1e-3)
    oloss, _ = onet.train_step(x[: B - 1], labels[: B - 1], 1e-3)
    net.train_step(xd[: B - 1], ld[: B - 1], 1e-3)
    assert abs(net.last_loss() - oloss) <= 1e-4 * max(1.0, abs(oloss)), (net.last_loss(), oloss)
    g = net.get_grads()
    for e in onet.layers:
        if e.get("n", 0) > 0:
            sl = slice(e["off"], e["off"] + e["n"])
            assert_close(g[sl], onet.grads[sl], REL_TOL, f"small batch after full ones: {e['kind']} gradients")
    assert_close(net.get_params(), onet.params, REL_TOL, "small batch after full ones: parameters")
    net.close()
