"""GPU parity of BatchNorm2D (cpu/src/batchnorm2d.cpp) against the CPU oracle, through the C ABI.

Floating-point bar: tensor-normalised 1e-4 (tests/util.py) for activations, gradients and statistics.
"""
import numpy as np
import pytest

from oracle import pyoracle as O
from tests.util import REL_TOL, assert_close, rel_err, uniform_pm1

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a device"
    from cnn_amd import capi

    assert capi.load().cnn_amd_device_arch().decode() == "gfx950"
    return torch


def dev(T, a):
    return T.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


# every BN site of AlexNet(batch_norm=true) (alexnet.cpp:13,17,20,23) at a small batch + ragged / tiny / long planes
BN_SHAPES = [(2, 16, 111, 111), (3, 32, 27, 27), (4, 64, 13, 13), (5, 128, 6, 6), (1, 1, 1, 1), (2, 3, 1, 5),
             (1, 2, 70, 71), (7, 5, 3, 3), (2, 4, 64, 64), (64, 32, 16, 16), (65, 32, 16, 16)]  # (last two: a channel = / > the LDS limit
             # of the one-workgroup-per-channel kernels)


@pytest.mark.parametrize("shape", BN_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_batchnorm_train_forward_backward_vs_oracle(T, shape, lib_option):
    from cnn_amd import capi

    B, C, H, W = shape
    x = (uniform_pm1(70, shape) * 2 + 0.5).astype(np.float32)
    gamma = (uniform_pm1(71, (C,)) + 1.5).astype(np.float32)
    beta = uniform_pm1(72, (C,)).astype(np.float32)
    mm0 = uniform_pm1(73, (C,)).astype(np.float32)
    mv0 = (uniform_pm1(74, (C,)) + 1.5).astype(np.float32)
    dy = uniform_pm1(75, shape).astype(np.float32)

    y_o, norm_o, sm_o, sv_o, mm_o, mv_o = O.batchnorm_forward(x, gamma, beta, mm0, mv0)
    dx_o, gg_o, gb_o = O.batchnorm_backward(x, dy, gamma, sm_o, sv_o)

    bn = capi.BatchNorm2d(B, C, H, W)
    xd, gd, bd, mmd, mvd = dev(T, x), dev(T, gamma), dev(T, beta), dev(T, mm0), dev(T, mv0)
    yd = T.empty_like(xd)
    bn.forward(xd, gd, bd, mmd, mvd, yd, training=True)
    assert_close(host(yd), y_o, what="y")
    assert_close(host(bn.saved_mean), sm_o, what="batch mean")
    assert_close(host(bn.saved_var), sv_o, what="batch var")
    assert_close(host(mmd), mm_o, what="moving mean")
    assert_close(host(mvd), mv_o, what="moving var")

    dyd = dev(T, dy)
    ggd, gbd = T.full((C,), 7.0, device="cuda"), T.full((C,), 7.0, device="cuda")  # overwritten, not accumulated (:101-102)
    bn.backward(xd, dyd, gd, ggd, gbd)
    if B * H * W > 1:
        assert_close(host(dyd), dx_o, what="dx (in place)")
    else:  # one element per channel: dx is pure cancellation noise around 0 in both implementations
        assert np.abs(host(dyd)).max() <= 1e-3 and np.abs(dx_o).max() <= 1e-3
    assert_close(host(ggd), gg_o, what="gamma grad")
    assert_close(host(gbd), gb_o, what="beta grad")

    # determinism: the same call again gives the same bits
    y2, dy2 = T.empty_like(xd), dev(T, dy)
    mm2, mv2 = dev(T, mm0), dev(T, mv0)
    bn.forward(xd, gd, bd, mm2, mv2, y2, training=True)
    g2, b2 = T.empty_like(ggd), T.empty_like(gbd)
    bn.backward(xd, dy2, gd, g2, b2)
    assert T.equal(y2, yd) and T.equal(dy2, dyd) and T.equal(g2, ggd) and T.equal(b2, gbd) and T.equal(mm2, mmd)
    # (round 6) the one-workgroup-per-channel backward keeps its channel in registers; the LDS-resident form it replaces (same element ->
    # thread map, same summation order) gives the same bits
    lib_option("BN_BWD_LDS", "1")
    dy3, g3, b3 = dev(T, dy), T.empty_like(ggd), T.empty_like(gbd)
    bn.backward(xd, dy3, gd, g3, b3)
    assert T.equal(dy3, dyd) and T.equal(g3, ggd) and T.equal(b3, gbd)
    lib_option("BN_BWD_LDS", None)

    # BatchNorm2D -> ReLU from one pass (cnn_batchnorm2d_forward_relu): y unchanged bit for bit, y_relu = relu(y) (relu.cpp:25),
    # in training and in evaluation
    for training in (True, False):
        y3, r3, yref = T.empty_like(xd), T.empty_like(xd), T.empty_like(xd)
        mm3, mv3, mm4, mv4 = dev(T, mm0), dev(T, mv0), dev(T, mm0), dev(T, mv0)
        bn.forward(xd, gd, bd, mm4, mv4, yref, training=training)
        bn.forward(xd, gd, bd, mm3, mv3, y3, training=training, y_relu=r3)
        assert T.equal(y3, yref) and T.equal(mm3, mm4) and T.equal(mv3, mv4)
        assert T.equal(r3, T.where(y3 >= 0, y3, T.zeros_like(y3)))


@pytest.mark.parametrize("shape", [(4, 8, 57, 57), (8, 32, 14, 14)], ids=lambda s: "x".join(map(str, s)))
def test_batchnorm_one_pass_statistics_far_from_zero_and_in_place(T, shape):
    """the one-pass statistics (bn_stats_pilot: sums around a member of the channel) on a channel whose mean is 1000 sigma away from 0 --
    where E[x^2] - E[x]^2 would lose every digit -- and with y ALIASING x (the API allows it: the pilot value is published by the
    statistics pass, not re-read from a tensor the apply pass is overwriting).  At this offset the fp32 ORACLE is the less
    accurate side (its sequential sum of 13 000 values near 1000 drifts by ~1e-3 of sigma: batchnorm2d.cpp:46-52), so the bar is the
    fp64 restatement: HIP within 1e-4 of it, and no further from it than the fp32 oracle is"""
    from cnn_amd import capi

    B, C, H, W = shape
    rs = np.random.RandomState(77)
    x = (rs.standard_normal(shape) + 1000.0 * (1 + np.arange(C).reshape(1, C, 1, 1) % 3)).astype(np.float32)
    gamma = (uniform_pm1(78, (C,)) + 1.5).astype(np.float32)
    beta = uniform_pm1(79, (C,)).astype(np.float32)
    z = np.zeros(C, np.float32)
    y32, _, sm32, sv32, _, _ = O.batchnorm_forward(x, gamma, beta, z, z)
    y64, _, sm64, sv64, _, _ = O.batchnorm_forward(x, gamma, beta, z, z, f64=True)
    bn = capi.BatchNorm2d(B, C, H, W)
    for in_place in (False, True):
        xd = dev(T, x)
        yd = xd if in_place else T.empty_like(xd)
        bn.forward(xd, dev(T, gamma), dev(T, beta), dev(T, z), dev(T, z), yd, training=True)
        for got, r32, r64, what in ((host(yd), y32, y64, "y"), (host(bn.saved_var), sv32, sv64, "batch var"), (host(bn.saved_mean), sm32, sm64, "batch mean")):
            e_hip, e_ora = rel_err(got, r64), rel_err(r32, r64)
            assert e_hip <= REL_TOL and e_hip <= max(2 * e_ora, 1e-6), (what, in_place, e_hip, e_ora)


@pytest.mark.parametrize("shape", BN_SHAPES[:4] + BN_SHAPES[5:7], ids=lambda s: "x".join(map(str, s)))
def test_batchnorm_eval_uses_moving_statistics(T, shape):
    from cnn_amd import capi

    B, C, H, W = shape
    x = uniform_pm1(80, shape).astype(np.float32)
    gamma = (uniform_pm1(81, (C,)) + 1.5).astype(np.float32)
    beta = uniform_pm1(82, (C,)).astype(np.float32)
    mm = uniform_pm1(83, (C,)).astype(np.float32)
    mv = (uniform_pm1(84, (C,)) + 1.5).astype(np.float32)
    y_o = O.batchnorm_forward(x, gamma, beta, mm, mv, training=False)[0]
    bn = capi.BatchNorm2d(B, C, H, W)
    mmd, mvd = dev(T, mm), dev(T, mv)
    yd = T.empty(shape, device="cuda")
    bn.forward(dev(T, x), dev(T, gamma), dev(T, beta), mmd, mvd, yd, training=False)
    # same arithmetic, no reduction: bit-exact with the oracle, and the moving statistics are untouched (:82-93)
    assert np.array_equal(host(yd), y_o)
    assert np.array_equal(host(mmd), mm) and np.array_equal(host(mvd), mv)


def test_batchnorm_full_size_properties(T):
    """BN after conv_layer_1 at the benchmark batch (256 x 16 x 111 x 111): normalised output has per-channel mean
    beta and variance gamma^2 * var/(var+eps); backward's dx sums to ~0 per channel and is orthogonal to norm."""
    from cnn_amd import capi

    B, C, H, W = 256, 16, 111, 111
    g = T.Generator(device="cuda").manual_seed(5)
    x = T.randn((B, C, H, W), device="cuda", generator=g) * 3 + 1
    gamma = T.rand(C, device="cuda", generator=g) + 0.5
    beta = T.rand(C, device="cuda", generator=g)
    mm, mv = T.zeros(C, device="cuda"), T.zeros(C, device="cuda")
    bn = capi.BatchNorm2d(B, C, H, W)
    y = T.empty_like(x)
    bn.forward(x, gamma, beta, mm, mv, y, training=True)
    xd = x.double()
    mean_ref, var_ref = xd.mean(dim=(0, 2, 3)), xd.var(dim=(0, 2, 3), unbiased=False)
    assert rel_err(host(bn.saved_mean), host(mean_ref)) < REL_TOL and rel_err(host(bn.saved_var), host(var_ref)) < REL_TOL
    assert rel_err(host(mm), 0.1 * host(mean_ref)) < REL_TOL and rel_err(host(mv), 0.1 * host(var_ref)) < REL_TOL
    yd = y.double()
    assert (yd.mean(dim=(0, 2, 3)) - beta.double()).abs().max().item() < 1e-4
    assert rel_err(host(yd.var(dim=(0, 2, 3), unbiased=False)), host(gamma.double() ** 2 * var_ref / (var_ref + 1e-5))) < REL_TOL
    dy = T.randn((B, C, H, W), device="cuda", generator=g)
    dy0 = dy.clone()
    gg, gb = T.empty(C, device="cuda"), T.empty(C, device="cuda")
    bn.backward(x, dy, gamma, gg, gb)
    norm = (xd - mean_ref[None, :, None, None]) / T.sqrt(var_ref + 1e-5)[None, :, None, None]
    assert rel_err(host(gb), host(dy0.double().sum(dim=(0, 2, 3)))) < REL_TOL
    assert rel_err(host(gg), host((dy0.double() * norm).sum(dim=(0, 2, 3)))) < REL_TOL
    # closed form of batchnorm2d.cpp:118-155 in fp64
    L = B * H * W
    dn = dy0.double() * gamma.double()[None, :, None, None]
    ref = (dn - dn.mean(dim=(0, 2, 3), keepdim=True) - norm * (dn * norm).sum(dim=(0, 2, 3), keepdim=True) / L) / T.sqrt(
        var_ref + 1e-5)[None, :, None, None]
    assert rel_err(host(dy), host(ref)) < REL_TOL


def test_batchnorm_rejects_bad_arguments(T):
    from cnn_amd import capi

    L = capi.load()
    x = T.zeros((2, 3, 4, 4), device="cuda")
    v = T.zeros(3, device="cuda")
    rc = L.cnn_batchnorm2d_forward(capi._ptr(x), capi._ptr(x), capi._ptr(v), capi._ptr(v), capi._ptr(v), capi._ptr(v),
                                   None, None, 2, 3, 4, 4, 1e-5, 0.1, 1, None, 0, None)
    assert rc != 0 and b"saved_mean" in L.cnn_amd_last_error()
    rc = L.cnn_batchnorm2d_forward(capi._ptr(x), capi._ptr(x), capi._ptr(v), capi._ptr(v), capi._ptr(v), capi._ptr(v),
                                   capi._ptr(v), capi._ptr(v), 2, 3, 4, 4, 1e-5, 0.1, 1, None, 0, None)
    assert rc != 0 and b"workspace" in L.cnn_amd_last_error()


@pytest.mark.parametrize("shape,splits", [((6, 16, 27, 27), (2, 4)), ((5, 3, 7, 9), (1, 3, 1)), ((4, 16, 13, 13), (4,))],
                         ids=["two_ranks", "three_uneven_ranks", "one_rank"])
def test_sync_batchnorm_sharded_batch_equals_full_batch_oracle(T, shape, splits, lib_option):
    """the split-phase (sync-BN) entry points with the batch sharded over simulated ranks -- per-rank partial sums, summed
    like an all-reduce -- reproduce the reference's full-batch BatchNorm2D forward and backward (batchnorm2d.cpp:24-158)"""
    from cnn_amd import capi

    B, C, H, W = shape
    assert sum(splits) == B
    x = (uniform_pm1(90, shape) * 2 + 0.25).astype(np.float32)
    gamma = (uniform_pm1(91, (C,)) + 1.5).astype(np.float32)
    beta = uniform_pm1(92, (C,)).astype(np.float32)
    mm0 = uniform_pm1(93, (C,)).astype(np.float32)
    mv0 = (uniform_pm1(94, (C,)) + 1.5).astype(np.float32)
    dy = uniform_pm1(95, shape).astype(np.float32)
    y_o, _, sm_o, sv_o, mm_o, mv_o = O.batchnorm_forward(x, gamma, beta, mm0, mv0)
    dx_o, gg_o, gb_o = O.batchnorm_backward(x, dy, gamma, sm_o, sv_o)

    # "ranks": shards of the batch, each with its own layer object / buffers; the all-reduce is emulated by summing the
    # rank tensors of one collective call and writing the total back into each of them
    bounds = np.cumsum((0,) + tuple(splits))
    ranks = []
    for r, n in enumerate(splits):
        sl = slice(bounds[r], bounds[r + 1])
        ranks.append(dict(bn=capi.BatchNorm2d(n, C, H, W), x=dev(T, x[sl]), dy=dev(T, dy[sl]), y=T.empty((n, C, H, W), device="cuda"),
                          g=dev(T, gamma), b=dev(T, beta), mm=dev(T, mm0), mv=dev(T, mv0), gg=T.empty(C, device="cuda"),
                          gb=T.empty(C, device="cuda")))
    count = float(B * H * W)

    L = capi.load()
    s1 = [T.empty(C, device="cuda") for _ in ranks]
    s2 = [T.empty(C, device="cuda") for _ in ranks]
    s4 = [T.empty(4 * C, device="cuda") for _ in ranks]

    def allreduce(ts):
        tot = T.stack(ts).sum(dim=0)
        for t in ts:
            t.copy_(tot)

    def dims(rk):
        return (rk["bn"].B, C, H, W)

    for rk, t in zip(ranks, s1):
        capi.check(L.cnn_batchnorm2d_partial_sums(capi._ptr(rk["x"]), None, 0.0, capi._ptr(t), *dims(rk), capi._ptr(rk["bn"].ws),
                                                  rk["bn"].ws_bytes, None), "sums1")
    allreduce(s1)
    for rk, t1, t2 in zip(ranks, s1, s2):
        capi.check(L.cnn_batchnorm2d_partial_sums(capi._ptr(rk["x"]), capi._ptr(t1), count, capi._ptr(t2), *dims(rk),
                                                  capi._ptr(rk["bn"].ws), rk["bn"].ws_bytes, None), "sums2")
    allreduce(s2)
    for rk, t1, t2 in zip(ranks, s1, s2):
        bn = rk["bn"]
        capi.check(L.cnn_batchnorm2d_forward_from_sums(capi._ptr(rk["x"]), capi._ptr(rk["y"]), capi._ptr(rk["g"]), capi._ptr(rk["b"]),
                                                       capi._ptr(rk["mm"]), capi._ptr(rk["mv"]), capi._ptr(bn.saved_mean),
                                                       capi._ptr(bn.saved_var), capi._ptr(t1), capi._ptr(t2), count, *dims(rk), bn.eps,
                                                       bn.momentum, None), "fwd")
    y = np.concatenate([host(rk["y"]) for rk in ranks])
    assert_close(y, y_o, what="y over the sharded batch")
    for rk in ranks:  # every rank holds the GLOBAL statistics
        assert_close(host(rk["bn"].saved_mean), sm_o, what="mean")
        assert_close(host(rk["bn"].saved_var), sv_o, what="var")
        assert_close(host(rk["mm"]), mm_o, what="moving mean")
        assert_close(host(rk["mv"]), mv_o, what="moving var")
    for rk, t in zip(ranks, s4):
        bn = rk["bn"]
        capi.check(L.cnn_batchnorm2d_backward_sums(capi._ptr(rk["x"]), capi._ptr(rk["dy"]), capi._ptr(rk["g"]), capi._ptr(bn.saved_mean),
                                                   capi._ptr(bn.saved_var), capi._ptr(t), *dims(rk), bn.eps, capi._ptr(bn.ws), bn.ws_bytes,
                                                   None), "bwd sums")
    allreduce(s4)
    for rk, t in zip(ranks, s4):
        bn = rk["bn"]
        capi.check(L.cnn_batchnorm2d_backward_from_sums(capi._ptr(rk["x"]), capi._ptr(rk["dy"]), capi._ptr(rk["g"]),
                                                        capi._ptr(bn.saved_mean), capi._ptr(bn.saved_var), capi._ptr(t), count,
                                                        capi._ptr(rk["gg"]), capi._ptr(rk["gb"]), *dims(rk), bn.eps, None), "bwd")
    dx = np.concatenate([host(rk["dy"]) for rk in ranks])
    assert_close(dx, dx_o, what="dx over the sharded batch")
    for rk in ranks:  # full-batch sums on every rank (batchnorm2d.cpp:123-124: not divided by the batch)
        assert_close(host(rk["gg"]), gg_o, what="gamma grad")
        assert_close(host(rk["gb"]), gb_o, what="beta grad")
    if len(splits) == 1:  # one rank: the same arithmetic as the single-device entry points on their general path (C < 32 here;
        # layers taken by the one-workgroup-per-channel kernels sum a channel in a different order) -- in their TWO-pass form: the
        # sharded statistics are two all-reduced sums (x, then (x - mean)^2), which is batchnorm2d.cpp:46-61's own order; the
        # single-device default since round 5 is ONE pass around a pilot value (bn_stats_pilot), held to the oracle at 1e-4 above
        lib_option("BN_TWO_PASS", "1")
        bn = capi.BatchNorm2d(B, C, H, W)
        xd, gd, bd = dev(T, x), dev(T, gamma), dev(T, beta)
        mm, mv, y1 = dev(T, mm0), dev(T, mv0), T.empty(shape, device="cuda")
        bn.forward(xd, gd, bd, mm, mv, y1, training=True)
        assert T.equal(y1, ranks[0]["y"]) and T.equal(mm, ranks[0]["mm"]) and T.equal(mv, ranks[0]["mv"])
        # the split-phase forward with the ReLU behind it from the same pass (cnn_batchnorm2d_forward_from_sums_relu)
        bn2 = capi.BatchNorm2d(B, C, H, W)
        mm2, mv2, y2, r2 = dev(T, mm0), dev(T, mv0), T.empty(shape, device="cuda"), T.empty(shape, device="cuda")
        bn2.forward_sync(xd, gd, bd, mm2, mv2, y2, lambda t: None, count, y_relu=r2)
        assert T.equal(y2, y1) and T.equal(mm2, mm) and T.equal(mv2, mv) and T.equal(r2, T.where(y2 >= 0, y2, T.zeros_like(y2)))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(6, 8, 25, 25), (6, 32, 12, 12), (64, 64, 56, 56)], ids=str)
def test_batchnorm_relu_only_output_and_its_recomputation(T, shape):
    """round 4: cnn_batchnorm2d_forward_relu / _from_sums_relu with y = NULL write ONLY relu(y) -- bit-identical to the pass that writes both,
    statistics included -- and the evaluation entry with the saved batch statistics in the place of the moving ones re-computes y bit for bit
    (what BatchNorm2D::get_output() of the host layer relies on); general path, channel-resident path and a full-size site of configs[4]"""
    from cnn_amd import capi

    B, C, H, W = shape
    g = T.Generator(device="cuda").manual_seed(5)
    x = T.rand(shape, generator=g, device="cuda") * 3 - 1
    gamma = T.rand((C,), generator=g, device="cuda") + 0.5
    beta = T.rand((C,), generator=g, device="cuda") - 0.5
    for sync in (False, True):
        res = []
        for y_given in (True, False):
            bn = capi.BatchNorm2d(B, C, H, W)
            mm, mv = T.zeros(C, device="cuda"), T.zeros(C, device="cuda")
            y = T.full(shape, 7.0, device="cuda") if y_given else None
            r = T.full(shape, 7.0, device="cuda")
            if sync:
                bn.forward_sync(x, gamma, beta, mm, mv, y, lambda t: t, float(B * H * W), y_relu=r)
            else:
                bn.forward(x, gamma, beta, mm, mv, y, True, y_relu=r)
            res.append((y, r, mm, mv, bn.saved_mean.clone(), bn.saved_var.clone(), bn))
        (y, r, mm, mv, sm, sv, bn), (_, r2, mm2, mv2, sm2, sv2, _) = res
        assert T.equal(r, T.clamp_min(y, 0)) and T.equal(r2, r)
        assert T.equal(mm, mm2) and T.equal(mv, mv2) and T.equal(sm, sm2) and T.equal(sv, sv2)
        again = T.full(shape, 7.0, device="cuda")
        bn.forward(x, gamma, beta, sm.clone(), sv.clone(), again, False)  # evaluation arithmetic on the saved batch statistics
        assert T.equal(again, y), float((again - y).abs().max())


POOLED_SHAPES = [(2, 8, 112, 112), (6, 5, 60, 52), (1, 2, 140, 120), (2, 3, 2, 4), (2, 4, 66, 128), (3, 2, 74, 76)]  # (every channel beyond the one-workgroup limit: B*H*W > 16 K, or tiny)


@pytest.mark.parametrize("shape", POOLED_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_batchnorm_backward_from_the_pooled_domain(T, shape):
    """cnn_batchnorm2d_backward_pooled (round 6): BatchNorm2D <- ReLU <- MaxPool2D(2, 2) backward (batchnorm2d.cpp:98-158, relu.cpp:35-40,
    pool2d.cpp:100-107) as two kernels that rebuild the delta between the layers from (dpool, mask, pooled): against the oracle's three
    backward passes, and BIT-IDENTICAL to cnn_maxpool2d_backward_relu + cnn_batchnorm2d_backward (same elements, same order, same arithmetic)"""
    from cnn_amd import capi

    B, C, H, W = shape
    x = (uniform_pm1(80, shape) * 2 + 0.3).astype(np.float32)
    gamma = (uniform_pm1(81, (C,)) + 1.5).astype(np.float32)
    beta = (uniform_pm1(82, (C,)) * 0.5).astype(np.float32)
    mm0 = np.zeros(C, np.float32)
    mv0 = np.ones(C, np.float32)
    dpool = uniform_pm1(83, (B, C, H // 2, W // 2)).astype(np.float32)
    # oracle: BN forward -> ReLU -> pool; backward pool -> ReLU' -> BN'
    y_o, _, sm_o, sv_o, _, _ = O.batchnorm_forward(x, gamma, beta, mm0, mv0)
    r_o = O.relu_forward(y_o)
    p_o, m_o = O.maxpool_forward(r_o, 2, 2)
    dr_o = O.maxpool_backward(dpool, m_o, shape, 2, 2)
    dyo = O.relu_backward(r_o, dr_o)
    dx_o, gg_o, gb_o = O.batchnorm_backward(x, dyo, gamma, sm_o, sv_o)

    bn = capi.BatchNorm2d(B, C, H, W)
    xd, gd, bd = dev(T, x), dev(T, gamma), dev(T, beta)
    yd, rd = T.empty_like(xd), T.empty_like(xd)
    bn.forward(xd, gd, bd, dev(T, mm0), dev(T, mv0), yd, training=True, y_relu=rd)
    pooled, mask = capi.maxpool_forward(rd, 2, 2)
    dpd = dev(T, dpool)
    if not bn.backward_pooled_supported():  # (a channel that fits one workgroup: the sequence stays three kernels there)
        assert B * H * W <= 16384
        return
    gg, gb, dx = T.full((C,), 7.0, device="cuda"), T.full((C,), 7.0, device="cuda"), T.full_like(xd, 7.0)
    capi.kernel_timing(1)
    bn.backward_pooled(xd, dpd, mask, pooled, gd, gg, gb, dx)
    T.cuda.synchronize()
    names = [k.split("|")[0] for k in capi.kernel_timing_report()]
    capi.kernel_timing(0)
    assert names == ["bn_bwd_stats+pool", "bn_bwd_apply+pool"], names
    # (decisions: a ReLU / pool decision made on the HIP forward tensors can differ from the oracle's on a rounding-distance tie; on these
    #  inputs none does -- asserted -- so the comparison is plain)
    assert np.array_equal(host(mask), m_o) and np.array_equal(host(rd) > 0, r_o > 0)
    assert_close(host(dx), dx_o, what="dx from the pooled domain")
    assert_close(host(gg), gg_o, what="gamma gradient")
    assert_close(host(gb), gb_o, what="beta gradient")
    # the sequence it replaces, bit for bit
    dy3 = capi.maxpool_backward_relu(dpd, mask, pooled, shape, 2, 2)
    gg3, gb3 = T.empty((C,), device="cuda"), T.empty((C,), device="cuda")
    bn.backward(xd, dy3, gd, gg3, gb3)
    u32 = lambda t: host(t).view(np.uint32)
    assert np.array_equal(u32(dx), u32(dy3)) and np.array_equal(u32(gg), u32(gg3)) and np.array_equal(u32(gb), u32(gb3))


@pytest.mark.parametrize("shape", POOLED_SHAPES + [(2, 3, 30, 22)], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("training", [True, False], ids=["train", "eval"])
def test_batchnorm_relu_maxpool_in_one_apply_pass(T, shape, training):
    """cnn_batchnorm2d_forward_relu_pool (round 6): BatchNorm2D -> ReLU -> MaxPool2D(2, 2) (batchnorm2d.cpp:24-95, relu.cpp:25,
    pool2d.cpp:60-83) with the pool inside the apply pass: pooled tensor, mask, statistics and -- when asked for -- y and the ReLU output,
    BIT-IDENTICAL to cnn_batchnorm2d_forward_relu + cnn_maxpool2d_forward; with neither y nor the ReLU output requested (the train step's
    form) the same pooled tensor and mask"""
    from cnn_amd import capi

    B, C, H, W = shape
    bn = capi.BatchNorm2d(B, C, H, W)
    if not capi.load().cnn_batchnorm2d_forward_relu_pool_supported(B, C, H, W):
        assert B * H * W <= 16384
        return
    x = (uniform_pm1(90, shape) * 2 + 0.3).astype(np.float32)
    gamma = (uniform_pm1(91, (C,)) + 1.5).astype(np.float32)
    beta = (uniform_pm1(92, (C,)) * 0.5).astype(np.float32)
    mm0 = (uniform_pm1(93, (C,)) * 0.2).astype(np.float32)
    mv0 = (uniform_pm1(94, (C,)) * 0.2 + 1.0).astype(np.float32)
    xd, gd, bd = dev(T, x), dev(T, gamma), dev(T, beta)
    u32 = lambda t: host(t).view(np.uint32)
    # the two-call sequence
    mm1, mv1 = dev(T, mm0), dev(T, mv0)
    y1, r1 = T.empty_like(xd), T.empty_like(xd)
    bn.forward(xd, gd, bd, mm1, mv1, y1, training=training, y_relu=r1)
    p1, m1 = capi.maxpool_forward(r1, 2, 2)
    sm1, sv1 = bn.saved_mean.clone(), bn.saved_var.clone()
    # one pass, everything written
    bn2 = capi.BatchNorm2d(B, C, H, W)
    mm2, mv2 = dev(T, mm0), dev(T, mv0)
    y2, r2 = T.full_like(xd, 7.0), T.full_like(xd, 7.0)
    p2, m2 = T.full_like(p1, 7.0), T.full_like(m1, -3)
    bn2.forward_relu_pool(xd, gd, bd, mm2, mv2, p2, m2, y=y2, y_relu=r2, training=training)
    assert np.array_equal(u32(y2), u32(y1)) and np.array_equal(u32(r2), u32(r1))
    assert np.array_equal(u32(p2), u32(p1)) and np.array_equal(host(m2), host(m1))
    assert np.array_equal(u32(mm2), u32(mm1)) and np.array_equal(u32(mv2), u32(mv1))
    if training:
        assert np.array_equal(u32(bn2.saved_mean), u32(sm1)) and np.array_equal(u32(bn2.saved_var), u32(sv1))
    # ... and the train step's form: only the pooled tensor and the mask
    bn3 = capi.BatchNorm2d(B, C, H, W)
    p3, m3 = T.full_like(p1, 7.0), T.full_like(m1, -3)
    bn3.forward_relu_pool(xd, gd, bd, dev(T, mm0), dev(T, mv0), p3, m3, training=training)
    assert np.array_equal(u32(p3), u32(p1)) and np.array_equal(host(m3), host(m1))
    # against the oracle
    if training:
        y_o, _, _, _, _, _ = O.batchnorm_forward(x, gamma, beta, mm0, mv0)
        p_o, _ = O.maxpool_forward(O.relu_forward(y_o), 2, 2)
        assert_close(host(p3), p_o, what="pooled")
