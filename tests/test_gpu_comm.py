"""The data-parallel exchange behind the C ABI (cnn_comm_* / cnn_allreduce_grads -> RCCL): what can be checked on one GPU, and
the two-replica run of the C++ container that needs two (skipped cleanly on a 1-GPU box)."""
import ctypes as C
import threading

import numpy as np
import pytest

from cnn_amd import stacks as S
from tests.util import he_init, uniform01

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a device"
    return torch


def test_rccl_is_bound_and_a_one_rank_communicator_works(T):
    """RCCL is loaded through libcnn_amd.so (dlopen by soname), a 1-rank communicator initialises on this device and the in-place
    sum over it is the identity; nccl_smoke-style sanity that the driver's GPU test run exercises the exchange entry points"""
    from cnn_amd import capi
    from cnn_amd.dp import RcclComm

    lib = capi.load()
    assert lib.cnn_comm_available() == 1
    assert lib.cnn_comm_version() >= 20000  # RCCL 2.x
    comm = RcclComm(None, 1, 0)
    x = T.arange(100003, device="cuda", dtype=T.float32) * 0.5 - 7.0
    ref = x.clone()
    comm.all_reduce(x)
    T.cuda.synchronize()
    assert T.equal(x, ref)
    w, r = C.c_int(), C.c_int()
    capi.check(lib.cnn_comm_info(comm.handle, C.byref(w), C.byref(r)), "cnn_comm_info")
    assert (w.value, r.value) == (1, 0)
    assert lib.cnn_allreduce_grads(None, capi._ptr(x), 4, None) != 0 and b"null" in lib.cnn_amd_last_error()
    comm.destroy()


def test_container_with_a_one_rank_communicator_equals_the_plain_step(T):
    """Sequential::set_comm(world = 1): the exchange path is a no-op and BatchNorm2D takes the single-process statistics"""
    from cnn_amd import hostapi
    from cnn_amd.dp import RcclComm

    spec = S.alexnet(3, batch_norm=True)
    layout = S.walk(spec)
    p0 = he_init(layout, 81)
    B = 4
    x = T.from_numpy(uniform01(82, (B, 3, 224, 224))).cuda()
    labels = T.from_numpy((np.arange(B) % 3).astype(np.int32)).cuda()
    comm = RcclComm(None, 1, 0)
    outs = []
    for use_comm in (False, True):
        net = hostapi.HostSequential(spec)
        net.set_params(p0)
        if use_comm:
            net.set_comm(comm.handle, 1)
        for _ in range(2):
            net.train_step(x, labels, 1e-3)
        outs.append((net.last_loss(), net.get_params()))
        net.close()
    comm.destroy()
    assert outs[0][0] == outs[1][0] and np.array_equal(outs[0][1], outs[1][1])


def test_two_bucket_exchange_of_the_fused_step_tail_with_one_rank(T, lib_option):
    """the reference net's train step with a communicator (Sequential::fused_tail): bucket 1 -- everything behind conv_layer_1 --
    is all-reduced on the side stream as soon as its reductions land, under conv_layer_1's weight gradient, followed by its SGD step
    and filter images; bucket 2 -- conv_layer_1's 448 floats -- behind that kernel.  Forced on with ONE rank (DP_FORCE_EXCHANGE:
    every sum is an identity): four steps must equal the plain fused step bit for bit; a bucket sent before its gradients are
    final, or an SGD step that does not wait for its bucket, would show"""
    from cnn_amd import hostapi
    from cnn_amd.dp import RcclComm

    B = 4
    x = T.from_numpy(uniform01(92, (B, 3, 224, 224))).cuda()
    labels = T.from_numpy((np.arange(B) % 3).astype(np.int32)).cuda()
    p0 = (np.random.RandomState(93).standard_normal(111267) * 0.1).astype(np.float32)
    comm = RcclComm(None, 1, 0)
    outs = []
    for use_comm in (False, True):
        net = hostapi.HostAlexNet(3)
        net.set_params(p0)
        if use_comm:
            net.set_comm(comm.handle, 1)
            lib_option("DP_FORCE_EXCHANGE", "1")
        losses = []
        for _ in range(4):
            net.train_step(x, labels, 1e-3)
            losses.append(net.last_loss())
        outs.append((losses, net.get_params(), net.get_grads(), net.input_delta((B, 3, 224, 224))))
        net.close()
        lib_option("DP_FORCE_EXCHANGE", None)
    comm.destroy()
    assert outs[0][0] == outs[1][0]
    for a, b in zip(outs[0][1:], outs[1][1:]):
        assert np.array_equal(a, b)


def test_bucketed_exchange_path_with_one_rank(T, monkeypatch):
    """the bucketed all-reduce of big arenas (Sequential::backward: buckets go out on the communication stream while the backward
    pass is still running) exercised on one GPU: with a 1-rank communicator every bucket's sum is the identity, so the step must
    equal the plain one bit for bit -- ordering bugs (a bucket sent before its gradients are final) would show"""
    import subprocess
    import sys
    import os

    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from cnn_amd import hostapi, stacks as S
from cnn_amd.dp import RcclComm
spec = S.vgg11()
p0 = S.he_init(S.walk(spec, 3, 64, 64), 7)
x = torch.rand((4, 3, 64, 64), generator=torch.Generator(device='cuda').manual_seed(3), device='cuda')
labels = (torch.arange(4, device='cuda') %% 3).to(torch.int32)
comm = RcclComm(None, 1, 0)
res = []
for use in (0, 1):
    net = hostapi.HostSequential(spec, (3, 64, 64))
    net.set_params(p0)
    if use: net.set_comm(comm.handle, 1)
    for _ in range(2): net.train_step(x, labels, 1e-3)
    res.append(net.get_params()); net.close()
assert np.array_equal(res[0], res[1]), float(np.abs(res[0]-res[1]).max())
print('BUCKETS_OK')
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CNN_AMD_DP_FORCE_BUCKETS="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "BUCKETS_OK" in out.stdout, out.stdout[-1000:] + out.stderr[-3000:]


@pytest.mark.parametrize("world", [1, 2])
@pytest.mark.parametrize("which", ["alexnet_bn", "resnet18"])
def test_replicas_of_the_cpp_container_equal_the_full_batch_step(T, which, world, lib_option):
    """configs[2] / [4] in miniature -- every collective the product issues, on one 2-GPU box: two replicas of the C++ container (one per
    GPU, one host thread each, ncclCommInitAll via cnn_comm_init_all) train on the two halves of a batch and end up with the parameters
    of ONE replica stepping on the whole batch.
      alexnet_bn: the reference net with BatchNorm2D (alexnet.cpp:13-23) -- sync-BN (two [C] all-reduces forward, one [C][4] backward,
                  batchnorm2d.cpp:46-61,129-147) and the small arena's exchange inside the fused step tail;
      resnet18:   the ResNet-18-shaped stack (configs[4]; 43 MB of gradients) -- the BUCKETED exchange (>= 8 MB buckets on the
                  communication stream while the backward pass is still running, one wait at the end) interleaved with seventeen sync-BN
                  layers on the same communicator, 1x1 / 7x7 / 3x3 stride-1 and stride-2 layers.
    world = 1 runs the SAME body on the 1-GPU test box (one replica on the whole batch through a 1-rank communicator, the bucketed path
    forced on): every call the two-replica run makes is made, every sum is an identity."""
    if T.cuda.device_count() < world:
        pytest.skip("needs two GPUs (the driver's 1-GPU test box has one)")
    from cnn_amd import capi, hostapi

    if world == 1:
        lib_option("DP_FORCE_BUCKETS", "1")

    lib = capi.load()
    if which == "alexnet_bn":
        spec, in_shape, GB = S.alexnet(3, batch_norm=True), (3, 224, 224), 8
    else:
        spec, in_shape, GB = S.resnet18(3), (3, 224, 224), 4
    layout = S.walk(spec, *in_shape)
    p0 = he_init(layout, 91)
    lr, steps, half = 1e-3, 2, GB // world
    x = uniform01(92, (GB,) + in_shape)
    labels = (np.arange(GB) % 3).astype(np.int32)
    comms = (C.c_void_p * world)()
    capi.check(lib.cnn_comm_init_all(comms, world, None), "cnn_comm_init_all")
    results, errors = [None] * world, []

    def replica(rank):
        try:
            T.cuda.set_device(rank)
            xs = T.from_numpy(x[rank * half:(rank + 1) * half]).cuda()
            ls = T.from_numpy(labels[rank * half:(rank + 1) * half]).cuda()
            net = hostapi.HostSequential(spec, in_shape)
            net.set_params(p0)
            net.set_comm(C.c_void_p(comms[rank]), world)
            for _ in range(steps):
                net.train_step(xs, ls, lr)
            T.cuda.synchronize()
            results[rank] = net.get_params()
            net.close()
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=replica, args=(r,)) for r in range(world)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors
    T.cuda.set_device(0)
    full = hostapi.HostSequential(spec, in_shape)
    full.set_params(p0)
    xf, lf = T.from_numpy(x).cuda(), T.from_numpy(labels).cuda()
    for _ in range(steps):
        full.train_step(xf, lf, lr)
    want = full.get_params()
    full.close()
    for c in comms:
        lib.cnn_comm_destroy(C.c_void_p(c))
    assert all(np.array_equal(results[0], r) for r in results[1:]), "replicas diverged"
    # (the two-replica sums differ from the one-replica sums by summation order only: conv2d.cpp:148's batch mean as two partial means,
    # batchnorm2d.cpp:46-61's statistics as two partial sums)
    err = np.abs(results[0] - want).max() / np.abs(want).max()
    assert err <= (1e-5 if which == "alexnet_bn" else 1e-4), err


def test_batch_stager_orders_producer_upload_and_consumer(T):
    """cnn_batch_stager_* (row n4): pinned slots uploaded on the stager's copy stream arrive intact, in order, while the consumer
    of the previous batch is still running; a slot is not handed back to the producer before its consumer released it"""
    from cnn_amd import capi

    n = 1 << 20
    stager = capi.BatchStager(n * 4, depth=2)
    outs = []
    for i in range(7):
        host, slot = stager.acquire()
        host[:] = np.arange(n, dtype=np.float32) * 0.5 - i  # negative for i > 0: ReLU makes the consumer's result data-dependent
        dev = stager.submit(slot)
        stager.wait(slot)
        y = T.empty(n, device="cuda")
        capi.check(capi.load().cnn_relu_forward(dev, capi._ptr(y), n, capi._stream()), "cnn_relu_forward")
        stager.release(slot)
        outs.append(y)
    T.cuda.synchronize()
    for i, y in enumerate(outs):
        want = np.maximum(np.arange(n, dtype=np.float32) * 0.5 - i, 0)
        assert np.array_equal(y.cpu().numpy(), want), i
    stager.close()


def test_published_kernels_order_another_stream(T):
    """cnn_amd_publish_next_kernel / cnn_amd_wait_published: a second stream that waits for the published kernel sees its result --
    both for a kernel that carries the event in its own dispatch packet (linear backward, a launch_pub site) and for one that falls
    back to a plain event record (ReLU); waiting without anything published, or before the armed kernel was launched, is an error"""
    from cnn_amd import capi

    lib = capi.load()
    other = T.cuda.Stream()
    with pytest.raises(capi.CnnAmdError):
        with T.cuda.stream(T.cuda.Stream()):  # (a fresh stream: whatever earlier tests published belongs to other streams / is consumed)
            capi.publish_next_kernel()
            capi.wait_published(other)  # armed, not launched yet
    # consume the arm so that it does not leak into the next launch of this thread on that stream
    B, n_in, n_out = 64, 4608, 3
    x = T.rand((B, n_in), device="cuda")
    dy = T.rand((B, n_out), device="cuda") - 0.5
    w = T.rand((n_in, n_out), device="cuda") - 0.5
    for kind in ("linear_bwd", "relu"):
        for rep in range(3):
            T.cuda.synchronize()
            big = T.rand((1 << 24,), device="cuda") - 0.5
            out = T.zeros_like(big)
            gw, gb, dx = T.empty_like(w), T.empty(n_out, device="cuda"), T.zeros((B, n_in), device="cuda")
            capi.publish_next_kernel()
            if kind == "relu":
                capi.check(lib.cnn_relu_forward(capi._ptr(big), capi._ptr(out), big.numel(), capi._stream()), "cnn_relu_forward")
            else:
                capi.linear_backward(x, dy, w, float(B), gw, gb, dx, relu_below=True)
            capi.wait_published(other)
            with T.cuda.stream(other):
                got = (out.clone() if kind == "relu" else dx.clone())
            other.synchronize()
            T.cuda.synchronize()
            ref = T.clamp(big, min=0) if kind == "relu" else dx
            assert T.equal(got, ref), (kind, rep)


def test_side_stream_handle_and_early_reduction_flush(T):
    """cnn_amd_side_stream_get hands out the (stable) side stream of this thread; cnn_amd_flush_reduces with nothing recorded is a
    no-op; a deferred weight gradient flushed on the side stream and joined equals the in-order call bit for bit"""
    from cnn_amd import capi

    s1, s2 = capi.side_stream(), capi.side_stream()
    assert s1.cuda_stream != 0 and s1.cuda_stream == s2.cuda_stream
    capi.flush_reduces()
    case = (8, 16, 55, 55, 32, 3, 2, 0)
    conv = capi.Conv2d(*case)
    x = T.rand((8, 16, 55, 55), device="cuda")
    w = T.randn((32, 16, 3, 3), device="cuda") * 0.1
    dy = T.rand(conv.out_shape(), device="cuda") - 0.5
    gw_ref, gb_ref, dx_ref = T.empty_like(w), T.empty(32, device="cuda"), T.empty_like(x)
    conv.backward(x, dy, w, 8.0, gw_ref, gb_ref, dx_ref)
    gw, gb, dx = T.full_like(w, 7.0), T.full((32,), 7.0, device="cuda"), T.empty_like(x)
    conv.backward(x, dy, w, 8.0, gw, gb, dx, defer_join=True)
    with T.cuda.stream(s1):
        capi.flush_reduces()  # the recorded reduction, now, on the side stream
    capi.side_stream_join()
    T.cuda.synchronize()
    assert T.equal(gw, gw_ref) and T.equal(gb, gb_ref) and T.equal(dx, dx_ref)


def test_second_communicator_and_broadcast_with_one_rank(T):
    """cnn_comm_split (the communicator BatchNorm2D's sync-BN reductions get inside a data-parallel container) and cnn_comm_broadcast
    (ships rank 0's measured kernel choices): with one rank both are identities, but every call goes through RCCL -- the split
    communicator reports the same world / rank, sums on the two communicators interleave on two streams without ordering each other"""
    from cnn_amd import capi
    from cnn_amd.dp import RcclComm

    lib = capi.load()
    comm = RcclComm(None, 1, 0)
    second = C.c_void_p()
    capi.check(lib.cnn_comm_split(comm.handle, 0, 0, C.byref(second)), "cnn_comm_split")
    w, r = C.c_int(), C.c_int()
    capi.check(lib.cnn_comm_info(second, C.byref(w), C.byref(r)), "cnn_comm_info")
    assert (w.value, r.value) == (1, 0) and second.value != comm.handle.value
    a = T.arange(1 << 20, device="cuda", dtype=T.float32)
    b = T.arange(4096, device="cuda", dtype=T.float32) * 3
    ra, rb = a.clone(), b.clone()
    side = T.cuda.Stream()
    for _ in range(8):  # a "bucket" on one stream / communicator, a small "sync-BN" sum on the other, issued alternately
        with T.cuda.stream(side):
            capi.check(lib.cnn_allreduce_grads(comm.handle, capi._ptr(a), a.numel(), C.c_void_p(side.cuda_stream)), "cnn_allreduce_grads")
        capi.check(lib.cnn_allreduce_grads(second, capi._ptr(b), b.numel(), capi._stream()), "cnn_allreduce_grads")
    buf = T.tensor([7, -1, 3, 227], device="cuda", dtype=T.int32)
    capi.check(lib.cnn_comm_broadcast(comm.handle, capi._ptr(buf), 16, 0, capi._stream()), "cnn_comm_broadcast")
    T.cuda.synchronize()
    assert T.equal(a, ra) and T.equal(b, rb) and buf.tolist() == [7, -1, 3, 227]
    assert lib.cnn_comm_broadcast(None, capi._ptr(buf), 16, 0, None) != 0
    capi.check(lib.cnn_comm_destroy(second), "cnn_comm_destroy")
    comm.destroy()


def test_measured_tile_choice_travels_between_processes(T):
    """cnn_conv2d_autotune_ws (scratch from the caller) + cnn_conv2d_tune_export / _import: what a data-parallel container ships from
    rank 0 to the other replicas so that all of them run the same kernels.  The choice is a table entry per (geometry, mode): exported
    after measuring, importable under another geometry's key, and a geometry that has been measured asks for no scratch again"""
    from cnn_amd import capi

    lib = capi.load()
    d = capi.ConvDesc(2, 20, 20, 20, 40, 5, 1, 2)          # generic (5x5: no row kernel): implicit GEMM in both directions
    first = capi.ConvDesc(2, 3, 224, 224, 16, 3, 2, 0)     # the reference net's first layer: specialised kernels, nothing to measure
    assert lib.cnn_conv2d_autotune_workspace_bytes(C.byref(first)) == 0
    capi.check(lib.cnn_conv2d_autotune_ws(C.byref(first), None, 0, capi._stream()), "cnn_conv2d_autotune_ws")  # (a no-op)
    none = -2**31
    out = (C.c_int32 * 4)()
    capi.check(lib.cnn_conv2d_tune_export(C.byref(d), out), "cnn_conv2d_tune_export")
    need = lib.cnn_conv2d_autotune_workspace_bytes(C.byref(d))
    if list(out)[:2] == [none, none]:  # (not measured yet in this process)
        assert need > 4 * (2 * 20 * 400 + 2 * 40 * 400 + 40 * 20 * 25)
        assert lib.cnn_conv2d_autotune_ws(C.byref(d), None, 0, capi._stream()) != 0  # scratch is the caller's
        scratch = T.empty(need, dtype=T.uint8, device="cuda")
        capi.check(lib.cnn_conv2d_autotune_ws(C.byref(d), capi._ptr(scratch), need, capi._stream()), "cnn_conv2d_autotune_ws")
        T.cuda.synchronize()
        capi.check(lib.cnn_conv2d_tune_export(C.byref(d), out), "cnn_conv2d_tune_export")
    assert out[0] != none and out[1] != none
    assert lib.cnn_conv2d_autotune_workspace_bytes(C.byref(d)) == 0  # measured once per process
    twin = capi.ConvDesc(3, 20, 20, 20, 40, 5, 1, 2)       # "another replica": same layer, its own table entry
    got = (C.c_int32 * 4)()
    capi.check(lib.cnn_conv2d_tune_export(C.byref(twin), got), "cnn_conv2d_tune_export")
    if list(got)[:2] == [none, none]:
        capi.check(lib.cnn_conv2d_tune_import(C.byref(twin), out), "cnn_conv2d_tune_import")
        capi.check(lib.cnn_conv2d_tune_export(C.byref(twin), got), "cnn_conv2d_tune_export")
        assert list(got) == list(out)
        assert lib.cnn_conv2d_autotune_workspace_bytes(C.byref(twin)) == 0  # pinned: not measured again
    # and the pinned tiles compute the same convolution (forward against the im2col fallback)
    conv = capi.Conv2d(3, 20, 20, 20, 40, 5, 1, 2)
    x = T.from_numpy(uniform01(97, (3, 20, 20, 20))).cuda()
    wgt = T.from_numpy(uniform01(98, (40, 20, 5, 5)) - 0.5).cuda()
    bias = T.from_numpy(uniform01(99, (40,))).cuda()
    y = conv.forward(x, wgt, bias)
    ref = conv.forward_im2col(x, wgt, bias)
    T.cuda.synchronize()
    assert float((y - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
