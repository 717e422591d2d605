"""worker of tests/test_data_parallel_gloo.py: one rank of a world_size-N gloo job on CPU.
Each rank computes the gradients of ITS shard with the oracle (divided by the local batch, like the kernels),
all-reduces the flat arena through cnn_amd.dp (the function bench.py uses), applies SGD with the 1/G scale and
rank 0 checks the result against the full-batch oracle step."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from cnn_amd import dp  # noqa: E402
from oracle import pyoracle as O  # noqa: E402
from tests.util import normal_scaled, uniform01  # noqa: E402


def syncbn_reference_protocol(x, dy, gamma, beta, allreduce, global_count, eps=1e-5):
    """numpy statement of the sync-BN exchange (include/cnn_amd.h, cnn_batchnorm2d_partial_sums ...): what each rank
    computes between the three collectives.  Used by the gloo test to pin the protocol against the full-batch oracle;
    the HIP entry points follow the same steps (tests/test_gpu_batchnorm.py)."""
    ax = (0, 2, 3)
    s1 = torch.from_numpy(x.sum(axis=ax, dtype=np.float32))
    allreduce(s1)
    mean = (s1.numpy() / np.float32(global_count)).astype(np.float32)
    xc = x - mean[None, :, None, None]
    s2 = torch.from_numpy((xc * xc).sum(axis=ax, dtype=np.float32))
    allreduce(s2)
    var = (s2.numpy() / np.float32(global_count)).astype(np.float32)
    inv = (1.0 / np.sqrt(var + np.float32(eps))).astype(np.float32)
    norm = xc * inv[None, :, None, None]
    y = gamma[None, :, None, None] * norm + beta[None, :, None, None]
    g = gamma[None, :, None, None]
    s4 = np.stack([(dy * norm).sum(axis=ax), dy.sum(axis=ax), ((dy * g) * xc * np.float32(-0.5) * (inv ** 3)[None, :, None, None]).sum(axis=ax),
                   xc.sum(axis=ax)], axis=1).astype(np.float32)
    t4 = torch.from_numpy(s4)
    allreduce(t4)
    s4 = t4.numpy()
    L = np.float32(global_count)
    inv_v = s4[:, 2] / L
    u_g = (s4[:, 1] * gamma) * (-inv) + inv_v * np.float32(-2) * s4[:, 3]
    dx = (dy * g) * inv[None, :, None, None] + (inv_v * 2)[None, :, None, None] * xc + (u_g / L)[None, :, None, None]
    return y.astype(np.float32), dx.astype(np.float32), s4[:, 0].copy(), s4[:, 1].copy(), mean, var


def main():
    world, rank, _ = dp.env_world()
    dist = dp.init_process_group("gloo")
    GB, Hh, lr = 4, 64, 1e-2
    x = uniform01(50, (GB, 3, Hh, Hh))
    labels = (np.arange(GB) % 3).astype(np.int32)
    lo, hi = dp.shard_bounds(GB, rank, world)
    net = O.Net(hi - lo, 3, Hh, Hh)
    p0 = normal_scaled(51, (net.n_params,))
    net.params[:] = p0
    probs = O.softmax(net.forward(x[lo:hi]))
    _, delta = O.cross_entropy_backward(probs, labels[lo:hi])
    net.backward(delta)  # local gradients = (1/B_local) * sum over the shard
    grads = torch.from_numpy(net.grads.copy())
    scale = dp.allreduce_grads(grads, dist, world)
    new_params = O.sgd_update(p0, grads.numpy() * np.float32(scale), lr)
    # every rank must hold the same reduced arena (replicas stay in lock-step without a broadcast)
    check = torch.from_numpy(new_params.copy())
    ref = check.clone()
    dist.broadcast(ref, 0)
    assert torch.equal(check, ref), "replicas diverged"
    # the fused step tail of architectures::Sequential (host/src/sequential.cpp, fused_tail) exchanges the arena in TWO buckets --
    # everything behind conv_layer_1 first (it is final one weight-gradient kernel earlier), conv_layer_1's 448 floats last -- each
    # followed by its own SGD range: the same sums (a ring / tree all-reduce adds an element's terms in an order that depends on where
    # the element sits in the message, so the last bits may differ from the one-call exchange), identical on every rank
    g2 = torch.from_numpy(net.grads.copy())
    lo1 = 16 * 3 * 9 + 16  # conv_layer_1: filters + biases (alexnet.cpp:12, checkpoint order)
    p2 = p0.copy()
    for a, b in ((lo1, net.n_params), (0, lo1)):
        part = g2[a:b].clone()
        s2 = dp.allreduce_grads(part, dist, world)
        p2[a:b] = O.sgd_update(p0[a:b], part.numpy() * np.float32(s2), lr)
    assert np.abs(p2 - new_params).max() <= 1e-6 * np.abs(new_params).max(), "two-bucket exchange differs from the one-call exchange"
    check2 = torch.from_numpy(p2.copy())
    ref2 = check2.clone()
    dist.broadcast(ref2, 0)
    assert torch.equal(check2, ref2), "replicas diverged after the two-bucket exchange"
    if rank == 0:
        full = O.Net(GB, 3, Hh, Hh)
        full.params[:] = p0
        full.train_step(x, labels, lr)
        err = np.abs(new_params - full.params).max() / np.abs(full.params).max()
        assert err < 1e-5, err
        print(f"DP_OK world={world} err={err:.2e}")
    # ---- sync-BN exchange (BatchNorm couples samples in forward AND backward) over the same collective ----
    Bn, C = 8, 5
    xb = (uniform01(60, (Bn, C, 6, 7)) * 3 - 1).astype(np.float32)
    dyb = (uniform01(61, (Bn, C, 6, 7)) * 2 - 1).astype(np.float32)
    gamma = (uniform01(62, (C,)) + 0.5).astype(np.float32)
    beta = uniform01(63, (C,)).astype(np.float32)
    lo, hi = dp.shard_bounds(Bn, rank, world)
    y, dx, gg, gb, mean, var = syncbn_reference_protocol(xb[lo:hi], dyb[lo:hi], gamma, beta, dp.sum_allreduce(dist, world),
                                                            Bn * 6 * 7)
    y_o, _, sm_o, sv_o, _, _ = O.batchnorm_forward(xb, gamma, beta, np.zeros(C, np.float32), np.zeros(C, np.float32))
    dx_o, gg_o, gb_o = O.batchnorm_backward(xb, dyb, gamma, sm_o, sv_o)
    for got, ref in ((y, y_o[lo:hi]), (dx, dx_o[lo:hi]), (gg, gg_o), (gb, gb_o), (mean, sm_o), (var, sv_o)):
        assert np.abs(got - ref).max() <= 1e-4 * max(1e-6, np.abs(ref).max()), "sync-BN shard differs from the full-batch oracle"
    if rank == 0:
        print(f"SYNCBN_OK world={world}")
    # ---- two communicators, the container's issue order (Sequential::set_comm: cnn_comm_split gives BatchNorm2D's sync-BN sums a
    # communicator of their own; Sequential::backward sends buckets of the gradient arena on the communication stream WHILE the walk
    # goes on) -- layers back to front: a BatchNorm2D layer reduces its [C][4] sums synchronously on communicator B in the middle of the
    # walk, finished buckets go out asynchronously on communicator A; nothing on A waits for anything on B or the other way round
    grp_a, grp_b = dist.new_group(list(range(world))), dist.new_group(list(range(world)))
    rs = np.random.RandomState(70)
    layers = [("bn", 5), ("conv", 300000), ("bn", 7), ("conv", 200000), ("bn", 3), ("conv", 50000)]
    arena = [torch.from_numpy((rs.standard_normal(n).astype(np.float32) * (rank + 1))) for _, n in layers]
    base = [a.clone() / (rank + 1) for a in arena]
    total = sum(r + 1 for r in range(world))
    pending, bucket = [], []
    for idx in range(len(layers) - 1, -1, -1):
        kind, n = layers[idx]
        if kind == "bn":
            dist.all_reduce(arena[idx], group=grp_b)  # the layer's backward waits for this one (it needs the global sums)
            assert torch.allclose(arena[idx], base[idx] * total, rtol=1e-6, atol=1e-6)
        else:
            bucket.append(idx)
            if sum(layers[i][1] for i in bucket) >= 250000:  # a full bucket leaves while the walk continues
                pending += [(i, dist.all_reduce(arena[i], group=grp_a, async_op=True)) for i in bucket]
                bucket = []
    pending += [(i, dist.all_reduce(arena[i], group=grp_a, async_op=True)) for i in bucket]
    for i, work in pending:
        work.wait()
        assert torch.allclose(arena[i], base[i] * total, rtol=1e-6, atol=1e-5)
    if rank == 0:
        print(f"TWO_COMMS_OK world={world}")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
