"""Data-parallel glue: one process per GPU, the batch sharded by sample, ONE all-reduce of the flat gradient arena
per step (RCCL over xGMI when the backend is "nccl"; gloo on CPU for the tests).

The reference has no parallelism at all; the only cross-sample coupling on the path is the
batch mean inside the weight / bias gradients (conv2d.cpp:148,157; linear.cpp:62,70).  Each rank's kernels divide by
its LOCAL batch B_local, the all-reduce SUMS the G arenas and the SGD kernel multiplies by 1/G:
    (1/G) * sum_ranks (1/B_local) * sum_local  ==  (1/B_global) * sum_all        (equal shards)
Identical initial weights + identical reduced gradients keep the replicas bit-identical without any broadcast.
"""
import os


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init_process_group(backend, device_id=None):
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    kw = {"device_id": device_id} if device_id is not None else {}
    dist.init_process_group(backend, **kw)
    return dist


def shard_bounds(global_batch, rank, world):
    """sample range [lo, hi) of `rank`: equal contiguous shards (the mean-of-means identity needs equal sizes)"""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} is not divisible by {world} ranks")
    per = global_batch // world
    return rank * per, (rank + 1) * per


def allreduce_grads(grads, dist, world):
    """in-place SUM of the flat gradient arena over all ranks; returns the scale the SGD step must apply (1/G)"""
    if world > 1:
        dist.all_reduce(grads)
    return 1.0 / world


def sum_allreduce(dist, world):
    """the `allreduce(t)` callable cnn_amd.capi.BatchNorm2d.forward_sync / backward_sync expect: in-place SUM of a small
    per-channel tensor over all ranks (no-op for one rank)"""

    def allreduce(t):
        if world > 1:
            dist.all_reduce(t)
        return t

    return allreduce


def syncbn_reference_protocol(x, dy, gamma, beta, allreduce, global_count, eps=1e-5):
    """numpy statement of the sync-BN exchange (include/cnn_amd.h, cnn_batchnorm2d_partial_sums ...): what each rank
    computes between the three collectives.  Used by the gloo test to pin the protocol against the full-batch oracle;
    the HIP entry points follow the same steps (tests/test_gpu_batchnorm.py)."""
    import numpy as np
    import torch

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
