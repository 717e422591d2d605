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
