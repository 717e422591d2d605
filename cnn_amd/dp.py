"""Data-parallel glue: one process per GPU, the batch sharded by sample, ONE all-reduce of the flat gradient arena
per step -- RCCL over xGMI through the C ABI (RcclComm -> cnn_allreduce_grads) on the GPU path; a torch.distributed
handle (gloo) on CPU for the tests.

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


class RcclComm:
    """The data path's exchange: an RCCL communicator owned through the C ABI (cnn_comm_* / cnn_allreduce_grads in
    include/cnn_amd.h), one rank per process / GPU.  torch.distributed is only the courier of the 128-byte id (any
    rendezvous would do).  `all_reduce(t)` has the signature allreduce_grads() expects, so an instance stands in for the
    torch.distributed handle; `.handle` is the void* architectures::Sequential::set_comm takes."""

    def __init__(self, dist, world, rank, device="cuda"):
        import ctypes as C

        import torch

        from . import capi

        self.capi, self.world, self.rank = capi, world, rank
        lib = capi.load()
        if not lib.cnn_comm_available():
            raise capi.CnnAmdError("librccl could not be bound by libcnn_amd.so: " + lib.cnn_amd_last_error().decode())
        raw = (C.c_char * 128)()
        if rank == 0:
            capi.check(lib.cnn_comm_unique_id(raw), "cnn_comm_unique_id")
        buf = torch.frombuffer(bytearray(raw.raw), dtype=torch.uint8).to(device)
        if world > 1:
            dist.broadcast(buf, 0)
        self.handle = C.c_void_p()
        capi.check(lib.cnn_comm_init_rank(C.byref(self.handle), world, rank, buf.cpu().numpy().tobytes()), "cnn_comm_init_rank")
        w, r = C.c_int(), C.c_int()
        capi.check(lib.cnn_comm_info(self.handle, C.byref(w), C.byref(r)), "cnn_comm_info")
        assert (w.value, r.value) == (world, rank), (w.value, r.value, world, rank)
        self.version = int(lib.cnn_comm_version())

    def all_reduce(self, t):
        """in-place fp32 sum over all ranks, enqueued on the current stream"""
        self.capi.check(self.capi.load().cnn_allreduce_grads(self.handle, self.capi._ptr(t), t.numel(), self.capi._stream()),
                        "cnn_allreduce_grads")
        return t

    def destroy(self):
        if self.handle:
            self.capi.load().cnn_comm_destroy(self.handle)
            self.handle = None


def sum_allreduce(dist, world):
    """the `allreduce(t)` callable cnn_amd.capi.BatchNorm2d.forward_sync / backward_sync expect: in-place SUM of a small
    per-channel tensor over all ranks (no-op for one rank)"""

    def allreduce(t):
        if world > 1:
            dist.all_reduce(t)
        return t

    return allreduce
