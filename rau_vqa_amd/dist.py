"""Data-parallel plumbing: one process per GPU, gradients averaged over RCCL/xGMI.

The reference is single-GPU, so this is new
work, not a port.  Samples are independent in forward/backward; the only
cross-sample coupling is the 1/B of the cross-entropy (SS:310), so the
data-parallel step is: every rank runs rau_forward/rau_backward on its shard
with 1/B_local scaling, then the three flat gradient buffers (SS:322-324) are
ALL-REDUCE-AVERAGED, then every rank applies the identical update.

torch.distributed is used as plumbing only (process group, RCCL collective);
the tensors it reduces are zero-copy views of librau's device buffers.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class _DevBuf:
    """Expose a raw device allocation through __cuda_array_interface__."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4",
                                         "data": (ptr, False), "version": 2}


def device_view(ptr: int, n: int, device_id: int) -> torch.Tensor:
    """Zero-copy float32 torch view of n floats at device address ptr."""
    return torch.as_tensor(_DevBuf(ptr, n), device=torch.device("cuda", device_id))


def shard_batch(batch, rank: int, world: int):
    """Rank's slice of a global batch (even split by sample; B % world == 0)."""
    B = batch["lens"].shape[0]
    if B % world:
        raise ValueError(f"global batch {B} not divisible by world size {world}")
    per = B // world
    sl = slice(rank * per, (rank + 1) * per)
    return {"feats": batch["feats"][sl], "tokens": batch["tokens"][:, sl],
            "lens": batch["lens"][sl], "labels": batch["labels"][sl]}


def allreduce_average(tensors, group=None):
    """In-place average over ranks, largest-first so the big `mult` bucket is in
    flight first.  Uses RCCL's AVG on GPU tensors, SUM + scale on gloo."""
    world = dist.get_world_size(group)
    if world == 1:
        return
    avg = dist.get_backend(group) == "nccl"   # RCCL has AVG; gloo only SUM
    works = []
    for t in sorted(tensors, key=lambda x: -x.numel()):
        op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
        works.append(dist.all_reduce(t, op=op, group=group, async_op=True))
    for w in works:
        w.wait()
    if not avg:
        for t in tensors:
            t.mul_(1.0 / world)


class GradAllReduce:
    """Averages a RAU ctx's three flat gradient buffers across ranks.

    Collectives are enqueued relative to the ctx's own HIP stream (wrapped as a
    torch ExternalStream), so no host synchronisation is needed between
    rau_backward, the all-reduce and rau_noise_clip_adam.
    """

    def __init__(self, rau, group=None):
        self.rau = rau
        self.group = group
        dev = rau.cfg.device_id
        self.stream = torch.cuda.ExternalStream(rau.stream(), device=torch.device("cuda", dev))
        self.grads = []
        for g in ("mult", "rnn", "embed"):   # mult finishes first in backward
            _, gp, n = rau.device_pointers(g)
            self.grads.append(device_view(gp, n, dev))

    def __call__(self):
        with torch.cuda.stream(self.stream):
            allreduce_average(self.grads, self.group)
