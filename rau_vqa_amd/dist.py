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

    The `mult` bucket (the largest) is final before the encoder BPTT has run
    (rau_wait_grads), so its all-reduce is issued on a side stream underneath the rest
    of the backward pass; `rnn` and `embed` follow when rau_backward's last kernel is
    done.  Everything is ordered with stream events -- no host synchronisation between
    rau_backward, the collectives and rau_noise_clip_adam; the ctx stream waits for the
    reduced gradients before whatever is enqueued next (the update).
    """

    def __init__(self, rau, group=None):
        self.rau = rau
        self.group = group
        dev = torch.device("cuda", rau.cfg.device_id)
        self.stream = torch.cuda.ExternalStream(rau.stream(), device=dev)
        self.comm = torch.cuda.Stream(device=dev)
        self.grads = {}
        for g in ("mult", "rnn", "embed"):
            _, gp, n = rau.device_pointers(g)
            self.grads[g] = device_view(gp, n, rau.cfg.device_id)

    def __call__(self):
        world = dist.get_world_size(self.group)
        if world == 1:
            return
        avg = dist.get_backend(self.group) == "nccl"   # RCCL has AVG; gloo only SUM
        op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
        works = []
        for names in (("mult",), ("rnn", "embed")):
            self.rau.wait_grads(names[0], self.comm.cuda_stream)
            with torch.cuda.stream(self.comm):
                for g in names:
                    works.append(dist.all_reduce(self.grads[g], op=op, group=self.group,
                                                 async_op=True))
        with torch.cuda.stream(self.comm):
            for w in works:
                w.wait()
            if not avg:
                for t in self.grads.values():
                    t.mul_(1.0 / world)
        self.stream.wait_stream(self.comm)


class NativeGradAllReduce:
    """Same exchange through librau's own RCCL binding (rau_comm_init / rau_allreduce_grads):
    what a host without torch.distributed (the LuaJIT shim) uses.  Here the communicator id
    travels over the already initialised torch.distributed group; any host channel works."""

    def __init__(self, rau, group=None):
        self.rau = rau
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [rau.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        rau.comm_init(world, rank, box[0])

    def __call__(self):
        self.rau.allreduce_grads()


def reduce_hop_stats(losses, argmax, labels, group=None):
    """Global per-hop loss and train accuracy from per-rank values (reference SS:491-492, 518:
    the per-hop correct counts and criterion outputs feval logs).  `losses` [H] are means over
    the LOCAL batch, `argmax` [H,B_local] 1-based answers, `labels` [B_local].  Sums of
    (loss * B_local, correct, B_local) are all-reduced; returns (loss [H], accuracy [H]) over
    the global batch.  Works without a process group (single rank)."""
    import numpy as np
    H, B = argmax.shape
    v = np.concatenate([np.asarray(losses, np.float64) * B,
                        (argmax == np.asarray(labels)[None, :]).sum(1).astype(np.float64),
                        [float(B)]])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        t = torch.from_numpy(v)
        if dist.get_backend(group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        v = t.cpu().numpy()
    n = v[-1]
    return v[:H] / n, v[H:2 * H] / n
