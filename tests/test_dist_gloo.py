"""N > 1 path on CPU: two gloo ranks, each with half of the batch.

Checks the data-parallel contract of rau_vqa_amd/dist.py: per-rank gradients
computed with 1/B_local cross-entropy scaling, ALL-REDUCE-AVERAGED over ranks,
equal the single-process gradients on the whole batch (the CE mean is the only
cross-sample coupling, SS:310).  Gradients come from the CPU oracle here (no GPU
in this container); on the GPU the same allreduce_average runs over RCCL on
zero-copy views of librau's gradient buffers."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import oracle
    from rau_vqa_amd.dist import allreduce_average, shard_batch
    from tests import util
    torch.set_num_threads(1)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = util.shapes(util.SMALL)           # global batch 8
    batch, params, _ = util.make_problem(sh, dtype=np.float64, scale=0.5)
    local = shard_batch(batch, rank, world)
    shl = util.shapes(util.SMALL, B=sh.B // world)
    out = oracle.step(shl, params, local["feats"], local["tokens"], local["lens"],
                      local["labels"], None, dtype=np.float64)
    ts = [torch.from_numpy(out[k]) for k in ("g_mult", "g_rnn", "g_embed")]
    allreduce_average(ts)
    loss = torch.from_numpy(out["losses"].copy())
    allreduce_average([loss])
    from rau_vqa_amd.dist import reduce_hop_stats
    gl, ga = reduce_hop_stats(out["losses"], out["argmax"], local["labels"])
    if rank == 0:
        q.put({"g_mult": ts[0].numpy(), "g_rnn": ts[1].numpy(), "g_embed": ts[2].numpy(),
               "losses": loss.numpy(), "hop_loss": gl, "hop_acc": ga})
    dist.destroy_process_group()


def test_two_rank_average_equals_full_batch():
    import torch.multiprocessing as mp
    import oracle
    from tests import util
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sh = util.shapes(util.SMALL)
    batch, params, _ = util.make_problem(sh, dtype=np.float64, scale=0.5)
    ref = oracle.step(sh, params, batch["feats"], batch["tokens"], batch["lens"],
                      batch["labels"], None, dtype=np.float64)
    for k in ("g_mult", "g_rnn", "g_embed", "losses"):
        assert np.allclose(got[k], ref[k], rtol=1e-10, atol=1e-13), k
    # per-hop loss sums and correct-counts reduced over the ranks (SS:491-492)
    assert np.allclose(got["hop_loss"], ref["losses"], rtol=1e-10)
    acc = (ref["argmax"] == batch["labels"][None, :]).mean(1)
    assert np.allclose(got["hop_acc"], acc)
