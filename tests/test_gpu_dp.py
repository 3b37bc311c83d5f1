"""Data-parallel path on the GPU box: two ranks share the one GPU (gloo moves the
device tensors), each with its own rau_ctx on a half batch; after GradAllReduce
the zero-copy gradient views must equal the single-ctx full-batch gradients.
RCCL itself needs >= 2 GPUs and is exercised by the driver's scaling run."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIMS = dict(B=8, T=6, V=50, E=8, Rq=16, D=24, S=12, M=40, A=20, R=16, K=12, H=2)


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
    from rau_vqa_amd.dist import GradAllReduce, shard_batch
    from rau_vqa_amd.model import RAU, Config
    from tests import util
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = util.shapes(DIMS)
    batch, params, _ = util.make_problem(sh, scale=0.5)
    local = shard_batch(batch, rank, world)
    m = RAU(Config(**dict(DIMS, B=DIMS["B"] // world)))
    m.set_params(params)
    m.evaluate()
    m.set_batch(**local)
    m.zero_grads()
    m.forward()
    m.backward(np.full(sh.H, float(sh.H), np.float32))
    red = GradAllReduce(m)
    red()
    torch.cuda.synchronize()
    m.sync()
    # the torch views alias librau's buffers: read back through the C ABI
    g = m.get_grads()
    if rank == 0:
        q.put({k: v.copy() for k, v in g.items()})
    m.close()
    dist.destroy_process_group()


def test_two_ranks_one_gpu_average_equals_full_batch():
    import torch.multiprocessing as mp
    from rau_vqa_amd.model import RAU, Config
    from tests import util
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    sh = util.shapes(DIMS)
    batch, params, _ = util.make_problem(sh, scale=0.5)
    m = RAU(Config(**DIMS))
    m.set_params(params)
    m.evaluate()
    m.set_batch(**batch)
    m.zero_grads()
    m.forward()
    m.backward(np.full(sh.H, float(sh.H), np.float32))
    ref = m.get_grads()
    m.close()
    for k in ref:
        assert util.rel_err(got[k], ref[k]) < 1e-5, k


def test_device_view_aliases_librau_buffers():
    import torch
    from rau_vqa_amd.dist import device_view
    from rau_vqa_amd.model import RAU, Config
    m = RAU(Config(**DIMS))
    m.init_uniform(seed=5)
    w, g, n = m.device_pointers("rnn")
    t = device_view(w, n, 0)
    assert t.shape == (n,) and t.dtype == torch.float32
    assert np.array_equal(t.cpu().numpy(), m.get_params()["rnn"])
    gt = device_view(g, n, 0)
    gt.fill_(0.25)                       # a write through torch is visible to librau
    torch.cuda.synchronize()
    assert np.all(m.get_grads()["rnn"] == 0.25)
    m.close()


def test_native_rccl_binding_single_rank():
    """rau_comm_* / rau_allreduce_grads (RCCL bound inside librau): with one rank the average is
    the identity, which exercises id creation, communicator init, the three collectives on the
    side stream, the event ordering against rau_backward and the update that follows.  More
    ranks need more GPUs (RCCL refuses two ranks on one device): the driver's scaling run."""
    from rau_vqa_amd.model import RAU, Config
    from tests import util
    sh = util.shapes(DIMS)
    batch, params, _ = util.make_problem(sh, scale=0.5)
    m = RAU(Config(**DIMS))
    m.set_params(params)
    m.evaluate()
    m.set_batch(**batch)
    hop_w = np.full(sh.H, float(sh.H), np.float32)
    m.zero_grads(); m.forward(); m.backward(hop_w)
    ref = m.get_grads()
    uid = RAU.comm_unique_id()
    assert len(uid) == 128
    m.comm_init(1, 0, uid)
    m.zero_grads(); m.forward(); m.backward(hop_w)
    m.allreduce_grads()
    got = m.get_grads()                          # ordered on the ctx stream behind the collectives
    norms = m.update(step_t=0, eta=0.0)          # the update that follows in a training step
    assert np.all(np.isfinite(norms))
    m.comm_destroy()
    with pytest.raises(Exception, match="rau_comm_init"):
        m.allreduce_grads()
    m.close()
    for k in ref:
        np.testing.assert_array_equal(got[k], ref[k])


def test_bench_flow_with_two_ranks_over_gloo():
    """bench.py's N > 1 path end to end (rendezvous, per-step gradient exchange, barriers of
    the timed region and of rank 0's extra measurements, max-over-ranks timing, one JSON line
    from rank 0) with two ranks sharing the one GPU over gloo: RCCL itself needs one GPU per
    rank, everything around it is the same code."""
    import json
    import subprocess
    env = dict(os.environ, RAU_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "32"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 64 and d["value"] > 0
    assert d["scaling"] == "weak" and "roofline" in d


def test_bench_self_launch_strong_scaling_two_ranks():
    """`python bench.py --gpus 2` with no rendezvous in the environment -- the form the driver
    uses -- starts its own ranks as a child torch.distributed.run and relays rank 0's line.
    Strong-scaling flavour: BASELINE.json configs[3] (Ours_MS weights, a GLOBAL batch split over
    the ranks), plus the cross-rank per-hop loss / accuracy reduction."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(RAU_DIST_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--global-batch", "64", "--variant", "MS", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["global_batch"] == 64 and d["config"]["batch_per_gpu"] == 32
    assert "2 ranks" in d["config"]["collective"]
    assert len(d["hop_loss_global"]) == 8 and all(np.isfinite(d["hop_loss_global"]))
    assert all(0.0 <= a <= 1.0 for a in d["hop_train_acc_global"])
    for k in ("mfma_frac", "hbm_frac", "kernel", "traffic"):
        assert k in d["roofline"]
