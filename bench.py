#!/usr/bin/env python3
"""bench.py -- QA-pairs/s of the RAU forward+backward on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): Ours_SS 8-step RAU, batch 256 PER GPU,
14x14x512 features, 26-token questions (all full length), fp32, training mode
(dropout active, Philox masks regenerated each step), SS hop weights (x nHop).
One "step" = zero grads + rau_forward + rau_backward (every parameter gradient
in the three flat buffers) [+ RCCL average of those buffers when N > 1].  Inputs
are resident in HBM before the timed region.  Noise/clip/Adam is NOT in the
metric; it is timed separately as step_incl_update.

N > 1: one rank per GPU over RCCL.  `python bench.py --gpus N` launches its own ranks
(`python -m torch.distributed.run` as a CHILD process, started before this process has
imported torch or touched the GPU) and relays rank 0's JSON line; started under
torch.distributed.run (WORLD_SIZE set) it is a rank.  Default = weak scaling (256 per GPU);
`--global-batch G` = strong scaling (G/N per GPU: BASELINE.json configs[3] = 512 with
--variant MS, configs[4] = 1024 with --variant Full --D 2048).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md, f32-input MFMA
MFMA_BF16_PEAK_TFLOPS = 2500.0 # dense bf16 MFMA
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md, HBM3E (about 6300 achievable)
# kernel classes that can be "the dominant kernel" (bulk MFMA GEMMs with FLOP and byte counts)
BULK_CLASSES = ("conv_embed_fwd", "conv_embed_wgrad", "conv_att_pre", "conv_att_dgrad",
                "conv_att_wgrad")


def usable_cores():
    """CPU share of this process: cgroup quota if set, else the affinity mask,
    capped at 16 (the GPU box's per-GPU share; more threads only oversubscribe)."""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_baseline(cfgd, budget_s=15.0):
    """Reference graph restated on PyTorch-CPU (oracle/ref_torch.py), fp32, all cores,
    on a bounded sample of the same workload: batch 16 of the same shapes."""
    import torch
    import oracle
    from oracle import ref_torch
    from rau_vqa_amd import synth
    d = dict(cfgd)
    d["B"] = 16
    sh = oracle.Shapes(**d)
    ne, nr, nm = oracle.group_sizes(sh)
    batch = synth.make_batch(sh.B, sh.T, sh.V, sh.D, sh.S, sh.K, lens="full")
    params = synth.make_params({"embed": ne, "rnn": nr, "mult": nm})
    masks = synth.make_masks(oracle.mask_shapes(sh), {k: 0.5 for k in oracle.MASK_SITES})
    cores = usable_cores()
    torch.set_num_threads(cores)
    run = lambda: ref_torch.step(sh, params, batch["feats"], batch["tokens"], batch["lens"],
                                 batch["labels"], masks, backward=True, dtype=torch.float32)
    run()  # warm-up
    t0 = time.time()
    n = 0
    while True:
        run()
        n += 1
        if time.time() - t0 > budget_s or n >= 200:
            break
    dt = time.time() - t0
    return {"value": sh.B * n / dt, "unit": "QA-pairs/s", "cores": cores, "kind": "port",
            "sample": f"{n} fwd+bwd steps at batch {sh.B} of the same shapes "
                      f"(PyTorch-CPU fp32 restatement of the reference graph, {dt:.1f}s)"}


def cpu_baseline_cxx(cfgd, budget_s=12.0):
    """CPU-B: the C++ oracle (oracle/rau_cpu.cc, OpenMP, f32) on the same bounded
    sample -- a sanity floor next to the PyTorch-CPU restatement, not a tuned baseline."""
    import oracle
    from rau_vqa_amd import synth
    d = dict(cfgd)
    d["B"] = 16
    sh = oracle.Shapes(**d)
    ne, nr, nm = oracle.group_sizes(sh)
    batch = synth.make_batch(sh.B, sh.T, sh.V, sh.D, sh.S, sh.K, lens="full")
    params = synth.make_params({"embed": ne, "rnn": nr, "mult": nm})
    masks = synth.make_masks(oracle.mask_shapes(sh), {k: 0.5 for k in oracle.MASK_SITES})
    cores = usable_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)
    t0 = time.time()
    n = 0
    while True:
        oracle.step(sh, params, batch["feats"], batch["tokens"], batch["lens"], batch["labels"],
                    masks, dtype=np.float32)
        n += 1
        if time.time() - t0 > budget_s or n >= 50:
            break
    dt = time.time() - t0
    return {"value": sh.B * n / dt, "unit": "QA-pairs/s", "cores": cores, "kind": "port",
            "sample": f"{n} fwd+bwd steps at batch {sh.B} (C++ oracle, OpenMP f32, {dt:.1f}s)"}


def roofline_objects(prof, nprof, dtype, shape):
    """`roofline` (dominant bulk kernel) and `bulk_kernels` (all five conv classes) from a HIP-event
    profile of `nprof` steps (rau_prof_*: events on the stream each kernel is launched on).
    shape = (batch, D, dtype) selects the committed PMC pass the `traffic` figure is read from."""
    # dominant kernel = the bulk-GEMM class with the most measured device time in THIS run
    bulk = {k: prof[k] for k in BULK_CLASSES if k in prof and prof[k]["launches"]}
    dom_name = max(bulk, key=lambda k: bulk[k]["ms"])
    dom = bulk[dom_name]
    avg_ms = dom["ms"] / dom["launches"]
    flops_per_launch = dom["flops"] / dom["launches"]
    bytes_per_launch = dom["bytes"] / dom["launches"]
    tfl = flops_per_launch / (avg_ms * 1e-3) / 1e12
    gbs = bytes_per_launch / (avg_ms * 1e-3) / 1e9
    # HBM bytes per launch of that kernel: NOT measured by this run -- taken from the newest committed
    # PMC passes of the same command (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs,
    # FETCH_SIZE doubled per the gfx950 correction) when one exists for this kernel and shape
    traffic, traffic_src = None, None
    for rnd in ("r04", "r03", "r02", "r01"):
        tag = {(256, 512, "f32"): rnd, (256, 2048, "bf16"): rnd + "_bf16"}.get(shape)
        pmc = os.path.join(ROOT, "profiles", f"{tag}_pmc_{dom_name}.json")
        if tag and os.path.exists(pmc):
            traffic = json.load(open(pmc))["hbm_bytes_per_launch"]
            traffic_src = f"profiles/{tag}_pmc_{dom_name}.json (separate rocprofv3 --pmc passes)"
            break
    mfma_peak = MFMA_F32_PEAK_TFLOPS if dtype == "f32" else MFMA_BF16_PEAK_TFLOPS
    both = {"mfma_frac": tfl / mfma_peak, "hbm_frac": gbs / HBM_PEAK_GBS,
            "mfma_tflops": tfl, "hbm_gbs_algorithmic": gbs}
    if dtype == "f32":   # f32 MFMA-bound
        roof = {"bound": "mfma", "achieved": tfl, "peak": mfma_peak, "unit": "TFLOP/s",
                "frac": tfl / mfma_peak}
    else:                # bf16 operands: HBM-bound; algorithmic bytes = operands read
        roof = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": gbs / HBM_PEAK_GBS}
    roof.update({"traffic": traffic, "traffic_source": traffic_src, "kernel": dom_name,
                 "avg_launch_ms": avg_ms, "launches_per_step": dom["launches"] / nprof,
                 "flops_per_launch": flops_per_launch, "bytes_per_launch": bytes_per_launch, **both})
    # the same two fractions for every bulk class (which one is "dominant" can flip by run)
    bulk_kernels = {
        k: {"ms_per_step": round(v["ms"] / nprof, 4),
            "mfma_frac": round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / mfma_peak, 4),
            "hbm_frac": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        for k, v in bulk.items()}
    return {"roofline": roof, "bulk_kernels": bulk_kernels}


WORKLOADS = {("SS", 512, "f32"): "configs[1]", ("ResNet", 2048, "bf16"): "configs[2]",
             ("MS", 512, "f32"): "configs[3]", ("Full", 2048, "f32"): "configs[4]",
             ("Full", 2048, "bf16"): "configs[4] (bf16 operands)"}


# The other single-GPU workloads the default run puts on the driver's clock.
OTHER_CONFIGS = {
    "configs[2]": dict(batch=256, D=2048, dtype="bf16", variant="ResNet", steps=10),
    "configs[3] shard": dict(batch=64, D=512, dtype="f32", variant="MS", steps=20,
                             note=" -- one rank's 64-sample shard of the 512-sample global batch at 8 GPUs, "
                                  "without the gradient all-reduce"),
}


def other_config_child(name):
    """One other_configs leg in a CHILD process of its own (`bench.py --other-config NAME`, started with
    subprocess -- never an exec -- while this process idles).  Why not in this process: HIP deals streams to
    a handful of hardware queues by creation order and priority; with the headline context's streams (and
    the copy stream of its H2D leg) alive, a second context's bulk and weight-gradient streams can land on
    ONE hardware queue and serialise -- measured: configs[2] 9.13 instead of 7.96 ms/step, the 64-sample
    shard 4.42 instead of 4.16, in the same run of this file.  A failing leg is reported, not fatal: the
    headline line was measured before."""
    cmd = [sys.executable, os.path.abspath(__file__), "--other-config", name]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"error": f"no result (exit {r.returncode}): {r.stderr.strip()[-300:]}"}
    except Exception as e:   # timeout, spawn failure
        return {"error": repr(e)}


def other_config(name, base_cfgd, device_id, *, batch, D, dtype, variant, steps, warmup=3, note=""):
    """One more BASELINE.json config on the driver's clock: its own context, the
    same step (zero grads + forward + backward, train mode, Philox masks per step, inputs resident in
    HBM), timed over `steps` steps between host synchronisations, then a 3-step HIP-event profile for
    its own roofline object.  Single GPU; the headline line is not affected (it was timed before)."""
    import torch
    from rau_vqa_amd import synth
    from rau_vqa_amd.model import RAU, Config, hop_weights
    cfgd = dict(base_cfgd, B=batch, D=D)
    cfg = Config(device_id=device_id, dtype=dtype, **cfgd)
    m = RAU(cfg)
    try:
        m.init_uniform(seed=123)
        m.set_batch(**synth.make_batch(cfg.B, cfg.T, cfg.V, cfg.D, cfg.S, cfg.K, seed=123, lens="full"))
        m.training()
        hop_w = hop_weights(variant, cfg.H, 0)

        def step(i):
            m.set_dropout_seed(123, i)
            m.zero_grads()
            m.forward()
            m.backward(hop_w)

        def fence():
            m.sync()
            torch.cuda.synchronize()
        for i in range(warmup):
            step(i)
        fence()
        t0 = time.perf_counter()
        for i in range(steps):
            step(warmup + i)
        fence()
        dt = time.perf_counter() - t0
        assert np.all(np.isfinite(m.losses()))
        m.prof_reset()
        m.prof_enable(True)
        nprof = 3
        for i in range(nprof):
            step(1000 + i)
        m.sync()
        prof = m.prof()
        m.prof_enable(False)
        fmap = f"14x14x{cfg.D}"
        out = {"metric": f"QA-pairs/sec fwd+bwd, Ours_{variant} 8-step RAU, batch {cfg.B}, {fmap}",
               "value": cfg.B * steps / dt, "unit": "QA-pairs/s", "ms_per_step": dt / steps * 1e3,
               "steps": steps, "warmup": warmup, "dtype": dtype, "n_gpus": 1,
               "config": {"workload": f"Ours_{variant} 8-step RAU fwd+bwd, {fmap}, "
                                      f"{WORKLOADS.get((variant, D, dtype), 'not a BASELINE config')}{note}",
                          "batch_per_gpu": cfg.B, "T": cfg.T, "feature_map": fmap, "hops": cfg.H,
                          "hop_weights": variant,
                          "arithmetic": "f32 operands, f32 MFMA accumulate" if dtype == "f32"
                                        else "bf16-rounded operands in every conv and Linear GEMM, f32 accumulate; rest f32",
                          "launch": "eager, 3 streams"}}
        out.update(roofline_objects(prof, nprof, dtype, (batch, D, dtype)))
        gf_per_qa = {512: 3.656, 2048: 8.588}.get(D)
        if gf_per_qa:
            out["step_tflops_useful"] = gf_per_qa * cfg.B * steps / dt / 1e3
        return out
    finally:
        m.close()


def self_launch(args):
    """--gpus N > 1 without a rendezvous in the environment: start the ranks as a child
    torch.distributed.run (never an exec, and before anything here has touched the GPU),
    relay its output, exit with its code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env)
    sys.exit(r.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (weak scaling)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="strong scaling: total batch split evenly over the ranks "
                         "(configs[3]: 512 --variant MS; configs[4]: 1024 --variant Full --D 2048)")
    ap.add_argument("--D", type=int, default=512)
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32",
                    help="bf16 = BASELINE.json configs[2] mode (use with --D 2048): bf16-operand "
                         "conv GEMMs, f32 accumulate; never the default metric")
    ap.add_argument("--variant", choices=("SS", "MS", "Full", "ResNet"), default="SS",
                    help="per-hop loss weights of the four training scripts (SS:569, MS:568-570, "
                         "Full/ResNet: epoch-gated 0|1, see --epoch)")
    ap.add_argument("--epoch", type=int, default=0, help="epoch for the Full/ResNet gating vector")
    ap.add_argument("--graph", action="store_true",
                    help="replay each step as one hipGraph launch (rau_graph_step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--quick", action="store_true",
                    help="A/B sweeps: the timed steps and the 3-step kernel profile only (no update, H2D, "
                         "inference, other_configs or CPU-baseline legs)")
    ap.add_argument("--other-config", choices=sorted(OTHER_CONFIGS), default=None,
                    help="(internal) run ONE other_configs leg in this process and print its JSON object")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the other_configs legs (configs[2], the configs[3] shard) of the default run")
    args = ap.parse_args()
    if args.quick:
        args.no_cpu_baseline = args.no_other_configs = True
    if args.other_config:   # child of the default run: one leg, its own process, nothing else
        import torch  # before librau.so, so both share one HIP runtime
        base = dict(B=256, T=26, V=14000, E=200, Rq=512, D=512, S=196, M=512, A=256, R=512, K=1000, H=8)
        print(json.dumps(other_config(args.other_config, base, int(os.environ.get("LOCAL_RANK", "0")),
                                      **OTHER_CONFIGS[args.other_config])), flush=True)
        return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if args.global_batch:
        if args.global_batch % world:
            raise SystemExit(f"--global-batch {args.global_batch} not divisible by {world} ranks")
        args.batch = args.global_batch // world

    if args.graph:
        # ROCm 7.2's graph executor spreads a captured graph's branches over its own queues; with its
        # default the three-stream step replays 1.5-2x slower than the eager calls, with two queues
        # within 9-13 %.  The runtime reads this at its first HIP call, so it
        # is set here, for this process only, before torch / librau.so are loaded.
        os.environ.setdefault("DEBUG_HIP_FORCE_GRAPH_QUEUES", "2")
    import torch  # before librau.so, so both share one HIP runtime
    import torch.distributed as dist
    from rau_vqa_amd import synth
    from rau_vqa_amd.model import RAU, Config, hop_weights

    # RAU_DIST_BACKEND=gloo: rehearse the N>1 code path with several ranks on ONE GPU (RCCL
    # refuses two ranks per device); the default is RCCL, one rank per GPU
    backend = os.environ.get("RAU_DIST_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend)

    cfgd = dict(B=args.batch, T=26, V=14000, E=200, Rq=512, D=args.D, S=196, M=512, A=256,
                R=512, K=1000, H=8)
    cfg = Config(device_id=local_rank, dtype=args.dtype, **cfgd)
    m = RAU(cfg)
    m.init_uniform(seed=123)                      # uniform(-0.08, 0.08), SS:352-354
    batch = synth.make_batch(cfg.B, cfg.T, cfg.V, cfg.D, cfg.S, cfg.K, seed=123 + rank,
                             lens="full")
    m.set_batch(**batch)                          # resident in HBM from here on
    m.training()
    hop_w = hop_weights(args.variant, cfg.H, args.epoch)
    m_prof = [False]   # per-launch event profiling needs the eager path
    reducer = None
    if world > 1:
        from rau_vqa_amd.dist import GradAllReduce, NativeGradAllReduce
        # RAU_DP=native: the C ABI's own RCCL binding (rau_allreduce_grads) instead of
        # torch.distributed collectives on views of the same buffers
        reducer = (NativeGradAllReduce if os.environ.get("RAU_DP") == "native" else GradAllReduce)(m)

    def step(i):
        # every rank draws its OWN dropout masks for its shard (the noise key of the update
        # stays identical on all ranks: same update everywhere)
        m.set_dropout_seed(123 + 7919 * rank, i)
        if args.graph and not m_prof[0]:
            m.graph_step(hop_w)
        else:
            m.zero_grads()
            m.forward()
            m.backward(hop_w)
        if reducer is not None:
            reducer()

    def fence():
        m.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()


    def h2d_leg():
        """Fresh inputs every step (what SS:434-439 does): the batch goes through the two pinned slots
        of the asynchronous upload, batch i+1 uploading while step i runs.  Not the metric (its timed
        region starts with the inputs resident); reported beside it.  Every rank runs it (the steps
        carry the gradient collective)."""
        if args.graph:
            return None
        for sl in (0, 1):
            v = m.batch_slot(sl)
            b2 = batch if sl == 0 else synth.make_batch(cfg.B, cfg.T, cfg.V, cfg.D, cfg.S, cfg.K,
                                                        seed=977 + rank, lens="full")
            for k in ("feats", "tokens", "lens", "labels"):
                v[k][...] = np.asarray(b2[k]).reshape(v[k].shape)
        m.set_batch_async(0)
        for i in range(3):
            m.use_batch(i & 1)
            m.set_batch_async((i + 1) & 1)
            step(3000 + i)
        fence()
        t3 = time.perf_counter()
        nh2d = 10
        for i in range(nh2d):
            m.use_batch((i + 1) & 1)
            m.set_batch_async(i & 1)          # next batch: B*D*S*4 bytes over PCIe under this step
            step(3100 + i)
        fence()
        ms = (time.perf_counter() - t3) / nh2d * 1e3
        m.set_batch(**batch)
        return ms

    for i in range(args.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    losses = m.losses()
    assert os.environ.get("RAU_DEV_SKIP") or np.all(np.isfinite(losses)), losses

    extra = {}
    if rank == 0:
        # per-kernel-class device time (HIP events on the ctx stream, outside the timed region)
        m.prof_reset()
        m.prof_enable(True)
        m_prof[0] = True
        nprof = 3
        for i in range(nprof):
            step(1000 + i)
        m.sync()
        prof = m.prof()
        m.prof_enable(False)
        m_prof[0] = False
        extra.update(roofline_objects(prof, nprof, args.dtype, (args.batch, args.D, args.dtype)))
        tot = sum(v["ms"] for v in prof.values())
        extra["kernel_classes_ms_per_step"] = {
            k: round(v["ms"] / nprof, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
        extra["device_ms_per_step_profiled"] = tot / nprof
        # whole-step useful FLOPs: 3.656 GFLOP/QA at D=512, 8.588 at D=2048
        gf_per_qa = {512: 3.656, 2048: 8.588}.get(args.D)
        if gf_per_qa:
            extra["step_tflops_useful"] = gf_per_qa * cfg.B * world * args.steps / dt / 1e3
        if not args.quick:
            # update cost, reported separately (not part of the metric)
            fence()
            t1 = time.perf_counter()
            for i in range(5):
                step(2000 + i)
                m.update(step_t=i)
            fence()
            extra["step_incl_update_ms"] = (time.perf_counter() - t1) / 5 * 1e3
            h2d_ms = h2d_leg()
            if h2d_ms is not None:
                extra["step_incl_h2d_ms"] = h2d_ms
                extra["h2d_bytes_per_step"] = int(cfg.B * cfg.D * cfg.S * 4)
            # inference (predict_result, SS:633-705: evaluate mode, forward only, i_embed / ifeatproj
            # hoisted out of the hop loop), reported separately
            m.evaluate()
            for i in range(2):
                m.forward()
            fence()
            t2 = time.perf_counter()
            for i in range(10):
                m.forward()
            fence()
            inf_s = (time.perf_counter() - t2) / 10
            extra["inference_qa_per_s"] = cfg.B / inf_s
            # Roofline of the evaluate-mode forward (i_embed / ifeatproj once, encoder, H hops).  Per QA pair:
            # FLOPs = convs + encoder + hops (Linears + attention), bytes = X read, I and P written once and
            # read once per hop (f32).  At the reference's sizes the MFMA floor (0.41 GFLOP/QA / 157 TFLOP/s)
            # is 3.5x the HBM floor, so the bound is the matrix pipe, not HBM (SURVEY 8f next-2 assumed HBM).
            def inference_roofline(c, secs):
                fl = (2.0 * c.S * (c.M * c.D + c.A * c.M)
                      + c.T * (2.0 * 4 * c.Rq * c.E + 3 * 2.0 * 4 * c.Rq * c.Rq)
                      + 2.0 * c.Q * c.M
                      + c.H * (2.0 * c.R * (c.M + c.S + 4 * c.R) + 2.0 * c.M * c.A + 2.0 * c.S * c.M
                               + 2.0 * c.M * 4 * c.R + 2.0 * c.R * c.M + 2.0 * c.M * c.K
                               + 2.0 * c.S * (c.A + c.M))) * c.B
                by = 4.0 * c.B * c.S * (c.D + c.M + c.A + c.H * (c.M + c.A))
                t_mfma, t_hbm = fl / (MFMA_F32_PEAK_TFLOPS * 1e12), by / (HBM_PEAK_GBS * 1e9)
                return {"bound": "mfma" if t_mfma >= t_hbm else "hbm",
                        "achieved": fl / secs / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": fl / secs / 1e12 / MFMA_F32_PEAK_TFLOPS,
                        "flops_per_batch": fl, "algorithmic_bytes_per_batch": by,
                        "hbm_gbs_algorithmic": by / secs / 1e9, "hbm_frac": by / secs / 1e9 / HBM_PEAK_GBS}
            extra["inference"] = {"qa_per_s": cfg.B / inf_s, "batch": cfg.B, "ms_per_batch": inf_s * 1e3,
                                  "roofline": inference_roofline(cfg, inf_s),
                                  "note": "whole evaluate-mode forward, wall time over 10 batches; at this batch the "
                                          "27 encoder steps of two launches and 8 hops of seven (~120 dependent launches) bound it"}
            m.training()
            # the same forward at a serving batch (the recurrences' launch chain does not grow with the batch)
            if world == 1 and args.dtype == "f32" and not args.graph and args.batch == 256:
                from dataclasses import replace
                cfg_b = replace(cfg, B=1024)
                mb = RAU(cfg_b)
                mb.set_params(m.get_params())
                mb.set_batch(**synth.make_batch(cfg_b.B, cfg.T, cfg.V, cfg.D, cfg.S, cfg.K, seed=5, lens="full"))
                mb.evaluate()
                for i in range(2):
                    mb.forward()
                mb.sync()
                t4 = time.perf_counter()
                for i in range(5):
                    mb.forward()
                mb.sync()
                inf_b = (time.perf_counter() - t4) / 5
                mb.close()
                extra["inference_b1024"] = {"qa_per_s": cfg_b.B / inf_b, "batch": cfg_b.B,
                                            "ms_per_batch": inf_b * 1e3,
                                            "roofline": inference_roofline(cfg_b, inf_b)}
            m.training()
        # the other single-GPU workloads of BASELINE.json on the same clock (the headline's timed region
        # is over): configs[2] and the 64-sample strong-scaling shard of configs[3] (512 samples / 8 GPUs)
        headline_run = ((args.dtype, args.D, args.variant, args.batch) == ("f32", 512, "SS", 256)
                        and world == 1 and not args.graph and not args.global_batch)
        if headline_run and not args.no_other_configs:
            extra["other_configs"] = {name: other_config_child(name) for name in OTHER_CONFIGS}
        if world == 1 and not args.no_cpu_baseline:
            extra["cpu_baseline"] = cpu_baseline(cfgd, budget_s=12.0)
            extra["cpu_baseline_cxx"] = cpu_baseline_cxx(cfgd, budget_s=8.0)
    elif world > 1:
        # keep ranks in lock-step with rank 0's extra (collective-carrying) steps
        for i in range(3):
            step(1000 + i)
        fence()
    if rank != 0 and world > 1 and not args.quick:
        for i in range(5):
            step(2000 + i)
            m.update(step_t=i)
        fence()
        h2d_leg()
        m.evaluate()
        for i in range(2):
            m.forward()
        fence()
        for i in range(10):
            m.forward()
        fence()
        m.training()

    # global per-hop loss and train accuracy (SS:491-492, 518): per-rank sums reduced over the
    # ranks (SURVEY 8e "also reduce"); host-side logging, outside the timed region
    from rau_vqa_amd.dist import reduce_hop_stats
    hop_loss, hop_acc = reduce_hop_stats(m.losses(), m.argmax(), batch["labels"])

    if rank == 0:
        qa = cfg.B * world * args.steps / dt
        headline = (args.dtype, args.D, args.variant, cfg.B) == ("f32", 512, "SS", 256)
        fmap = f"14x14x{cfg.D}"
        metric = (f"QA-pairs/sec fwd+bwd, Ours_{args.variant} 8-step RAU, batch {cfg.B}"
                  f"{' per GPU' if world > 1 else ''}, {fmap}")
        if headline and world == 1:
            metric = "QA-pairs/sec fwd+bwd, Ours_SS 8-step RAU, batch 256, 14x14x512"
        cfgs = WORKLOADS
        line = {"metric": metric,
                "value": qa, "unit": "QA-pairs/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                "higher_is_better": True, "scaling": "strong" if args.global_batch else "weak",
                "vs_baseline": None,
                "dtype": args.dtype,
                "data": "synthetic",
                "config": {"workload": f"Ours_{args.variant} 8-step RAU fwd+bwd, {fmap}, "
                                       f"{cfgs.get((args.variant, args.D, args.dtype), 'not a BASELINE config')}",
                           "batch_per_gpu": cfg.B, "global_batch": cfg.B * world, "T": cfg.T,
                           "feature_map": fmap, "hops": cfg.H,
                           "parallelism": f"dp{world}",
                           "collective": (f"{backend} all-reduce(avg) of 3 flat grad buckets, "
                                          f"{world} ranks" if world > 1 else "none"),
                           "hop_weights": {"SS": "SS (x nHop)", "MS": "MS (x 1)"}.get(
                               args.variant, f"{args.variant} gating, epoch {args.epoch}"),
                           "arithmetic": "f32 operands, f32 MFMA accumulate" if args.dtype == "f32"
                                         else "bf16-rounded operands in every conv and Linear GEMM, f32 accumulate; rest f32",
                           "dropout": "train mode, Philox masks per step and rank",
                           "launch": "hipGraph replay" if args.graph else "eager, 3 streams"},
                "hop_loss_global": [round(float(x), 5) for x in hop_loss],
                "hop_train_acc_global": [round(float(x), 5) for x in hop_acc]}
        line.update(extra)
        print(json.dumps(line), flush=True)
    m.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
