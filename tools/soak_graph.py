"""One-off soak: hipGraph replay == eager calls bitwise on random shapes and random step sequences
(new lengths, gated hops, train/eval, dtypes).  usage: python tools/soak_graph.py [n] [seed0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rau_vqa_amd import synth
from rau_vqa_amd.model import RAU, Config
from tests import util
from tests.test_gpu_fuzz import draw
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
t0 = time.time()
for seed in range(s0, s0 + n):
    rng = np.random.default_rng(11000 + seed)
    dims = draw(rng)
    if seed % 2:
        dims["B"] = int(rng.integers(65, 100)); dims["S"] = int(rng.choice([196, 49, dims["S"]]))
    dtype = ["f32", "bf16", "f32"][seed % 3]
    sh = util.shapes(dims)
    _, params, _ = util.make_problem(sh, seed=seed, scale=0.3)
    ms = []
    for _ in range(2):
        m = RAU(Config(**{k: getattr(sh, k) for k in ("B", "T", "V", "E", "Rq", "D", "S", "M", "A", "R", "K", "H")},
                       dtype=dtype))
        m.set_params(params); ms.append(m)
    try:
        for it in range(5):
            lens = rng.integers(0, dims["T"] + 1, dims["B"]).astype(np.int32)
            if lens.max() == 0: lens[0] = max(1, dims["T"] // 2)
            batch = synth.make_batch(sh.B, sh.T, sh.V, sh.D, sh.S, sh.K, seed=1000 * seed + it, lens=lens)
            hop_w = rng.choice([0.0, 1.0, float(dims["H"])], dims["H"]).astype(np.float32)
            if not hop_w.any(): hop_w[0] = 1.0
            train = bool(rng.integers(0, 3))
            outs = []
            for m, use_graph in ((ms[0], False), (ms[1], True)):
                m.training() if train else m.evaluate()
                m.set_batch(batch["feats"], batch["tokens"], batch["lens"], batch["labels"])
                m.set_dropout_seed(11 + seed, it)
                if use_graph:
                    m.graph_step(hop_w)
                else:
                    m.zero_grads(); m.forward(); m.backward(hop_w)
                g = m.get_grads()
                outs.append((m.losses(), m.logits(), g["embed"], g["rnn"], g["mult"]))
            for i, (a, b) in enumerate(zip(*outs)):
                if not np.array_equal(a, b):
                    raise AssertionError(f"step {it} train={train} output {i} differs (max {np.max(np.abs(a - b))})")
    except Exception as e:
        bad += 1
        print("FAIL seed", seed, dtype, dims, str(e)[:400], flush=True)
    for m in ms: m.close()
    if (seed - s0) % 5 == 4: print(f"{seed - s0 + 1} cases, {bad} failures, {time.time() - t0:.0f}s", flush=True)
print("done", n, "cases", bad, "failures")
sys.exit(1 if bad else 0)
