"""One-off soak for stale state: ONE context runs a random sequence of steps -- train / evaluate, new
question lengths (the unroll length changes), new hop weights (gated hops), explicit masks or Philox masks
read back from the device, gradients zeroed or accumulated -- and every step is compared with the fp64
autograd oracle run from scratch on that step's inputs.  usage: python tools/soak_sequence.py [n_ctx] [seed0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import ref_torch
from rau_vqa_amd import synth
from rau_vqa_amd.model import RAU, Config
from tests import util
from tests.test_gpu_fuzz import draw
import oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
TOL = 1e-4
bad = 0
t0 = time.time()
for seed in range(s0, s0 + n):
    rng = np.random.default_rng(9000 + seed)
    dims = draw(rng)
    if seed % 2:
        dims["B"] = int(rng.integers(65, 100)); dims["S"] = int(rng.choice([196, 49, dims["S"]]))
    sh = util.shapes(dims)
    _, params, _ = util.make_problem(sh, seed=seed, scale=0.3)
    m = RAU(Config(**{k: getattr(sh, k) for k in ("B", "T", "V", "E", "Rq", "D", "S", "M", "A", "R", "K", "H",
                                                   "p_we", "p_rnn", "p_q", "p_x", "p_mf")}))
    m.set_params(params)
    acc = None
    try:
        for it in range(6):
            lens = rng.integers(0, dims["T"] + 1, dims["B"]).astype(np.int32)
            if lens.max() == 0: lens[0] = max(1, dims["T"] // 2)
            batch = synth.make_batch(sh.B, sh.T, sh.V, sh.D, sh.S, sh.K, seed=1000 * seed + it, lens=lens)
            hop_w = rng.choice([0.0, 1.0, float(dims["H"])], dims["H"]).astype(np.float32)
            if not hop_w.any(): hop_w[int(rng.integers(0, dims["H"]))] = 1.0
            train = bool(rng.integers(0, 3))
            masks = None
            if train:
                m.training()
                if rng.integers(0, 2):   # explicit masks
                    probs = {k: getattr(sh, "p_" + k) for k in oracle.MASK_SITES}
                    masks = synth.make_masks(oracle.mask_shapes(sh), probs, seed=77 * seed + it)
                    m.set_masks(masks)
                else:                     # device Philox masks, read back for the oracle
                    m.set_dropout_seed(500 + seed, it)
                    masks = {k: m.get_mask(k) for k in oracle.MASK_SITES}
            else:
                m.evaluate()
            m.set_batch(batch["feats"], batch["tokens"], batch["lens"], batch["labels"])
            zero = acc is None or bool(rng.integers(0, 2))
            if zero: m.zero_grads()
            m.forward(); out = m.outputs(); m.backward(hop_w); g = m.get_grads()
            ref = ref_torch.step(sh, params, batch["feats"], batch["tokens"], batch["lens"], batch["labels"],
                                 masks, hop_w)
            new = {k: ref["g_" + k].astype(np.float64) for k in ("embed", "rnn", "mult")}
            acc = new if zero else {k: acc[k] + new[k] for k in new}
            errs = {k: util.rel_err(out[k], ref[k]) for k in util.OUT_KEYS}
            for k in acc:
                errs["g_" + k] = util.rel_err(g[k], acc[k]) if np.max(np.abs(acc[k])) > 1e-12 else \
                    float(np.max(np.abs(g[k] - acc[k])))
            worst = {k: v for k, v in errs.items() if not v < TOL}
            if worst:
                raise AssertionError(f"step {it} train={train} zero={zero} hop_w={hop_w}: {worst}")
    except Exception as e:
        bad += 1
        print("FAIL seed", seed, dims, str(e)[:500], flush=True)
    m.close()
    if (seed - s0) % 5 == 4: print(f"{seed - s0 + 1} contexts, {bad} failures, {time.time() - t0:.0f}s", flush=True)
print("done", n, "contexts", bad, "failures")
sys.exit(1 if bad else 0)
