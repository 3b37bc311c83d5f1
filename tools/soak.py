"""One-off soak: the fuzz test's shape generator over many more seeds, plus larger batches
(both attention families, both launch-group policies).  usage: python tools/soak.py [n] [first_seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import util
from tests.test_gpu_parity import check
from tests.test_gpu_fuzz import draw
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0
t0 = time.time()
for seed in range(s0, s0 + n):
    rng = np.random.default_rng(1000 + seed)
    dims = draw(rng)
    if seed % 2:   # every other case above the 64-sample switch
        dims["B"] = int(rng.integers(65, 150))
        dims["S"] = int(rng.choice([49, 196, int(rng.integers(1, 60))]))
    sh = util.shapes(dims)
    lens = rng.integers(0, dims["T"] + 1, dims["B"]).astype(np.int32)
    if lens.max() == 0: lens[0] = dims["T"]
    hop_w = rng.choice([0.0, 1.0, float(dims["H"])], dims["H"]).astype(np.float32)
    if not hop_w.any(): hop_w[0] = 1.0
    mode = "train" if seed % 3 else "eval"
    try:
        check(sh, seed=seed, lens=lens, mode=mode, hop_w=hop_w, scale=0.3, torch_oracle=True)
    except Exception as e:
        bad += 1
        print("FAIL seed", seed, dims, mode, str(e)[:300], flush=True)
    if (seed - s0) % 10 == 9: print(f"{seed - s0 + 1} cases, {bad} failures, {time.time() - t0:.0f}s", flush=True)
print("done", n, "cases", bad, "failures")
sys.exit(1 if bad else 0)
