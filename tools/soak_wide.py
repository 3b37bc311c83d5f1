"""One-off soak with wider layers than the fuzz test draws (up to 600 channels, 2048-wide feature maps,
14x14 and 7x7 and odd maps), small batches on both sides of 64.  usage: python tools/soak_wide.py [n] [seed0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import util
from tests.test_gpu_parity import check
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
t0 = time.time()
for seed in range(s0, s0 + n):
    rng = np.random.default_rng(7000 + seed)
    m4 = lambda lo, hi: int(rng.integers(lo, hi + 1)) * 4
    dims = dict(B=int(rng.choice([1, 2, 5, 17, 33, 64, 65, 66, 97, 130])), T=int(rng.integers(1, 12)),
                V=int(rng.integers(5, 300)), E=m4(1, 60), Rq=m4(1, 140), D=int(rng.choice([m4(1, 150), 512, 2048])),
                S=int(rng.choice([196, 196, 49, 100, int(rng.integers(1, 220))])), M=m4(1, 150), A=m4(1, 80),
                R=m4(1, 140), K=m4(1, 275), H=int(rng.integers(1, 5)))
    lens = rng.integers(0, dims["T"] + 1, dims["B"]).astype(np.int32)
    if lens.max() == 0: lens[0] = dims["T"]
    hop_w = rng.choice([0.0, 1.0, float(dims["H"])], dims["H"]).astype(np.float32)
    if not hop_w.any(): hop_w[0] = 1.0
    try:
        check(util.shapes(dims), seed=seed, lens=lens, mode="train" if seed % 3 else "eval", hop_w=hop_w,
              scale=None, torch_oracle=True)
    except Exception as e:
        bad += 1
        print("FAIL seed", seed, dims, str(e)[:500], flush=True)
    if (seed - s0) % 10 == 9: print(f"{seed - s0 + 1} cases, {bad} failures, {time.time() - t0:.0f}s", flush=True)
print("done", n, "cases", bad, "failures")
sys.exit(1 if bad else 0)
