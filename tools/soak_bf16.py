"""One-off soak of the bf16-operand mode: random shapes against the rounding-emulating oracle
(tests/test_gpu_bf16.run).  usage: python tools/soak_bf16.py [n] [first_seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_gpu_bf16 import run
from tests.test_gpu_fuzz import draw
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 500
bad = 0
t0 = time.time()
for seed in range(s0, s0 + n):
    rng = np.random.default_rng(1000 + seed)
    dims = draw(rng)
    dims["S"] = int(rng.choice([196, 196, 49, int(rng.integers(1, 60))]))
    if seed % 2: dims["B"] = int(rng.integers(65, 120))
    dims["H"] = min(dims["H"], 3)
    try:
        run(dims, 0.3, mode="train" if seed % 3 else "eval")
    except Exception as e:
        bad += 1
        print("FAIL seed", seed, dims, str(e)[:400], flush=True)
    if (seed - s0) % 10 == 9: print(f"{seed - s0 + 1} cases, {bad} failures, {time.time() - t0:.0f}s", flush=True)
print("done", n, "cases", bad, "failures")
sys.exit(1 if bad else 0)
