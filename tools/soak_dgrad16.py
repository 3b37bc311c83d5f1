"""Soak of the bf16 mode on shapes dgrad16.hip takes (14x14 maps, M % 128 == 0, A % 32 == 0): random
sizes against the rounding-emulating oracle (tests/test_gpu_bf16.run).
usage: python tools/soak_dgrad16.py [n] [first_seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_gpu_bf16 import run
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
t0 = time.time()
for seed in range(s0, s0 + n):
    rng = np.random.default_rng(7000 + seed)
    dims = dict(B=int(rng.integers(1, 90)), T=int(rng.integers(1, 6)), V=50, E=int(rng.choice([8, 20, 64])),
                Rq=int(rng.choice([16, 32, 64])), D=int(rng.choice([32, 64, 256])), S=196,
                M=int(rng.choice([128, 256, 384])), A=int(rng.choice([32, 64, 96, 256])),
                R=int(rng.choice([16, 32, 64])), K=int(rng.choice([12, 40, 200])), H=int(rng.integers(1, 4)))
    try:
        run(dims, 0.3, mode="train" if seed % 3 else "eval")
    except Exception as e:
        bad += 1
        print("FAIL seed", seed, dims, str(e)[:400], flush=True)
    if (seed - s0) % 5 == 4: print(f"{seed - s0 + 1} cases, {bad} failures, {time.time() - t0:.0f}s", flush=True)
print("done", n, "cases", bad, "failures")
sys.exit(1 if bad else 0)
