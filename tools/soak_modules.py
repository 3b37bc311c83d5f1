"""One-off soak of the module-level calls (tests/test_gpu_fuzz.py's feval case over more seeds).
usage: python tools/soak_modules.py [n] [first_seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_fuzz import test_random_shapes_module_level_feval as case
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 10
bad = 0
t0 = time.time()
for seed in range(s0, s0 + n):
    try:
        case(seed)
    except Exception as e:
        bad += 1
        print("FAIL seed", seed, type(e).__name__, str(e)[:400], flush=True)
    if (seed - s0) % 10 == 9: print(f"{seed - s0 + 1} cases, {bad} failures, {time.time() - t0:.0f}s", flush=True)
print("done", n, "cases", bad, "failures")
sys.exit(1 if bad else 0)
