"""One-off: the model's real dimensions at awkward batch sizes (edge tiles beside interior ones, both
sides of the 64-sample switch), f32 against the fp64 autograd oracle and bf16 against the emulating one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import util
from tests.test_gpu_parity import check
from tests.test_gpu_bf16 import run as run_bf16
FULL = dict(T=26, V=14000, E=200, Rq=512, D=512, S=196, M=512, A=256, R=512, K=1000, H=8)
bad = 0
for B in (1, 3, 63, 65, 129, 200):
    t = time.time()
    try:
        check(util.shapes(dict(FULL, B=B)), scale=None, torch_oracle=True)
        print("f32 B", B, "ok", f"{time.time() - t:.0f}s", flush=True)
    except Exception as e:
        bad += 1; print("FAIL f32 B", B, str(e)[:500], flush=True)
for B, D in ((5, 2048), (67, 2048), (130, 512)):
    t = time.time()
    try:
        run_bf16(dict(FULL, B=B, D=D, H=3), None)
        print("bf16 B", B, "D", D, "ok", f"{time.time() - t:.0f}s", flush=True)
    except Exception as e:
        bad += 1; print("FAIL bf16 B", B, D, str(e)[:500], flush=True)
print("done", bad, "failures")
sys.exit(1 if bad else 0)
