"""One parity check of the step path at 14x14 maps, run as its own process by tests/test_gpu_knobs.py
(librau reads most A/B variables once per process).  Prints OK or raises."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import util
from tests.test_gpu_parity import check

B = int(sys.argv[1]) if len(sys.argv) > 1 else 72
if len(sys.argv) > 2 and sys.argv[2] == "bf16":
    # RAU_BF16 mode at widths the LDS-DMA bf16 weight gradient (wgrad16.hip) accepts, against the
    # autograd restatement that rounds the same operands (tests/test_gpu_bf16.py)
    from tests.test_gpu_bf16 import run
    run(dict(B=B, T=5, V=60, E=64, Rq=64, D=256, S=196, M=128, A=64, R=64, K=200, H=3), 0.2)
    print("OK")
    sys.exit(0)
# widths the wide conv tiling, the per-sample tiling, the fused attention kernels and the grouped
# weight gradients all accept; B > 64 takes the large-batch policies, B <= 64 the small-batch ones
dims = dict(B=B, T=7, V=120, E=200, Rq=64, D=64, S=196, M=128, A=64, R=64, K=1000, H=4)
check(util.shapes(dims), scale=0.2, torch_oracle=True)
check(util.shapes(dims), scale=0.2, mode="eval", torch_oracle=True)
print("OK")
