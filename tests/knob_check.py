"""One parity check of the step path at 14x14 maps, run as its own process by tests/test_gpu_knobs.py
(librau reads most A/B variables once per process).  Prints OK or raises."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import util
from tests.test_gpu_parity import check

B = int(sys.argv[1]) if len(sys.argv) > 1 else 72
kind = sys.argv[2] if len(sys.argv) > 2 else ""
if kind == "bf16":
    # RAU_BF16 mode at widths the LDS-DMA bf16 weight gradient (wgrad16.hip: both row counts % 256)
    # and dgrad16.hip (M % 128, A % 32) accept, against the autograd restatement that rounds the same
    # operands (tests/test_gpu_bf16.py)
    from tests.test_gpu_bf16 import run
    run(dict(B=B, T=5, V=60, E=64, Rq=64, D=256, S=196, M=256, A=64, R=64, K=200, H=3), 0.2)
    print("OK")
    sys.exit(0)
if kind == "bf16ws":
    # RAU_BF16 mode at the question-LSTM width the persistent encoder takes (enc_ws.hip's rounding form)
    from tests.test_gpu_bf16 import run
    dims = dict(B=B, T=6, V=120, E=200, Rq=512, D=64, S=196, M=128, A=64, R=64, K=200, H=2)
    run(dims, 0.2)     # (train mode; the evaluate-mode forward takes the same kernel)
    print("OK")
    sys.exit(0)
if kind == "ws":
    # the reference's question-LSTM width: the only one the weight-stationary persistent encoder
    # (enc_ws.hip: Rq == 512, B % 16 == 0) takes, so RAU_ENC_WS=0|1 compares two different paths
    dims = dict(B=B, T=6, V=120, E=200, Rq=512, D=64, S=196, M=128, A=64, R=64, K=200, H=2)
elif kind == "wg":
    # row counts the LDS-DMA f32 conv weight gradients take (wgrad_dma.hip: A, M, D multiples of 128)
    dims = dict(B=B, T=5, V=120, E=64, Rq=64, D=128, S=196, M=128, A=128, R=64, K=200, H=3)
else:
    # widths the wide conv tiling, the per-sample tiling, the fused attention kernels and the grouped
    # weight gradients all accept; B > 64 takes the large-batch policies, B <= 64 the small-batch ones
    dims = dict(B=B, T=7, V=120, E=200, Rq=64, D=64, S=196, M=128, A=64, R=64, K=1000, H=4)
check(util.shapes(dims), scale=0.2, torch_oracle=True)
check(util.shapes(dims), scale=0.2, mode="eval", torch_oracle=True)
print("OK")
