"""GPU gradient parity at the MODEL'S REAL DIMENSIONS (reference SS:209-229: E=200, Rq=512,
M=512, A=256, R=512, K=1000, nhop 8, T=26, 14x14 maps), through the C ABI, against the fp64
oracle -- every output and every parameter tensor's gradient, 1e-4 max-norm relative.

Why these shapes: the bulk GEMMs only take their unpredicated interior-tile paths
(gemm_core.h LoadRC/LoadSC FAST branches, the SC_DTANH operand with its bias row sums, the
128x128 EPI_OUTER dgrad, the split-K counts conv_wgrad_splits / tn_splits / skinny_splits
choose) when M >= 128, D >= 128 and S % 28 == 0 hold together, i.e. at the benchmarked
dimensions.  BASELINE.json configs[0] (batch 16) is where the oracle is affordable: the first test
uses the C++ oracle (seconds), the others the autograd restatement in fp64 (BLAS-backed, so the
whole file stays well inside the GPU test budget on a box with few host cores); the two oracles
agree to 1e-10 (tests/test_oracle_agree.py).
"""
import numpy as np
import pytest

from rau_vqa_amd.model import hop_weights
from tests import util
from tests.test_gpu_parity import check

pytestmark = pytest.mark.gpu

REAL = dict(T=26, V=14000, E=200, Rq=512, S=196, M=512, A=256, R=512, K=1000, H=8)


def test_config0_b16_d512_ss_weights_every_gradient():
    """BASELINE.json configs[0]: Ours_SS 8 hops, batch 16, 14x14x512, ragged lengths, train mode
    with explicit masks, SS hop weights (x nHop, SS:569); params uniform(-0.08, 0.08) (SS:352-354)."""
    errs = check(util.shapes(dict(REAL, B=16, D=512)))
    assert max(errs.values()) < 1e-4


def test_config0_ms_weights_and_full_gating():
    """Same shape under the other scripts' per-hop loss weights: MS x1 (MS:568-570) and the Full
    schedule at epoch 20 (hops 4.. gated off, Full:414-426,587-589: skipped-hop backward)."""
    sh = util.shapes(dict(REAL, B=16, D=512))
    check(sh, hop_w=hop_weights("MS", 8), seed=7, torch_oracle=True)
    w = hop_weights("Full", 8, epoch=20)
    assert w.tolist() == [1, 1, 1, 0, 0, 0, 0, 0]
    check(sh, hop_w=w, seed=8, torch_oracle=True)


def test_resnet_d2048_b16_f32():
    """Ours_ResNet feature width (ResNet:38,217; run script -cnnout_dim 2048): K = 2048 reduction
    in i_embed, 2048-row weight-gradient tiles."""
    check(util.shapes(dict(REAL, B=16, D=2048)), torch_oracle=True)


def test_b144_interior_and_edge_tiles_together():
    """B = 144: 144*196 = 28224 flattened columns = 220.5 column tiles (interior + one edge tile
    per launch), 3 row tiles of 64 in every skinny GEMM with a ragged last one, other split-K
    counts than B=16/256.  H=4 -> hop launch groups 2,1,1."""
    check(util.shapes(dict(REAL, B=144, D=512, H=4)), torch_oracle=True)


def test_fast_dtanh_tile_small():
    """Smallest shape on the interior SC_DTANH path (M >= 128, D >= 128, S = 196): compared per
    element, cheap enough to run with several seeds."""
    dims = dict(B=6, T=5, V=60, E=200, Rq=32, D=128, S=196, M=128, A=128, R=32, K=1000, H=2)
    for seed in (1, 2):
        check(util.shapes(dims), seed=seed, scale=0.2)


def test_eval_mode_full_dims_gradients():
    """evaluate mode (dropout = identity): I and P are hop-invariant, computed once and shared by
    all hops' backward."""
    check(util.shapes(dict(REAL, B=16, D=512, H=3)), mode="eval", torch_oracle=True)


def test_resnet_d2048_bf16_vs_emulating_oracle():
    """configs[2] dims in RAU_BF16 mode at B=16 against the autograd restatement that rounds the
    same GEMM operands to bf16 (5e-4), and against the exact one (3e-2)."""
    from tests.test_gpu_bf16 import run
    run(dict(REAL, B=16, D=2048, H=4), None)
