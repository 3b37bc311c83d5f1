"""Both attention kernel families at shapes where the other one is the default.

rau_create picks the per-sample 16-wave fused kernels above 64 samples and the row-chunk split
kernels (8 chunks per sample) up to 64; RAU_ATT_FUSED / RAU_ATT_SPLIT (read when the context is
created) force one family.  Same bar as tests/test_gpu_parity.py: 1e-4 against the fp64 oracle on
every output and every layer's gradient, in train and evaluate mode, on 14x14 and pitched 7x7 maps.
"""
import pytest

from tests import util
from tests.test_gpu_parity import check

pytestmark = pytest.mark.gpu

SEVEN = dict(B=6, T=5, V=40, E=8, Rq=16, D=24, S=49, M=40, A=20, R=16, K=12, H=3)
WIDE = dict(B=80, T=6, V=50, E=16, Rq=32, D=64, S=196, M=128, A=64, R=32, K=40, H=3)


@pytest.mark.parametrize("dims,scale", [(util.SMALL, 0.5), (util.MEDIUM, 0.2), (SEVEN, 0.5)])
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_fused_attention_kernels_at_small_batches(monkeypatch, dims, scale, mode):
    monkeypatch.setenv("RAU_ATT_FUSED", "1")
    check(util.shapes(dims), scale=scale, mode=mode)


@pytest.mark.parametrize("chunks", ["4", "8"])
def test_split_attention_kernels_above_64_samples(monkeypatch, chunks):
    monkeypatch.setenv("RAU_ATT_SPLIT", "1")
    monkeypatch.setenv("RAU_ATT_CHUNKS", chunks)
    check(util.shapes(WIDE), scale=0.3, torch_oracle=True)


def test_default_family_at_80_samples_is_the_fused_one_and_agrees():
    check(util.shapes(WIDE), scale=0.3, torch_oracle=True)
