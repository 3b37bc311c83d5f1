"""Host-side logic that needs no GPU: hop-weight schedules, synthetic batches,
flat-parameter sizes, batch sharding."""
import numpy as np
import pytest

import oracle
from rau_vqa_amd import synth
from rau_vqa_amd.model import Config, hop_weights


def test_hop_weights_follow_the_four_scripts():
    assert np.all(hop_weights("SS", 8) == 8.0)                  # Ours_SS:569 dpred:mul(nHop)
    assert np.all(hop_weights("MS", 8) == 1.0)                  # Ours_MS:568-570
    assert np.all(hop_weights("Full", 8, epoch=0) == 1.0)       # Ours_Full:414-426
    w = hop_weights("Full", 8, epoch=20)   # stops {1000,35,25,20,18,16,16,16}
    assert list(w) == [1, 1, 1, 0, 0, 0, 0, 0]
    w = hop_weights("ResNet", 8, epoch=16)  # {1000,30,24,20,18,16,16,15}
    assert list(w) == [1, 1, 1, 1, 1, 0, 0, 0]


def test_group_sizes_match_baseline_md():
    """rnn 3,563,520; mult 4,916,142 (D=512) / 5,702,574 (D=2048)."""
    sh = oracle.Shapes(B=1, T=1, V=14000, E=200, Rq=512, D=512, S=196, M=512, A=256, R=512,
                       K=1000, H=8)
    ne, nr, nm = oracle.group_sizes(sh)
    assert (ne, nr, nm) == (14000 * 200, 3563520, 4916142)
    sh.D = 2048
    assert oracle.group_sizes(sh)[2] == 5702574


def test_synthetic_batch_contract():
    b = synth.make_batch(B=32, T=26, V=14000, D=8, S=196, K=1000, lens="ragged")
    assert b["feats"].shape == (32, 8, 196) and b["feats"].min() >= 0      # post-ReLU-like
    assert b["tokens"].shape == (26, 32) and b["tokens"].min() >= 1
    assert b["lens"].min() >= 3 and b["lens"].max() <= 26
    for k in range(32):   # right-padded with ZEROPAD = 1 (loader.lua:1393)
        assert np.all(b["tokens"][b["lens"][k]:, k] == 1)
        assert np.all(b["tokens"][:b["lens"][k], k] >= 2)
    assert b["labels"].min() >= 1 and b["labels"].max() <= 1000
    full = synth.make_batch(B=4, T=26, V=50, D=8, S=4, K=10, lens="full")
    assert np.all(full["lens"] == 26)


def test_config_mask_shapes_match_oracle():
    c = Config(B=4, T=5, H=3)
    sh = oracle.Shapes(B=4, T=5, H=3, V=c.V, E=c.E, Rq=c.Rq, D=c.D, S=c.S, M=c.M, A=c.A,
                       R=c.R, K=c.K)
    assert c.mask_shapes() == oracle.mask_shapes(sh)


def test_shard_batch_even_split():
    from rau_vqa_amd.dist import shard_batch
    b = synth.make_batch(B=8, T=5, V=20, D=4, S=4, K=6, lens="ragged")
    parts = [shard_batch(b, r, 4) for r in range(4)]
    assert np.array_equal(np.concatenate([p["feats"] for p in parts]), b["feats"])
    assert np.array_equal(np.concatenate([p["tokens"] for p in parts], axis=1), b["tokens"])
    with pytest.raises(ValueError):
        shard_batch(b, 0, 3)
