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


def test_split_guard_fails_closed():
    """consumers of split-K partial sums (lin_reduce_epilogue, the LSTM cell
    kernels, the attention kernels, the criterion head, splitk_reduce_acc) read slab[split][..]
    with no bound of their own; their launchers check the span against the registered workspace
    first and return RAU_ERR_STATE instead of launching.  The same check, through the C ABI, on a
    workspace of the real size (16 * B * 4R floats at B = 256) with good and corrupted state."""
    from rau_vqa_amd import _lib
    l = _lib.lib()
    B, R4 = 256, 2048
    ws = 16 * B * R4
    per = B * R4                       # one partial of the attention-LSTM gate GEMM
    OK, ERR_STATE = 0, -3
    assert l.rau_split_guard_check(ws, 0, 4, per) == OK
    assert l.rau_split_guard_check(ws, ws // 2, 8, per) == OK            # second half, exactly full
    assert l.rau_split_guard_check(ws, 0, 0, per) == OK                  # no partials: nothing is read
    # a stale split count (one more than fits), a stale offset, a count from another shape
    assert l.rau_split_guard_check(ws, ws // 2, 9, per) == ERR_STATE
    assert b"nothing would be launched" in l.rau_last_error()
    assert l.rau_split_guard_check(ws, ws - per + 1, 1, per) == ERR_STATE
    assert l.rau_split_guard_check(ws, ws, 1, per) == ERR_STATE          # one past the end
    assert l.rau_split_guard_check(ws, 0, 17, per) == ERR_STATE
    assert l.rau_split_guard_check(ws, 0, -1, per) == ERR_STATE          # garbage count
    assert l.rau_split_guard_check(ws, 0, 5000, 1) == ERR_STATE          # above the library's split cap
    assert l.rau_split_guard_check(ws, 0, 1, 0) == ERR_STATE             # degenerate partial size


def test_persistent_encoder_needs_its_whole_grid_resident():
    """enc_ws.hip's workgroups wait on progress counters of OTHER workgroups of
    the same launch, so rau_create only selects it when every one of them can be resident at once --
    counted at ONE workgroup per CU, because in the step each CU also holds a bulk tile."""
    from rau_vqa_amd import _lib
    l = _lib.lib()
    assert l.rau_enc_ws_coresident(64, 2, 256) == 1      # MI355X, SPX: 192 workgroups on 256 CUs
    assert l.rau_enc_ws_coresident(32, 1, 256) == 1
    assert l.rau_enc_ws_coresident(16, 1, 96) == 1       # B = 16: one sample half, 96 workgroups
    assert l.rau_enc_ws_coresident(64, 2, 128) == 0      # a 128-CU partition: 192 do not fit at one per CU
    assert l.rau_enc_ws_coresident(64, 1, 32) == 0       # CPX partition
    assert l.rau_enc_ws_coresident(64, 0, 256) == 0      # the occupancy query says it does not fit at all
    assert l.rau_enc_ws_coresident(24, 2, 256) == 0      # not a shape the kernel takes
