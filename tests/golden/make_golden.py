"""Generates the golden fixtures in this directory from the CPU oracle (fp64).

    python tests/golden/make_golden.py

PARITY UNPINNED: the reference (Lua/Torch7) ships no golden vectors and cannot run
in the build image, so these vectors come from the build's own oracle
(oracle/rau_cpu.cc), which is cross-validated against the independent autograd
restatement oracle/ref_torch.py (tests/test_oracle_agree.py).  A fixture holds
inputs (seeded synthetic batch, parameters, dropout masks) and expected outputs
(per-hop losses / logits / argmax / attention, question state, flat gradients).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from tests import util  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (shape dict, lens, hop weights, train mode, param scale)
    "tiny_train_ss": (dict(B=4, T=5, V=30, E=8, Rq=8, D=12, S=8, M=16, A=8, R=8, K=12, H=2),
                      "ragged", "SS", True, 0.5),
    "tiny_eval_ms": (dict(B=3, T=4, V=20, E=8, Rq=8, D=8, S=12, M=12, A=8, R=8, K=8, H=3),
                     [4, 1, 3], "MS", False, 0.5),
    "s196_k1000_gated": (dict(B=6, T=6, V=60, E=200, Rq=16, D=16, S=196, M=24, A=12, R=16,
                              K=1000, H=3), "ragged", [1.0, 0.0, 1.0], True, 0.3),
}


def build(name):
    dims, lens, hw, train, scale = CASES[name]
    sh = util.shapes(dims)
    batch, params, masks = util.make_problem(sh, seed=123, lens=lens, dtype=np.float32, scale=scale)
    if hw == "SS":
        hop_w = np.full(sh.H, float(sh.H), np.float32)
    elif hw == "MS":
        hop_w = np.ones(sh.H, np.float32)
    else:
        hop_w = np.asarray(hw, np.float32)
    ref = oracle.step(sh, params, batch["feats"], batch["tokens"], batch["lens"],
                      batch["labels"], masks if train else None, hop_w, dtype=np.float64)
    out = {"dims": np.array([dims[k] for k in sorted(dims)], np.int32),
           "dim_names": np.array(sorted(dims)), "train": np.array(train), "hop_w": hop_w}
    for k, v in batch.items():
        out["in_" + k] = v
    for k, v in params.items():
        out["p_" + k] = v
    for k, v in masks.items():
        out["m_" + k] = np.packbits(v.reshape(-1))
    for k in ("losses", "logits", "att", "q", "dopred", "att_c", "att_h"):
        out["o_" + k] = ref[k].astype(np.float32)
    out["o_argmax"] = ref["argmax"]
    for k in ("g_embed", "g_rnn", "g_mult"):
        out["o_" + k] = ref[k].astype(np.float32)
    return out


if __name__ == "__main__":
    for name in CASES:
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **build(name))
        print("wrote", name, os.path.getsize(os.path.join(HERE, name + ".npz")), "bytes")
