"""rau_dtype RAU_F32S: f32 operands of the five conv GEMMs split into three bf16 terms (hi + mid +
lo = all 24 significand bits), six bf16 MFMA products per operand pair, f32 accumulate.

The claim to check is "f32-grade": against the fp64 oracle the split mode must meet the same
1e-4 bar as the exact-f32 path on every output and every layer's gradient, its error must be of
the same order as the exact path's (here: within 4x layer by layer, and below 2e-5 overall), and it
must be orders of magnitude closer than the bf16-rounded mode."""
import numpy as np
import pytest

import oracle
from tests import util

pytestmark = pytest.mark.gpu


def run_mode(sh, batch, params, masks, hop_w, dtype):
    from rau_vqa_amd.model import RAU, Config
    cfg = Config(**{k: getattr(sh, k) for k in
                    ("B", "T", "V", "E", "Rq", "D", "S", "M", "A", "R", "K", "H",
                     "p_we", "p_rnn", "p_q", "p_x", "p_mf")}, dtype=dtype)
    m = RAU(cfg)
    m.set_params(params)
    m.training()
    m.set_masks(masks)
    m.set_batch(batch["feats"], batch["tokens"], batch["lens"], batch["labels"])
    m.zero_grads()
    m.forward()
    out = m.outputs()
    m.backward(hop_w)
    g = m.get_grads()
    layouts = {k: m.layout(k) for k in ("embed", "rnn", "mult")}
    m.close()
    return out, g, layouts


def errors(out, g, layouts, ref):
    errs = {k: util.rel_err(out[k], ref[k]) for k in util.OUT_KEYS}
    for grp in layouts:
        for name, sl in util.layer_slices(layouts[grp]):
            r = ref["g_" + grp][sl]
            errs[name] = (float(np.max(np.abs(g[grp][sl] - r))) if np.max(np.abs(r)) < 1e-12
                          else util.rel_err(g[grp][sl], r))
    return errs


@pytest.mark.parametrize("dims,scale", [
    (util.SMALL, 0.5), (util.MEDIUM, 0.2),
    (dict(B=6, T=4, V=40, E=200, Rq=32, D=2048, S=196, M=64, A=32, R=32, K=1000, H=2), 0.05),
    (dict(B=6, T=5, V=40, E=8, Rq=16, D=24, S=49, M=40, A=20, R=16, K=12, H=3), 0.5),
    (dict(B=16, T=26, V=14000, E=200, Rq=512, D=512, S=196, M=512, A=256, R=512, K=1000, H=8), None),
])
def test_split_mode_is_f32_grade(dims, scale):
    sh = util.shapes(dims)
    batch, params, masks = util.make_problem(sh, scale=scale)
    hop_w = np.full(sh.H, float(sh.H), np.float32)
    ref = oracle.step(sh, params, batch["feats"], batch["tokens"], batch["lens"],
                      batch["labels"], masks, hop_w, dtype=np.float64)
    e = {}
    for dt in ("f32", "f32s", "bf16"):
        out, g, layouts = run_mode(sh, batch, params, masks, hop_w, dt)
        e[dt] = errors(out, g, layouts, ref)
        if dt != "bf16":
            ok, _, _ = util.argmax_margin_ok(ref["logits"], out["argmax"], ref["argmax"])
            assert ok, dt
    worst = {dt: max(v.values()) for dt, v in e.items()}
    print("max rel err vs fp64 oracle:", {k: f"{v:.2e}" for k, v in worst.items()})
    bad = {k: v for k, v in e["f32s"].items() if not v < 1e-4}
    assert not bad, bad
    # same order as the exact path, layer by layer (floor: the exact path's own noise level)
    for k, v in e["f32s"].items():
        assert v <= 4.0 * max(e["f32"][k], 5e-7), (k, v, e["f32"][k])
    assert worst["f32s"] < 2e-5
    if worst["bf16"] > 1e-4:
        assert worst["f32s"] < worst["bf16"] / 50


def test_split_mode_is_deterministic_and_differs_from_the_exact_path():
    sh = util.shapes(util.MEDIUM)
    batch, params, masks = util.make_problem(sh, scale=0.2)
    hop_w = np.full(sh.H, float(sh.H), np.float32)
    a = run_mode(sh, batch, params, masks, hop_w, "f32s")
    b = run_mode(sh, batch, params, masks, hop_w, "f32s")
    c = run_mode(sh, batch, params, masks, hop_w, "f32")
    for k in a[1]:
        assert np.array_equal(a[1][k], b[1][k])
    assert not np.array_equal(a[1]["mult"], c[1]["mult"])   # not the fmaf chain, and says so
    assert util.rel_err(a[1]["mult"], c[1]["mult"]) < 1e-5
