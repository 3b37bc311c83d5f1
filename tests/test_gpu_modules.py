"""Module-level entry points (rau_embed_* / rau_deeplstm_* / rau_multimodal_* /
rau_criterion_*, include/rau.h) against the oracle and against the step-level path.

The reference's feval drives its clones one :forward / :backward at a time
(SS:443-596); ``rau_vqa_amd.modules.feval`` re-states those loops over the C ABI's
module-level calls.  Bar: same as test_gpu_parity (1e-4 max-norm relative vs the
fp64 oracle, answer indices exact); single modules with arbitrary gradOutputs
(including the do_pred / attprob gradients feval leaves at zero and the feature-map
gradient it discards) are checked against the autograd restatement.
"""
import numpy as np
import pytest

import oracle
from tests import util
from tests.test_gpu_parity import TOL

pytestmark = pytest.mark.gpu


def make_model(sh, params, masks=None, mode="train"):
    from rau_vqa_amd.model import RAU, Config
    cfg = Config(**{k: getattr(sh, k) for k in
                    ("B", "T", "V", "E", "Rq", "D", "S", "M", "A", "R", "K", "H",
                     "p_we", "p_rnn", "p_q", "p_x", "p_mf")})
    m = RAU(cfg)
    m.set_params(params)
    if mode == "train":
        m.training()
        if masks is not None:
            m.set_masks(masks)
    else:
        m.evaluate()
    return m


def cuda(a, dtype=None):
    import torch
    t = torch.as_tensor(np.ascontiguousarray(a)).cuda()
    return t if dtype is None else t.to(dtype)


def grad_errs(got, ref, layouts):
    errs = {}
    for grp in layouts:
        for name, sl in util.layer_slices(layouts[grp]):
            r = ref["g_" + grp][sl]
            if np.max(np.abs(r)) < 1e-12:
                errs[name] = float(np.max(np.abs(got[grp][sl] - r)))
            else:
                errs[name] = util.rel_err(got[grp][sl], r)
    return errs


@pytest.mark.parametrize("dims,scale,lens", [
    (util.SMALL, 0.5, "ragged"),
    (util.MEDIUM, 0.2, "ragged"),
    (util.SMALL, 0.5, np.array([0, 6, 1, 3, 6, 2, 0, 4], np.int32)),
])
def test_feval_over_module_calls_matches_oracle_and_step_path(dims, scale, lens):
    import torch
    from rau_vqa_amd import modules
    sh = util.shapes(dims)
    batch, params, masks = util.make_problem(sh, lens=lens, scale=scale)
    hop_w = np.full(sh.H, float(sh.H), np.float32)
    ref = oracle.step(sh, params, batch["feats"], batch["tokens"], batch["lens"],
                      batch["labels"], masks, hop_w, dtype=np.float64)
    m = make_model(sh, params, masks)
    layouts = {k: m.layout(k) for k in ("embed", "rnn", "mult")}
    m.zero_grads()
    losses, answers = modules.feval(m, cuda(batch["feats"]), cuda(batch["tokens"], torch.int32),
                                    cuda(batch["lens"], torch.int32),
                                    cuda(batch["labels"], torch.int32), hop_w)
    m.sync()
    g_mod = m.get_grads()
    errs = grad_errs(g_mod, ref, layouts)
    errs["losses"] = util.rel_err(losses.numpy(), ref["losses"])
    bad = {k: v for k, v in errs.items() if not v < TOL}
    assert not bad, f"module-level feval vs oracle above {TOL}: {bad}"
    ok, _, _ = util.argmax_margin_ok(ref["logits"], answers.cpu().numpy(), ref["argmax"])
    assert ok
    # the step-level path on the same ctx (same masks): same numbers up to summation order
    m.set_batch(batch["feats"], batch["tokens"], batch["lens"], batch["labels"])
    m.zero_grads()
    m.forward()
    m.backward(hop_w)
    g_step = m.get_grads()
    for k in g_step:
        assert util.rel_err(g_mod[k], g_step[k]) < 1e-5, k
    m.close()


def test_feval_over_module_calls_with_device_philox_masks():
    """No explicit masks: every clone draws its slice of the (seed, step) Philox tensors,
    i.e. the same masks the step-level path uses -> same gradients."""
    import torch
    from rau_vqa_amd import modules
    sh = util.shapes(util.SMALL)
    batch, params, _ = util.make_problem(sh, scale=0.5)
    hop_w = np.ones(sh.H, np.float32)
    m = make_model(sh, params)
    m.set_dropout_seed(77, 5)
    m.zero_grads()
    modules.feval(m, cuda(batch["feats"]), cuda(batch["tokens"], torch.int32),
                  cuda(batch["lens"], torch.int32), cuda(batch["labels"], torch.int32), hop_w)
    m.sync()
    g_mod = m.get_grads()
    m.set_batch(batch["feats"], batch["tokens"], batch["lens"], batch["labels"])
    m.set_dropout_seed(77, 5)
    m.zero_grads()
    m.forward()
    m.backward(hop_w)
    g_step = m.get_grads()
    for k in g_step:
        assert np.max(np.abs(g_step[k])) > 0
        assert util.rel_err(g_mod[k], g_step[k]) < 1e-5, k
    m.close()


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_multimodal_clone_with_full_grad_outputs(mode):
    """One multimodal clone, every gradOutput non-zero (d_do_pred and d_attprob are zeros in
    feval, SS:566,573) and the feature-map gradient requested (discarded at SS:579):
    against autograd on the module-granular restatement."""
    import torch
    from oracle import ref_torch as RT
    from rau_vqa_amd import modules
    sh = util.shapes(util.SMALL)
    batch, params, masks = util.make_problem(sh, scale=0.5)
    rng = np.random.default_rng(5)
    h = 1
    q = rng.standard_normal((sh.B, sh.Q)).astype(np.float32) * 0.5
    c0 = rng.standard_normal((sh.B, sh.R)).astype(np.float32) * 0.5
    h0 = np.tanh(rng.standard_normal((sh.B, sh.R))).astype(np.float32) * 0.5
    gouts = [rng.standard_normal(s).astype(np.float32) * 0.3 for s in
             [(sh.B, sh.K), (sh.B,), (sh.B, sh.S), (sh.B, sh.R), (sh.B, sh.R)]]
    # ---- autograd reference in float64
    t64 = lambda a: torch.as_tensor(a, dtype=torch.float64)
    flat = t64(params["mult"]).clone().requires_grad_(True)
    Pm = RT._split(flat, RT.mult_specs(sh))
    ins = [t64(q).requires_grad_(True),
           t64(batch["feats"]).reshape(sh.B, sh.D, sh.S, 1).requires_grad_(True),
           t64(c0).requires_grad_(True), t64(h0).requires_grad_(True)]
    mk = (lambda k: torch.as_tensor(masks[k][h])) if mode == "train" else (lambda k: None)
    mx = mk("x")
    outs = RT.multimodal(sh, Pm, ins[0], ins[1], ins[2], ins[3], mk("q"),
                         None if mx is None else mx.reshape(sh.B, sh.D, sh.S, 1), mk("mf"))
    torch.autograd.backward(outs, [t64(g) for g in gouts])
    # ---- the clone
    m = make_model(sh, params, masks, mode)
    layouts = {"mult": m.layout("mult")}
    m.zero_grads()
    clone = modules.MultimodalClone(m, h)
    dq, dX, dc, dh = None, None, None, None
    ext = torch.cuda.ExternalStream(m.stream())
    with torch.cuda.stream(ext):
        args = [cuda(q), cuda(batch["feats"]), cuda(c0), cuda(h0)]
        got = clone.forward(*args)
        dq, dX, dc, dh = clone.backward(*args, *[cuda(g) for g in gouts], want_dX=True)
    m.sync()
    for name, a, b in zip(("logits", "do_pred", "attprob", "c", "h"), got, outs):
        assert util.rel_err(a.cpu().numpy(), b.detach().numpy()) < TOL, name
    for name, a, b in (("d_q", dq, ins[0].grad), ("d_X", dX, ins[1].grad.reshape(sh.B, sh.D, sh.S)),
                       ("d_c", dc, ins[2].grad), ("d_h", dh, ins[3].grad)):
        assert util.rel_err(a.cpu().numpy(), b.numpy()) < TOL, name
    g = m.get_grads()["mult"]
    gref = flat.grad.numpy()
    errs = grad_errs({"mult": g}, {"g_mult": gref}, layouts)   # (attscore bias grad is 0 analytically)
    bad = {k: v for k, v in errs.items() if not v < TOL}
    assert not bad, bad
    m.close()


def test_deeplstm_and_embed_clones_against_autograd():
    import torch
    from oracle import ref_torch as RT
    from rau_vqa_amd import modules
    sh = util.shapes(util.SMALL)
    batch, params, masks = util.make_problem(sh, scale=0.5)
    rng = np.random.default_rng(9)
    t = 2
    state = rng.standard_normal((sh.B, sh.Q)).astype(np.float32) * 0.5
    gstate = rng.standard_normal((sh.B, sh.Q)).astype(np.float32) * 0.3
    tok = batch["tokens"][t].copy()
    tok[1] = tok[0]   # a repeated token inside one clone: accumulation order
    t64 = lambda a: torch.as_tensor(a, dtype=torch.float64)
    emb = t64(params["embed"]).clone().requires_grad_(True)
    flat = t64(params["rnn"]).clone().requires_grad_(True)
    Pr = RT._split(flat, RT.rnn_specs(sh))
    st = t64(state).requires_grad_(True)
    we = torch.tanh(RT._drop(emb.view(sh.V, sh.E)[torch.as_tensor(tok).long() - 1],
                             torch.as_tensor(masks["we"][t]), sh.p_we))
    out = RT.deep_lstm(sh, Pr, we, st, torch.as_tensor(masks["rnn"][t]))
    out.backward(t64(gstate))
    m = make_model(sh, params, masks)
    m.zero_grads()
    ext = torch.cuda.ExternalStream(m.stream())
    with torch.cuda.stream(ext):
        e, r = modules.EmbedClone(m, t), modules.DeepLSTMClone(m, t)
        x_t, s_in = cuda(tok, torch.int32), cuda(state)
        we_g = e.forward(x_t)
        so = r.forward(we_g, s_in)
        d_x, d_s = r.backward(we_g, s_in, cuda(gstate))
        e.backward(x_t, d_x)
    m.sync()
    assert util.rel_err(we_g.cpu().numpy(), we.detach().numpy()) < TOL
    assert util.rel_err(so.cpu().numpy(), out.detach().numpy()) < TOL
    assert util.rel_err(d_s.cpu().numpy(), st.grad.numpy()) < TOL
    g = m.get_grads()
    assert util.rel_err(g["rnn"], flat.grad.numpy()) < TOL
    assert util.rel_err(g["embed"], emb.grad.numpy()) < TOL
    assert np.max(np.abs(g["mult"])) == 0
    m.close()


def test_criterion_clone_and_argument_checks():
    import torch
    import ctypes as C
    from rau_vqa_amd import modules, _lib
    sh = util.shapes(util.SMALL)
    batch, params, _ = util.make_problem(sh, scale=0.5)
    m = make_model(sh, params, mode="eval")
    rng = np.random.default_rng(2)
    lg = rng.standard_normal((sh.B, sh.K)).astype(np.float32)
    y = batch["labels"]
    crit = modules.CriterionClone(m, 0)
    ext = torch.cuda.ExternalStream(m.stream())
    with torch.cuda.stream(ext):
        lgd, yd = cuda(lg), cuda(y, torch.int32)
        loss = crit.forward(lgd, yd)
        dl = crit.backward(lgd, yd, scale=3.0)
    m.sync()
    lt = torch.as_tensor(lg, dtype=torch.float64).requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(lt, torch.as_tensor(y).long() - 1)
    (3.0 * ref).backward()
    ref = ref.detach()
    assert abs(loss - ref.item()) < 1e-5 * max(1.0, abs(ref.item()))
    assert util.rel_err(dl.cpu().numpy(), lt.grad.numpy()) < TOL
    # clone index out of range / missing batch are reported, not crashed on
    out = C.c_void_p()
    assert m._lib.rau_embed_forward(m._h, sh.T, None, C.byref(out)) == -1
    assert m._lib.rau_embed_forward(m._h, 0, None, C.byref(out)) == -3   # no batch resident
    assert b"no tokens" in m._lib.rau_last_error()
    assert m._lib.rau_multimodal_forward(m._h, sh.H, None, None, None, None, None, None, None,
                                         None, None) == -1
    m.close()


@pytest.mark.parametrize("row_loop", [False, True])
def test_feval_with_device_tensor_glue_only(row_loop):
    """The Lua shim's route (bindings/rau.lua RAU.Tensor): feval's loops over the module-level ABI
    with NOTHING but rau_dev_* calls for the tensor algebra in between -- row selection by question
    length (one call, or the reference's row-by-row loop), uni_pred accumulation, first-max
    argmax, correct counts.  Same gradients as the oracle."""
    from rau_vqa_amd import modules
    from rau_vqa_amd.modules import DevTensor
    sh = util.shapes(util.SMALL)
    lens = np.array([0, 6, 1, 3, 6, 2, 0, 4], np.int32)
    batch, params, masks = util.make_problem(sh, lens=lens, scale=0.5)
    hop_w = np.full(sh.H, float(sh.H), np.float32)
    ref = oracle.step(sh, params, batch["feats"], batch["tokens"], batch["lens"],
                      batch["labels"], masks, hop_w, dtype=np.float64)
    m = make_model(sh, params, masks)
    layouts = {k: m.layout(k) for k in ("embed", "rnn", "mult")}
    feats = DevTensor.zeros(m, sh.B, sh.D, sh.S).copy(batch["feats"])
    x = [DevTensor.ints(m, batch["tokens"][t]) for t in range(sh.T)]
    y = DevTensor.ints(m, batch["labels"])
    m.zero_grads()
    losses, correct, uni = modules.feval_dev(m, feats, x, batch["lens"], y, hop_w, row_loop=row_loop)
    m.sync()
    errs = grad_errs(m.get_grads(), ref, layouts)
    errs["losses"] = util.rel_err(np.array(losses), ref["losses"])
    errs["uni"] = util.rel_err(uni.numpy(), ref["logits"].sum(0))
    bad = {k: v for k, v in errs.items() if not v < TOL}
    assert not bad, bad
    want = [(ref["argmax"][h] == batch["labels"]).sum() for h in range(sh.H)]
    assert correct == [int(v) for v in want]
    m.close()


def test_device_tensor_ops():
    from rau_vqa_amd.modules import DevTensor
    sh = util.shapes(util.EDGE)
    _, params, _ = util.make_problem(sh)
    m = make_model(sh, params, mode="eval")
    rng = np.random.default_rng(0)
    a = rng.standard_normal((6, 10)).astype(np.float32)
    a[2, 3] = a[2, 7] = a[2].max() + 1.0                     # a tie: the FIRST index wins
    t = DevTensor.zeros(m, 6, 10).copy(a)
    v, i = t.max(2)
    assert np.array_equal(i.numpy("int32").ravel(), a.argmax(1) + 1)
    assert np.array_equal(v.numpy().ravel(), a.max(1))
    assert abs(t.sum() - float(a.astype(np.float64).sum())) < 1e-5
    u = DevTensor.zeros(m, 6, 10).fill(2.0).add(0.5, t).mul(2.0)
    assert np.allclose(u.numpy(), (2.0 + 0.5 * a) * 2.0)
    key = DevTensor.ints(m, np.array([1, 3, 3, 0, 3, 2], np.int32))
    w = DevTensor.zeros(m, 6, 10).select_rows(t, key, 3)
    want = np.where((np.array([1, 3, 3, 0, 3, 2]) == 3)[:, None], a, 0.0)
    assert np.array_equal(w.numpy(), want)
    w[0] = t[5]                                              # row view / row assign
    assert np.array_equal(w.numpy()[0], a[5])
    assert key.eq_sum(DevTensor.ints(m, np.array([1, 0, 3, 0, 3, 9], np.int32))) == 4
    w.free()
    from rau_vqa_amd import _lib as L
    with pytest.raises(L.RauError, match="rau_dev_free"):     # not one of the ctx's allocations
        L.check(m._lib.rau_dev_free(m._h, 12345 * 16))
    m.close()


def test_reference_adam_on_flat_device_vectors():
    """SS:770-772 above the C ABI: `adam(x, dx, lr, alpha, beta, epsilon, state)` of
    utils/optim_updates.lua:59-87 on each group's flat (param, grad) device vectors -- as one fused
    pass (rau_dev_adam, what bindings/rau.lua's RAU.adam calls) and as the reference's own five
    tensor statements (mul / add / addcmul / sqrt / add(scalar) / addcdiv, one rau_dev_* call each)
    -- against the oracle's Adam (noise off, clip off) over three steps; and m:updateParameters'
    plain SGD.  Also: repeated max() allocates nothing (a ring of context-owned slots)."""
    from rau_vqa_amd.modules import DevTensor, adam, adam_statements, flat
    sh = util.shapes(util.SMALL)
    _, params, _ = util.make_problem(sh)
    rng = np.random.default_rng(11)
    for fn in (adam, adam_statements):
        m = make_model(sh, params, mode="eval")
        ref = {k: v.astype(np.float64) for k, v in params.items()}
        mom = {k: (np.zeros(v.size), np.zeros(v.size)) for k, v in params.items()}
        state = {k: {} for k in params}
        for t in range(1, 4):
            grads = {k: (rng.standard_normal(v.size) * 0.01).astype(np.float32) for k, v in params.items()}
            m.set_grads(grads)
            for k in ("embed", "rnn", "mult"):
                lr = 3e-4 if k == "mult" else 3e-3                      # SS:43-46, 770-772
                x, dx = flat(m, k)
                fn(x, dx, lr, 0.9, 0.999, 1e-8, state[k])
                oracle.noise_clip_adam(ref[k], grads[k].astype(np.float64), mom[k][0], mom[k][1], None,
                                       t - 1, t, lr, eta=0.0, clip=1e30)
            got = m.get_params()
            for k in ref:
                assert util.rel_err(got[k], ref[k]) < 1e-5, (fn.__name__, t, k)
        assert state["mult"]["t"] == 3
        m.close()
    # nn.Module:updateParameters(lr) = x - lr dx on the flat vectors
    m = make_model(sh, params, mode="eval")
    grads = {k: (rng.standard_normal(v.size) * 0.01).astype(np.float32) for k, v in params.items()}
    m.set_grads(grads)
    for k in ("embed", "rnn", "mult"):
        x, dx = flat(m, k)
        x.add(-0.5, dx)
    got = m.get_params()
    for k in params:
        assert np.allclose(got[k], params[k] - 0.5 * grads[k], rtol=1e-6, atol=1e-7), k
    t = DevTensor.zeros(m, 5, 7).copy(rng.standard_normal((5, 7)).astype(np.float32))
    seen = set()
    for _ in range(12):
        v, i = t.max(2)
        seen.add((v.ptr, i.ptr))
    assert len(seen) == 4                                             # four slots, reused
    m.close()


def test_module_level_calls_on_7x7_maps():
    """S = 49 (the scripts' default feature grid, SS:36-37) through the MODULE-level entry points:
    callers see dense [.., 49] tensors, the library re-pitches to 52 internally."""
    import torch
    from rau_vqa_amd import modules
    dims = dict(B=6, T=5, V=40, E=8, Rq=16, D=24, S=49, M=40, A=20, R=16, K=12, H=3)
    sh = util.shapes(dims)
    batch, params, masks = util.make_problem(sh, scale=0.5)
    hop_w = np.full(sh.H, float(sh.H), np.float32)
    ref = oracle.step(sh, params, batch["feats"], batch["tokens"], batch["lens"],
                      batch["labels"], masks, hop_w, dtype=np.float64)
    m = make_model(sh, params, masks)
    layouts = {k: m.layout(k) for k in ("embed", "rnn", "mult")}
    m.zero_grads()
    losses, answers = modules.feval(m, cuda(batch["feats"]), cuda(batch["tokens"], torch.int32),
                                    cuda(batch["lens"], torch.int32),
                                    cuda(batch["labels"], torch.int32), hop_w)
    m.sync()
    errs = grad_errs(m.get_grads(), ref, layouts)
    errs["losses"] = util.rel_err(losses.numpy(), ref["losses"])
    bad = {k: v for k, v in errs.items() if not v < TOL}
    assert not bad, bad
    # one clone with every gradOutput set, attprob in and out dense [B,49], d_X on request
    q = cuda(np.random.default_rng(1).standard_normal((sh.B, 4 * sh.Rq)).astype(np.float32))
    X = cuda(batch["feats"])
    mm = modules.MultimodalClone(m, 1)
    lg, dp, att, cn, hn = mm.forward(q, X, None, None)
    assert att.shape == (sh.B, 49)
    assert torch.allclose(att.sum(1), torch.ones(sh.B, device=att.device), atol=1e-5)
    d_att = torch.ones_like(att)
    dq, dX, dc, dh = mm.backward(q, X, None, None, torch.zeros_like(lg), None, d_att, None, None,
                                 want_dX=True)
    assert dX.shape == (sh.B, sh.D, 49) and bool(torch.isfinite(dX).all())
    # softmax Jacobian: a constant gradient at the attention output gives no gradient upstream
    assert float(dq.abs().max()) < 1e-5 and float(dX.abs().max()) < 1e-5
    m.close()


def test_out_of_range_ids_in_device_tensors_are_clamped_not_faulted():
    """Module-level calls take DEVICE id tensors the library cannot range-check (include/rau.h): token
    ids and labels are clamped into [1, V] / [1, K] by the kernels, so 0, negative and too-large ids
    behave like the nearest valid id instead of reading or updating memory outside the tables."""
    import torch
    from rau_vqa_amd import modules
    sh = util.shapes(util.SMALL)
    batch, params, _ = util.make_problem(sh, scale=0.5)
    m = make_model(sh, params, mode="eval")
    bad = np.array([0, -7, sh.V + 1, 2 ** 30, 1, sh.V, 3, 2][:sh.B], np.int32)
    good = np.clip(bad, 1, sh.V).astype(np.int32)
    rng = np.random.default_rng(4)
    d_we = rng.standard_normal((sh.B, sh.E)).astype(np.float32)
    grads = []
    ext = torch.cuda.ExternalStream(m.stream())
    outs = []
    for ids in (bad, good):
        m.zero_grads()
        with torch.cuda.stream(ext):
            e = modules.EmbedClone(m, 0)
            x = cuda(ids, torch.int32)
            outs.append(e.forward(x).clone())
            e.backward(x, cuda(d_we))
        m.sync()
        grads.append(m.get_grads()["embed"].copy())
    assert torch.equal(outs[0], outs[1])
    assert np.array_equal(grads[0], grads[1])
    # labels: the criterion treats 0 / K+5 like 1 / K
    lg = rng.standard_normal((sh.B, sh.K)).astype(np.float32)
    ybad = np.array([0, sh.K + 5, 1, sh.K, 2, 3, 4, 5][:sh.B], np.int32)
    ygood = np.clip(ybad, 1, sh.K).astype(np.int32)
    crit = modules.CriterionClone(m, 0)
    res = []
    for y in (ybad, ygood):
        with torch.cuda.stream(ext):
            lgd, yd = cuda(lg), cuda(y, torch.int32)
            loss = crit.forward(lgd, yd)
            dl = crit.backward(lgd, yd).clone()
        m.sync()
        res.append((loss, dl))
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1])
    m.close()
