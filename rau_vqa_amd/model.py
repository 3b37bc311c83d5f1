"""Host-side driver of the RAU hot path: a thin object over the C ABI.

``RAU`` owns one ``rau_ctx`` (one GPU) and exposes the step the reference's
``feval`` performs (experiments/Ours_SS/LstmAttCtrlGradNoiseDontSelect.lua:428-596):
``set_batch`` (next_batch_feat + H2D, SS:434-439), ``forward`` (SS:443-520),
``backward(hop_w)`` (SS:561-596), ``update`` (SS:597-630 + adam, SS:770-772).
All arithmetic happens in librau.so; this file only moves pointers.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib as L


@dataclass
class Config:
    """Network sizes; defaults are the reference's hard-coded locals (SS:202-229)."""
    B: int = 100
    T: int = 26
    V: int = 14000
    E: int = 200
    Rq: int = 512
    D: int = 512
    S: int = 196
    M: int = 512
    A: int = 256
    R: int = 512
    K: int = 1000
    H: int = 8
    p_we: float = 0.5
    p_rnn: float = 0.5
    p_q: float = 0.5
    p_x: float = 0.5
    p_mf: float = 0.5
    device_id: int = 0
    dtype: str = "f32"      # "f32" | "bf16" (bf16-rounded operands in every conv and Linear GEMM) | "f32s" (convs: 3 x bf16 split)

    @property
    def Q(self) -> int:
        return 4 * self.Rq

    def mask_shapes(self):
        return {"we": (self.T, self.B, self.E), "rnn": (self.T, self.B, self.Rq),
                "q": (self.H, self.B, self.Q), "x": (self.H, self.B, self.D, self.S),
                "mf": (self.H, self.B, self.M)}


def hop_weights(variant: str, H: int, epoch: int = 0):
    """Per-hop scale of the criterion gradient for the four training scripts.

    SS: x nHop (Ours_SS:569); MS: x1 (Ours_MS:568-570); Full / ResNet: x1 until
    ``epoch >= tab_multhop_stop_timing[h]``, then x0 (Ours_Full:414-426,587-589;
    Ours_ResNet:418-427).
    """
    if variant == "SS":
        return np.full(H, float(H), np.float32)
    if variant == "MS":
        return np.ones(H, np.float32)
    sched = {"Full": [1000, 35, 25, 20, 18, 16, 16, 16, 16, 1000],
             "ResNet": [1000, 30, 24, 20, 18, 16, 16, 15, 1000, 1000]}[variant]
    w = np.ones(H, np.float32)
    for h in range(H):
        stop = sched[h] if h < len(sched) else 1000
        if epoch >= stop:
            w[h] = 0.0
    return w


class RAU:
    def __init__(self, cfg: Config):
        self.cfg = cfg
        self._lib = L.lib()
        c = L.RauConfig(B=cfg.B, T=cfg.T, V=cfg.V, E=cfg.E, Rq=cfg.Rq, D=cfg.D, S=cfg.S,
                        M=cfg.M, A=cfg.A, R=cfg.R, K=cfg.K, H=cfg.H, p_we=cfg.p_we,
                        p_rnn=cfg.p_rnn, p_q=cfg.p_q, p_x=cfg.p_x, p_mf=cfg.p_mf,
                        dtype={"f32": 0, "bf16": 1, "f32s": 2}[cfg.dtype],
                        device_id=cfg.device_id)
        h = C.c_void_p()
        L.check(self._lib.rau_create(C.byref(c), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rau_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- parameters (m:getParameters(), SS:322-324)
    def group_size(self, group: str) -> int:
        n = C.c_size_t()
        L.check(self._lib.rau_params(self._h, L.GROUPS[group], None, None, C.byref(n)))
        return n.value

    def group_sizes(self):
        return {g: self.group_size(g) for g in L.GROUPS}

    def device_pointers(self, group: str):
        """(weights_ptr, grads_ptr, n): raw device addresses of the flat buffers."""
        w, g, n = C.c_void_p(), C.c_void_p(), C.c_size_t()
        L.check(self._lib.rau_params(self._h, L.GROUPS[group], C.byref(w), C.byref(g), C.byref(n)))
        return w.value, g.value, n.value

    def layout(self, group: str):
        out = []
        for i in range(self._lib.rau_layout_count(self._h, L.GROUPS[group])):
            name, off, r, c = C.c_char_p(), C.c_size_t(), C.c_int32(), C.c_int32()
            L.check(self._lib.rau_layout_entry(self._h, L.GROUPS[group], i, C.byref(name),
                                               C.byref(off), C.byref(r), C.byref(c)))
            out.append((name.value.decode(), off.value, r.value, c.value))
        return out

    def set_params(self, params):
        for g, a in params.items():
            a = np.ascontiguousarray(a, np.float32)
            L.check(self._lib.rau_set_params(self._h, L.GROUPS[g], a.ctypes.data, a.size))

    def _get(self, fn, group):
        a = np.empty(self.group_size(group), np.float32)
        L.check(fn(self._h, L.GROUPS[group], a.ctypes.data, a.size))
        return a

    def get_params(self):
        return {g: self._get(self._lib.rau_get_params, g) for g in L.GROUPS}

    def get_grads(self):
        return {g: self._get(self._lib.rau_get_grads, g) for g in L.GROUPS}

    def set_grads(self, grads):
        for g, a in grads.items():
            a = np.ascontiguousarray(a, np.float32)
            L.check(self._lib.rau_set_grads(self._h, L.GROUPS[g], a.ctypes.data, a.size))

    # ---- snapshots (torch.save / torch.load of SS:1188-1197, Eval.lua:113-114,344-347)
    def save_snapshot(self, path, it, epoch, opt, cuda=True):
        from . import t7
        self.sync()
        t7.save_snapshot(path, self.get_params(), it, epoch, opt, cuda=cuda)

    def load_snapshot(self, path):
        """embed_param:copy(snap.params[1]) ...; returns (it, epoch, opt)."""
        from . import t7
        it, epoch, opt, params = t7.load_snapshot(path)
        for g, a in params.items():
            if a.size != self.group_size(g):
                raise ValueError(f"snapshot group {g} has {a.size} floats, model has "
                                 f"{self.group_size(g)}")
        self.set_params(params)
        return it, epoch, opt

    def init_uniform(self, seed=123, lo=-0.08, hi=0.08):
        L.check(self._lib.rau_init_uniform(self._h, seed, lo, hi))

    def zero_grads(self):
        L.check(self._lib.rau_zero_grads(self._h))

    # ---- mode / dropout
    def training(self):
        L.check(self._lib.rau_set_mode(self._h, L.MODE_TRAIN))

    def evaluate(self):
        L.check(self._lib.rau_set_mode(self._h, L.MODE_EVAL))

    def set_dropout_seed(self, seed: int, step: int = 0):
        L.check(self._lib.rau_set_dropout_seed(self._h, seed, step))

    def set_masks(self, masks):
        for k, m in masks.items():
            m = np.ascontiguousarray(m, np.uint8)
            L.check(self._lib.rau_set_mask(self._h, L.MASK_SITES[k], m.ctypes.data, m.size))

    def get_mask(self, site: str):
        shape = self.cfg.mask_shapes()[site]
        m = np.empty(int(np.prod(shape)), np.uint8)
        L.check(self._lib.rau_get_mask(self._h, L.MASK_SITES[site], m.ctypes.data, m.size))
        return m.reshape(shape)

    # ---- batch + the hot path
    def set_batch(self, feats, tokens, lens, labels=None):
        c = self.cfg
        feats = np.ascontiguousarray(feats, np.float32)
        tokens = np.ascontiguousarray(tokens, np.int32)
        lens = np.ascontiguousarray(lens, np.int32)
        if feats.size != c.B * c.D * c.S or tokens.shape != (c.T, c.B) or lens.shape != (c.B,):
            raise ValueError("batch shapes do not match the config")
        lp = None
        if labels is not None:
            labels = np.ascontiguousarray(labels, np.int32)
            if labels.shape != (c.B,):
                raise ValueError("labels shape")
            lp = labels.ctypes.data
        L.check(self._lib.rau_set_batch(self._h, feats.ctypes.data, tokens.ctypes.data,
                                        lens.ctypes.data, lp))

    # asynchronous, double-buffered upload (rau_batch_slot / rau_set_batch_async / rau_use_batch)
    def batch_slot(self, slot):
        """numpy views of slot's PINNED staging: {feats [B,D,S], tokens [T,B], lens [B], labels [B]}.
        A loader fills them in place; set_batch_async(slot) then uploads without a host copy."""
        c = self.cfg
        p = [C.c_void_p() for _ in range(4)]
        L.check(self._lib.rau_batch_slot(self._h, slot, *[C.byref(x) for x in p]))

        def view(ptr, n, ct, dt, shape):
            return np.frombuffer((ct * n).from_address(ptr.value), dtype=dt).reshape(shape)
        return {"feats": view(p[0], c.B * c.D * c.S, C.c_float, np.float32, (c.B, c.D, c.S)),
                "tokens": view(p[1], c.T * c.B, C.c_int32, np.int32, (c.T, c.B)),
                "lens": view(p[2], c.B, C.c_int32, np.int32, (c.B,)),
                "labels": view(p[3], c.B, C.c_int32, np.int32, (c.B,))}

    def set_batch_async(self, slot, feats=None, tokens=None, lens=None, labels=None, has_labels=True):
        """Enqueue the upload of a batch into `slot` on the copy stream and return.  Arrays left None
        are taken from the slot's staging (filled in place through batch_slot)."""
        c = self.cfg

        def ptr(a, dt, n):
            if a is None:
                return None, None
            a = np.ascontiguousarray(a, dt)
            if a.size != n:
                raise ValueError("batch shapes do not match the config")
            return a.ctypes.data, a
        fp, fk = ptr(feats, np.float32, c.B * c.D * c.S)
        tp, tk = ptr(tokens, np.int32, c.T * c.B)
        lp, lk = ptr(lens, np.int32, c.B)
        yp, yk = ptr(labels, np.int32, c.B)
        L.check(self._lib.rau_set_batch_async(self._h, slot, fp, tp, lp, yp, int(bool(has_labels))))

    def use_batch(self, slot):
        L.check(self._lib.rau_use_batch(self._h, slot))

    def forward(self):
        L.check(self._lib.rau_forward(self._h))

    def backward(self, hop_w):
        w = np.ascontiguousarray(hop_w, np.float32)
        if w.shape != (self.cfg.H,):
            raise ValueError("hop_w must have H entries")
        L.check(self._lib.rau_backward(self._h, w.ctypes.data))

    def graph_step(self, hop_w, zero_grads=True):
        """zero_grads + forward + backward as one hipGraph launch (captured on first use)."""
        w = np.ascontiguousarray(hop_w, np.float32)
        if w.shape != (self.cfg.H,):
            raise ValueError("hop_w must have H entries")
        L.check(self._lib.rau_graph_step(self._h, w.ctypes.data, int(zero_grads)))

    def sync(self):
        L.check(self._lib.rau_sync(self._h))

    def update(self, step_t, lr=3e-3, mult_lr=3e-4, beta1=0.9, beta2=0.999, eps=1e-8,
               eta=0.01, gamma=0.55, clip=0.1, noise_seed=0):
        norms = np.zeros(3, np.float32)
        L.check(self._lib.rau_noise_clip_adam(self._h, step_t, lr, mult_lr, beta1, beta2, eps,
                                              eta, gamma, clip, noise_seed, norms.ctypes.data))
        return norms

    # ---- results
    def _out(self, fn, shape, dtype=np.float32):
        a = np.empty(shape, dtype)
        L.check(fn(self._h, a.ctypes.data))
        return a

    def losses(self):
        return self._out(self._lib.rau_get_losses, (self.cfg.H,))

    def argmax(self):
        return self._out(self._lib.rau_get_argmax, (self.cfg.H, self.cfg.B), np.int32)

    def logits(self):
        return self._out(self._lib.rau_get_logits, (self.cfg.H, self.cfg.B, self.cfg.K))

    def dopred(self):
        return self._out(self._lib.rau_get_dopred, (self.cfg.H, self.cfg.B))

    def attention(self):
        return self._out(self._lib.rau_get_attention, (self.cfg.H, self.cfg.B, self.cfg.S))

    def question_state(self):
        return self._out(self._lib.rau_get_question_state, (self.cfg.B, self.cfg.Q))

    def att_state(self):
        c = np.empty((self.cfg.H, self.cfg.B, self.cfg.R), np.float32)
        h = np.empty_like(c)
        L.check(self._lib.rau_get_att_state(self._h, c.ctypes.data, h.ctypes.data))
        return c, h

    def outputs(self):
        c, h = self.att_state()
        return {"losses": self.losses(), "argmax": self.argmax(), "logits": self.logits(),
                "dopred": self.dopred(), "att": self.attention(), "q": self.question_state(),
                "att_c": c, "att_h": h}

    # ---- timing
    def stream(self) -> int:
        s = C.c_void_p()
        L.check(self._lib.rau_stream(self._h, C.byref(s)))
        return s.value or 0

    # ---- native data-parallel exchange (RCCL inside librau, no torch.distributed needed)
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        L.check(L.lib().rau_comm_unique_id(buf, 128))
        return buf.raw

    def comm_init(self, nranks: int, rank: int, uid: bytes):
        L.check(self._lib.rau_comm_init(self._h, nranks, rank, uid, len(uid)))

    def allreduce_grads(self):
        """Average the three flat gradient buffers over all ranks, in place, ordered on the
        ctx stream (mult bucket overlapped with the encoder BPTT)."""
        L.check(self._lib.rau_allreduce_grads(self._h))

    def comm_destroy(self):
        L.check(self._lib.rau_comm_destroy(self._h))

    def wait_grads(self, group: str, hip_stream: int):
        """Order `hip_stream` after the last backward's gradients of `group` (no host sync)."""
        L.check(self._lib.rau_wait_grads(self._h, L.GROUPS[group], C.c_void_p(hip_stream)))

    def timer_begin(self):
        L.check(self._lib.rau_timer_begin(self._h))

    def timer_end(self) -> float:
        ms = C.c_float()
        L.check(self._lib.rau_timer_end(self._h, C.byref(ms)))
        return ms.value

    def prof_enable(self, on=True):
        L.check(self._lib.rau_prof_enable(self._h, int(on)))

    def prof_reset(self):
        L.check(self._lib.rau_prof_reset(self._h))

    def prof(self):
        out = {}
        n = self._lib.rau_prof_count(self._h)
        for i in range(n):
            name, cnt = C.c_char_p(), C.c_int64()
            ms, fl, by = C.c_double(), C.c_double(), C.c_double()
            L.check(self._lib.rau_prof_entry(self._h, i, C.byref(name), C.byref(cnt),
                                             C.byref(ms), C.byref(fl), C.byref(by)))
            out[name.value.decode()] = {"launches": cnt.value, "ms": ms.value,
                                        "flops": fl.value, "bytes": by.value}
        return out
