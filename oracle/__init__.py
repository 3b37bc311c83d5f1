"""CPU oracle for the RAU forward/backward path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product path (``rau_vqa_amd``) never does.

PARITY UNPINNED: the reference (Lua/Torch7) cannot run in this image and ships
no golden vectors; the C++ restatement (``rau_cpu.cc``) is cross-validated
against the independently written autograd restatement (``ref_torch.py``).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, fields

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librau_oracle.so")


class OracleCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("B", "T", "V", "E", "Rq", "D", "S", "M", "A", "R", "K", "H")] + \
               [(n, C.c_float) for n in ("p_we", "p_rnn", "p_q", "p_x", "p_mf")]


@dataclass
class Shapes:
    """Shapes of one RAU problem (names as in SURVEY.md section 8)."""
    B: int = 4
    T: int = 5
    V: int = 50
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

    @property
    def Q(self) -> int:
        return 4 * self.Rq

    def c(self) -> OracleCfg:
        return OracleCfg(**{f.name: getattr(self, f.name) for f in fields(self)})


def build(force: bool = False) -> str:
    """Compile oracle/librau_oracle.so with the committed Makefile."""
    src = [os.path.join(_HERE, f) for f in ("rau_cpu.cc", "rau_oracle.h")]
    stale = (not os.path.exists(_LIB_PATH) or
             any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "librau_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        for n in ("rau_oracle_n_embed", "rau_oracle_n_rnn", "rau_oracle_n_mult"):
            getattr(_lib, n).restype = C.c_size_t
            getattr(_lib, n).argtypes = [C.POINTER(OracleCfg)]
        _lib.rau_oracle_fill_mask.restype = None
        _lib.rau_oracle_fill_mask.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32,
                                              C.c_float, C.c_size_t, C.c_void_p]
    return _lib


def group_sizes(sh: Shapes):
    cfg = sh.c()
    l = lib()
    return (l.rau_oracle_n_embed(cfg), l.rau_oracle_n_rnn(cfg), l.rau_oracle_n_mult(cfg))


MASK_SITES = {"we": 0, "rnn": 1, "q": 2, "x": 3, "mf": 4}


def mask_shapes(sh: Shapes):
    return {"we": (sh.T, sh.B, sh.E), "rnn": (sh.T, sh.B, sh.Rq),
            "q": (sh.H, sh.B, sh.Q), "x": (sh.H, sh.B, sh.D, sh.S),
            "mf": (sh.H, sh.B, sh.M)}


def philox_masks(sh: Shapes, seed: int, step: int = 0):
    """The dropout masks the HIP path generates for (seed, step), as uint8 keep flags."""
    out = {}
    for name, shape in mask_shapes(sh).items():
        n = int(np.prod(shape))
        m = np.empty(n, np.uint8)
        p = getattr(sh, "p_" + name)
        lib().rau_oracle_fill_mask(seed, MASK_SITES[name], step, p, n,
                                   m.ctypes.data_as(C.c_void_p))
        out[name] = m.reshape(shape)
    return out


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def step(sh: Shapes, params, feats, tokens, lens, labels, masks=None, hop_w=None,
         backward=True, dtype=np.float32):
    """One forward(+backward) step on the CPU oracle.

    params: dict embed/rnn/mult flat arrays.  masks: dict we/rnn/q/x/mf of uint8
    keep flags or None (evaluate mode).  Returns a dict of outputs and grads.
    """
    dt = np.dtype(dtype)
    fn = lib().rau_oracle_step_f32 if dt == np.float32 else lib().rau_oracle_step_f64
    fn.restype = C.c_int
    cvt = lambda a: np.ascontiguousarray(a, dtype=dt)
    emb, rnn, mult = cvt(params["embed"]), cvt(params["rnn"]), cvt(params["mult"])
    ne, nr, nm = group_sizes(sh)
    assert emb.size == ne and rnn.size == nr and mult.size == nm, \
        (emb.size, ne, rnn.size, nr, mult.size, nm)
    feats = cvt(feats)
    assert feats.size == sh.B * sh.D * sh.S
    tokens = np.ascontiguousarray(tokens, np.int32)
    lens = np.ascontiguousarray(lens, np.int32)
    assert tokens.shape == (sh.T, sh.B) and lens.shape == (sh.B,)
    labels_c = None if labels is None else np.ascontiguousarray(labels, np.int32)
    m = {}
    for k, shape in mask_shapes(sh).items():
        if masks is None or masks.get(k) is None:
            m[k] = None
        else:
            m[k] = np.ascontiguousarray(masks[k], np.uint8)
            assert m[k].shape == shape, (k, m[k].shape, shape)
    if hop_w is None:
        hop_w = np.full(sh.H, float(sh.H))  # SS:569
    hop_w = cvt(hop_w)
    out = {
        "losses": np.zeros(sh.H, dt), "argmax": np.zeros((sh.H, sh.B), np.int32),
        "logits": np.zeros((sh.H, sh.B, sh.K), dt), "dopred": np.zeros((sh.H, sh.B), dt),
        "att": np.zeros((sh.H, sh.B, sh.S), dt), "q": np.zeros((sh.B, sh.Q), dt),
        "att_c": np.zeros((sh.H, sh.B, sh.R), dt), "att_h": np.zeros((sh.H, sh.B, sh.R), dt),
    }
    g = {}
    if backward:
        g = {"g_embed": np.zeros(ne, dt), "g_rnn": np.zeros(nr, dt), "g_mult": np.zeros(nm, dt)}
    cfg = sh.c()
    rc = fn(C.byref(cfg), _ptr(emb), _ptr(rnn), _ptr(mult), _ptr(feats), _ptr(tokens),
            _ptr(lens), _ptr(labels_c), _ptr(m["we"]), _ptr(m["rnn"]), _ptr(m["q"]),
            _ptr(m["x"]), _ptr(m["mf"]), _ptr(hop_w), _ptr(out["losses"]),
            _ptr(out["argmax"]), _ptr(out["logits"]), _ptr(out["dopred"]),
            _ptr(out["att"]), _ptr(out["q"]), _ptr(out["att_c"]), _ptr(out["att_h"]),
            _ptr(g.get("g_embed")), _ptr(g.get("g_rnn")), _ptr(g.get("g_mult")))
    if rc != 0:
        raise ValueError(f"rau_oracle_step failed rc={rc}")
    out.update(g)
    return out


def noise_clip_adam(x, g, m, v, noise, step_t, adam_t, lr, beta1=0.9, beta2=0.999,
                    eps=1e-8, eta=0.01, gamma=0.55, clip=0.1):
    """In-place update of one flat group; returns the pre-clip gradient norm."""
    dt = x.dtype
    suf = "f32" if dt == np.float32 else "f64"
    real = C.c_float if dt == np.float32 else C.c_double
    fn = getattr(lib(), "rau_oracle_noise_clip_adam_" + suf)
    fn.restype = None
    fn.argtypes = [C.c_size_t] + [C.c_void_p] * 5 + [C.c_int64, C.c_int64] + [real] * 7 + [C.c_void_p]
    norm = np.zeros(1, dt)
    fn(x.size, _ptr(x), _ptr(g), _ptr(m), _ptr(v), _ptr(noise), step_t, adam_t,
       lr, beta1, beta2, eps, eta, gamma, clip, _ptr(norm))
    return float(norm[0])
