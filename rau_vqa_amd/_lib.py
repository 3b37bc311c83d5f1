"""ctypes binding of librau.so (the C ABI declared in include/rau.h).

The library is the product: if it is missing, or no gfx950 device is usable,
loading / rau_create raise -- there is no CPU or PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# Nothing here touches the process environment.  (rau_graph_step replays fastest with the HIP runtime
# variable DEBUG_HIP_FORCE_GRAPH_QUEUES=2; a host that wants that sets it itself
# before its first HIP call -- `bench.py --graph` does, for its own process only.)
# RAU_LIB overrides the library path (A/B runs of two builds on one GPU box)
LIB_PATH = os.environ.get("RAU_LIB") or os.path.join(_HERE, "librau.so")
CSRC = os.path.join(_HERE, "csrc")

GROUP_EMBED, GROUP_RNN, GROUP_MULT = 0, 1, 2
GROUPS = {"embed": GROUP_EMBED, "rnn": GROUP_RNN, "mult": GROUP_MULT}
MODE_TRAIN, MODE_EVAL = 0, 1
MASK_SITES = {"we": 0, "rnn": 1, "q": 2, "x": 3, "mf": 4}


class RauConfig(C.Structure):
    """Mirror of ``rau_config`` (include/rau.h)."""
    _fields_ = [(n, C.c_int32) for n in
                ("B", "T", "V", "E", "Rq", "D", "S", "M", "A", "R", "K", "H")] + \
               [(n, C.c_float) for n in ("p_we", "p_rnn", "p_q", "p_x", "p_mf")] + \
               [("dtype", C.c_int32), ("device_id", C.c_int32)]


class RauError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compile librau.so for gfx950 with the in-tree Makefile (hipcc)."""
    args = ["make", "-C", CSRC, "-j4"]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    if not os.path.exists(LIB_PATH):
        raise RauError("build did not produce " + LIB_PATH)
    return LIB_PATH


_SIGS = {
    # name: (restype, argtypes)
    "rau_default_config": (None, [C.POINTER(RauConfig)]),
    "rau_last_error": (C.c_char_p, []),
    "rau_abi_version": (C.c_int, []),
    "rau_create": (C.c_int, [C.POINTER(RauConfig), C.POINTER(C.c_void_p)]),
    "rau_destroy": (None, [C.c_void_p]),
    "rau_params": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                             C.POINTER(C.c_size_t)]),
    "rau_layout_count": (C.c_int, [C.c_void_p, C.c_int]),
    "rau_layout_entry": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_char_p),
                                   C.POINTER(C.c_size_t), C.POINTER(C.c_int32),
                                   C.POINTER(C.c_int32)]),
    "rau_set_params": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
    "rau_get_params": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
    "rau_get_grads": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
    "rau_set_grads": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
    "rau_init_uniform": (C.c_int, [C.c_void_p, C.c_uint64, C.c_float, C.c_float]),
    "rau_zero_grads": (C.c_int, [C.c_void_p]),
    "rau_set_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "rau_set_dropout_seed": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32]),
    "rau_set_mask": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
    "rau_get_mask": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
    "rau_set_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rau_batch_feats": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "rau_batch_slot": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "rau_set_batch_async": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int]),
    "rau_use_batch": (C.c_int, [C.c_void_p, C.c_int]),
    "rau_forward": (C.c_int, [C.c_void_p]),
    "rau_backward": (C.c_int, [C.c_void_p, C.c_void_p]),
    # module-level entry points: device pointers in, pointers to ctx-owned slots out
    "rau_embed_forward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "rau_embed_backward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "rau_deeplstm_forward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                       C.POINTER(C.c_void_p)]),
    "rau_deeplstm_backward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "rau_multimodal_forward": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 4 +
                               [C.POINTER(C.c_void_p)] * 5),
    "rau_multimodal_backward": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 9 +
                                [C.POINTER(C.c_void_p)] * 4),
    "rau_criterion_forward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                        C.POINTER(C.c_float)]),
    "rau_criterion_backward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float,
                                         C.POINTER(C.c_void_p)]),
    # device tensors (the tensor algebra feval does between module calls)
    "rau_dev_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "rau_dev_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rau_dev_fill": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_float]),
    "rau_dev_copy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rau_dev_axpy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float]),
    "rau_dev_scale": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_float]),
    "rau_dev_addcmul": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rau_dev_addcdiv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rau_dev_sqrt": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "rau_dev_add_scalar": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_float]),
    "rau_dev_adam": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                               C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32]),
    "rau_dev_select_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                      C.c_void_p, C.c_int32]),
    "rau_dev_rowmax": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                 C.c_void_p]),
    "rau_dev_sum": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_double)]),
    "rau_dev_count_eq": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                   C.POINTER(C.c_int32)]),
    "rau_dev_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rau_dev_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rau_graph_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "rau_sync": (C.c_int, [C.c_void_p]),
    "rau_get_losses": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rau_get_argmax": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rau_get_logits": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rau_get_dopred": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rau_get_attention": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rau_get_question_state": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rau_get_att_state": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "rau_noise_clip_adam": (C.c_int, [C.c_void_p, C.c_int64] + [C.c_float] * 8 +
                            [C.c_uint64, C.c_void_p]),
    "rau_stream": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "rau_wait_grads": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "rau_comm_unique_id": (C.c_int, [C.c_void_p, C.c_size_t]),
    "rau_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]),
    "rau_allreduce_grads": (C.c_int, [C.c_void_p]),
    "rau_comm_destroy": (C.c_int, [C.c_void_p]),
    "rau_timer_begin": (C.c_int, [C.c_void_p]),
    "rau_timer_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "rau_prof_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "rau_prof_reset": (C.c_int, [C.c_void_p]),
    "rau_prof_count": (C.c_int, [C.c_void_p]),
    "rau_prof_entry": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p),
                                 C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                 C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "rau_split_guard_check": (C.c_int, [C.c_size_t, C.c_size_t, C.c_int, C.c_size_t]),
    "rau_enc_ws_coresident": (C.c_int, [C.c_int, C.c_int, C.c_int]),
}

_lib = None


def lib() -> C.CDLL:
    """Load librau.so and bind every symbol include/rau.h declares."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RauError(f"{LIB_PATH} not found: build it with "
                           "`python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)")
        # PyTorch wheels bundle their own libamdhip64; if librau pulled in the system copy
        # first, a later `import torch` would bring up a SECOND HIP runtime in the process,
        # which then finds no GPU.  Loading torch first makes both share one runtime.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            if os.environ.get("RAU_LIB") and not hasattr(l, name):
                continue           # A/B runs against an older build (development only)
            fn = getattr(l, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        raise RauError(f"librau error {rc}: {lib().rau_last_error().decode()}")
