"""Diagnostic (not a test): print per-tensor relative errors GPU vs oracle for one shape."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import oracle  # noqa: E402
from tests import util  # noqa: E402
from tests.test_gpu_parity import run_gpu  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "SMALL"
mode = sys.argv[2] if len(sys.argv) > 2 else "train"
sh = util.shapes(getattr(util, name))
batch, params, masks = util.make_problem(sh, scale=0.5 if name != "MEDIUM" else 0.2)
hop_w = np.full(sh.H, float(sh.H), np.float32)
t0 = time.time()
ref = oracle.step(sh, params, batch["feats"], batch["tokens"], batch["lens"], batch["labels"],
                  masks if mode == "train" else None, hop_w, dtype=np.float64)
t1 = time.time()
got, layouts = run_gpu(sh, batch, params, masks, hop_w, mode)
t2 = time.time()
print(f"{name} {mode}: oracle {t1-t0:.2f}s gpu {t2-t1:.2f}s")
for k in util.OUT_KEYS:
    print(f"  {k:28s} rel {util.rel_err(got[k], ref[k]):.3e}  ref max {np.max(np.abs(ref[k])):.3e}")
print("  argmax equal:", np.array_equal(got["argmax"], ref["argmax"]))
for grp in ("embed", "rnn", "mult"):
    for n, sl in util.layer_slices(layouts[grp]):
        r = ref["g_" + grp][sl]
        print(f"  d {n:36s} rel {util.rel_err(got['g_' + grp][sl], r):.3e}  ref max {np.max(np.abs(r)):.3e}")
