"""Repeat one seeded training step at the model's real dimensions and compare every output and
gradient bitwise with the first run (development tool: a data race in a bulk kernel shows up as a
run that differs).  usage: python tools/stress_wide.py [B] [runs] [f32|bf16] [D]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rau_vqa_amd import synth
from rau_vqa_amd.model import RAU, Config

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dtype = sys.argv[3] if len(sys.argv) > 3 else "f32"
D = int(sys.argv[4]) if len(sys.argv) > 4 else 512
dims = dict(B=B, T=26, V=14000, E=200, Rq=512, D=D, S=196, M=512, A=256, R=512, K=1000, H=8)
m = RAU(Config(dtype=dtype, **dims))
m.init_uniform(seed=123)
batch = synth.make_batch(B, 26, 14000, D, 196, 1000, lens="ragged")
m.set_batch(**batch)
m.training()
hop_w = np.full(8, 8.0, np.float32)

def run():
    m.set_dropout_seed(5, 1)
    m.zero_grads()
    m.forward()
    out = m.outputs()
    m.backward(hop_w)
    g = m.get_grads()
    return {**out, **{"g_" + k: v for k, v in g.items()}}

ref = run()
bad = 0
for i in range(runs):
    r = run()
    diff = [k for k in ref if not np.array_equal(ref[k], r[k])]
    if diff:
        bad += 1
        print("run", i, "differs in", diff, flush=True)
print(f"B={B} {dtype} D={D}: {bad} of {runs} runs differ from the first")
m.close()
sys.exit(1 if bad else 0)
