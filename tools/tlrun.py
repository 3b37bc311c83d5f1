"""Dump a HIP-event timeline of one training step (development tool).
usage: RAU_PROF_TIMELINE=gpurun_out/tl.csv python tools/tlrun.py [B] [D]
RAU_TL_SPARSE=1 brackets only the bulk / weight-gradient streams and the chain's phase markers;
RAU_TL_MODE=eval dumps an evaluate-mode forward (the predict_result path) instead; RAU_TL_VARIANT=MS|Full"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rau_vqa_amd import synth
from rau_vqa_amd.model import RAU, Config, hop_weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
D = int(sys.argv[2]) if len(sys.argv) > 2 else 512
m = RAU(Config(B=B, D=D, dtype=os.environ.get('RAU_TL_DTYPE', 'f32')))
m.init_uniform(seed=123)
m.set_batch(**synth.make_batch(B, 26, 14000, D, 196, 1000, lens="full"))
EVAL = os.environ.get('RAU_TL_MODE') == 'eval'
m.evaluate() if EVAL else m.training()
w = hop_weights(os.environ.get('RAU_TL_VARIANT', 'SS'), 8)
def step(i):
    if EVAL:
        m.forward(); return
    m.set_dropout_seed(123, i); m.zero_grads(); m.forward(); m.backward(w)
for i in range(4): step(i)
m.sync()
import time
t=time.perf_counter()
for i in range(10): step(10+i)
m.sync()
print("ms/step", (time.perf_counter()-t)*100)
m.prof_reset(); m.prof_enable(2 if os.environ.get('RAU_TL_SPARSE') else True)
for i in range(2): step(100 + i)
m.sync(); m.prof(); m.prof_enable(False)
m.close()
