"""Development probe: evaluate-mode forward with N contexts in flight (independent batches on their own
streams).  usage: python tools/exp_infer2.py [batch] [nctx]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # before librau.so
from rau_vqa_amd import synth
from rau_vqa_amd.model import RAU, Config

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfgd = dict(B=B, T=26, V=14000, E=200, Rq=512, D=512, S=196, M=512, A=256, R=512, K=1000, H=8)
ms = []
for i in range(N):
    m = RAU(Config(device_id=0, dtype="f32", **cfgd))
    if i == 0:
        m.init_uniform(seed=123)
    else:
        m.set_params(ms[0].get_params())
    m.set_batch(**synth.make_batch(B, 26, 14000, 512, 196, 1000, seed=123 + i, lens="full"))
    m.evaluate()
    ms.append(m)
def fence():
    for m in ms:
        m.sync()
for n in range(1, N + 1):
    act = ms[:n]
    for _ in range(3):
        for m in act:
            m.forward()
    fence()
    t = time.perf_counter()
    R = 20
    for _ in range(R):
        for m in act:
            m.forward()
    fence()
    dt = time.perf_counter() - t
    print(f"contexts in flight {n}: {dt / R * 1e3:.3f} ms per round, {B * n * R / dt:.0f} QA-pairs/s", flush=True)
