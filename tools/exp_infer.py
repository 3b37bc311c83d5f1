"""Development probe: evaluate-mode forward throughput.  usage: python tools/exp_infer.py [batch] [dtype]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # before librau.so
from rau_vqa_amd import synth
from rau_vqa_amd.model import RAU, Config

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dt = sys.argv[2] if len(sys.argv) > 2 else "f32"
D = int(sys.argv[3]) if len(sys.argv) > 3 else 512
m = RAU(Config(device_id=0, dtype=dt, B=B, T=26, V=14000, E=200, Rq=512, D=D, S=196, M=512, A=256, R=512, K=1000, H=8))
m.init_uniform(seed=123)
m.set_batch(**synth.make_batch(B, 26, 14000, D, 196, 1000, seed=123, lens="full"))
m.evaluate()
for _ in range(3):
    m.forward()
m.sync()
best = 1e9
for rep in range(3):
    t = time.perf_counter()
    for _ in range(20):
        m.forward()
    m.sync()
    best = min(best, (time.perf_counter() - t) / 20)
print(f"B={B} {dt} D={D}: {best * 1e3:.3f} ms per batch, {B / best:.0f} QA-pairs/s", flush=True)
