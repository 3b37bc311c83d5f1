"""Host enqueue time of a training step against its device time (development tool).
usage: python tools/hosttime.py [B] [D]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rau_vqa_amd import synth
from rau_vqa_amd.model import RAU, Config, hop_weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
D = int(sys.argv[2]) if len(sys.argv) > 2 else 512
m = RAU(Config(B=B, D=D))
m.init_uniform(seed=123)
m.set_batch(**synth.make_batch(B, 26, 14000, D, 196, 1000, lens="full"))
m.training()
w = hop_weights("SS", 8)
def step(i):
    m.set_dropout_seed(123, i); m.zero_grads(); m.forward(); m.backward(w)
for i in range(5): step(i)
m.sync()
N = 20
t0 = time.perf_counter()
for i in range(N): step(10 + i)
t1 = time.perf_counter()
m.sync()
t2 = time.perf_counter()
print(f"B={B}: host enqueue {1e3*(t1-t0)/N:.3f} ms/step, enqueue+drain {1e3*(t2-t0)/N:.3f} ms/step")
# one step at a time: enqueue, then wait
hs, ds = [], []
for i in range(N):
    a = time.perf_counter(); step(40 + i); b = time.perf_counter(); m.sync(); c = time.perf_counter()
    hs.append(b - a); ds.append(c - a)
print(f"  single steps: host {1e3*sum(hs)/N:.3f} ms, until idle {1e3*sum(ds)/N:.3f} ms")
m.close()
