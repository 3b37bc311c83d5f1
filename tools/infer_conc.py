"""Evaluate-mode forward throughput: one context vs several contexts stepped from their own host
threads (development tool).  usage: python tools/infer_conc.py [B] [nctx]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rau_vqa_amd import synth
from rau_vqa_amd.model import RAU, Config
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
NC = int(sys.argv[2]) if len(sys.argv) > 2 else 2
N = 20
ms = []
for i in range(NC):
    m = RAU(Config(B=B, D=512))
    m.init_uniform(seed=123)
    m.set_batch(**synth.make_batch(B, 26, 14000, 512, 196, 1000, lens="full"))
    m.evaluate()
    for _ in range(3): m.forward()
    m.sync()
    ms.append(m)
t = time.perf_counter()
for _ in range(N): ms[0].forward()
ms[0].sync()
one = (time.perf_counter() - t) / N
print(f"B={B}: one context {one*1e3:.3f} ms/batch = {B/one:.0f} QA/s")
def work(m):
    for _ in range(N): m.forward()
    m.sync()
th = [threading.Thread(target=work, args=(m,)) for m in ms]
t = time.perf_counter()
for x in th: x.start()
for x in th: x.join()
dt = time.perf_counter() - t
print(f"B={B}: {NC} contexts concurrently {dt/N/NC*1e3:.3f} ms/batch = {B*N*NC/dt:.0f} QA/s")
for m in ms: m.close()
