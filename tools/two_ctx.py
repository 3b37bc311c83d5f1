"""Experiment: N independent contexts of B/N samples each, stepped concurrently from N host threads
(one GPU), against one context of B samples.  usage: python tools/two_ctx.py [B] [D] [N]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rau_vqa_amd import synth
from rau_vqa_amd.model import RAU, Config, hop_weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
D = int(sys.argv[2]) if len(sys.argv) > 2 else 512
N = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dt = os.environ.get('RAU_TL_DTYPE', 'f32')
w = hop_weights("SS", 8)
def make(b):
    m = RAU(Config(B=b, D=D, dtype=dt))
    m.init_uniform(seed=123)
    m.set_batch(**synth.make_batch(b, 26, 14000, D, 196, 1000, lens="full"))
    m.training()
    return m
def run(m, n, off):
    for i in range(n):
        m.set_dropout_seed(123, off + i); m.zero_grads(); m.forward(); m.backward(w)
    m.sync()
def timed(ms, n):
    ts = [threading.Thread(target=run, args=(m, n, 10)) for m in ms]
    t = time.perf_counter()
    for x in ts: x.start()
    for x in ts: x.join()
    return (time.perf_counter() - t) / n * 1e3
one = make(B)
run(one, 4, 0)
print(f"1 x B={B}: {timed([one], 10):.3f} ms/step")
one.close()
ms = [make(B // N) for _ in range(N)]
for m in ms: run(m, 4, 0)
print(f"{N} x B={B // N} concurrent: {timed(ms, 10):.3f} ms per {B} samples")
print(f"1 x B={B // N} alone: {timed(ms[:1], 10):.3f} ms/step")
for m in ms: m.close()
