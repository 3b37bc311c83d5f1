import sys, time
sys.path.insert(0, '.')
import numpy as np
from rau_vqa_amd import synth
from rau_vqa_amd.model import RAU, Config, hop_weights
cfg = Config(B=256, T=26, V=14000, D=512, H=8)
m = RAU(cfg); m.init_uniform(123)
m.set_batch(**synth.make_batch(cfg.B, cfg.T, cfg.V, cfg.D, cfg.S, cfg.K, lens="full"))
m.training(); w = hop_weights("SS", 8)
for i in range(3):
    m.set_dropout_seed(1, i); m.zero_grads(); m.forward(); m.backward(w)
m.sync()
enq = []; tot = []
for i in range(10):
    t0 = time.perf_counter()
    m.set_dropout_seed(1, i); m.zero_grads(); m.forward()
    t1 = time.perf_counter()
    m.backward(w)
    t2 = time.perf_counter()
    m.sync()
    t3 = time.perf_counter()
    enq.append((t1 - t0, t2 - t1)); tot.append(t3 - t0)
print("enqueue fwd ms", np.median([e[0] for e in enq]) * 1e3, "bwd ms", np.median([e[1] for e in enq]) * 1e3, "total ms", np.median(tot) * 1e3)
