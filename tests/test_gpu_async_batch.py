"""Asynchronous, double-buffered batch upload (rau_batch_slot / rau_set_batch_async / rau_use_batch):
what SS:434-439 does every iteration, behind the loader's prefetch
(utils/vqa_prepro_loader.lua:931-958).  Steps fed through alternating slots must give exactly the
results of the synchronous rau_set_batch on the same batches."""
import numpy as np
import pytest

from rau_vqa_amd import _lib as L
from rau_vqa_amd import synth
from tests import util

pytestmark = pytest.mark.gpu

DIMS = dict(B=12, T=9, V=300, E=200, Rq=64, D=72, S=196, M=136, A=132, R=68, K=1000, H=3)


def make(dims):
    from rau_vqa_amd.model import RAU, Config
    m = RAU(Config(**dims))
    m.init_uniform(seed=5)
    m.training()
    return m


def step(m, i, hop_w):
    m.set_dropout_seed(77, i)
    m.zero_grads()
    m.forward()
    out = m.outputs()
    m.backward(hop_w)
    g = m.get_grads()
    return {**out, **{"g_" + k: v for k, v in g.items()}}


def same(a, b):
    return all(np.array_equal(a[k], b[k]) for k in a)


def test_alternating_async_uploads_equal_synchronous_set_batch():
    d = DIMS
    batches = [synth.make_batch(d["B"], d["T"], d["V"], d["D"], d["S"], d["K"], seed=s, lens="ragged")
               for s in (1, 2, 3, 4, 5)]
    hop_w = np.full(d["H"], float(d["H"]), np.float32)
    m = make(d)
    want = []
    for i, b in enumerate(batches):
        m.set_batch(**b)
        want.append(step(m, i, hop_w))
    assert not same(want[0], want[1])                      # the batches do differ

    # same context, now through the two slots: batch i+1 is uploaded while step i is enqueued
    def fill(slot, b, in_place):
        if in_place:                                       # the loader writes the pinned staging itself
            v = m.batch_slot(slot)
            for k in ("feats", "tokens", "lens", "labels"):
                v[k][...] = np.asarray(b[k]).reshape(v[k].shape)
            m.set_batch_async(slot)
        else:                                              # or hands over its own arrays
            m.set_batch_async(slot, **b)
    fill(0, batches[0], True)
    for i in range(len(batches)):
        m.use_batch(i & 1)
        if i + 1 < len(batches):
            fill((i + 1) & 1, batches[i + 1], in_place=(i % 2 == 1))
        got = step(m, i, hop_w)
        assert same(got, want[i]), f"step {i} differs from the synchronous upload"
    # the synchronous call still works afterwards (it writes the current slot)
    m.set_batch(**batches[2])
    assert same(step(m, 2, hop_w), want[2])
    m.close()


def test_uploads_overlap_running_steps_without_any_host_sync():
    """the test above reads results after every step, so an upload never actually runs
    beside a step.  Here N steps go out back to back through alternating slots with NO host
    synchronisation in between (gradients accumulate over all of them: one zero_grads at the start),
    each next batch -- 19 MB of features -- enqueued for upload while the previous steps are still
    executing; only the final accumulated gradients and the last step's outputs are read.  They must
    equal, bit for bit, the same N steps fed by the synchronous rau_set_batch.  Exercises the
    consumed / uploaded event ordering with the host running ahead of the device."""
    d = dict(B=48, T=9, V=300, E=200, Rq=64, D=512, S=196, M=128, A=64, R=64, K=1000, H=3)
    N = 6
    batches = [synth.make_batch(d["B"], d["T"], d["V"], d["D"], d["S"], d["K"], seed=40 + s, lens="ragged")
               for s in range(N)]
    hop_w = np.full(d["H"], float(d["H"]), np.float32)

    def run(m, feed):
        m.zero_grads()
        for i in range(N):
            feed(m, i)
            m.set_dropout_seed(91, i)
            m.forward()
            m.backward(hop_w)
        out = m.outputs()                                  # first host synchronisation of the run
        g = m.get_grads()
        return {**out, **{"g_" + k: v for k, v in g.items()}}

    m = make(d)
    want = run(m, lambda m, i: m.set_batch(**batches[i]))

    def feed_async(m, i):
        if i == 0:
            m.set_batch_async(0, **batches[0])
        m.use_batch(i & 1)
        if i + 1 < N:
            k = (i + 1) & 1
            if i % 2 == 0:                                 # in place: rau_batch_slot before EVERY refill
                v = m.batch_slot(k)
                for key in ("feats", "tokens", "lens", "labels"):
                    v[key][...] = np.asarray(batches[i + 1][key]).reshape(v[key].shape)
                m.set_batch_async(k)
            else:
                m.set_batch_async(k, **batches[i + 1])
    got = run(m, feed_async)
    bad = [k for k in want if not np.array_equal(want[k], got[k])]
    assert not bad, f"asynchronously fed steps differ from the synchronous ones in {bad}"
    m.close()


def test_async_upload_argument_and_state_errors():
    d = dict(DIMS, B=4, H=1)
    m = make(d)
    with pytest.raises(L.RauError, match="holds no batch"):
        m.use_batch(1)
    with pytest.raises(L.RauError, match="slot"):
        m.use_batch(2)
    b = synth.make_batch(d["B"], d["T"], d["V"], d["D"], d["S"], d["K"], seed=9, lens="ragged")
    bad = dict(b, tokens=b["tokens"].copy())
    bad["tokens"][0, 0] = d["V"] + 1
    with pytest.raises(L.RauError, match="token"):       # ids are range-checked on the host, as in rau_set_batch
        m.set_batch_async(0, **bad)
    m.set_batch_async(0, **b)
    m.use_batch(0)
    m.forward()
    with pytest.raises(L.RauError, match="other slot"):   # the resident batch of an open forward pass
        m.set_batch_async(0, **b)
    m.set_batch_async(1, **b)                             # the other slot is free
    m.backward(np.ones(1, np.float32))
    m.close()


def test_inference_batches_without_labels_through_the_slots():
    d = dict(DIMS, B=6)
    m = make(d)
    m.evaluate()
    b = synth.make_batch(d["B"], d["T"], d["V"], d["D"], d["S"], d["K"], seed=3, lens="ragged")
    m.set_batch(b["feats"], b["tokens"], b["lens"])
    m.forward()
    want = m.logits()
    m.set_batch_async(1, b["feats"], b["tokens"], b["lens"], None, has_labels=False)
    m.use_batch(1)
    m.forward()
    assert np.array_equal(m.logits(), want)
    with pytest.raises(L.RauError, match="labels"):
        m.backward(np.ones(d["H"], np.float32))
    m.close()


from tests.test_loader import dataset, D as LD, W as LW, H as LH, T as LT   # noqa: E402,F401  (fixture)


def test_slot_feeder_on_device_equals_synchronous_feed(dataset):   # noqa: F811
    """The loader joined to the upload slots (loader.SlotFeeder) against loader.feed + rau_set_batch
    on the same synthetic dataset: identical per-step results over an epoch wrap."""
    from rau_vqa_amd import loader
    root, fdir, q, lens, feats = dataset
    B = 4
    dims = dict(B=B, T=LT, V=9, E=8, Rq=8, D=LD, S=LW * LH, M=8, A=8, R=8, K=12, H=2)
    hop_w = np.full(2, 2.0, np.float32)
    m = make(dims)
    ref = loader.load_data(str(root), batch_size=B).train_data
    want = []
    for i in range(8):
        loader.feed(m, ref.next_batch_feat(fdir, LD, LW, LH))
        want.append(step(m, i, hop_w))
    v = loader.load_data(str(root), batch_size=B)
    feeder = loader.SlotFeeder(m, v.train_data, fdir, LD, LW, LH)
    for i in range(8):
        got = step(m, i, hop_w)               # enqueue the step on the resident batch ...
        assert same(got, want[i]), i
        if i < 7:
            feeder.next()                     # ... then hand over the prefetched one
    m.close()
