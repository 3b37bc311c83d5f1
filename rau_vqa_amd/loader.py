"""Host-side batch producer with the reference loader's contract.

Restates utils/vqa_prepro_loader.lua of the reference:
  * ``load_data`` (lines 1294-1473): question tensors + vocabulary from ``data_prepro.h5`` /
    ``data_prepro.json``; word ids are shifted by one so that id 1 is ZEROPAD (1335, 1393);
  * batch order options 1-4 (1219-1291): shuffle / inorder / sort by length / randsort;
  * ``next_batch_feat`` (837-1010): returns ``feats [B,D,W,H]``, ``x [T,B]`` (transposed),
    ``x_len [B]``, answers ``[B]`` (train) or multiple-choice ids ``[B,nMC]`` (test) and
    ``qids [B]``; per-image features are ``torch.save``d FloatTensors named after the image
    (``COCO_*.t7``), looked up in ``tab_featpaths[datatype]``; the order is re-drawn when fewer
    than one batch is left; with prefetch the NEXT batch's feature files are read by one
    background worker while the current batch is being used (the reference's ``threads`` pool).

What differs, and why: the HDF5 question file needs ``h5py``, which is not installed in the build
image -- ``load_data`` reads it when h5py is importable and otherwise an ``.npz`` with the same
dataset names; Torch's ``randperm`` / unstable ``sort`` streams cannot be reproduced, numpy's
seeded generator and a stable sort are used instead (same distribution, different draws).
"""
from __future__ import annotations

import json
import os
import threading
from dataclasses import dataclass

import numpy as np

from . import t7


@dataclass
class QuestionSet:
    question: np.ndarray      # [N, T] int32, ids already +1 (1 = ZEROPAD)
    lengths_q: np.ndarray     # [N]
    img_list: np.ndarray      # [N] 1-based index into the image-name list
    question_id: np.ndarray   # [N]
    answers: np.ndarray = None    # [N] 1-based answer id (train)
    mc_ans: np.ndarray = None     # [N, nMC] (test)
    datatype: np.ndarray = None   # [N] 1-based index into tab_featpaths


def feature_name(img_path: str) -> str:
    """'train2014/COCO_train2014_000000357413.jpg' -> 'COCO_train2014_000000357413.t7'."""
    base = os.path.basename(img_path)
    stem, ext = os.path.splitext(base)
    return stem + ".t7" if ext else base + ".t7"


class DataClass:
    """One split's batch iterator (the reference's ``dataclass``)."""

    def __init__(self, qs: QuestionSet, img_names, batch_size, split="train", prefetch=False,
                 seed=123):
        self.qs, self.img_names, self.batch_size, self.split = qs, list(img_names), batch_size, split
        self.n = int(qs.question.shape[0])
        if self.n < batch_size:
            raise ValueError(f"{self.n} examples < batch_size {batch_size}")
        self.seq_len = int(qs.question.shape[1])
        self.opt_prefetch = prefetch
        self.opt_batch_order = 2
        self.rng = np.random.default_rng(seed)
        self.batch_index = 0
        self.batch_order = np.arange(self.n)
        self._job = None          # (key, thread, result holder)
        self._next_dest = None    # SlotFeeder: where the prefetch worker assembles the next batch

    # ---- batch order options, loader.lua:1219-1291
    def set_batch_order_option(self, opt):
        if opt not in (1, 2, 3, 4):
            raise ValueError("batch order option must be 1 (shuffle), 2 (inorder), 3 (sort), 4 (randsort)")
        self.opt_batch_order = opt

    def reorder(self):
        self.batch_index = 0
        self._job = None
        lens = np.asarray(self.qs.lengths_q)
        if self.opt_batch_order == 1:
            self.batch_order = self.rng.permutation(self.n)
        elif self.opt_batch_order == 2:
            self.batch_order = np.arange(self.n)
        elif self.opt_batch_order == 3:
            self.batch_order = np.argsort(lens, kind="stable")
        else:   # sorted by length, ties shuffled
            order = np.argsort(lens, kind="stable")
            s = lens[order]
            i = 0
            while i < self.n:
                j = i
                while j < self.n and s[j] == s[i]:
                    j += 1
                order[i:j] = order[i:j][self.rng.permutation(j - i)]
                i = j
            self.batch_order = order

    def reset_batch_pointer(self):
        self.batch_index = 0

    # ---- features
    def _paths(self, start, tab_featpaths):
        idx = self.batch_order[start:start + self.batch_size]
        dt = self.qs.datatype[idx] if self.qs.datatype is not None else np.ones(len(idx), int)
        return [os.path.join(tab_featpaths[int(d) - 1],
                             feature_name(self.img_names[int(self.qs.img_list[i]) - 1]))
                for i, d in zip(idx, dt)]

    @staticmethod
    def _load_feats(paths, D, W, H, out=None):
        """Per-image feature files into one [B,D,W,H] array; `out` = a caller-owned destination
        (the pinned staging of an upload slot: the batch is assembled where the H2D copy reads it)."""
        if out is None:
            out = np.zeros((len(paths), D, W, H), np.float32)
        else:
            out = out.reshape(len(paths), D, W, H)
        for i, p in enumerate(paths):
            out[i] = t7.load_feature(p, D, W, H).reshape(D, W, H)   # asserts the three sizes
        return out

    def _start_prefetch(self, tab_featpaths, D, W, H):
        paths = self._paths(self.batch_index, tab_featpaths)
        holder = {}
        dest = self._next_dest() if self._next_dest is not None else None

        def work():
            try:
                holder["feats"] = self._load_feats(paths, D, W, H, dest)
            except Exception as e:   # surfaced on the consumer side
                holder["error"] = e
        th = threading.Thread(target=work, daemon=True)
        th.start()
        self._job = ((self.batch_index, tuple(paths)), th, holder)

    def next_batch_feat(self, tab_featpaths, feat_dim, feat_w=1, feat_h=1):
        """-> feats [B,D,W,H] f32, x [T,B] i32, x_len [B] i32, a [B] | [B,nMC] i32, qids [B]."""
        if isinstance(tab_featpaths, (str, os.PathLike)):
            tab_featpaths = [tab_featpaths]
        B = self.batch_size
        idx = self.batch_order[self.batch_index:self.batch_index + B]
        paths = self._paths(self.batch_index, tab_featpaths)
        feats = None
        if self.opt_prefetch and self._job is not None:
            key, th, holder = self._job
            th.join()                                    # pool:synchronize()
            if "error" in holder:
                raise holder["error"]
            if key == (self.batch_index, tuple(paths)):
                feats = holder["feats"]
        if feats is None:
            feats = self._load_feats(paths, feat_dim, feat_w, feat_h)
        x = np.ascontiguousarray(self.qs.question[idx].T, np.int32)          # transpose(1,2)
        x_len = np.ascontiguousarray(self.qs.lengths_q[idx], np.int32)
        qids = np.ascontiguousarray(self.qs.question_id[idx])
        src = self.qs.answers if self.split == "train" else self.qs.mc_ans
        a = np.ascontiguousarray(src[idx], np.int32)
        self.batch_index += B
        if self.batch_index + B > self.n:                # loader.lua:911-913
            self.reorder()
        if self.opt_prefetch:
            self._start_prefetch(tab_featpaths, feat_dim, feat_w, feat_h)
        return feats, x, x_len, a, qids


class VqaData:
    """What ``vqa_prepro_loader.load_data`` returns: vocabulary + train / test iterators."""


def _read_questions(vqa_dir):
    h5 = os.path.join(vqa_dir, "data_prepro.h5")
    npz = os.path.join(vqa_dir, "data_prepro.npz")
    if os.path.exists(h5):
        try:
            import h5py   # not in the build image; used when present
        except ImportError as e:
            raise RuntimeError("data_prepro.h5 needs h5py, which is not installed; convert it to "
                               "data_prepro.npz with the same dataset names") from e
        with h5py.File(h5, "r") as f:
            return {k: np.asarray(f[k]) for k in f.keys()}
    with np.load(npz, allow_pickle=False) as f:
        return {k: f[k] for k in f.files}


def load_data(vqa_dir, batch_size, prefetch=False, test_batch_size=None, seed=123):
    """loader.lua:1294-1473 (without the valid_ratio split)."""
    with open(os.path.join(vqa_dir, "data_prepro.json")) as f:
        info = json.load(f)
    d = _read_questions(vqa_dir)
    one = lambda n: np.ones(n, np.int64)
    train = QuestionSet(question=d["ques_train"].astype(np.int32) + 1,     # zero padding -> 1
                        lengths_q=d["ques_length_train"], img_list=d["img_pos_train"],
                        question_id=d["question_id_train"], answers=d["answers"],
                        datatype=d.get("datatype_train", one(len(d["answers"]))))
    test = QuestionSet(question=d["ques_test"].astype(np.int32) + 1,
                       lengths_q=d["ques_length_test"], img_list=d["img_pos_test"],
                       question_id=d["question_id_test"], mc_ans=d["MC_ans_test"],
                       datatype=one(len(d["question_id_test"])))
    v = VqaData()
    as_list = lambda m: [m[k] for k in sorted(m, key=int)] if isinstance(m, dict) else list(m)
    v.img_train, v.img_test = as_list(info["unique_img_train"]), as_list(info["unique_img_test"])
    v.vocab_dict = {1: "ZEROPAD", **{int(i) + 1: w for i, w in info["ix_to_word"].items()}}
    v.vocab_map = {w: i for i, w in v.vocab_dict.items()}
    v.answer_dict = {int(i): w for i, w in info["ix_to_ans"].items()}
    v.answer_map = {w: i for i, w in v.answer_dict.items()}
    v.vocab_size = len(info["ix_to_word"]) + 1            # including ZEROPAD
    v.answer_size = len(info["ix_to_ans"])
    v.seq_len = v.max_sentence_len = int(train.question.shape[1])
    v.train_data = DataClass(train, v.img_train, batch_size, "train", prefetch, seed)
    v.test_data = DataClass(test, v.img_test, test_batch_size or batch_size, "test", prefetch, seed)
    return v


def feed(rau, batch):
    """next_batch_feat's tuple -> rau_set_batch (the H2D of SS:434-439); returns qids."""
    feats, x, x_len, a, qids = batch
    B, D = feats.shape[0], feats.shape[1]
    labels = a if a.ndim == 1 else None                   # test batches carry MC ids, no labels
    rau.set_batch(feats.reshape(B, D, -1), x, x_len, labels)
    return qids


class SlotFeeder:
    """The loader's prefetch joined to the ctx's two upload slots (rau_batch_slot /
    rau_set_batch_async / rau_use_batch): what SS:434-439 + vqa_prepro_loader.lua:931-958 do every
    iteration, without a host copy or a host wait on the step's path.

    The prefetch worker reads the NEXT batch's feature files straight into the pinned staging of
    the slot the device is not using; ``next()`` (called right after the current step has been
    enqueued) hands that slot to the copy stream -- the 100 MB transfer runs under the step still
    executing -- makes it the resident batch for the following step, and points the worker at the
    slot just left.  Usage::

        feeder = SlotFeeder(rau, data, featdir, D, W, H)      # batch 0 resident on return
        for it in range(n):
            rau.forward(); rau.backward(w); ...              # enqueue step `it`
            qids = feeder.next()                              # batch it+1 resident for the next step
    """

    def __init__(self, rau, data: DataClass, tab_featpaths, feat_dim, feat_w=1, feat_h=1):
        self.rau, self.data = rau, data
        self.args = (tab_featpaths, feat_dim, feat_w, feat_h)
        self.slot = 0                  # the slot the NEXT batch is assembled in
        data.opt_prefetch = True
        data._job = None               # any batch prefetched before now went to ordinary memory
        data._next_dest = lambda: self.rau.batch_slot(self.slot)["feats"]
        self.qids = self._advance()    # batch 0: read synchronously (nothing to overlap with yet)

    def _advance(self):
        d, rau, s = self.data, self.rau, self.slot
        view = rau.batch_slot(s)                      # (host-waits until the slot's last upload has left)
        self.slot = s ^ 1                             # the worker started by next_batch_feat fills the other
        feats, x, x_len, a, qids = d.next_batch_feat(*self.args)
        if not np.shares_memory(feats, view["feats"]):   # first batch / a re-drawn order: not prefetched in place
            view["feats"][...] = feats.reshape(view["feats"].shape)
        view["tokens"][...] = x
        view["lens"][...] = x_len
        labels = a.ndim == 1                          # test batches carry MC ids, no labels
        if labels:
            view["labels"][...] = a
        rau.set_batch_async(s, has_labels=labels)     # staging filled in place: no host copy
        rau.use_batch(s)
        return qids

    def next(self):
        """Upload the prefetched batch and make it resident; returns its question ids."""
        self.qids = self._advance()
        return self.qids
