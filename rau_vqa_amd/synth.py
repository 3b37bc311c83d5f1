"""Synthetic batches and parameters of the shapes BASELINE.md section 2.1 names.

There is no dataset or checkpoint in the build image, so every test and the
benchmark run on seeded synthetic data that follows the loader's output contract
(reference utils/vqa_prepro_loader.lua:837-1010 ``next_batch_feat``):
``feats [B,D,14,14]`` float, ``x [T,B]`` token ids padded with 1 (ZEROPAD,
loader.lua:1393), ``x_len [B]``, ``y [B]`` answer ids in 1..K.
"""
from __future__ import annotations

import numpy as np


def make_batch(B, T, V, D, S, K, seed=123, lens="full", dtype=np.float32):
    """lens: "full" (all T, benchmark worst case) or "ragged" (uniform [min(3,T), T])."""
    rng = np.random.default_rng(seed)
    feats = (np.abs(rng.standard_normal((B, D, S))) * 0.5).astype(dtype)  # post-ReLU-like
    if isinstance(lens, str):
        if lens == "full":
            x_len = np.full(B, T, np.int32)
        elif lens == "ragged":
            x_len = rng.integers(min(3, T), T + 1, size=B).astype(np.int32)
        else:
            raise ValueError(lens)
    else:
        x_len = np.asarray(lens, np.int32)
    tokens = rng.integers(2, V + 1, size=(T, B)).astype(np.int32)
    tokens[np.arange(T)[:, None] >= x_len[None, :]] = 1
    labels = rng.integers(1, K + 1, size=B).astype(np.int32)
    return {"feats": feats, "tokens": tokens, "lens": x_len, "labels": labels}


def make_params(sizes, seed=123, lo=-0.08, hi=0.08, dtype=np.float32):
    """uniform(-0.08, 0.08) on each flat group (reference SS:352-354)."""
    rng = np.random.default_rng(seed + 1)
    return {k: rng.uniform(lo, hi, size=n).astype(dtype) for k, n in sizes.items()}


def make_masks(mask_shapes, probs, seed=123):
    """Explicit Bernoulli keep masks (uint8) for the five dropout sites."""
    rng = np.random.default_rng(seed + 2)
    return {k: (rng.random(shape) >= probs[k]).astype(np.uint8) for k, shape in mask_shapes.items()}
