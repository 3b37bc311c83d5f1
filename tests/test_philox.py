"""Dropout mask stream: oracle's Philox4x32-10 against an independent numpy
implementation and the published known-answer vectors of Random123."""
import numpy as np

import oracle

M0, M1 = 0xD2511F53, 0xCD9E8D57
W0, W1 = 0x9E3779B9, 0xBB67AE85


def philox4x32_10(ctr, key):
    c = [int(x) for x in ctr]
    k = [int(x) for x in key]
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k[0]) & 0xFFFFFFFF, p1 & 0xFFFFFFFF,
             ((p0 >> 32) ^ c[3] ^ k[1]) & 0xFFFFFFFF, p0 & 0xFFFFFFFF]
        k = [(k[0] + W0) & 0xFFFFFFFF, (k[1] + W1) & 0xFFFFFFFF]
    return c


def test_numpy_philox_known_answers():
    # Random123 kat_vectors: philox4x32-10
    assert philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2) == \
        [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
                         [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def keep_ref(seed, site, step, p, n):
    thr = int(round(p * 256))
    out = np.zeros(n, np.uint8)
    for blk in range((n + 15) // 16):
        o = philox4x32_10([blk & 0xFFFFFFFF, blk >> 32, site, step],
                          [seed & 0xFFFFFFFF, seed >> 32])
        for j in range(16):
            i = blk * 16 + j
            if i < n:
                out[i] = ((o[j >> 2] >> (8 * (j & 3))) & 0xFF) >= thr
    return out


def test_oracle_masks_match_numpy_philox():
    import ctypes as C
    for seed, site, step, p, n in [(7, 3, 0, 0.5, 1000), (2 ** 40 + 5, 1, 9, 0.25, 333),
                                   (123, 4, 2 ** 31, 0.5, 16)]:
        got = np.empty(n, np.uint8)
        oracle.lib().rau_oracle_fill_mask(seed, site, step, p, n, got.ctypes.data_as(C.c_void_p))
        assert np.array_equal(got, keep_ref(seed, site, step, p, n))


def test_mask_statistics():
    sh = oracle.Shapes(B=8, T=4, V=10, E=64, Rq=32, D=32, S=16, M=32, A=16, R=16, K=8, H=2)
    m = oracle.philox_masks(sh, seed=11, step=1)
    for k, v in m.items():
        assert abs(v.mean() - 0.5) < 0.03, k
    m2 = oracle.philox_masks(sh, seed=11, step=2)
    assert not np.array_equal(m["q"], m2["q"])          # step changes the stream
    assert np.array_equal(m["q"], oracle.philox_masks(sh, seed=11, step=1)["q"])
