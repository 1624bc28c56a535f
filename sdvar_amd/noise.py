"""Counter-based sampling noise (Philox4x32-10), the host statement of the stream the HIP sampler generates.

The reference draws tokens with torch.multinomial (/root/reference/models/helpers.py:19), which is exactly
argmax(p / q) with q ~ Exp(1) of shape (B*l, V).  torch's CPU exponential_ stream depends on the
host CPU vendor (MKL VSL on Intel, mt19937 elsewhere) and on CUDA/HIP devices it is a different Philox layout, so
token ids can only be compared across machines when q is an explicit, portable input.  This module defines it:

    counter = (v >> 2, token, image_global, draw)   key = (seed_lo, seed_hi ^ 0x5D5A17AB)
    x       = Philox4x32-10(counter, key)[v & 3]
    u       = ((x >> 9) + 0.5) * 2^-23          in (0, 1), exactly representable in float32
    q       = -log(u)                           (host: float64 log rounded to float32)

`draw` is the index of the sampler call in the run (one per drafted stage), `image_global` the index of the image in
the whole job, so shards of a batch generate the same noise a single process would.
"""
from __future__ import annotations

import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_KEY_XOR = 0x5D5A17AB
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0: int, k1: int):
    """Vectorised Philox4x32 with 10 rounds; c* are uint32 arrays (broadcastable); returns 4 uint32 arrays."""
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint64) for c in np.broadcast_arrays(c0, c1, c2, c3)]
    k0 &= 0xFFFFFFFF; k1 &= 0xFFFFFFFF
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)), lo1, (hi0 ^ c3 ^ np.uint64(k1)), lo0
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return [c.astype(np.uint32) for c in (c0, c1, c2, c3)]


def exponential_noise(seed: int, draw: int, B: int, l: int, V: int, image_offset: int = 0) -> np.ndarray:
    """q of shape (B, l, V) float32 for sampler call number `draw` (images image_offset .. image_offset+B-1)."""
    assert V % 4 == 0
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    k0, k1 = seed & 0xFFFFFFFF, ((seed >> 32) ^ _KEY_XOR) & 0xFFFFFFFF
    v4 = np.arange(V // 4, dtype=np.uint32)[None, None, :]
    tok = np.arange(l, dtype=np.uint32)[None, :, None]
    img = (np.arange(B, dtype=np.uint32) + np.uint32(image_offset))[:, None, None]
    x = philox4x32_10(v4, tok, img, np.uint32(draw), k0, k1)
    x = np.stack(x, axis=-1).reshape(B, l, V)
    u = ((x >> np.uint32(9)).astype(np.float64) + 0.5) * (2.0 ** -23)
    return (-np.log(u)).astype(np.float32)
