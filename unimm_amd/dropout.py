"""Counter-based dropout shared by every kernel (one hash per two neighbouring columns).

The kernels never store a mask: the backward pass regenerates it from (key, thr).  `key` is one
32-bit word per (seed, step, site); `keep_mask` is the bit-exact host mirror of the device hash
(unimm_amd/csrc/common.h: mix32) so that tests can replay the device masks inside the CPU oracle."""
from __future__ import annotations

import numpy as np

_M1, _M2 = 0x7FEB352D, 0x846CA68B


def mix32_int(x: int) -> int:
    x &= 0xFFFFFFFF
    x ^= x >> 16
    x = (x * _M1) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * _M2) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def site_key(seed: int, site: int) -> int:
    """The per-(seed, site) part of a dropout key: what a launch carries as its argument."""
    return mix32_int(mix32_int(seed * 0x9E3779B1 + 0x51ED270B) ^ (site * 0x85EBCA77 + 0x165667B1))


def step_salt(seed: int, step: int) -> int:
    """The per-step part: one word for the whole step, XORed into every site's key.  A replayed launch sequence reads it
    from device memory (its arguments are frozen at capture); an eager launch gets the combined key as its argument."""
    return mix32_int(mix32_int(seed * 0xC2B2AE3D + 0x27D4EB2F) + step * 0x9E3779B1)


def make_key(seed: int, step: int, site: int) -> int:
    """One word per dropout site and step (site ids are assigned by the engine)."""
    return site_key(seed, site) ^ step_salt(seed, step)


def drop_arg(p: float, key: int):
    """(key, thr, scale) triple the C ABI takes; p == 0 disables."""
    if p <= 0.0:
        return (0, 0, 1.0)
    thr = min(int(p * 4294967296.0), 0xFFFFFFFF)
    return (key & 0xFFFFFFFF, thr, 1.0 / (1.0 - p))


def keep_mask2d(key: int, thr: int, rows: int, cols: int) -> np.ndarray:
    """Host mirror of the device rule (unimm_amd/csrc/common.h): element (row, col) of a [rows, cols]
    activation is kept iff the 16-bit field (col & 1) of drop_word(key, w = row * ceil(cols / 2) + (col >> 1)) -- the affine stage
    w * M1 + key followed by one xorshift / multiply / xorshift round with key * 0x9E3779B1 XORed in before the multiply -- is
    >= thr >> 16.  Returns a boolean [rows, cols] array."""
    half = (cols + 1) // 2
    r = np.arange(rows, dtype=np.uint64)[:, None]
    c = np.arange(cols, dtype=np.uint64)[None, :]
    key &= 0xFFFFFFFF
    w = (r * np.uint64(half) + (c >> np.uint64(1))) & np.uint64(0xFFFFFFFF)
    x = (w * np.uint64(_M1) + np.uint64(key)) & np.uint64(0xFFFFFFFF)             # drop_lin: affine in the word index
    x ^= x >> np.uint64(15)                                                         # drop_fin
    x ^= np.uint64((key * 0x9E3779B1) & 0xFFFFFFFF)                                 # the key's second entry
    x = (x * np.uint64(_M2)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    field = np.where((c & np.uint64(1)) == 1, x >> np.uint64(16), x & np.uint64(0xFFFF))
    return field >= np.uint64(thr >> 16)


def keep_mask_nd(key: int, thr: int, shape) -> np.ndarray:
    """Keep mask of an activation of the given shape: the last dimension is the column."""
    shape = tuple(int(d) for d in shape)
    rows = int(np.prod(shape[:-1], dtype=np.int64)) if len(shape) > 1 else 1
    return keep_mask2d(key, thr, rows, shape[-1]).reshape(shape)


def keep_mask(key: int, thr: int, n: int) -> np.ndarray:
    """1-D activation of n elements (one row)."""
    return keep_mask2d(key, thr, 1, n)[0]
