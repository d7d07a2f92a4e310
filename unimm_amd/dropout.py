"""Counter-based dropout shared by every kernel: keep(idx) = mix32(idx ^ key) >= thr.

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


def make_key(seed: int, step: int, site: int) -> int:
    """One word per dropout site and step (site ids are assigned by the engine)."""
    return mix32_int(mix32_int(seed * 0x9E3779B1 + step) ^ (site * 0x85EBCA77 + 0x165667B1))


def drop_arg(p: float, key: int):
    """(key, thr, scale) triple the C ABI takes; p == 0 disables."""
    if p <= 0.0:
        return (0, 0, 1.0)
    thr = min(int(p * 4294967296.0), 0xFFFFFFFF)
    return (key & 0xFFFFFFFF, thr, 1.0 / (1.0 - p))


def keep_mask(key: int, thr: int, n: int) -> np.ndarray:
    """Host mirror: boolean keep mask for linear element indices 0..n-1."""
    x = np.arange(n, dtype=np.uint64) ^ np.uint64(key)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(_M1)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(_M2)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return x >= np.uint64(thr)
