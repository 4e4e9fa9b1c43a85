"""Philox4x32-10 counter-based RNG in numpy - TEST INFRASTRUCTURE (oracle side only).

Restates the published algorithm (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as
1, 2, 3", SC'11; multipliers 0xD2511F53 / 0xCD9E8D57, Weyl constants 0x9E3779B9 / 0xBB67AE85) that
``openrl_amd/csrc/orl_common.h`` implements on the device, so that sampled actions and synthetic
observations of the HIP engine can be reproduced bit-for-bit on the CPU.

Known-answer vectors from the Random123 distribution (kat_vectors, philox4x32-10) are checked in
``tests/test_oracle_cpu.py``.
"""
from __future__ import annotations

import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(seed, c0, c1, c2, c3):
    """Vectorised over counters.  ``seed`` is a python int (64 bit); c* are uint32 arrays/scalars.

    Returns four uint32 arrays (x, y, z, w)."""
    c0, c1, c2, c3 = np.broadcast_arrays(*(np.asarray(c, dtype=np.uint64) & MASK32 for c in (c0, c1, c2, c3)))
    c0, c1, c2, c3 = c0.copy(), c1.copy(), c2.copy(), c3.copy()
    k0 = int(seed) & 0xFFFFFFFF
    k1 = (int(seed) >> 32) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK32
        n0 = hi1 ^ c1 ^ np.uint64(k0)
        n1 = lo1
        n2 = hi0 ^ c3 ^ np.uint64(k1)
        n3 = lo0
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def u01(x):
    """[0,1) with 24 bits, float32 - device ``u01``."""
    return ((np.asarray(x, dtype=np.uint32) >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)).astype(
        np.float32)


def u01_open0(x):
    """(0,1] - device ``u01_open0``."""
    return (((np.asarray(x, dtype=np.uint32) >> np.uint32(8)).astype(np.float32) + np.float32(1.0))
            * np.float32(1.0 / 16777216.0)).astype(np.float32)


def box_muller(x0, x1):
    """Two standard normals per uint32 pair, float32 math like the device (tolerance ~1e-6)."""
    u1 = u01_open0(x0)
    u2 = u01(x1)
    rad = np.sqrt(np.float32(-2.0) * np.log(u1)).astype(np.float32)
    ang = (np.float32(6.28318530717958647692) * u2).astype(np.float32)
    return (rad * np.cos(ang)).astype(np.float32), (rad * np.sin(ang)).astype(np.float32)


def _fmix32(x):
    x = x & MASK32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & MASK32
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & MASK32
    x ^= x >> np.uint64(16)
    return x


def feistel_perm(n: int, seed: int, stream_id: int) -> np.ndarray:
    """Device ``orl_perm_feistel``: 4-round balanced Feistel over ceil(log2 n) bits with cycle walking; round
    keys = one Philox block of (stream_id) under ``seed``; round function = murmur3 fmix32(half ^ key)."""
    bits = 1
    while (1 << bits) < n:
        bits += 1
    hb = (bits + 1) // 2
    hmask = np.uint64((1 << hb) - 1)
    keys = [np.uint64(int(k)) for k in philox4x32_10(seed, stream_id & 0xFFFFFFFF, (stream_id >> 32) & 0xFFFFFFFF,
                                                     0x5EED5EED, 0)]
    out = np.empty(n, dtype=np.int64)
    pending = np.arange(n)
    cur = np.arange(n, dtype=np.uint64)
    while pending.size:
        lft = cur >> np.uint64(hb)
        rgt = cur & hmask
        for r in range(4):
            fv = _fmix32(rgt ^ keys[r]) & hmask
            lft, rgt = rgt, lft ^ fv
        cur = (lft << np.uint64(hb)) | rgt
        done = cur < np.uint64(n)
        out[pending[done]] = cur[done].astype(np.int64)
        pending = pending[~done]
        cur = cur[~done]
    return out
