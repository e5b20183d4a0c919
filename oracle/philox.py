"""ORACLE (test infrastructure only) -- numpy Philox4x32-10, the published counter-based RNG
(Salmon et al., SC'11) that csrc/oige_math.hpp implements for the sync-free reset path.

The reference draws its reset noise with torch.rand (tasks/shared/locomotion.py:137,139); that
stream cannot be replayed inside a fused kernel, so the HIP path keys Philox by
(seed, step, env id).  Tests regenerate the very same uniforms here and feed them to the oracle,
which makes the sync-free path checkable bit-for-bit too.
"""
from __future__ import annotations

import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0: int, k1: int):
    """Vectorised over counter arrays (uint32); returns four uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) for c in np.broadcast_arrays(c0, c1, c2, c3))
    k0 &= 0xFFFFFFFF
    k1 &= 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)) & MASK
        n1 = p1 & MASK
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)) & MASK
        n3 = p0 & MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def u01(x: np.ndarray) -> np.ndarray:
    """Top 24 bits -> fp32 in [0, 1)."""
    return (x >> np.uint32(8)).astype(np.float32) * np.float32(2.0**-24)


def uniforms(env_ids: np.ndarray, num_draws: int, seed: int, step: int, stream: int = 0) -> np.ndarray:
    """(len(env_ids), num_draws) fp32 uniforms: draw j of env i = word (j % 4) of the Philox block
    with counter (i, (j // 4) | stream << 16, step_lo, step_hi) and key (seed_lo, seed_hi)."""
    env_ids = np.asarray(env_ids, dtype=np.uint64)
    nblk = (num_draws + 3) // 4
    out = np.zeros((env_ids.shape[0], nblk * 4), dtype=np.float32)
    for blk in range(nblk):
        w = philox4x32_10(env_ids & MASK, np.uint64(blk | (stream << 16)), np.uint64(step & 0xFFFFFFFF), np.uint64((step >> 32) & 0xFFFFFFFF),
                          seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
        for q in range(4):
            out[:, 4 * blk + q] = u01(w[q])
    return out[:, :num_draws]
