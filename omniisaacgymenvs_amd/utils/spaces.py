"""Minimal ``gym.spaces.Box`` stand-in (gym is not installed here; rl_task.py:137-150 only
needs low/high/shape/dtype)."""
from __future__ import annotations

import numpy as np


class Box:
    def __init__(self, low, high, dtype=np.float32):
        self.low = np.asarray(low, dtype=dtype)
        self.high = np.asarray(high, dtype=dtype)
        assert self.low.shape == self.high.shape
        self.shape = self.low.shape
        self.dtype = np.dtype(dtype)

    def __repr__(self) -> str:
        return f"Box(shape={self.shape}, dtype={self.dtype})"

    def sample(self, rng: np.random.Generator = None) -> np.ndarray:
        rng = rng or np.random.default_rng()
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return rng.uniform(lo, hi).astype(self.dtype)
