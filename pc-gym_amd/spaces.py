"""Minimal Box space: the subset of gymnasium.spaces.Box the reference touches
(pcgym.py:68-92: low / high / shape / sample()).  gymnasium is not a dependency;
when it is importable the real Box is used so SB3-style callers keep working."""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - gymnasium is absent in the build image
    from gymnasium.spaces import Box as _GymBox
except Exception:  # noqa: BLE001
    _GymBox = None


class _Box:
    def __init__(self, low, high, dtype=np.float64, seed=None):
        self.low = np.asarray(low, dtype=np.float64)
        self.high = np.asarray(high, dtype=np.float64)
        self.shape = self.low.shape
        self.dtype = np.dtype(dtype)
        self._rng = np.random.default_rng(seed)

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1e6)
        hi = np.where(np.isfinite(self.high), self.high, 1e6)
        return self._rng.uniform(lo, hi).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return f"Box({self.low}, {self.high}, {self.shape}, {self.dtype})"


def Box(low, high, dtype=np.float64):
    if _GymBox is not None:
        with np.errstate(all="ignore"):
            return _GymBox(low=np.asarray(low, dtype=dtype), high=np.asarray(high, dtype=dtype), dtype=dtype)
    return _Box(low, high, dtype=dtype)
