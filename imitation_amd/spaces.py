"""Minimal observation/action space objects.

The reference takes `gymnasium.spaces` objects (not installed in this image).
Only the surface the GAIL/AIRL path touches is provided:
`shape`, `dtype`, `low`/`high` (Box), `n` (Discrete), `sample`, `contains`.
Reference call sites: `rewards/reward_nets.py:416-424` (flattened dim),
`data/buffer.py:284-309` (shape/dtype), SB3 `collect_rollouts` (Box clipping).
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np


class Space:
    shape: Optional[Tuple[int, ...]]
    dtype: Optional[np.dtype]

    def __init__(self, shape=None, dtype=None, seed: Optional[int] = None):
        self.shape = None if shape is None else tuple(int(s) for s in shape)
        self.dtype = None if dtype is None else np.dtype(dtype)
        self._rng = np.random.default_rng(seed)

    def seed(self, seed: Optional[int] = None):
        self._rng = np.random.default_rng(seed)
        return [seed]

    def sample(self):
        raise NotImplementedError

    def contains(self, x) -> bool:
        raise NotImplementedError

    def __contains__(self, x) -> bool:
        return self.contains(x)


class Box(Space):
    """Continuous box `[low, high]^shape`."""

    def __init__(self, low, high, shape: Optional[Sequence[int]] = None, dtype=np.float32, seed=None):
        if shape is None:
            shape = np.broadcast(np.asarray(low), np.asarray(high)).shape
        super().__init__(shape, dtype, seed)
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return self._rng.uniform(lo, hi).astype(self.dtype)

    def contains(self, x) -> bool:
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

    def __eq__(self, other):
        return (
            isinstance(other, Box)
            and self.shape == other.shape
            and np.array_equal(self.low, other.low)
            and np.array_equal(self.high, other.high)
        )


class Discrete(Space):
    """`{0, ..., n-1}`."""

    def __init__(self, n: int, seed=None):
        super().__init__((), np.int64, seed)
        self.n = int(n)

    def sample(self):
        return np.int64(self._rng.integers(self.n))

    def contains(self, x) -> bool:
        x = np.asarray(x)
        return x.shape == () and np.issubdtype(x.dtype, np.integer) and 0 <= int(x) < self.n

    def __repr__(self):
        return f"Discrete({self.n})"

    def __eq__(self, other):
        return isinstance(other, Discrete) and self.n == other.n


def flatdim(space: Space) -> int:
    """Flattened feature width after SB3-style preprocessing:
    Box -> prod(shape), Discrete(n) -> n (one-hot)."""
    if isinstance(space, Box):
        return int(np.prod(space.shape))
    if isinstance(space, Discrete):
        return space.n
    raise NotImplementedError(f"unsupported space {space!r}")
