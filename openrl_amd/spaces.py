"""Minimal observation/action spaces with gymnasium's class NAMES.

The reference dispatches on ``space.__class__.__name__`` (buffers/utils/util.py:59-85,
replay_data.py:148, act.py:14-25), so real ``gymnasium.spaces`` objects work here unchanged; these
classes exist because gymnasium is not installable offline.
"""
from __future__ import annotations

import numpy as np


class Space:
    def __init__(self, shape=None, dtype=None):
        self._shape = None if shape is None else tuple(int(s) for s in shape)
        self.dtype = dtype

    @property
    def shape(self):
        return self._shape

    def contains(self, x) -> bool:
        raise NotImplementedError

    def seed(self, seed=None):
        self._np_random = np.random.default_rng(seed)
        return [seed]

    @property
    def np_random(self):
        if getattr(self, "_np_random", None) is None:
            self.seed()
        return self._np_random

    def sample(self, mask=None):
        raise NotImplementedError

    def __contains__(self, x) -> bool:
        return self.contains(x)

    def __repr__(self):
        return "%s(%s)" % (self.__class__.__name__, self._shape)


class Box(Space):
    def __init__(self, low=-np.inf, high=np.inf, shape=None, dtype=np.float32):
        if shape is None:
            shape = np.shape(low)
        super().__init__(shape, dtype)
        self.low = np.broadcast_to(np.asarray(low, dtype=dtype), self._shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=dtype), self._shape).copy()

    def sample(self, mask=None):
        """Uniform inside finite bounds, standard normal where a side is unbounded (gymnasium's Box.sample uses
        shifted exponentials for the half-bounded cases; only boundedness matters to the callers here)."""
        lo, hi = self.low.astype(np.float64), self.high.astype(np.float64)
        fin = np.isfinite(lo) & np.isfinite(hi)
        x = self.np_random.normal(size=self._shape)
        u = self.np_random.uniform(size=self._shape)
        x = np.where(fin, np.where(fin, lo, 0.0) + u * np.where(fin, hi - lo, 0.0), np.clip(x, lo, hi))
        return x.astype(self.dtype)

    def contains(self, x) -> bool:
        """gymnasium.spaces.Box.contains: castable to the dtype, same shape, inside the bounds."""
        x = np.asarray(x)
        return bool(np.can_cast(x.dtype, self.dtype) and x.shape == self._shape
                    and np.all(x >= self.low) and np.all(x <= self.high))


class Discrete(Space):
    def __init__(self, n, start=0):
        super().__init__((), np.int64)
        self.n = int(n)
        self.start = int(start)

    def sample(self, mask=None):
        """gymnasium.spaces.Discrete.sample: uniform over the legal entries of ``mask`` (int8 [n]) when given."""
        if mask is not None:
            legal = np.flatnonzero(np.asarray(mask).reshape(-1)[: self.n])
            if legal.size:
                return int(self.start + self.np_random.choice(legal))
            return int(self.start)
        return int(self.start + self.np_random.integers(self.n))

    def contains(self, x) -> bool:
        x = np.asarray(x)
        return bool(x.shape == () and np.issubdtype(x.dtype, np.integer) and self.start <= int(x) < self.start + self.n)

    def __repr__(self):
        return "Discrete(%d)" % self.n


class MultiDiscrete(Space):
    """gymnasium.spaces.MultiDiscrete(nvec): the reference reads ``high - low + 1`` per component (act.py:28)."""

    def __init__(self, nvec):
        self.nvec = np.asarray(nvec, dtype=np.int64).reshape(-1)
        super().__init__(self.nvec.shape, np.int64)
        self.low = np.zeros_like(self.nvec)
        self.high = self.nvec - 1

    def sample(self, mask=None):
        return (self.np_random.random(self.nvec.shape) * self.nvec).astype(np.int64)

    def contains(self, x) -> bool:
        x = np.asarray(x)
        return bool(x.shape == self.nvec.shape and np.issubdtype(x.dtype, np.integer) and np.all(x >= 0)
                    and np.all(x < self.nvec))

    def __repr__(self):
        return "MultiDiscrete(%s)" % self.nvec.tolist()


class Tuple(Space):  # noqa: A001 - the name is the contract
    """gymnasium.spaces.Tuple.  As an ACTION space the reference reads it as "continuous + discrete" (its mixed / agar
    branch: ``action_space[0]`` a Box, ``action_space[1]`` a Discrete - act.py:33-43, buffers/utils/util.py:83-84)."""

    def __init__(self, spaces):
        super().__init__(None, None)
        self.spaces = tuple(spaces)

    def __getitem__(self, i):
        return self.spaces[i]

    def __len__(self):
        return len(self.spaces)

    def sample(self, mask=None):
        return tuple(s.sample() for s in self.spaces)

    def __repr__(self):
        return "Tuple(%s)" % ", ".join(repr(s) for s in self.spaces)


class Dict(Space):  # noqa: A001 - the name is the contract
    def __init__(self, spaces=None, **kw):
        super().__init__(None, None)
        self.spaces = dict(spaces or {})
        self.spaces.update(kw)

    def __getitem__(self, k):
        return self.spaces[k]

    def keys(self):
        return self.spaces.keys()


def kind(space) -> str:
    return space.__class__.__name__


def policy_obs_space(obs_space):
    """buffers/utils/util.py:42-53: Dict{"policy","critic"} selects per-tower spaces."""
    if kind(obs_space) == "Dict" and "policy" in obs_space.spaces:
        return obs_space["policy"]
    return obs_space


def critic_obs_space(obs_space):
    if kind(obs_space) == "Dict" and "critic" in obs_space.spaces:
        return obs_space["critic"]
    return obs_space


def obs_dim(space) -> int:
    """Flat observation width of a Box / Discrete space (get_shape_from_obs_space_v2, util.py:56-71)."""
    k = kind(space)
    if k == "Box":
        if len(space.shape) != 1:
            raise NotImplementedError("openrl_amd builds MLP towers only: Box obs must be 1-D, got %s" % (space.shape,))
        return int(space.shape[0])
    if k == "Discrete":
        return int(space.n)
    raise NotImplementedError("obs space type %s not built in the MI355X engine" % k)


def act_shape(space) -> int:
    """Stored action width (get_shape_from_act_space, buffers/utils/util.py:74-85)."""
    k = kind(space)
    if k == "Discrete":
        return 1
    if k == "Box":
        return int(space.shape[0])
    if k == "MultiDiscrete":  # one stored column per component (buffers/utils/util.py:80-81: space.shape)
        return int(np.asarray(space.shape).reshape(-1)[0])
    if k == "Tuple":  # the mixed branch: continuous dims + ONE stored column for the discrete action (util.py:83-84)
        return int(space[0].shape[0]) + 1
    raise NotImplementedError("action space type %s not built in the MI355X engine (Discrete / Box / MultiDiscrete / Tuple)"
                              % k)
