"""``ObsData`` - the dict-of-arrays view the reference hands to code that touches Dict observations
(``openrl/buffers/utils/obs_data.py:22-62``, a ``treevalue.TreeValue`` there; a plain mapping here - treevalue is not a
dependency of this package).  Host-side numpy only: the device buffer (``buffers/replay_data.py``) keeps the
``{"policy", "critic"}`` parts as two device tensors and never goes through this class."""
from __future__ import annotations

from typing import Dict

import numpy as np


class ObsData(dict):
    """Arrays laid out ``[step, env, agent, ...]`` per key.  Leaves are shared with the dict passed in, not copied."""

    def __init__(self, data: Dict[str, np.ndarray]):
        super().__init__(data)

    def __getattr__(self, key):  # tree-style attribute access: obs.policy
        try:
            return dict.__getitem__(self, key)
        except KeyError:
            raise AttributeError(key) from None

    def flatten(self) -> np.ndarray:
        """All leaves concatenated along axis 0, in key order (obs_data.py:23-24)."""
        return np.concatenate(list(self.values()))

    @staticmethod
    def prepare_input(obs):
        """[env][agent, ...] -> [env * agent, ...] per key (obs_data.py:26-34)."""
        if isinstance(obs, dict):
            return {k: np.concatenate(v) for k, v in obs.items()}
        return np.concatenate(obs, axis=0)

    def step_batch(self, step: int) -> Dict[str, np.ndarray]:
        """The [env * agent, ...] batch of one step per key (obs_data.py:36-40)."""
        return {k: np.concatenate(dict.__getitem__(self, k)[step]) for k in self.keys()}

    def all_batch(self, min: int, max: int) -> Dict[str, np.ndarray]:  # noqa: A002 - the reference's argument names
        """Steps [min, max) flattened to [steps * env * agent, ...] per key (obs_data.py:42-48)."""
        out = {}
        for k in self.keys():
            v = dict.__getitem__(self, k)
            out[k] = v[min:max].reshape((-1, *v.shape[3:]))
        return out

    def __getitem__(self, key):
        if isinstance(key, (int, np.integer)):
            return self.step_batch(int(key))
        return dict.__getitem__(self, key)

    def step_flatten(self, step: int) -> np.ndarray:
        """One step's leaves concatenated along the last axis (what obs_data.py:56-62 computes; the reference drops the
        result - it has no ``return``)."""
        return np.concatenate([dict.__getitem__(self, k)[step] for k in self.keys()], -1)
