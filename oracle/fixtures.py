"""Seeded synthetic inputs shared by ``oracle/gen_golden.py`` (which feeds them to the REAL reference) and the
parity tests (which feed the same arrays to the HIP engine) - TEST INFRASTRUCTURE.

Full-size cases (BASELINE.json configs[1]: 4096 envs x 128 steps = 524 288 rows) are too large to commit as
buffers, so only the reference's OUTPUTS (final weights, train_info, ValueNorm state) are stored under
``tests/golden/`` and the inputs are regenerated here from a seed.  ``numpy.random.RandomState`` (MT19937 +
the legacy polar ``randn``) is bit-stable across numpy versions and platforms.
"""
from __future__ import annotations

import numpy as np


def synth_update_buffer(seed: int, N: int, T: int, D: int, n_act: int, A: int = 1):
    """A filled rollout buffer for a Discrete(n_act) policy with observation width D: every field the PPO update
    reads, in the reference's ``ReplayData`` shapes.  ``returns`` are NOT included - both
    sides run their own ``compute_returns`` on these inputs (that is part of what is compared)."""
    rs = np.random.RandomState(seed)
    f32 = np.float32
    out = dict(
        policy_obs=rs.randn(T + 1, N, A, D).astype(f32),
        rewards=rs.rand(T, N, A, 1).astype(f32),
        value_preds=(0.3 * rs.randn(T + 1, N, A, 1)).astype(f32),
        masks=(rs.rand(T + 1, N, A, 1) > 0.02).astype(f32),
        active_masks=np.ones((T + 1, N, A, 1), f32),
        bad_masks=np.ones((T + 1, N, A, 1), f32),
        actions=rs.randint(0, n_act, (T, N, A, 1)).astype(f32),
    )
    out["action_log_probs"] = (np.log(1.0 / n_act) + 0.05 * rs.randn(T, N, A, 1)).astype(f32)
    out["action_masks"] = np.ones((T + 1, N, A, n_act), f32)
    out["next_value"] = (0.3 * rs.randn(N, A, 1)).astype(f32)
    return out
