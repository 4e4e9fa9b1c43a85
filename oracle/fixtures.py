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


def synth_update_buffer_general(seed: int, N: int, T: int, Dp: int, Dc: int, kind: str, n_act: int, A: int = 1,
                                legal_masks: bool = False, rnn_hidden: int = 0):
    """Like ``synth_update_buffer`` for the other BASELINE.json shapes (configs[2..4] at FULL size): ``kind`` is
    ``"discrete"`` (Discrete(n_act), optionally with random legal-move masks that always contain the taken action)
    or ``"box"`` (Box(n_act,): real-valued actions with per-dimension log-probs, ``FixedNormal.log_probs``); separate policy / critic
    observations when ``Dp != Dc``; ``rnn_hidden > 0`` adds the two recurrent-state arrays ``[T+1, N, A, 1, H]``."""
    rs = np.random.RandomState(seed)
    f32 = np.float32
    out = dict(policy_obs=rs.randn(T + 1, N, A, Dp).astype(f32))
    out["critic_obs"] = out["policy_obs"] if Dc == Dp else rs.randn(T + 1, N, A, Dc).astype(f32)
    out["rewards"] = rs.rand(T, N, A, 1).astype(f32)
    out["value_preds"] = (0.3 * rs.randn(T + 1, N, A, 1)).astype(f32)
    env_alive = (rs.rand(T + 1, N, 1, 1) > 0.02)
    out["masks"] = np.broadcast_to(env_alive, (T + 1, N, A, 1)).astype(f32).copy()
    out["active_masks"] = np.ones((T + 1, N, A, 1), f32)
    if A > 1:  # some agents finish before their env does (onpolicy_driver.py:110-124)
        out["active_masks"] = (rs.rand(T + 1, N, A, 1) > 0.05).astype(f32)
    out["bad_masks"] = np.ones((T + 1, N, A, 1), f32)
    if kind == "discrete":
        act = rs.randint(0, n_act, (T, N, A, 1))
        out["actions"] = act.astype(f32)
        am = np.ones((T + 1, N, A, n_act), f32)
        if legal_masks:
            am = (rs.rand(T + 1, N, A, n_act) > 0.4).astype(f32)
            np.put_along_axis(am[:T], act, 1.0, axis=-1)  # the taken action was legal
            am[T, ..., 0] = 1.0
        out["action_masks"] = am
        # a near-uniform behaviour policy over the legal moves: ratios start close to 1
        out["action_log_probs"] = (-np.log(am[:T].sum(-1, keepdims=True)) + 0.05 * rs.randn(T, N, A, 1)).astype(f32)
    else:
        out["actions"] = rs.randn(T, N, A, n_act).astype(f32)
        # FixedNormal.log_probs is PER DIMENSION (distributions.py:30-33): a unit-variance behaviour policy, ratios near 1
        out["action_log_probs"] = (-0.9189385 - 0.5 * out["actions"] ** 2 + 0.05 * rs.randn(T, N, A, n_act)).astype(f32)
    if rnn_hidden:
        out["rnn_states"] = (0.5 * rs.randn(T + 1, N, A, 1, rnn_hidden)).astype(f32)
        out["rnn_states_critic"] = (0.5 * rs.randn(T + 1, N, A, 1, rnn_hidden)).astype(f32)
    out["next_value"] = (0.3 * rs.randn(N, A, 1)).astype(f32)
    return out
