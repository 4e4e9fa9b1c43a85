"""CPU restatement of MPE ``simple_spread`` (3 agents, 3 landmarks) - TEST INFRASTRUCTURE (see ppo_oracle.py).

Follows, in float64 like the reference:
  * action decoding        openrl/envs/mpe/multiagent_env.py:268-310 (Discrete(5): u = [a1 - a2, a3 - a4] * 5.0)
  * World.step             openrl/envs/mpe/core.py:216-291 (action force, soft collision force between colliding
                           movable entities, damping 0.25, dt 0.1) - only agent/agent pairs collide in this scenario
  * reward / observation   openrl/envs/mpe/scenarios/simple_spread.py:84-125 (reward counts the agent's collision
                           with ITSELF: -1 per step, as the reference does), shared reward = sum over agents
                           (multiagent_env.py:191-194), critic obs = all agents' obs concatenated (:213-221)
  * done                   multiagent_env.py:255-260: every agent done when current_step >= world_length (25)
Pinned against the reference's own ``World`` / ``Scenario`` classes by ``oracle/gen_golden.py`` (mpe_spread.npz).
The device env (csrc/orl_mpe.hip) draws reset positions from Philox instead of numpy's PCG64 - positions are
test inputs here, never compared across generators.
"""
from __future__ import annotations

import numpy as np

from . import philox as px

N_AGENTS, N_LANDMARKS = 3, 3
AGENT_SIZE, DT, DAMPING, CONTACT_FORCE, CONTACT_MARGIN, SENSITIVITY = 0.15, 0.1, 0.25, 1e2, 1e-3, 5.0
WORLD_LENGTH = 25


def decode_action(a: np.ndarray) -> np.ndarray:
    """[..., ] int action -> [..., 2] force (multiagent_env.py:289-310)."""
    a = np.asarray(a).astype(np.int64)
    u = np.zeros(a.shape + (2,), dtype=np.float64)
    u[..., 0] = (a == 1).astype(np.float64) - (a == 2)
    u[..., 1] = (a == 3).astype(np.float64) - (a == 4)
    return u * SENSITIVITY


def world_step(pos: np.ndarray, vel: np.ndarray, actions: np.ndarray):
    """pos, vel [A, 2] float64 -> new (pos, vel) after one core.World.step."""
    force = decode_action(actions)  # mass 1, accel None
    for a in range(N_AGENTS):
        for b in range(a + 1, N_AGENTS):
            delta = pos[a] - pos[b]
            dist = np.sqrt(np.sum(np.square(delta)))
            k = CONTACT_MARGIN
            pen = np.logaddexp(0, -(dist - 2 * AGENT_SIZE) / k) * k
            f = CONTACT_FORCE * delta / dist * pen
            force[a] = f + force[a]
            force[b] = -f + force[b]
    vel = vel * (1 - DAMPING) + force * DT
    pos = pos + vel * DT
    return pos, vel


def rewards(pos: np.ndarray, lm: np.ndarray) -> np.ndarray:
    """[A] shared reward (every agent gets the sum of the individual rewards)."""
    ind = np.zeros(N_AGENTS)
    for i in range(N_AGENTS):
        r = 0.0
        for l in range(N_LANDMARKS):
            r -= min(np.sqrt(np.sum(np.square(pos[a] - lm[l]))) for a in range(N_AGENTS))
        for a in range(N_AGENTS):
            if np.sqrt(np.sum(np.square(pos[a] - pos[i]))) < 2 * AGENT_SIZE:
                r -= 1
        ind[i] = r
    return np.full(N_AGENTS, ind.sum())


def observations(pos: np.ndarray, vel: np.ndarray, lm: np.ndarray):
    """policy obs [A, 18] = [vel, pos, landmarks - pos, others - pos, comm zeros]; critic obs [A, 54]."""
    obs = []
    for i in range(N_AGENTS):
        parts = [vel[i], pos[i]] + [lm[l] - pos[i] for l in range(N_LANDMARKS)]
        parts += [pos[o] - pos[i] for o in range(N_AGENTS) if o != i]
        parts += [np.zeros(2) for o in range(N_AGENTS) if o != i]
        obs.append(np.concatenate(parts))
    p = np.stack(obs)
    return p, np.tile(p.reshape(1, -1), (N_AGENTS, 1))


def device_reset_positions(seed: int, env: int, episode: int):
    """The DEVICE env's reset draw (csrc/orl_mpe.hip mpe_reset_state): three Philox blocks keyed
    (seed; env, 0x3D9E0000 + k, episode, 0): agents uniform(-1,1), landmarks 0.8 * uniform(-1,1), fp32."""
    u = []
    for k in range(3):
        r = px.philox4x32_10(seed, env, 0x3D9E0000 + k, episode, 0)
        u += [px.u01(int(x)) for x in r]
    u = np.array(u, dtype=np.float32)
    ag = (u[:6] * np.float32(2.0) - np.float32(1.0)).reshape(3, 2)
    lm = (np.float32(0.8) * (u[6:] * np.float32(2.0) - np.float32(1.0))).reshape(3, 2)
    return ag, lm
