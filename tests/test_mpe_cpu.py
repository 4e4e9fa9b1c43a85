"""MPE simple_spread restatement (oracle/mpe_oracle.py) against trajectories of the reference's own World / Scenario
classes (tests/golden/mpe_spread.npz, minted by oracle/gen_golden.py::mpe_case)."""
import numpy as np

from oracle import mpe_oracle as mo
from tests import helpers as H


def test_mpe_restatement_matches_reference_trajectories():
    g = H.load_golden("mpe_spread")
    E, S = g["actions"].shape[:2]
    crowded = 0
    for e in range(E):
        pos, vel, lm = g["pos0"][e].copy(), np.zeros((3, 2)), g["lm0"][e]
        for s in range(S):
            pos, vel = mo.world_step(pos, vel, g["actions"][e, s])
            assert np.array_equal(pos, g["pos"][e, s]) and np.array_equal(vel, g["vel"][e, s])
            r = mo.rewards(pos, lm)
            assert np.array_equal(r, g["rewards"][e, s])
            p, c = mo.observations(pos, vel, lm)
            assert np.array_equal(p, g["obs"][e, s]) and c.shape == (3, 54)
            assert np.array_equal(c[0], p.reshape(-1)) and np.array_equal(c[2], c[0])
            d = np.linalg.norm(pos[:, None] - pos[None], axis=-1)[np.triu_indices(3, 1)]
            crowded += int((d < 0.3).any())
    assert crowded >= 5  # the golden trajectories do exercise the contact forces


def test_action_decoding_and_self_collision_term():
    u = mo.decode_action(np.array([0, 1, 2, 3, 4]))
    assert np.array_equal(u, np.array([[0, 0], [5, 0], [-5, 0], [0, 5], [0, -5]], dtype=np.float64))
    pos = np.array([[0.0, 0.0], [1.0, 1.0], [-1.0, 1.0]])
    r = mo.rewards(pos, pos.copy())  # every landmark covered, nobody touching: only the 3 self-collisions remain
    assert np.array_equal(r, np.full(3, -3.0))
