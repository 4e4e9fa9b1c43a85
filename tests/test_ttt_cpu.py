"""Rules of the tic-tac-toe restatement (oracle/ttt_oracle.py) pinned by exhaustive properties - runs without a GPU."""
import itertools

import numpy as np

from oracle import ttt_oracle as to


def test_every_line_wins_and_nothing_else_does():
    for board in itertools.product((0, 1, 2), repeat=9):
        for p in (1, 2):
            want = any(all(board[c] == p for c in line) for line in
                       ((0, 1, 2), (3, 4, 5), (6, 7, 8), (0, 3, 6), (1, 4, 7), (2, 5, 8), (0, 4, 8), (2, 4, 6)))
            assert to.wins(board, p) == want
    assert len(to.LINES) == 8 and len(set(to.LINES)) == 8


def test_random_games_terminate_with_consistent_rewards_and_masks():
    rs = np.random.RandomState(0)
    outcomes = {1.0: 0, -1.0: 0, 0.0: 0}
    starts_second = 0
    for env in range(200):
        g = to.Game(seed=3, env=env)
        starts_second += int(sum(g.board) != 0)
        for _ in range(40):
            mask = g.mask()
            assert mask.sum() >= 1 and np.array_equal(g.obs()[0::2] + g.obs()[1::2], 1.0 - mask)
            # the opponent has moved as often as the agent, or once more when it opened
            n_me, n_op = int(g.obs()[0::2].sum()), int(g.obs()[1::2].sum())
            assert n_op - n_me in (0, 1)
            ep = g.episode
            rew, done = g.step(int(rs.choice(np.flatnonzero(mask))))
            assert done == (g.episode == ep + 1) and (done or rew == 0.0)
            if done:
                outcomes[rew] += 1
    assert all(v > 0 for v in outcomes.values()) and 60 < starts_second < 140
    # an illegal move loses at once
    g = to.Game(seed=1, env=0)
    taken = int(np.flatnonzero(g.mask() == 0)[0]) if (g.mask() == 0).any() else None
    if taken is None:
        g.step(4)
        taken = 4 if g.board[4] != 0 else int(np.flatnonzero(g.mask() == 0)[0])
    assert g.step(taken) == (-1.0, True)


def test_two_phase_step_with_the_random_reply_equals_the_one_phase_step():
    """Oracle self-consistency: agent_move + opponent_move(reply) == step when the reply is the uniformly random legal
    move step() itself would draw (same Philox key) - ties the self-play restatement to the random-opponent one."""
    import copy

    from oracle import philox as px

    rs = np.random.RandomState(5)
    for env in range(60):
        a_game, b_game = to.Game(seed=8, env=env), to.Game(seed=8, env=env)
        for _ in range(30):
            assert a_game.board == b_game.board and a_game.episode == b_game.episode
            act = int(rs.choice(np.flatnonzero(a_game.mask())))
            want = a_game.step(act)
            rew, done, pending = b_game.agent_move(act)
            reply = 0
            if pending:
                empty = [c for c in range(9) if b_game.board[c] == 0]
                x, _, _, _ = px.philox4x32_10(8, env, 0x77C70000 + b_game.moves, b_game.episode, 0)
                k = int(np.float32(px.u01(x)) * np.float32(len(empty)))
                reply = empty[min(k, len(empty) - 1)]
            got = b_game.opponent_move(reply)
            assert got == want and a_game.board == b_game.board and a_game.moves == b_game.moves
    del copy
