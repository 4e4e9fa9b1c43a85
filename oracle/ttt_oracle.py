"""TEST INFRASTRUCTURE ONLY - CPU restatement of the device tic-tac-toe env (``openrl_amd/csrc/orl_ttt.hip``).

What it restates: the game the reference trains on in ``examples/selfplay`` - PettingZoo ``tictactoe_v3`` (a third-party
package, not vendored under /root/reference and not installed here; rules restated from its published behaviour:
3x3 board, three in a row wins, winner +1 / loser -1 / draw 0, an illegal move loses at once) seen through
``RandomOpponentWrapper`` (``openrl/selfplay/wrappers/random_opponent_wrapper.py:27-43``: the opponent samples
uniformly from the legal moves) and ``BaseMultiPlayerWrapper`` (``base_multiplayer_wrapper.py:85-150``: ``self_player``
is drawn at random at every reset, the opponent moves until it is the agent's turn, a step returns after the
opponent's reply).  Parity of the RULES is pinned by exhaustive properties (``tests/test_ttt_cpu.py``: every one of the
8 lines wins, full boards without a line draw, the opponent only ever takes empty cells); the random streams are the
engine's own Philox keys, so device and oracle are compared bit for bit.
"""
from __future__ import annotations

import numpy as np

from . import philox as px

LINES = ((0, 1, 2), (3, 4, 5), (6, 7, 8), (0, 3, 6), (1, 4, 7), (2, 5, 8), (0, 4, 8), (2, 4, 6))


def wins(board, p: int) -> bool:
    return any(all(board[c] == p for c in line) for line in LINES)


def opponent_move(board, seed: int, env: int, episode: int, move: int) -> None:
    empty = [c for c in range(9) if board[c] == 0]
    x, _, _, _ = px.philox4x32_10(seed, env, 0x77C70000 + move, episode, 0)
    k = int(np.float32(px.u01(x)) * np.float32(len(empty)))
    board[empty[min(k, len(empty) - 1)]] = 2


def begin(seed: int, env: int, episode: int):
    board = [0] * 9
    x, _, _, _ = px.philox4x32_10(seed, env, 0x77C7FFFF, episode, 0)
    moves = 0
    if int(x) & 1:
        opponent_move(board, seed, env, episode, 0)
        moves = 1
    return board, moves


class Game:
    """One env of the batch: ``step(a) -> (reward, done)``; ``obs()`` / ``mask()`` as the device env writes them."""

    def __init__(self, seed: int, env: int):
        self.seed, self.env, self.episode = seed, env, 0
        self.board, self.moves = begin(seed, env, 0)

    def obs(self) -> np.ndarray:
        o = np.zeros(18, np.float32)
        for c in range(9):
            o[2 * c] = self.board[c] == 1
            o[2 * c + 1] = self.board[c] == 2
        return o

    def mask(self) -> np.ndarray:
        return np.array([self.board[c] == 0 for c in range(9)], np.float32)

    def step(self, a: int):
        rew, done = 0.0, False
        if not (0 <= a < 9) or self.board[a] != 0:
            rew, done = -1.0, True
        else:
            self.board[a] = 1
            if wins(self.board, 1):
                rew, done = 1.0, True
            elif 0 not in self.board:
                done = True
            else:
                opponent_move(self.board, self.seed, self.env, self.episode, self.moves)
                self.moves += 1
                if wins(self.board, 2):
                    rew, done = -1.0, True
                elif 0 not in self.board:
                    done = True
        if done:
            self.episode += 1
            self.board, self.moves = begin(self.seed, self.env, self.episode)
        return rew, done

    # ---- two-phase step (orl_ttt_agent_move / orl_ttt_opponent_move): the opponent's reply is supplied by the caller
    def agent_move(self, a: int):
        """Returns (reward, done, pending): pending = the game is open and waits for the opponent's reply."""
        self._rew, self._phase = 0.0, 1
        if not (0 <= a < 9) or self.board[a] != 0:
            self._rew, self._phase = -1.0, 2
        else:
            self.board[a] = 1
            if wins(self.board, 1):
                self._rew, self._phase = 1.0, 2
            elif 0 not in self.board:
                self._phase = 2
        return self._rew, self._phase == 2, self._phase == 1

    def opponent_view(self):
        o, m = np.zeros(18, np.float32), np.zeros(9, np.float32)
        if self._phase == 1:
            for c in range(9):
                o[2 * c] = self.board[c] == 2
                o[2 * c + 1] = self.board[c] == 1
                m[c] = self.board[c] == 0
        else:
            m[0] = 1.0
        return o, m

    def opponent_move(self, a: int):
        rew, done = self._rew, self._phase == 2
        if self._phase == 1:
            if not (0 <= a < 9) or self.board[a] != 0:
                a = min(c for c in range(9) if self.board[c] == 0)
            self.board[a] = 2
            self.moves += 1
            if wins(self.board, 2):
                rew, done = -1.0, True
            elif 0 not in self.board:
                done = True
        if done:
            self.episode += 1
            self.board, self.moves = begin(self.seed, self.env, self.episode)
        return rew, done
