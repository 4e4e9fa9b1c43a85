// orl_ttt.h - rules of the device tic-tac-toe env shared by the stand-alone step kernel (orl_ttt.hip) and the fused
// rollout kernel (orl_act.hip, ORL_ENV_TTT).  See orl_ttt.hip for the env's contract.
#pragma once
#include "orl_common.h"

namespace orl {

constexpr int TTT_STATE_W = 12;  // board[9] (0 empty, 1 agent, 2 opponent), opponent moves this episode, episode, -

__device__ inline bool ttt_wins(const int (&b)[9], int p) {
  return (b[0] == p && b[1] == p && b[2] == p) || (b[3] == p && b[4] == p && b[5] == p) ||
         (b[6] == p && b[7] == p && b[8] == p) || (b[0] == p && b[3] == p && b[6] == p) ||
         (b[1] == p && b[4] == p && b[7] == p) || (b[2] == p && b[5] == p && b[8] == p) ||
         (b[0] == p && b[4] == p && b[8] == p) || (b[2] == p && b[4] == p && b[6] == p);
}

__device__ inline int ttt_empty(const int (&b)[9]) {
  int n = 0;
#pragma unroll
  for (int c = 0; c < 9; ++c) n += b[c] == 0;
  return n;
}

// uniformly random legal move: the k-th empty cell, k = floor(u * n_empty)
__device__ inline void ttt_opponent_move(int (&b)[9], uint64_t seed, uint32_t env, uint32_t episode, uint32_t move) {
  const int n = ttt_empty(b);
  const u4 r = philox4x32_10(seed, env, 0x77C70000u + move, episode, 0u);
  int k = (int)(u01(r.x) * (float)n);
  k = k < n - 1 ? k : n - 1;
  int seen = 0;
#pragma unroll
  for (int c = 0; c < 9; ++c) {
    if (b[c] == 0) {
      if (seen == k) b[c] = 2;
      ++seen;
    }
  }
}

// empty board; if the agent plays second the opponent opens.  Returns the number of opponent moves made (0 / 1).
__device__ inline int ttt_begin(int (&b)[9], uint64_t seed, uint32_t env, uint32_t episode) {
#pragma unroll
  for (int c = 0; c < 9; ++c) b[c] = 0;
  const u4 r = philox4x32_10(seed, env, 0x77C7FFFFu, episode, 0u);
  if (r.x & 1u) {
    ttt_opponent_move(b, seed, env, episode, 0u);
    return 1;
  }
  return 0;
}

// ---- bitboard forms (the fused rollout kernel): cell c = bit c of a 9-bit mask per player; the same rules and the same
// draws as the array forms above (tests/test_ttt_gpu.py holds the fused rollout bit-exact against the step kernel) ----
__device__ inline bool ttt_wins_bits(int m) {
  const int rows = m & (m >> 1) & (m >> 2) & 0x49;  // (0,1,2) (3,4,5) (6,7,8)
  const int cols = m & (m >> 3) & (m >> 6) & 0x7;   // (0,3,6) (1,4,7) (2,5,8)
  return (rows | cols) != 0 || (m & 0x111) == 0x111 || (m & 0x54) == 0x54;
}

// ttt_opponent_move's choice given its Philox word: the k-th empty cell (one-hot), k = floor(u * n_empty)
__device__ inline int ttt_pick_empty_bits(int empty, uint32_t rx) {
  const int n = __popc((unsigned)empty);
  int k = (int)(u01(rx) * (float)n);
  k = k < n - 1 ? k : n - 1;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (i < k) empty &= empty - 1;
  return empty & -empty;
}

// Philox counter words of the env's three draws (ttt_opponent_move / ttt_begin above): which = 0 the opponent's reply to
// move `move` of `episode`, 1 = who opens `episode`, 2 = the opening move of `episode`
__device__ inline uint32_t ttt_draw(uint64_t seed, uint32_t env, int which, uint32_t episode, uint32_t move) {
  const uint32_t c1 = which == 1 ? 0x77C7FFFFu : 0x77C70000u + (which == 0 ? move : 0u);
  return philox4x32_10(seed, env, c1, episode, 0u).x;
}

}  // namespace orl
