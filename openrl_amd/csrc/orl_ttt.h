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

}  // namespace orl
