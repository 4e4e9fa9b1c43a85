// orl_ttt.hip - device-resident batched tic-tac-toe against a uniformly random legal-move opponent, for gfx950:
// BASELINE config 5's env (examples/selfplay: PettingZoo tictactoe_v3 behind RandomOpponentWrapper,
// openrl/selfplay/wrappers/base_multiplayer_wrapper.py:85-150 + random_opponent_wrapper.py:27-43).
//
// One lane owns one game.  The learning agent is "player 1" of the state; at every reset a Philox bit decides whether
// it moves first or second (BaseMultiPlayerWrapper.reset picks self_player at random and lets the opponent move until
// it is the agent's turn).  A step = the agent's move, then - if the game is still open - the opponent's reply.
// Rewards are PettingZoo's: +1 win, -1 loss, 0 draw, -1 for an illegal move (which also ends the game).  Observation:
// 18 floats, obs[2*cell] = agent's mark, obs[2*cell+1] = opponent's mark (the two 3x3 planes of tictactoe_v3 in
// action-index order); action mask: 9 floats, 1 = empty cell.  Finished games restart in the same step (auto-reset,
// envs/vec_env/sync_venv.py:217-222): the returned observation / mask are the first ones of the next game.
// Integer work throughout; Philox streams: start bit (seed, env, 0x77C7FFFF, episode), opponent move k of an episode
// (seed, env, 0x77C70000 + k, episode).
#include "orl_common.h"
#include "orl_ttt.h"

namespace orl {

__device__ inline void ttt_write(const int (&b)[9], int moves, int episode, float* __restrict__ st, float* __restrict__ obs,
                                 float* __restrict__ amask) {
#pragma unroll
  for (int c = 0; c < 9; ++c) {
    st[c] = (float)b[c];
    obs[2 * c] = b[c] == 1 ? 1.f : 0.f;
    obs[2 * c + 1] = b[c] == 2 ? 1.f : 0.f;
    amask[c] = b[c] == 0 ? 1.f : 0.f;
  }
  st[9] = (float)moves; st[10] = (float)episode; st[11] = 0.f;
}

__global__ void ttt_reset_kernel(float* __restrict__ st, float* __restrict__ ep_stats, float* __restrict__ obs,
                                 float* __restrict__ amask, int N, uint64_t seed) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  int b[9];
  const int moves = ttt_begin(b, seed, (uint32_t)n, 0u);
  ttt_write(b, moves, 0, st + (size_t)n * TTT_STATE_W, obs + (size_t)n * 18, amask + (size_t)n * 9);
  if (ep_stats != nullptr) {
#pragma unroll
    for (int k = 0; k < 4; ++k) ep_stats[(size_t)n * 4 + k] = 0.f;
  }
}

__global__ void ttt_step_kernel(float* __restrict__ st, float* __restrict__ ep_stats, const float* __restrict__ actions,
                                float* __restrict__ obs, float* __restrict__ amask, float* __restrict__ rewards,
                                uint8_t* __restrict__ dones, int N, uint64_t seed) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float* s = st + (size_t)n * TTT_STATE_W;
  int b[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) b[c] = (int)s[c];
  int moves = (int)s[9], episode = (int)s[10];
  const int a = (int)actions[n];
  float rew = 0.f;
  bool done = false;
  bool legal = false;
#pragma unroll
  for (int c = 0; c < 9; ++c) legal = legal || (c == a && b[c] == 0);
  if (!legal) {  // TerminateIllegalWrapper: the mover loses
    rew = -1.f;
    done = true;
  } else {
#pragma unroll
    for (int c = 0; c < 9; ++c)
      if (c == a) b[c] = 1;
    if (ttt_wins(b, 1)) { rew = 1.f; done = true; }
    else if (ttt_empty(b) == 0) done = true;
    else {
      ttt_opponent_move(b, seed, (uint32_t)n, (uint32_t)episode, (uint32_t)moves);
      ++moves;
      if (ttt_wins(b, 2)) { rew = -1.f; done = true; }
      else if (ttt_empty(b) == 0) done = true;
    }
  }
  if (ep_stats != nullptr) {
    float* e = ep_stats + (size_t)n * 4;
    e[0] += rew; e[1] += 1.f;
    if (done) { e[2] += e[0]; e[3] += 1.f; e[0] = 0.f; e[1] = 0.f; }
  }
  if (done) {
    ++episode;
    moves = ttt_begin(b, seed, (uint32_t)n, (uint32_t)episode);
  }
  ttt_write(b, moves, episode, s, obs + (size_t)n * 18, amask + (size_t)n * 9);
  rewards[n] = rew;
  dones[n] = done ? 1 : 0;
}

// ---- two-phase step for a LEARNED opponent (self-play): the agent's move, then - after the caller has run the
// opponent's policy on `opp_obs` / `opp_masks` (the board from the opponent's side) - the opponent's move.
// state[11]: 0 = agent to move, 1 = waiting for the opponent's reply, 2 = game ended by the agent's move.
__global__ void ttt_agent_move_kernel(float* __restrict__ st, const float* __restrict__ actions,
                                      float* __restrict__ opp_obs, float* __restrict__ opp_masks,
                                      float* __restrict__ rewards, uint8_t* __restrict__ dones, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float* s = st + (size_t)n * TTT_STATE_W;
  int b[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) b[c] = (int)s[c];
  const int a = (int)actions[n];
  float rew = 0.f;
  int phase = 1;
  bool legal = false;
#pragma unroll
  for (int c = 0; c < 9; ++c) legal = legal || (c == a && b[c] == 0);
  if (!legal) { rew = -1.f; phase = 2; }
  else {
#pragma unroll
    for (int c = 0; c < 9; ++c)
      if (c == a) b[c] = 1;
    if (ttt_wins(b, 1)) { rew = 1.f; phase = 2; }
    else if (ttt_empty(b) == 0) phase = 2;
  }
#pragma unroll
  for (int c = 0; c < 9; ++c) {
    s[c] = (float)b[c];
    opp_obs[(size_t)n * 18 + 2 * c] = (phase == 1 && b[c] == 2) ? 1.f : 0.f;      // the opponent's own marks
    opp_obs[(size_t)n * 18 + 2 * c + 1] = (phase == 1 && b[c] == 1) ? 1.f : 0.f;  // the agent's marks
    opp_masks[(size_t)n * 9 + c] = phase == 1 ? (b[c] == 0 ? 1.f : 0.f) : (c == 0 ? 1.f : 0.f);
  }
  s[11] = (float)phase;
  rewards[n] = rew;
  dones[n] = phase == 2 ? 1 : 0;
}

__global__ void ttt_opponent_move_kernel(float* __restrict__ st, float* __restrict__ ep_stats,
                                         const float* __restrict__ opp_actions, float* __restrict__ obs,
                                         float* __restrict__ amask, float* __restrict__ rewards,
                                         uint8_t* __restrict__ dones, int N, uint64_t seed) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float* s = st + (size_t)n * TTT_STATE_W;
  int b[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) b[c] = (int)s[c];
  int moves = (int)s[9], episode = (int)s[10];
  const int phase = (int)s[11];
  float rew = rewards[n];
  bool done = phase == 2;
  if (phase == 1) {
    int a = (int)opp_actions[n];
    bool legal = false;
#pragma unroll
    for (int c = 0; c < 9; ++c) legal = legal || (c == a && b[c] == 0);
    if (!legal) {  // cannot happen under the mask; keep the game well-defined: first empty cell
      a = -1;
#pragma unroll
      for (int c = 8; c >= 0; --c)
        if (b[c] == 0) a = c;
    }
#pragma unroll
    for (int c = 0; c < 9; ++c)
      if (c == a) b[c] = 2;
    ++moves;
    if (ttt_wins(b, 2)) { rew = -1.f; done = true; }
    else if (ttt_empty(b) == 0) done = true;
  }
  if (ep_stats != nullptr) {
    float* e = ep_stats + (size_t)n * 4;
    e[0] += rew; e[1] += 1.f;
    if (done) { e[2] += e[0]; e[3] += 1.f; e[0] = 0.f; e[1] = 0.f; }
  }
  if (done) {
    ++episode;
    moves = ttt_begin(b, seed, (uint32_t)n, (uint32_t)episode);  // an opening move of the opponent stays uniform
  }
  ttt_write(b, moves, episode, s, obs + (size_t)n * 18, amask + (size_t)n * 9);
  rewards[n] = rew;
  dones[n] = done ? 1 : 0;
}

}  // namespace orl

using namespace orl;

extern "C" {

int orl_ttt_state_width(void) { return TTT_STATE_W; }

int orl_ttt_reset(float* env_state, float* ep_stats, float* obs, float* action_masks, int N, uint64_t env_seed,
                  void* stream) {
  ORL_REQUIRE(env_state && obs && action_masks && N > 0, "orl_ttt_reset: bad arguments");
  hipLaunchKernelGGL(ttt_reset_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, env_state, ep_stats,
                     obs, action_masks, N, env_seed);
  return launch_status("orl_ttt_reset");
}

int orl_ttt_step(float* env_state, float* ep_stats, const float* actions, float* obs, float* action_masks,
                 float* rewards, uint8_t* dones, int N, uint64_t env_seed, void* stream) {
  ORL_REQUIRE(env_state && actions && obs && action_masks && rewards && dones && N > 0, "orl_ttt_step: bad arguments");
  hipLaunchKernelGGL(ttt_step_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, env_state, ep_stats,
                     actions, obs, action_masks, rewards, dones, N, env_seed);
  return launch_status("orl_ttt_step");
}

int orl_ttt_agent_move(float* env_state, const float* actions, float* opp_obs, float* opp_masks, float* rewards,
                       uint8_t* dones, int N, void* stream) {
  ORL_REQUIRE(env_state && actions && opp_obs && opp_masks && rewards && dones && N > 0, "orl_ttt_agent_move: bad arguments");
  hipLaunchKernelGGL(ttt_agent_move_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, env_state, actions,
                     opp_obs, opp_masks, rewards, dones, N);
  return launch_status("orl_ttt_agent_move");
}

int orl_ttt_opponent_move(float* env_state, float* ep_stats, const float* opp_actions, float* obs, float* action_masks,
                          float* rewards, uint8_t* dones, int N, uint64_t env_seed, void* stream) {
  ORL_REQUIRE(env_state && opp_actions && obs && action_masks && rewards && dones && N > 0,
              "orl_ttt_opponent_move: bad arguments");
  hipLaunchKernelGGL(ttt_opponent_move_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, env_state,
                     ep_stats, opp_actions, obs, action_masks, rewards, dones, N, env_seed);
  return launch_status("orl_ttt_opponent_move");
}

}  // extern "C"
