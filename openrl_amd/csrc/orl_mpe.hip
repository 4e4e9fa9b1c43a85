// orl_mpe.hip - device-resident batched MPE `simple_spread` (3 agents, 3 landmarks) for gfx950
// (reference: openrl/envs/mpe/core.py:216-291, multiagent_env.py:167-310,
// scenarios/simple_spread.py:46-125, auto-reset semantics of envs/vec_env/sync_venv.py:178-247).
//
// One lane owns one env (3 agents): 24 floats of state, ~40 flops of physics, 3 x (18 + 54) floats of
// observations out - a pure HBM-streaming kernel; state rows are 96 B, observations leave as the
// [N, 3, 18] / [N, 3, 54] arrays the buffer insert consumes.  fp32 (the reference integrates in float64).
// The world itself (orl_mpe.h) is shared with the fused recurrent rollout.
#include "orl_common.h"
#include "orl_mpe.h"

namespace orl {

__device__ inline void mpe_write_obs(const MpeWorld& w, float* __restrict__ op, float* __restrict__ oc) {
  float o[3][MPE_OBS];
  mpe_obs(w, o);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < MPE_OBS; ++k) op[i * MPE_OBS + k] = o[i][k];
  if (oc != nullptr) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int k = 0; k < MPE_OBS; ++k) oc[i * MPE_COBS + j * MPE_OBS + k] = o[j][k];
  }
}

__global__ void mpe_reset_kernel(float* __restrict__ st, float* __restrict__ ep_stats, float* __restrict__ obs_p,
                                 float* __restrict__ obs_c, int N, uint64_t seed) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  MpeWorld w;
  mpe_reset_state(seed, (uint32_t)n, 0u, w.pos, w.vel, w.lm);
  w.step = 0.f;
  w.episode = 0.f;
  float* s = st + (size_t)n * MPE_STATE_W;
  mpe_store(s, w);
  s[20] = 0.f; s[21] = 0.f; s[22] = 0.f; s[23] = 0.f;
  if (ep_stats != nullptr) {
#pragma unroll
    for (int k = 0; k < 4; ++k) ep_stats[(size_t)n * 4 + k] = 0.f;
  }
  mpe_write_obs(w, obs_p + (size_t)n * MPE_A * MPE_OBS, obs_c ? obs_c + (size_t)n * MPE_A * MPE_COBS : nullptr);
}

__global__ void mpe_step_kernel(float* __restrict__ st, float* __restrict__ ep_stats, const float* __restrict__ actions,
                                float* __restrict__ obs_p, float* __restrict__ obs_c, float* __restrict__ rewards,
                                uint8_t* __restrict__ dones, int N, uint64_t seed, int world_length) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float* s = st + (size_t)n * MPE_STATE_W;
  MpeWorld w;
  mpe_load(s, w);
  int act[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) act[i] = (int)actions[(size_t)n * MPE_A + i];
  float rew;
  bool done;
  mpe_advance(w, act, seed, (uint32_t)n, world_length, rew, done);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    rewards[(size_t)n * MPE_A + i] = rew;
    dones[(size_t)n * MPE_A + i] = done ? 1 : 0;
  }
  if (ep_stats != nullptr) {
    float* e = ep_stats + (size_t)n * 4;
    float ev[4] = {e[0], e[1], e[2], e[3]};
    mpe_ep_stats(ev, rew, done);
    e[0] = ev[0]; e[1] = ev[1]; e[2] = ev[2]; e[3] = ev[3];
  }
  mpe_store(s, w);
  mpe_write_obs(w, obs_p + (size_t)n * MPE_A * MPE_OBS, obs_c ? obs_c + (size_t)n * MPE_A * MPE_COBS : nullptr);
}

}  // namespace orl

using namespace orl;

extern "C" {

int orl_mpe_state_width(void) { return MPE_STATE_W; }

int orl_mpe_reset(float* env_state, float* ep_stats, float* obs_policy, float* obs_critic, int N, uint64_t env_seed,
                  void* stream) {
  ORL_REQUIRE(env_state && obs_policy && N > 0, "orl_mpe_reset: bad arguments");
  hipLaunchKernelGGL(mpe_reset_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, env_state, ep_stats,
                     obs_policy, obs_critic, N, env_seed);
  return launch_status("orl_mpe_reset");
}

int orl_mpe_step(float* env_state, float* ep_stats, const float* actions, float* obs_policy, float* obs_critic,
                 float* rewards, uint8_t* dones, int N, uint64_t env_seed, int world_length, void* stream) {
  ORL_REQUIRE(env_state && actions && obs_policy && rewards && dones && N > 0 && world_length > 0,
              "orl_mpe_step: bad arguments");
  hipLaunchKernelGGL(mpe_step_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, env_state, ep_stats,
                     actions, obs_policy, obs_critic, rewards, dones, N, env_seed, world_length);
  return launch_status("orl_mpe_step");
}

}  // extern "C"
