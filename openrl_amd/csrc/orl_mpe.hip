// orl_mpe.hip - device-resident batched MPE `simple_spread` (3 agents, 3 landmarks) for gfx950
// (reference: openrl/envs/mpe/core.py:216-291, multiagent_env.py:167-310,
// scenarios/simple_spread.py:46-125, auto-reset semantics of envs/vec_env/sync_venv.py:178-247).
//
// One lane owns one env (3 agents): 24 floats of state, ~40 flops of physics, 3 x (18 + 54) floats of
// observations out - a pure HBM-streaming kernel; state rows are 96 B, observations leave as the
// [N, 3, 18] / [N, 3, 54] arrays the buffer insert consumes.  fp32 (the reference integrates in float64).
#include "orl_common.h"

namespace orl {

constexpr int MPE_A = 3, MPE_L = 3, MPE_STATE_W = 24;  // pos[3][2] vel[3][2] landmark[3][2] step episode - - - -
constexpr int MPE_OBS = 18, MPE_COBS = 54;

__device__ inline void mpe_reset_state(uint64_t seed, uint32_t env, uint32_t episode, float (&pos)[3][2],
                                       float (&vel)[3][2], float (&lm)[3][2]) {
  float u[12];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const u4 r = philox4x32_10(seed, env, 0x3D9E0000u + k, episode, 0u);
    u[4 * k + 0] = u01(r.x); u[4 * k + 1] = u01(r.y); u[4 * k + 2] = u01(r.z); u[4 * k + 3] = u01(r.w);
  }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      pos[i][d] = u[2 * i + d] * 2.0f - 1.0f;          // np_random.uniform(-1, +1)
      vel[i][d] = 0.f;
      lm[i][d] = 0.8f * (u[6 + 2 * i + d] * 2.0f - 1.0f);  // 0.8 * uniform(-1, +1)
    }
}

// observation of every agent: [vel, pos, landmarks - pos, other agents - pos, comm (zeros)]
__device__ inline void mpe_write_obs(const float (&pos)[3][2], const float (&vel)[3][2], const float (&lm)[3][2],
                                     float* __restrict__ op, float* __restrict__ oc) {
  float o[3][MPE_OBS];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    o[i][0] = vel[i][0]; o[i][1] = vel[i][1]; o[i][2] = pos[i][0]; o[i][3] = pos[i][1];
#pragma unroll
    for (int l = 0; l < 3; ++l) {
      o[i][4 + 2 * l] = lm[l][0] - pos[i][0];
      o[i][5 + 2 * l] = lm[l][1] - pos[i][1];
    }
    int k = 10;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (j == i) continue;
      o[i][k++] = pos[j][0] - pos[i][0];
      o[i][k++] = pos[j][1] - pos[i][1];
    }
    o[i][14] = 0.f; o[i][15] = 0.f; o[i][16] = 0.f; o[i][17] = 0.f;
  }
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
  float pos[3][2], vel[3][2], lm[3][2];
  mpe_reset_state(seed, (uint32_t)n, 0u, pos, vel, lm);
  float* s = st + (size_t)n * MPE_STATE_W;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      s[2 * i + d] = pos[i][d];
      s[6 + 2 * i + d] = vel[i][d];
      s[12 + 2 * i + d] = lm[i][d];
    }
  s[18] = 0.f; s[19] = 0.f; s[20] = 0.f; s[21] = 0.f; s[22] = 0.f; s[23] = 0.f;
  if (ep_stats != nullptr) {
#pragma unroll
    for (int k = 0; k < 4; ++k) ep_stats[(size_t)n * 4 + k] = 0.f;
  }
  mpe_write_obs(pos, vel, lm, obs_p + (size_t)n * MPE_A * MPE_OBS, obs_c ? obs_c + (size_t)n * MPE_A * MPE_COBS : nullptr);
}

__global__ void mpe_step_kernel(float* __restrict__ st, float* __restrict__ ep_stats, const float* __restrict__ actions,
                                float* __restrict__ obs_p, float* __restrict__ obs_c, float* __restrict__ rewards,
                                uint8_t* __restrict__ dones, int N, uint64_t seed, int world_length) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float* s = st + (size_t)n * MPE_STATE_W;
  float pos[3][2], vel[3][2], lm[3][2], f[3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      pos[i][d] = s[2 * i + d];
      vel[i][d] = s[6 + 2 * i + d];
      lm[i][d] = s[12 + 2 * i + d];
    }
  // action force: Discrete(5) one-hot, u = [a1 - a2, a3 - a4] * 5 (multiagent_env.py:289-310), mass 1
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int a = (int)actions[(size_t)n * MPE_A + i];
    f[i][0] = 5.0f * ((a == 1 ? 1.f : 0.f) - (a == 2 ? 1.f : 0.f));
    f[i][1] = 5.0f * ((a == 3 ? 1.f : 0.f) - (a == 4 ? 1.f : 0.f));
  }
  // soft collision forces between agents (core.py:293-323): contact_force 1e2, contact_margin 1e-3, sizes 0.15
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = a + 1; b < 3; ++b) {
      const float dx = pos[a][0] - pos[b][0], dy = pos[a][1] - pos[b][1];
      const float dist = sqrtf(dx * dx + dy * dy);
      const float x = -(dist - 0.3f) / 1e-3f;
      const float pen = (fmaxf(x, 0.f) + log1pf(expf(-fabsf(x)))) * 1e-3f;  // logaddexp(0, x) * k
      const float sc = 1e2f / dist * pen;
      f[a][0] += sc * dx; f[a][1] += sc * dy;
      f[b][0] -= sc * dx; f[b][1] -= sc * dy;
    }
  // integrate (core.py:271-291): damping 0.25, dt 0.1, no max_speed
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      vel[i][d] = vel[i][d] * 0.75f + f[i][d] * 0.1f;
      pos[i][d] += vel[i][d] * 0.1f;
    }
  // reward (simple_spread.py:84-100): -sum_l min_a dist - #collisions incl. the agent itself; shared = sum
  float cover = 0.f;
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    float md = 3.0e38f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float dx = pos[a][0] - lm[l][0], dy = pos[a][1] - lm[l][1];
      md = fminf(md, sqrtf(dx * dx + dy * dy));
    }
    cover += md;
  }
  float coll = 3.f;  // every agent "collides" with itself (distance 0 < 0.3)
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = a + 1; b < 3; ++b) {
      const float dx = pos[a][0] - pos[b][0], dy = pos[a][1] - pos[b][1];
      if (sqrtf(dx * dx + dy * dy) < 0.3f) coll += 2.f;  // counted once for a and once for b
    }
  const float rew = -(3.f * cover) - coll;  // sum over the 3 agents of (-cover - own collisions)
  const float step = s[18] + 1.f;
  const bool done = step >= (float)world_length;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    rewards[(size_t)n * MPE_A + i] = rew;
    dones[(size_t)n * MPE_A + i] = done ? 1 : 0;
  }
  float episode = s[19];
  if (ep_stats != nullptr) {
    float* e = ep_stats + (size_t)n * 4;
    const float ret = e[0] + rew, len = e[1] + 1.f;
    if (done) { e[2] += ret; e[3] += 1.f; e[0] = 0.f; e[1] = 0.f; }
    else { e[0] = ret; e[1] = len; }
  }
  float nstep = step;
  if (done) {  // auto-reset (sync_venv.py:217-222): the returned observation is the new episode's first one
    episode += 1.f;
    nstep = 0.f;
    mpe_reset_state(seed, (uint32_t)n, (uint32_t)episode, pos, vel, lm);
  }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      s[2 * i + d] = pos[i][d];
      s[6 + 2 * i + d] = vel[i][d];
      s[12 + 2 * i + d] = lm[i][d];
    }
  s[18] = nstep;
  s[19] = episode;
  mpe_write_obs(pos, vel, lm, obs_p + (size_t)n * MPE_A * MPE_OBS, obs_c ? obs_c + (size_t)n * MPE_A * MPE_COBS : nullptr);
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
