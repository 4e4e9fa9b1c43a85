// orl_mpe.h - MPE `simple_spread` world as device functions (3 agents, 3 landmarks), shared by the stepwise env
// kernels (orl_mpe.hip) and the fused recurrent rollout (orl_rnn_rollout.hip) so that both advance a world with the
// same arithmetic.  Reference: openrl/envs/mpe/core.py:216-323, multiagent_env.py:167-310,
// scenarios/simple_spread.py:46-125; auto-reset semantics of envs/vec_env/sync_venv.py:178-247.
#pragma once
#include "orl_common.h"

namespace orl {

constexpr int MPE_A = 3, MPE_L = 3, MPE_STATE_W = 24;  // pos[3][2] vel[3][2] landmark[3][2] step episode - - - -
constexpr int MPE_OBS = 18, MPE_COBS = 54;

// The world is stepped from two translation units built with -ffast-math (orl_mpe.hip, orl_rnn_rollout.hip).  Its
// arithmetic is pinned here - no reassociation, no reciprocal / approximate forms, no cross-statement contraction - so
// that both contexts (state reloaded from memory per step vs kept in registers over a rollout) round identically.
#define ORL_MPE_FP \
  _Pragma("clang fp reassociate(off)") _Pragma("clang fp reciprocal(off)") _Pragma("clang fp contract(off)")

struct MpeWorld {
  float pos[3][2], vel[3][2], lm[3][2];
  float step, episode;
};

__device__ inline void mpe_reset_state(uint64_t seed, uint32_t env, uint32_t episode, float (&pos)[3][2],
                                       float (&vel)[3][2], float (&lm)[3][2]) {
  ORL_MPE_FP
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

__device__ inline void mpe_load(const float* __restrict__ s, MpeWorld& w) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      w.pos[i][d] = s[2 * i + d];
      w.vel[i][d] = s[6 + 2 * i + d];
      w.lm[i][d] = s[12 + 2 * i + d];
    }
  w.step = s[18];
  w.episode = s[19];
}

__device__ inline void mpe_store(float* __restrict__ s, const MpeWorld& w) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      s[2 * i + d] = w.pos[i][d];
      s[6 + 2 * i + d] = w.vel[i][d];
      s[12 + 2 * i + d] = w.lm[i][d];
    }
  s[18] = w.step;
  s[19] = w.episode;
}

// observation of every agent: [vel, pos, landmarks - pos, other agents - pos, comm (zeros)]
__device__ inline void mpe_obs(const MpeWorld& w, float (&o)[3][MPE_OBS]) {
  ORL_MPE_FP
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    o[i][0] = w.vel[i][0]; o[i][1] = w.vel[i][1]; o[i][2] = w.pos[i][0]; o[i][3] = w.pos[i][1];
#pragma unroll
    for (int l = 0; l < 3; ++l) {
      o[i][4 + 2 * l] = w.lm[l][0] - w.pos[i][0];
      o[i][5 + 2 * l] = w.lm[l][1] - w.pos[i][1];
    }
    int k = 10;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (j == i) continue;
      o[i][k++] = w.pos[j][0] - w.pos[i][0];
      o[i][k++] = w.pos[j][1] - w.pos[i][1];
    }
    o[i][14] = 0.f; o[i][15] = 0.f; o[i][16] = 0.f; o[i][17] = 0.f;
  }
}

// One world step with the agents' Discrete(5) actions; returns the shared reward and the done flag, applies the
// auto-reset (the world then already holds the new episode's first state).
__device__ inline void mpe_advance(MpeWorld& w, const int (&act)[3], uint64_t seed, uint32_t env, int world_length,
                                   float& rew, bool& done) {
  ORL_MPE_FP
  float f[3][2];
  // action force: Discrete(5) one-hot, u = [a1 - a2, a3 - a4] * 5 (multiagent_env.py:289-310), mass 1
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int a = act[i];
    f[i][0] = 5.0f * ((a == 1 ? 1.f : 0.f) - (a == 2 ? 1.f : 0.f));
    f[i][1] = 5.0f * ((a == 3 ? 1.f : 0.f) - (a == 4 ? 1.f : 0.f));
  }
  // soft collision forces between agents (core.py:293-323): contact_force 1e2, contact_margin 1e-3, sizes 0.15
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = a + 1; b < 3; ++b) {
      const float dx = w.pos[a][0] - w.pos[b][0], dy = w.pos[a][1] - w.pos[b][1];
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
      w.vel[i][d] = w.vel[i][d] * 0.75f + f[i][d] * 0.1f;
      w.pos[i][d] += w.vel[i][d] * 0.1f;
    }
  // reward (simple_spread.py:84-100): -sum_l min_a dist - #collisions incl. the agent itself; shared = sum
  float cover = 0.f;
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    float md = 3.0e38f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float dx = w.pos[a][0] - w.lm[l][0], dy = w.pos[a][1] - w.lm[l][1];
      md = fminf(md, sqrtf(dx * dx + dy * dy));
    }
    cover += md;
  }
  float coll = 3.f;  // every agent "collides" with itself (distance 0 < 0.3)
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = a + 1; b < 3; ++b) {
      const float dx = w.pos[a][0] - w.pos[b][0], dy = w.pos[a][1] - w.pos[b][1];
      if (sqrtf(dx * dx + dy * dy) < 0.3f) coll += 2.f;  // counted once for a and once for b
    }
  rew = -(3.f * cover) - coll;  // sum over the 3 agents of (-cover - own collisions)
  const float step = w.step + 1.f;
  done = step >= (float)world_length;
  w.step = step;
  if (done) {  // auto-reset (sync_venv.py:217-222): the returned observation is the new episode's first one
    w.episode += 1.f;
    w.step = 0.f;
    mpe_reset_state(seed, env, (uint32_t)w.episode, w.pos, w.vel, w.lm);
  }
}

// ep_stats row [4]: running episode return, length, sum of finished returns, finished count
__device__ inline void mpe_ep_stats(float (&e)[4], float rew, bool done) {
  ORL_MPE_FP
  const float ret = e[0] + rew, len = e[1] + 1.f;
  if (done) { e[2] += ret; e[3] += 1.f; e[0] = 0.f; e[1] = 0.f; }
  else { e[0] = ret; e[1] = len; }
}

}  // namespace orl
